// fft_length 8192 on the wave-private 1024-point core: one wave = one real frame.
//
// The frame rides as 4096 complex points c[n] = (x[2n], x[2n+1]) (real-2x packing, like fft_length 2048 / 4096), and the
// 4096-point complex transform is a decimation-in-time split into FOUR passes through the 1024-point core:
//     Y_r = FFT_1024(c[4 n' + r]),  r = 0..3          (lane l reads the pair x[8 (l + 64 s) + 2 r], +1: one 8-byte load)
//     Z[k + 1024 m] = sum_r w_4096^(r k) w_4^(r m) Y_r[k]            (lane-local radix-4: the four Y_r share the core's bin layout)
// followed by the real-input untangle  E = (Z[k] + conj Z[4096 - k]) / 2,  O = -i (Z[k] - conj Z[4096 - k]) / 2,
//     X[k] = E + w_8192^k O,   X[k + 4096] = E - w_8192^k O.
// The partner of bin k + 1024 m is (1024 - k) + 1024 (3 - m): the SAME partner lane as the 1024-point untangle (64 - l /
// 63 - l, ds_bpermute) with the register index m' = 3 - m (bin 0: m' = (4 - m) mod 4), so nothing goes through LDS.
// w_8192^(k + 1024 m) = w_8192^k w_8^m: one 1024-entry table and three constant rotations; w_4096^(2k), ^(3k) are formed from
// w_4096^k.  Every wave instruction stores 1 KiB of the frame's row contiguously (16-byte "sc1 nt" stores, 64 per frame).
// 8 waves per workgroup share 58 KB of tables (the window is held pre-permuted, [r][l + 64 s] -> w[8 (l + 64 s) + 2 r], +1, so
// that the LDS reads are conflict-free); ~200 VGPRs: 2 waves per SIMD.
#include "wave_stft.hpp"

namespace nxsig {

struct Wave8kArgs {
  const float* x;
  int64_t batch_stride, L, lo, M;
  int32_t N, hop, reflect, batch;
  int64_t total_frames, chunk;      // frames of THIS launch (interior or edge set) and frames per workgroup
  int64_t per_row, m_split, m_add0, m_add1;  // frame j of a row in this launch is frame j + (j < m_split ? m_add0 : m_add1)
  const v2f* wperm;   // [4][1024]: (w[8 i + 2 r], w[8 i + 2 r + 1]) at [r][i]
  const v2f* twB;
  const v2f* twC;
  const v2f* tw4k;    // w_4096^k, k < 1024
  const v2f* tw8k;    // w_8192^k, k < 1024
  float div;
  int32_t aligned8;   // every frame start is 8-byte aligned: pairs travel as one 8-byte load
  v2f* z;
};

// GENERAL = false: every sample the frame's 8192-sample span reads lies inside the signal (plain loads, no bounds math);
// GENERAL = true : the few frames at the stream ends / under padding modes / short frames near the end: per-sample fetch.
// NPRED: frame_length < 8192 (samples past the frame are loaded by the streaming kernel but must not reach the transform).
template <bool SCALE, int W, bool GENERAL, bool NPRED>
__global__ __launch_bounds__(64 * W) void k_stft_wave_8k(Wave8kArgs a) {
  constexpr int K = 1024, NQ = 8, P = 16, XCH = K + K / 16 + 16, KOUT = 8192;
  v2f* s_wp = reinterpret_cast<v2f*>(g_wave_smem);
  v2f* s_twB = s_wp + 4 * K;
  v2f* s_twC = s_twB + 256;
  v2f* s_t4 = s_twC + 4 * 256;
  v2f* s_t8 = s_t4 + K;
  v2f* s_x = s_t8 + K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4 * K; i += 64 * W) s_wp[i] = a.wperm[i];
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < 4 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  for (int i = tid; i < K; i += 64 * W) { s_t4[i] = a.tw4k[i]; s_t8[i] = a.tw8k[i]; }
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  const int64_t f_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t f_end = f_begin + a.chunk;
  if (f_end > a.total_frames) f_end = a.total_frames;
  const int src0 = ((64 - lane) & 63) << 2, src1 = (63 - lane) << 2;

  for (int64_t fr = f_begin + wave; fr < f_end; fr += W) {
    const int64_t row = fr / a.per_row, jf = fr - row * a.per_row;
    const int64_t m_fr = jf + (jf < a.m_split ? a.m_add0 : a.m_add1);
    const float* xr = a.x + (size_t)row * a.batch_stride;
    const int64_t q0 = m_fr * a.hop;                    // padded index of the frame's first sample
    const float* xf = xr + (q0 - a.lo);
    int n_frame = a.N;
    asm volatile("" : "+s"(n_frame));  // opaque per frame: 128 loop-invariant compare masks would otherwise be hoisted into (spilled) SGPRs
    v2f y[4][2][NQ];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v2f d[P];
#pragma unroll
      for (int s = 0; s < P; ++s) {
        const int n2 = 8 * (lane + 64 * s) + 2 * r;     // sample index of the complex point's real part
        float va, vb;
        if (!GENERAL) {
          if (a.aligned8) { const v2f t = *reinterpret_cast<const v2f*>(xf + n2); va = t.x; vb = t.y; }
          else { va = xf[n2]; vb = xf[n2 + 1]; }
        } else {
          WaveArgs g;  // fetch_any reads L, lo, reflect only
          g.L = a.L; g.lo = a.lo; g.reflect = a.reflect;
          va = n2 < n_frame ? fetch_any(xr, g, q0 + n2) : 0.0f;
          vb = n2 + 1 < n_frame ? fetch_any(xr, g, q0 + n2 + 1) : 0.0f;
        }
        const v2f w = s_wp[r * K + lane + 64 * s];
        d[s] = v2f{va * w.x, vb * w.y};                 // exact f32 products (lib/nx_signal.ex:101)
        if (NPRED && !GENERAL) {                        // samples past a short frame never reach the transform (0 x Inf)
          if (n2 >= n_frame) d[s].x = 0.0f;
          if (n2 + 1 >= n_frame) d[s].y = 0.0f;
        }
      }
      wave_fft_core<K>(d, y[r], xb, s_twB, s_twC, lane);
      __builtin_amdgcn_sched_barrier(0);  // keep the next quarter's loads from being hoisted above this core (register pressure)
    }
    // ---- radix-4 combine: y[m][par][q] <- Z[k + 1024 m],  k = 2 lane + par + 128 q
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const v2f t1 = s_t4[2 * lane + e + 128 * q];
        const v2f t2 = wcmul(t1, t1), t3 = wcmul(t2, t1);
        v2f a0 = y[0][e][q], a1 = wcmul(y[1][e][q], t1), a2 = wcmul(y[2][e][q], t2), a3 = wcmul(y[3][e][q], t3);
        dft4(a0, a1, a2, a3);
        y[0][e][q] = a0; y[1][e][q] = a1; y[2][e][q] = a2; y[3][e][q] = a3;
      }
    // ---- untangle the real spectrum + store: 16 bytes = bins (k, k + 1) of one frame row
    const StreamRow zs(a.z + ((size_t)row * a.M + m_fr) * KOUT, KOUT * 8);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        // partner of bin k + 1024 m: (1024 - k) + 1024 (3 - m) on lane 64 - l / 63 - l; bin 0 of lane 0: m' = (4 - m) mod 4
        const v2f own0 = q == 0 ? y[(4 - m) % 4][0][0] : y[3 - m][0][(NQ - q) % NQ];  // (% NQ: the q == 0 arm must stay in bounds)
        v2f p0, p1;
        p0.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src0, __float_as_int(y[3 - m][0][NQ - 1 - q].x)));
        p0.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src0, __float_as_int(y[3 - m][0][NQ - 1 - q].y)));
        p1.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src1, __float_as_int(y[3 - m][1][NQ - 1 - q].x)));
        p1.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src1, __float_as_int(y[3 - m][1][NQ - 1 - q].y)));
        if (lane == 0) p0 = own0;
        const v2f z0 = y[m][0][q], z1 = y[m][1][q];
        const v4f E = v4f{z0.x + p0.x, z0.y - p0.y, z1.x + p1.x, z1.y - p1.y} * 0.5f;
        const v4f O = v4f{z0.y + p0.y, p0.x - z0.x, z1.y + p1.y, p1.x - z1.x} * 0.5f;
        const v4f t = *reinterpret_cast<const v4f*>(&s_t8[2 * lane + 128 * q]);   // w_8192^k, w_8192^(k+1)
        v2f o0 = wcmul(v2f{O.x, O.y}, v2f{t.x, t.y}), o1 = wcmul(v2f{O.z, O.w}, v2f{t.z, t.w});
        if (m == 1) { o0 = rot45<false>(o0); o1 = rot45<false>(o1); }             // x w_8^m
        if (m == 2) { o0 = rot90<false>(o0); o1 = rot90<false>(o1); }
        if (m == 3) { o0 = rot135<false>(o0); o1 = rot135<false>(o1); }
        const v4f to = v4f{o0.x, o0.y, o1.x, o1.y};
        v4f lo4 = E + to, hi4 = E - to;
        lo4 = fft_eps0(lo4); hi4 = fft_eps0(hi4);  // Nx.fft's clean-up (:102) precedes the scaling
        if (SCALE) { lo4 = lo4 / a.div; hi4 = hi4 / a.div; }
        zs.st16(lo4, lane * 16 + 1024 * q + 8192 * m);
        zs.st16(hi4, lane * 16 + 1024 * q + 8192 * m + 32768);
        if (q & 1) __builtin_amdgcn_sched_barrier(0);  // two bin groups in flight: the scheduler would otherwise hoist all 128 partner fetches
      }
  }
}

int launch_stft_wave_8k(Ctx* c, const StftLaunch& s, bool* handled) {
  *handled = false;
  constexpr int W = 8, K = 1024, XCH = K + K / 16 + 16;
  if (s.K != 8192 || s.fr.N > 8192 || tune(c, kT_DISABLE_8K, 0)) return NXSIG_OK;
  if ((int)c->memo_win.size() != s.fr.N) return NXSIG_OK;  // the host copy of this call's window (ctx_window) is needed
  int rc = ensure_wave_tables(c, K);
  if (rc) return rc;
  *handled = true;
  Wave8kArgs a;
  a.x = s.x; a.batch_stride = s.batch_stride; a.L = s.fr.L; a.lo = s.fr.lo; a.M = s.fr.M;
  a.N = s.fr.N; a.hop = s.fr.hop; a.reflect = s.fr.reflect; a.batch = s.batch;
  a.div = s.inv_scale_div;
  a.z = reinterpret_cast<v2f*>(s.z);
  a.aligned8 = ((reinterpret_cast<uintptr_t>(s.x) & 7) == 0 && (s.batch_stride & 1) == 0 && (s.fr.hop & 1) == 0 && (s.fr.lo & 1) == 0) ? 1 : 0;
  Ctx::WaveTables& wt = c->wave_tables[K];
  a.twB = reinterpret_cast<const v2f*>(wt.twB);
  a.twC = reinterpret_cast<const v2f*>(wt.twC);
  {
    std::vector<float2> wp((size_t)4 * K);
    for (int r = 0; r < 4; ++r)
      for (int i = 0; i < K; ++i) {
        const int n = 8 * i + 2 * r;
        wp[(size_t)r * K + i] = make_float2(n < s.fr.N ? c->memo_win[n] : 0.0f, n + 1 < s.fr.N ? c->memo_win[n + 1] : 0.0f);
      }
    const void* d = nullptr;
    if ((rc = ctx_table(c, 0x8B10ull, wp.data(), wp.size() * sizeof(float2), &d))) return rc;
    a.wperm = reinterpret_cast<const v2f*>(d);
  }
  {
    auto hit = c->memo.find(0x8B11000000000000ull);
    if (hit != c->memo.end()) { a.tw4k = reinterpret_cast<const v2f*>(hit->second[0]); a.tw8k = reinterpret_cast<const v2f*>(hit->second[1]); }
    else {
      const double two_pi = 6.283185307179586476925286766559;
      std::vector<float2> t4(K), t8(K);
      for (int k = 0; k < K; ++k) {
        t4[k] = make_float2((float)std::cos(-two_pi * k / 4096.0), (float)std::sin(-two_pi * k / 4096.0));
        t8[k] = make_float2((float)std::cos(-two_pi * k / 8192.0), (float)std::sin(-two_pi * k / 8192.0));
      }
      const void *d4 = nullptr, *d8 = nullptr;
      if ((rc = ctx_table(c, 0x8B12ull, t4.data(), t4.size() * sizeof(float2), &d4))) return rc;
      if ((rc = ctx_table(c, 0x8B13ull, t8.data(), t8.size() * sizeof(float2), &d8))) return rc;
      c->memo[0x8B11000000000000ull] = {reinterpret_cast<uint64_t>(d4), reinterpret_cast<uint64_t>(d8)};
      a.tw4k = reinterpret_cast<const v2f*>(d4); a.tw8k = reinterpret_cast<const v2f*>(d8);
    }
  }
  // interior frames [m_lo, m_hi): the whole 8192-sample span lies inside the signal (whatever the padding mode)
  const int64_t hop = s.fr.hop, lo = s.fr.lo, M = s.fr.M;
  int64_t m_lo = lo > 0 ? (lo + hop - 1) / hop : 0;
  int64_t m_hi = (s.fr.L + lo - 8192 >= 0) ? (s.fr.L + lo - 8192) / hop + 1 : 0;
  if (m_hi > M) m_hi = M;
  if (m_lo > M) m_lo = M;
  if (m_hi < m_lo) m_hi = m_lo;
  const size_t lds = (size_t)(4 * K + 256 + 4 * 256 + 2 * K) * 8 + (size_t)W * XCH * 8;
  auto go = [&](auto kernel, int64_t per_row, int64_t split, int64_t add0, int64_t add1, int fpw) -> int {
    if (per_row <= 0) return NXSIG_OK;
    a.per_row = per_row; a.m_split = split; a.m_add0 = add0; a.m_add1 = add1;
    a.total_frames = per_row * s.batch;
    a.chunk = (int64_t)W * fill_units_per_wave(c, a.total_frames, W, fpw < 1 ? 1 : fpw);
    const int64_t blocks = (a.total_frames + a.chunk - 1) / a.chunk;
    if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "stft: too many frames for one launch");
    NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note(split < ((int64_t)1 << 61) ? "stft.8k.edge" : "stft.8k");
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  const int fpw = 2;
  const int64_t big = (int64_t)1 << 62;
  if (s.fr.N < 8192)
    rc = s.has_scale ? go(k_stft_wave_8k<true, W, false, true>, m_hi - m_lo, big, m_lo, m_lo, fpw)
                     : go(k_stft_wave_8k<false, W, false, true>, m_hi - m_lo, big, m_lo, m_lo, fpw);
  else
    rc = s.has_scale ? go(k_stft_wave_8k<true, W, false, false>, m_hi - m_lo, big, m_lo, m_lo, fpw)
                     : go(k_stft_wave_8k<false, W, false, false>, m_hi - m_lo, big, m_lo, m_lo, fpw);
  if (rc) return rc;
  return s.has_scale ? go(k_stft_wave_8k<true, W, true, true>, m_lo + (M - m_hi), m_lo, 0, m_hi - m_lo, 1)
                     : go(k_stft_wave_8k<false, W, true, true>, m_lo + (M - m_hi), m_lo, 0, m_hi - m_lo, 1);
}

}  // namespace nxsig
