// Wave kernels, part 9 (round 6): FIR filters of 1 026 ... 32 769 taps as a UNIFORMLY PARTITIONED FREQUENCY-DOMAIN DELAY LINE
// (Convolution.fftconvolve with a long second operand, lib/nx_signal/convolution.ex:252-298: room impulse responses are its use case;
// the reference transforms the whole row once and has no tap limit).
//
// The partitioned form of round 5 (api.cpp: launch_fir_partitioned) ran every partition of <= 1 025 taps as its own overlap-save FIR —
// P x (forward + inverse transform) per block and a summing pass: 3.1 ms for 4 097 taps on BASELINE config 5's shard, 0.018 of the
// roofline at 16 385.  Here every input block is transformed ONCE, the partitions meet it in the frequency domain, and every output block
// is inverted ONCE:
//
//   h = h_0 | h_1 | ... | h_{P-1},  h_p = h[1024 p .. 1024 p + 1024)  (the last one may hold 1 025 taps),  P = ceil((taps - 1) / 1024)
//   window m of a row = the 2 048 samples x[n0 + 1024 m - 1024 .. n0 + 1024 m + 1024)   (n0 = out_start, zeros outside the row)
//   k_fir_dline_fwd :  Z_m = FFT_1024(z),  z[j] = window[2 j] + i window[2 j + 1]                          one transform per 1 024 samples
//   k_fir_dline_mac :  W_k[b] = sum_p  A_p[b] Z_{k-p}[b] + B_p[b] conj Z_{k-p}[(1024 - b) mod 1024]       the delay line: a thread owns the
//                      bin pair (b, 1024 - b) and walks the blocks k of a run with the last P spectra of its two bins in registers;
//                      A_p, B_p = k_fir_r2k's fused untangle x H_p x re-tangle coefficients (kernels_wave.hip), 1 / 1024 folded in
//   k_fir_dline_inv :  (y[2 j], y[2 j + 1]) = IFFT_1024(W_k)[j];  outputs 1 024 ... 2 047 of the block are y[1024 k .. 1024 k + 1024)
//
// The sum over the partitions happens BEFORE the one inverse transform, so Nx.ifft's clean-up (|y| <= 1e-10 -> 0, convolution.ex:282)
// applies to finished samples exactly as in the reference (the round-5 form scaled the taps by 2^20 to keep it off its partial sums).
// Traffic: 8 + 8 (fwd) + 8 + 8 (mac) + 8 + 4 (inv) bytes per sample whatever the tap count (44 against the algorithmic 8: the three passes
// run at ~5 TB/s, so the ceiling of this form is ~0.18 of the roofline; fusing the delay line into the inverse pass is the next step);
// rows are cut into segments whose spectra (Z and W) fit 1 GB of scratch.  More than 16 partitions: the delay line runs in passes of 16
// partitions that accumulate into W.
#include "wave_stft.hpp"

#include <vector>

namespace nxsig {

struct DlineArgs {
  const float* x;
  int64_t L, batch_stride;
  int32_t batch, P;
  int64_t n0, out_len;        // y[i] = (x * h)[n0 + i], i < out_len
  int64_t m_first, nwin;      // windows m_first .. m_first + nwin - 1 of every row are in Z (m_first = k_first - P + 1)
  int64_t k_first, nblk;      // output blocks of this segment
  const v2f* twB;
  const v2f* twC;
  const v2f* coef;            // c64[2][P][1024][2]: (A_p[b], B_p[b]), then (A_p[b], conj B_p[(1024 - b) mod 1024])
  v2f* Z;                     // c64[batch][nwin][1024]
  v2f* Wt;                    // c64[batch][nblk][1024]
  float* y;
  int* row_flags;
};

// ---- forward: one wave per window
template <int W>
__global__ __launch_bounds__(64 * W) void k_fir_dline_fwd(DlineArgs a, int64_t total) {
  constexpr int K = 1024, P = 16, R3 = 4, NQ = 8, XCH = K + K / 16 + 16;
  v2f* s_twB = reinterpret_cast<v2f*>(g_wave_smem);
  v2f* s_twC = s_twB + 256;
  v2f* s_x = s_twC + R3 * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < R3 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  for (int64_t u = (int64_t)blockIdx.x * W + wave; u < total; u += (int64_t)gridDim.x * W) {
    const int64_t row = u / a.nwin, wi = u - row * a.nwin;
    const int64_t s0 = a.n0 + (a.m_first + wi) * 1024 - 1024;      // first sample of the window (index into the row)
    const float* xr = a.x + (size_t)row * a.batch_stride;
    v2f zz[2][NQ];
    if (s0 >= 0 && s0 + 2048 <= a.L && ((reinterpret_cast<uintptr_t>(xr + s0) & 15) == 0)) {   // wave-uniform
      const v4f* p = reinterpret_cast<const v4f*>(xr + s0) + lane;
#pragma unroll
      for (int q = 0; q < NQ; ++q) { const v4f r = p[64 * q]; zz[0][q] = v2f{r.x, r.y}; zz[1][q] = v2f{r.z, r.w}; }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t i = s0 + 4 * lane + 256 * q + e;
          r[e] = (i >= 0 && i < a.L) ? xr[i] : 0.0f;
        }
        zz[0][q] = v2f{r[0], r[1]}; zz[1][q] = v2f{r[2], r[3]};
      }
    }
    v2f d[P];
    wave_fft_core_T<K>(zz, d, xb, s_twB, s_twC, lane);   // d[s] = Z[lane + 64 s]
    v2f* zo = a.Z + ((size_t)row * a.nwin + wi) * K + lane;
#pragma unroll
    for (int s = 0; s < P; ++s) zo[64 * s] = d[s];
  }
}

// ---- the delay line: a thread owns bin b of a (row, run) and walks the run's blocks with the last PMAX spectra of its bin AND of the
// partner bin (1024 - b) mod 1024 in registers (the partner's thread holds the mirror image: every spectrum value is loaded twice, the
// second time from cache; one bin per thread keeps the coefficient registers at 4 per partition).  The ring of PMAX slots is indexed
// statically: the loop over blocks is unrolled by PMAX, block j of a round puts its window into slot j, partition p reads slot j - p.
template <int PMAX>
__global__ __launch_bounds__(256) void k_fir_dline_mac(DlineArgs a, int p_lo, int np, int accumulate, int64_t run_len, int64_t runs_per_row) {
  const int64_t bid = blockIdx.x;
  const int quarter = (int)(bid & 3);
  const int64_t rr = bid >> 2, row = rr / runs_per_row, run = rr - row * runs_per_row;
  const int kb_ = quarter * 256 + threadIdx.x;          // this thread's bin
  const int kp = (1024 - kb_) & 1023;                   // its partner (bins 0 and 512 are their own)
  v2f cA[PMAX], cB[PMAX];
#pragma unroll
  for (int p = 0; p < PMAX; ++p) {
    if (p < np) {
      const v4f c0 = *reinterpret_cast<const v4f*>(a.coef + ((size_t)(p_lo + p) * 1024 + kb_) * 2);
      cA[p] = v2f{c0.x, c0.y}; cB[p] = v2f{c0.z, c0.w};
    } else {
      cA[p] = v2f{0.f, 0.f}; cB[p] = v2f{0.f, 0.f};
    }
  }
  v2f za[PMAX], zb[PMAX];   // ring: own bin / partner bin of the last PMAX windows
#pragma unroll
  for (int p = 0; p < PMAX; ++p) { za[p] = v2f{0.f, 0.f}; zb[p] = v2f{0.f, 0.f}; }
  const int64_t k0 = a.k_first + run * run_len;
  int64_t k1 = k0 + run_len;
  if (k1 > a.k_first + a.nblk) k1 = a.k_first + a.nblk;
  if (k0 >= k1) return;
  const v2f* zrow = a.Z + (size_t)row * a.nwin * 1024;
  v2f* wrow = a.Wt + (size_t)row * a.nblk * 1024;
  // window of (block k, partition p_lo + p) = k - p_lo - p; its index in Z is that minus m_first (>= 0 by construction); a run warms the
  // ring up over the np - 1 windows in front of its first block
  constexpr int D = PMAX < 4 ? PMAX : 4;   // windows in flight
  const int64_t kw0 = k0 - (np - 1);
  v2f pa[D], pb[D];
  auto zidx = [&](int64_t k) { int64_t i = k - p_lo - a.m_first; return i < 0 ? (int64_t)0 : (i >= a.nwin ? a.nwin - 1 : i); };
#pragma unroll
  for (int dd = 0; dd < D; ++dd) {
    const v2f* zp = zrow + (size_t)zidx(kw0 + dd) * 1024;
    pa[dd] = zp[kb_]; pb[dd] = zp[kp];
  }
  for (int64_t kk = kw0; kk < k1; kk += PMAX) {
    // (generic lambda + integral_constant: the ring indices must be compile-time constants; a 16-fold `#pragma unroll` of this body is
    // beyond the unroller's budget and the ring went to scratch)
    auto step = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j < PMAX) {
        const int64_t k = kk + j;
        za[j] = pa[j % D]; zb[j] = pb[j % D];
        {   // refill the slot with window k + D
          const v2f* zp = zrow + (size_t)zidx(k + D) * 1024;
          pa[j % D] = zp[kb_]; pb[j % D] = zp[kp];
        }
        if (k >= k0 && k < k1) {
          v2f w = v2f{0.f, 0.f};
#pragma unroll
          for (int p = 0; p < PMAX; ++p) {
            const int sl = (j - p + PMAX) % PMAX;
            w += wcmul(cA[p], za[sl]) + wcmul_conj(cB[p], zb[sl]);   // A Z[b] + B conj Z[(1024 - b) mod 1024]
          }
          v2f* wp = wrow + (size_t)(k - a.k_first) * 1024 + kb_;
          if (accumulate) w += *wp;
          *wp = w;
        }
        __builtin_amdgcn_sched_barrier(0);   // one block at a time
      }
    };
#define NXSIG_STEP(J) step(std::integral_constant<int, J>{});
    NXSIG_STEP(0) NXSIG_STEP(1) NXSIG_STEP(2) NXSIG_STEP(3) NXSIG_STEP(4) NXSIG_STEP(5) NXSIG_STEP(6) NXSIG_STEP(7)
    NXSIG_STEP(8) NXSIG_STEP(9) NXSIG_STEP(10) NXSIG_STEP(11) NXSIG_STEP(12) NXSIG_STEP(13) NXSIG_STEP(14) NXSIG_STEP(15)
#undef NXSIG_STEP
  }
}

// ---- inverse: one wave per output block
template <int W>
__global__ __launch_bounds__(64 * W) void k_fir_dline_inv(DlineArgs a, int64_t total) {
  constexpr int K = 1024, P = 16, R3 = 4, NQ = 8, XCH = K + K / 16 + 16;
  v2f* s_twB = reinterpret_cast<v2f*>(g_wave_smem);
  v2f* s_twC = s_twB + 256;
  v2f* s_x = s_twC + R3 * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < R3 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  for (int64_t u = (int64_t)blockIdx.x * W + wave; u < total; u += (int64_t)gridDim.x * W) {
    const int64_t row = u / a.nblk, bi = u - row * a.nblk;
    const v2f* wp = a.Wt + ((size_t)row * a.nblk + bi) * K + lane;
    v2f d[P];
#pragma unroll
    for (int s = 0; s < P; ++s) d[s] = wp[64 * s];
    v2f y2[2][NQ];
    wave_fft_core<K, true, true>(d, y2, xb, s_twB, s_twC, lane);   // unscaled inverse (1 / 1024 rides in A, B): y2[par][q] = (y[2 j], y[2 j + 1]), j = 2 lane + par + 128 q
    // non-finite samples: every output of a block depends on every sample of its windows (FirLaunch::row_flags: the row is poisoned)
    if (wave_any_nonfinite(y2[0][NQ - 1].x, y2[0][NQ - 1].y) && lane == 0) atomicOr(a.row_flags + row, 1);
    const int64_t i0 = (a.k_first + bi) * 1024;     // y index of the block's first valid output (block sample 1 024)
    float* yr = a.y + (size_t)row * a.out_len;
#pragma unroll
    for (int q = NQ / 2; q < NQ; ++q) {
      const int64_t i = i0 + 4 * lane + 256 * (q - NQ / 2);
      const v4f o = fft_eps0(v4f{y2[0][q].x, y2[0][q].y, y2[1][q].x, y2[1][q].y});   // Nx.ifft's clean-up on the finished samples
      if (i + 3 < a.out_len && ((reinterpret_cast<uintptr_t>(yr + i) & 15) == 0)) {
        __builtin_nontemporal_store(o, (gv4f*)(yr + i));
      } else {
        if (i < a.out_len) yr[i] = o.x;
        if (i + 1 < a.out_len) yr[i + 1] = o.y;
        if (i + 2 < a.out_len) yr[i + 2] = o.z;
        if (i + 3 < a.out_len) yr[i + 3] = o.w;
      }
    }
  }
}

// ---- delay line FUSED into the inverse pass (<= 4 partitions, i.e. <= 4 097 taps): one wave per output block reads the block's np
// windows (8 KB each: the np - 1 older ones were read by the neighbouring blocks a moment ago, L2), forms W_k in the core's input layout —
// the partner bins conj Z[(1024 - b) mod 1024] come from the partner lane like in k_fir_r2k — and inverts it; W never exists in memory:
// 8 + 8 (fwd) + 8 + 4 bytes per sample instead of 44.  The coefficients of all partitions sit in LDS (16 KB per partition, shared by the
// workgroup's eight waves).
template <int W, int PMAX>
__global__ __launch_bounds__(64 * W) void k_fir_dline_macinv(DlineArgs a, int64_t total, int bpw) {
  constexpr int K = 1024, P = 16, R3 = 4, NQ = 8, XCH = K + K / 16 + 16;
  v2f* s_twB = reinterpret_cast<v2f*>(g_wave_smem);
  v2f* s_twC = s_twB + 256;
  v4f* s_cf = reinterpret_cast<v4f*>(s_twC + R3 * 256);          // [PMAX][1024] (A, B)
  v2f* s_x = reinterpret_cast<v2f*>(s_cf + PMAX * K);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < R3 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  for (int i = tid; i < a.P * K; i += 64 * W) s_cf[i] = reinterpret_cast<const v4f*>(a.coef)[a.P * K + i];   // second table: (A_p[b], conj B_p[-b])
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  const int src = ((64 - lane) & 63) << 2;
  // a wave takes bpw consecutive blocks of one row (their windows overlap: cache hits)
  const int64_t u0 = ((int64_t)blockIdx.x * W + wave) * bpw;
  for (int64_t u = u0; u < u0 + bpw && u < total; ++u) {
    const int64_t row = u / a.nblk, bi = u - row * a.nblk;
    const v2f* zrow = a.Z + (size_t)row * a.nwin * K + lane;
    // window of (block k_first + bi, partition p) sits at index bi + (P_total - 1) - p of the segment's Z
    v2f acc[P];
#pragma unroll
    for (int s = 0; s < P; ++s) acc[s] = v2f{0.f, 0.f};
    v2f db[2][P];   // the window in use and the one in flight trade places (static indices: the loop over the partitions is unrolled)
    {
      const v2f* zp = zrow + (size_t)(bi + a.P - 1) * K;
#pragma unroll
      for (int s = 0; s < P; ++s) db[0][s] = zp[64 * s];
    }
    // W[b] = sum_p A_p[b] Z_p[b] + B_p[b] conj Z_p[-b] = U[b] + conj V[-b],  V[b] = sum_p conj(B_p[-b]) Z_p[b]: both sums run over the
    // lane's OWN bins (the table holds (A_p[b], conj B_p[-b])), and the partner lane is visited once, at the end, instead of once per partition
    v2f vv[P];
#pragma unroll
    for (int s = 0; s < P; ++s) vv[s] = v2f{0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
      if (p < a.P) {   // wave-uniform
        if (p + 1 < a.P) {
          const v2f* zp = zrow + (size_t)(bi + a.P - 2 - p) * K;
#pragma unroll
          for (int s = 0; s < P; ++s) db[(p + 1) & 1][s] = zp[64 * s];
        }
#pragma unroll
        for (int s = 0; s < P; ++s) {
          const v4f cf = s_cf[p * K + lane + 64 * s];
          acc[s] += wcmul(v2f{cf.x, cf.y}, db[p & 1][s]);
          vv[s] += wcmul(v2f{cf.z, cf.w}, db[p & 1][s]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < P; ++s) {
      v2f pv;
      pv.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(vv[P - 1 - s].x)));
      pv.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(vv[P - 1 - s].y)));
      if (lane == 0) pv = vv[(P - s) % P];
      acc[s] += v2f{pv.x, -pv.y};
    }
    v2f y2[2][NQ];
    wave_fft_core<K, true, true>(acc, y2, xb, s_twB, s_twC, lane);
    if (wave_any_nonfinite(y2[0][NQ - 1].x, y2[0][NQ - 1].y) && lane == 0) atomicOr(a.row_flags + row, 1);
    const int64_t i0 = (a.k_first + bi) * 1024;
    float* yr = a.y + (size_t)row * a.out_len;
#pragma unroll
    for (int q = NQ / 2; q < NQ; ++q) {
      const int64_t i = i0 + 4 * lane + 256 * (q - NQ / 2);
      const v4f o = fft_eps0(v4f{y2[0][q].x, y2[0][q].y, y2[1][q].x, y2[1][q].y});
      if (i + 3 < a.out_len && ((reinterpret_cast<uintptr_t>(yr + i) & 15) == 0)) {
        __builtin_nontemporal_store(o, (gv4f*)(yr + i));
      } else {
        if (i < a.out_len) yr[i] = o.x;
        if (i + 1 < a.out_len) yr[i + 1] = o.y;
        if (i + 2 < a.out_len) yr[i + 2] = o.z;
        if (i + 3 < a.out_len) yr[i + 3] = o.w;
      }
    }
  }
}

static void host_fft_f64(std::vector<double>& re, std::vector<double>& im) {   // radix-2, in place, power-of-two length
  const size_t n = re.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = -6.283185307179586476925286766559 / (double)len;
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < len / 2; ++k) {
        const double wr = std::cos(ang * (double)k), wi = std::sin(ang * (double)k);
        const size_t u = i + k, v = i + k + len / 2;
        const double tr = re[v] * wr - im[v] * wi, ti = re[v] * wi + im[v] * wr;
        re[v] = re[u] - tr; im[v] = im[u] - ti;
        re[u] += tr; im[u] += ti;
      }
  }
}

// 1 026 ... 32 769 taps.  *handled stays false when the shape is not this path's (the caller falls back to the partitioned form).
int launch_fir_dline(Ctx* c, const FirLaunch& s, bool* handled) {
  *handled = false;
  if (s.taps < 1026 || s.taps > 32769 || s.out_len <= 0 || s.batch <= 0) return NXSIG_OK;
  const int P = (s.taps - 1 + 1023) / 1024;
  int rc = ensure_wave_tables(c, 1024);
  if (rc) return rc;
  *handled = true;
  dispatch_note("fir.dline");
  {  // many rows: the scratch of a segment (at least 64 blocks of every row) must stay within its budget, so the rows go in groups
    const int64_t per_row = (int64_t)(64 + P - 1) * 16384;
    int64_t rows_max = ((int64_t)1 << 30) / per_row;
    if (rows_max < 1) rows_max = 1;
    if (s.batch > rows_max) {
      for (int32_t r0 = 0; r0 < s.batch; r0 += (int32_t)rows_max) {
        FirLaunch g = s;
        g.batch = s.batch - r0 < rows_max ? s.batch - r0 : (int32_t)rows_max;
        g.x = s.x + (size_t)r0 * s.batch_stride;
        g.y = s.y + (size_t)r0 * s.out_len;
        g.row_flags = s.row_flags + r0;
        bool h2 = false;
        if ((rc = launch_fir_dline(c, g, &h2))) return rc;
      }
      return NXSIG_OK;
    }
  }
  // ---- coefficient table (per distinct filter: memoised by content)
  const uint64_t hkey = fnv1a(0xD11E0000ull ^ (uint64_t)s.taps, s.h_host, (size_t)s.taps * sizeof(float));
  const void* cd = nullptr;
  auto hit = c->memo.find(hkey);
  if (hit != c->memo.end()) cd = reinterpret_cast<const void*>(hit->second[0]);
  else {
    std::vector<float2> coef((size_t)P * 1024 * 4);   // [P][1024] (A_p[b], B_p[b]) for the three passes, then [P][1024] (A_p[b], conj B_p[-b]) for the fused pass
    std::vector<double> re(2048), im(2048);
    for (int p = 0; p < P; ++p) {
      std::fill(re.begin(), re.end(), 0.0); std::fill(im.begin(), im.end(), 0.0);
      const int lo = p * 1024, hi = (p == P - 1) ? s.taps : lo + 1024;     // the last partition takes up to 1 025 taps
      for (int i = lo; i < hi; ++i) re[i - lo] = (double)s.h_host[i];
      host_fft_f64(re, im);
      for (int k = 0; k < 1024; ++k) {   // k_fir_r2k's coefficients (kernels_wave.hip) of the 2 048-point spectrum of h_p
        const double sr = 0.5 * (re[k] + re[k + 1024]), si = 0.5 * (im[k] + im[k + 1024]);
        const double dr = 0.5 * (re[k] - re[k + 1024]), di = 0.5 * (im[k] - im[k + 1024]);
        const double th = 6.283185307179586476925286766559 * (double)k / 2048.0, sn = std::sin(th), cs = std::cos(th);
        coef[((size_t)p * 1024 + k) * 2] = make_float2((float)((sr - dr * sn) / 1024.0), (float)((si - di * sn) / 1024.0));
        coef[((size_t)p * 1024 + k) * 2 + 1] = make_float2((float)(-di * cs / 1024.0), (float)(dr * cs / 1024.0));
      }
      for (int k = 0; k < 1024; ++k) {
        const float2 A = coef[((size_t)p * 1024 + k) * 2], Bm = coef[((size_t)p * 1024 + ((1024 - k) & 1023)) * 2 + 1];
        coef[((size_t)(P + p) * 1024 + k) * 2] = A;
        coef[((size_t)(P + p) * 1024 + k) * 2 + 1] = make_float2(Bm.x, -Bm.y);
      }
    }
    if ((rc = ctx_table(c, 0xD11E1ull ^ ((uint64_t)s.taps << 20), coef.data(), coef.size() * sizeof(float2), &cd))) return rc;
    c->memo[hkey] = {reinterpret_cast<uint64_t>(cd)};
  }
  Ctx::WaveTables& wt = c->wave_tables[1024];
  DlineArgs a;
  a.x = s.x; a.L = s.L; a.batch_stride = s.batch_stride; a.batch = s.batch; a.P = P;
  a.n0 = s.out_start; a.out_len = s.out_len;
  a.twB = reinterpret_cast<const v2f*>(wt.twB); a.twC = reinterpret_cast<const v2f*>(wt.twC);
  a.coef = reinterpret_cast<const v2f*>(cd);
  a.y = s.y; a.row_flags = s.row_flags;
  const int64_t nblk_total = (s.out_len + 1023) / 1024;
  // <= 4 partitions: the delay line rides in the inverse pass (k_fir_dline_macinv: no W tensor); NXSIG_FIR_DLINE=2 keeps the three passes
  const bool fused = P <= 4 && tune(c, kT_FIR_DLINE, 1) != 2;
  // segment length: Z + W of a segment within 1 GB of scratch.  (Segments small enough for the Infinity Cache were measured and LOSE:
  // config 5's shard at 4 097 taps 3.39 / 2.62 / 2.15 / 1.99 / 1.86 / 1.78 ms with 32 / 64 / 128 / 192 / 1 024 / 4 096 MB — three launches per
  // segment cost more than the cache returns; 1 GB bounds the scratch, profiles/r06/fir_delay_line.txt)
  const int seg_mb = tune(c, kT_FIR_DLINE, 1) >= 16 ? tune(c, kT_FIR_DLINE, 1) : 1024;   // (NXSIG_FIR_DLINE >= 16: the segment budget in MB, for sweeps)
  int64_t seg = ((int64_t)seg_mb << 20) / ((int64_t)s.batch * (fused ? 8192 : 16384));
  if (seg < 64) seg = 64;
  if (seg > nblk_total) seg = nblk_total;
  void* scratch = nullptr;
  const size_t zbytes = (size_t)s.batch * (size_t)(seg + P - 1) * 8192, wbytes = fused ? 0 : (size_t)s.batch * (size_t)seg * 8192;
  if ((rc = ctx_scratch(c, 21, zbytes + wbytes, &scratch))) return rc;
  a.Z = reinterpret_cast<v2f*>(scratch);
  a.Wt = reinterpret_cast<v2f*>(static_cast<char*>(scratch) + zbytes);
  constexpr int W = 4;
  const size_t lds = (size_t)(256 + 4 * 256) * 8 + (size_t)W * (1024 + 64 + 16) * 8;
  for (int64_t kf = 0; kf < nblk_total; kf += seg) {
    a.k_first = kf;
    a.nblk = nblk_total - kf < seg ? nblk_total - kf : seg;
    a.m_first = kf - (P - 1);
    a.nwin = a.nblk + P - 1;
    {
      const int64_t total = (int64_t)s.batch * a.nwin;
      int64_t blocks = (total + W - 1) / W;
      if (blocks > (int64_t)c->num_cus * 64) blocks = (int64_t)c->num_cus * 64;
      hipLaunchKernelGGL(k_fir_dline_fwd<W>, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a, total);
      NXSIG_HIP_TRY(hipGetLastError());
    }
    if (fused) {
      constexpr int W8 = 8;
      const size_t lds8 = (size_t)(256 + 4 * 256) * 8 + (size_t)4 * 1024 * 16 + (size_t)W8 * (1024 + 64 + 16) * 8;
      const int64_t total = (int64_t)s.batch * a.nblk;
      const int bpw = 4;
      const int64_t blocks = (total + (int64_t)W8 * bpw - 1) / ((int64_t)W8 * bpw);
      if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "fir: signal too long for one launch");
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fir_dline_macinv<W8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
      dispatch_note("fir.dline.fused");
      hipLaunchKernelGGL((k_fir_dline_macinv<W8, 4>), dim3((unsigned)blocks), dim3(64 * W8), lds8, c->stream, a, total, bpw);
      NXSIG_HIP_TRY(hipGetLastError());
      continue;
    }
    // runs: enough (row, run) pairs to fill the chip, each at least 2 P blocks long (a run re-reads P - 1 windows to warm up)
    int64_t run_len = ((int64_t)s.batch * a.nblk + c->num_cus - 1) / c->num_cus;
    if (run_len < 2 * P) run_len = 2 * P;
    if (run_len > a.nblk) run_len = a.nblk;
    const int64_t runs_per_row = (a.nblk + run_len - 1) / run_len;
    const int64_t mac_blocks = (int64_t)s.batch * runs_per_row * 4;
    if (mac_blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "fir: too many rows for one launch");
    for (int p_lo = 0; p_lo < P; p_lo += 16) {
      const int np = P - p_lo < 16 ? P - p_lo : 16;
      const int acc = p_lo > 0 ? 1 : 0;
      if (np <= 4) hipLaunchKernelGGL(k_fir_dline_mac<4>, dim3((unsigned)mac_blocks), dim3(256), 0, c->stream, a, p_lo, np, acc, run_len, runs_per_row);
      else if (np <= 8) hipLaunchKernelGGL(k_fir_dline_mac<8>, dim3((unsigned)mac_blocks), dim3(256), 0, c->stream, a, p_lo, np, acc, run_len, runs_per_row);
      else hipLaunchKernelGGL(k_fir_dline_mac<16>, dim3((unsigned)mac_blocks), dim3(256), 0, c->stream, a, p_lo, np, acc, run_len, runs_per_row);
      NXSIG_HIP_TRY(hipGetLastError());
    }
    {
      const int64_t total = (int64_t)s.batch * a.nblk;
      int64_t blocks = (total + W - 1) / W;
      if (blocks > (int64_t)c->num_cus * 64) blocks = (int64_t)c->num_cus * 64;
      hipLaunchKernelGGL(k_fir_dline_inv<W>, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a, total);
      NXSIG_HIP_TRY(hipGetLastError());
    }
  }
  return NXSIG_OK;
}

}  // namespace nxsig
