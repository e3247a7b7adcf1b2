// Wave-private FFT kernels: the one-sided (Hermitian, "packed") inverse STFT — opt-in extension, SURVEY 8f-2 / 8f-3.
//
// The reference's STFT-domain filtering chain is stft -> z * H -> istft (guides/filtering.livemd:137-159) on the FULL two-sided
// spectrum, 8 KB + 2 KB of HBM traffic per frame at fft_length 1024 although the signal is real: half of z mirrors the other
// half and the imaginary part of the result is round-off.  The packed chain keeps what is independent:
//   nxsig_stft_packed_f32    c64[batch][M][K/2]: bins 0 .. K/2 - 1, with Re X[K/2] (the Nyquist bin, real for a real frame)
//                            riding in the imaginary part of bin 0 (which is zero for a real frame)      -> 4 KB per frame
//   nxsig_istft_packed_f32   the inverse of exactly that layout, REAL f32 output                            -> 1 KB per frame
// equal to Nx.real(NxSignal.istft(full Hermitian spectrum)) to fp32 round-off (tests/test_gpu_packed.py).
//
// k_istft_packed<R>: N = fft_length = 1024, hop = 1024 / R a multiple of 128.  A real 1024-point inverse transform is ONE
// 512-point complex inverse transform of Z[k] = Xe[k] + i Xo[k], Xe = (X[k] + conj X[512 - k]) / 2 (spectrum of the even
// samples), Xo = (X[k] - conj X[512 - k]) / 2 * e^(+2 pi i k / 1024) (of the odd samples): z[m] = x[2m] + i x[2m + 1].  Two
// frames share the 1024-point core exactly as in k_istft_wave_half (kernels_wave.hip): Y[k0] = Z0 + w^k0 Z1, Y[k0 + 512] =
// Z0 - w^k0 Z1, and the core returns z_f[n] for n = lane + 64 q in zz[f][q] — a PAIR of adjacent real samples per complex
// value, so the overlap-add (hop / 2 pairs per segment, a multiple of 64) stays in registers and every store is 8 bytes of
// two adjacent real samples.  The partner bin X[512 - k0] of k0 = lane + 64 s lives on lane 64 - lane (ds_bpermute).
#include "wave_stft.hpp"

namespace nxsig {

struct IstftPackedArgs {
  const v2f* z;               // c64[batch][M][512] packed
  int64_t M;
  int32_t batch, hop;
  int64_t segs_per_row;       // M + R - 1  (out_len = segs_per_row * hop)
  int64_t run_len, runs_per_row, total_runs;
  const float* wtab;          // f32[1024]
  const v2f* twB;             // conjugated tables: the core runs in inverse direction
  const v2f* twC;
  const v2f* twH;             // w_1024^k0 = exp(-2 pi i k0 / 1024), k0 < 512
  float scale;
  const float* den;           // f32[2R-1][hop]: reciprocal of the guarded OLA normaliser (istft_den_table)
  float* y;                   // f32[batch][segs_per_row * hop]
  float* dummy;
  int* nf_list;               // units (frame pairs) that hold a non-finite bin: redone frame by frame by k_istft_nf_fix
};

template <int R, bool SCALE, int W>
__global__ __launch_bounds__(64 * W) void k_istft_packed(IstftPackedArgs a) {
  constexpr int K = 1024, NH = 512, R3 = 4, NQ = 8, QS = NQ / R, XCH = K + K / 16 + 16;
  static_assert(NQ % R == 0, "hop must be a multiple of 128");
  float* s_w = reinterpret_cast<float*>(g_wave_smem);       // window, 1024 real values
  v2f* s_twB = reinterpret_cast<v2f*>(s_w + K);
  v2f* s_twC = s_twB + 256;
  v2f* s_twH = s_twC + R3 * 256;
  v2f* s_x = s_twH + NH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < K; i += 64 * W) s_w[i] = a.wtab[i];
  for (int i = tid; i < NH; i += 64 * W) s_twH[i] = a.twH[i];
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < R3 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  const int64_t run = (int64_t)blockIdx.x * W + wave;
  if (run >= a.total_runs) return;
  const int64_t row = run / a.runs_per_row;
  const int64_t j0 = (run - row * a.runs_per_row) * a.run_len;  // run_len is even
  int64_t j1 = j0 + a.run_len;
  if (j1 > a.segs_per_row) j1 = a.segs_per_row;
  int64_t m_start = j0 >= (R - 1) ? j0 - (R - 1) : 0;
  m_start &= ~(int64_t)1;                 // frame pairs start at even frames

  v2f wv[NQ], tw[NQ];                     // (w[2n], w[2n + 1]) and w_1024^k0 for n = k0 = lane + 64 q
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    wv[q] = *reinterpret_cast<const v2f*>(&s_w[2 * (lane + 64 * q)]);
    tw[q] = s_twH[lane + 64 * q];
  }
  const float invK = 1.0f / (float)K;
  v2f pend[R - 1 > 0 ? R - 1 : 1][QS];
#pragma unroll
  for (int i = 0; i < R - 1; ++i)
#pragma unroll
    for (int qq = 0; qq < QS; ++qq) pend[i][qq] = v2f{0.f, 0.f};

  const v2f* zrow = a.z + (size_t)row * a.M * NH + lane;
  v2f r0[NQ], r1[NQ];
  auto issue_loads = [&](int64_t m) {
    const int64_t last = a.M - 1;
    const v2f* p0 = zrow + (size_t)(m < last ? m : last) * NH;
    const v2f* p1 = zrow + (size_t)(m + 1 < last ? m + 1 : last) * NH;
#pragma unroll
    for (int q = 0; q < NQ; ++q) { r0[q] = __builtin_nontemporal_load(p0 + 64 * q); r1[q] = __builtin_nontemporal_load(p1 + 64 * q); }
  };
  // Z[k0] = Xe + i Xo of one frame from its packed half spectrum (registers r[s] = X[lane + 64 s])
  const int src = ((64 - lane) & 63) << 2;
  auto fold = [&](const v2f* r, v2f* zf) {
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
      // partner X[512 - k0]: lane >= 1 -> element 7 - s of lane 64 - lane; lane 0 -> own element 8 - s (s = 0: the Nyquist bin)
      v2f p;
      p.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(r[NQ - 1 - s].x)));
      p.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(r[NQ - 1 - s].y)));
      v2f x = r[s];
      if (lane == 0) {
        if (s == 0) { p = v2f{r[0].y, 0.f}; x = v2f{r[0].x, 0.f}; }   // bin 0 carries Re X[512] in its imaginary part
        else p = r[NQ - s];
      }
      const v2f xe = v2f{x.x + p.x, x.y - p.y} * 0.5f;                // (X + conj P) / 2
      const v2f xo = wcmul_conj(v2f{x.x - p.x, x.y + p.y} * 0.5f, tw[s]);   // (X - conj P) / 2 * e^(+2 pi i k0 / 1024)
      zf[s] = v2f{xe.x - xo.y, xe.y + xo.x};                          // Xe + i Xo
    }
  };
  v2f d[2 * NQ];
  bool nf_next = false;
  auto combine = [&]() {
    v2f z0[NQ], z1[NQ];
    fold(r0, z0);
    fold(r1, z1);
    v2f sum = v2f{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const v2f t = wcmul(z1[q], tw[q]);
      d[q] = z0[q] + t;
      d[q + NQ] = z0[q] - t;
      sum += d[q];
    }
    nf_next = wave_any_nonfinite(sum.x, sum.y);
  };
  issue_loads(m_start);
  combine();

  for (int64_t m = m_start; m < j1; m += 2) {
    // a pair that holds a non-finite bin shares it between its two frames; the reference inverts frame by frame (:609)
    if (__builtin_expect(nf_next, 0) && lane == 0) {
      const int i = atomicAdd(a.nf_list, 1);
      if (i < a.nf_list[1]) reinterpret_cast<int64_t*>(a.nf_list + 2)[i] = (row << 40) | m;
    }
    issue_loads(m + 2 < j1 ? m + 2 : m);
    __builtin_amdgcn_sched_barrier(0);
    v2f zz[2][NQ];
    wave_fft_core<K, true>(d, zz, xb, s_twB, s_twC, lane);
    __builtin_amdgcn_sched_barrier(0);
    combine();
    __builtin_amdgcn_sched_barrier(0);
    ifft_eps_cold<NQ>(zz, kFftEps * (float)K);   // Nx.ifft's clean-up (:609), cold form: wave_stft.hpp
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t j = m + e;              // frame index = index of the segment it completes
      const float live = j < a.M ? 1.0f : 0.0f;
      // (the phantom second half of an odd last pair, j == segs_per_row, is never stored: keep its table row inside the table)
      const int64_t jt = j < a.segs_per_row ? j : a.segs_per_row - 1;
      const int64_t trow = jt < R - 1 ? jt : (jt >= a.M ? R + (jt - a.M) : R - 1);
      const float* dp = a.den + trow * a.hop + 2 * lane;
      const bool store = (j >= j0) && (j < j1);
      float* yp = store ? a.y + (size_t)row * a.segs_per_row * a.hop + j * a.hop + 2 * lane : a.dummy + 2 * lane;
#pragma unroll
      for (int qq = 0; qq < QS; ++qq) {
        v2f f[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          v2f v = zz[e][i * QS + qq] * invK;
          if (SCALE) v = v * a.scale;
          f[i] = v * (wv[i * QS + qq] * live);
        }
        v2f out;
        if (R == 1) { out = f[0]; }
        else {
          out = pend[0][qq] + f[0];
#pragma unroll
          for (int i = 0; i + 1 < R - 1; ++i) pend[i][qq] = pend[i + 1][qq] + f[i + 1];
          pend[R - 2][qq] = f[R - 1];
        }
        const v2f rd = *reinterpret_cast<const v2f*>(dp + 128 * qq);
        __builtin_nontemporal_store(out * rd, (gv2f*)(yp + 128 * qq));
      }
    }
  }
}

int istft_den_table(Ctx* c, int R, int hop, const float* window_host, const float** out);   // kernels_wave.hip
int istft_nf_list(Ctx* c, int64_t capacity, int** list);                                    // kernels_generic.hip

template <int R>
static int launch_istft_packed_R(Ctx* c, const IstftLaunch& s, const float* window_host) {
  constexpr int K = 1024, W = 4, R3 = 4, XCH = K + K / 16 + 16;
  IstftPackedArgs a;
  a.z = reinterpret_cast<const v2f*>(s.z); a.M = s.M; a.batch = s.batch; a.hop = s.hop;
  a.segs_per_row = s.M + R - 1;
  a.wtab = s.window;
  Ctx::WaveTables& wt = c->wave_tables[K];
  a.twB = reinterpret_cast<const v2f*>(wt.twBi);
  a.twC = reinterpret_cast<const v2f*>(wt.twCi);
  a.scale = s.scale_mul;
  { int rc = istft_den_table(c, R, s.hop, window_host, &a.den); if (rc) return rc; }
  a.y = reinterpret_cast<float*>(s.y);
  void* dummy = nullptr;
  { int rc = ctx_scratch(c, 3, (size_t)8192 * sizeof(float2), &dummy); if (rc) return rc; }
  a.dummy = reinterpret_cast<float*>(dummy);
  {
    const void* dh = nullptr;
    auto hit = c->memo.find(0x774800000000ull ^ (uint64_t)K);   // shared with launch_istft_wave_R's pair kernel: built once per context
    if (hit != c->memo.end()) dh = reinterpret_cast<const void*>(hit->second[0]);
    else {
      std::vector<float2> twH((size_t)K / 2);
      for (int k0 = 0; k0 < K / 2; ++k0) {
        const double ang = -6.283185307179586476925286766559 * (double)k0 / (double)K;
        twH[k0] = make_float2((float)std::cos(ang), (float)std::sin(ang));
      }
      int rc = ctx_table(c, 0x7748ull ^ (uint64_t)K, twH.data(), twH.size() * sizeof(float2), &dh);
      if (rc) return rc;
      c->memo[0x774800000000ull ^ (uint64_t)K] = {reinterpret_cast<uint64_t>(dh)};
    }
    a.twH = reinterpret_cast<const v2f*>(dh);
  }
  const int64_t total_segs = a.segs_per_row * s.batch;
  const int waves_per_cu = tune(c, kT_ISTFT_RUNS_PER_CU, 12);
  int64_t run_len = (total_segs + (int64_t)c->num_cus * waves_per_cu - 1) / ((int64_t)c->num_cus * waves_per_cu);
  if (run_len < 8) run_len = 8;
  run_len = (run_len + 1) & ~(int64_t)1;
  a.run_len = run_len;
  a.runs_per_row = (a.segs_per_row + run_len - 1) / run_len;
  a.total_runs = a.runs_per_row * s.batch;
  const int64_t blocks = (a.total_runs + W - 1) / W;
  { int rc = istft_nf_list(c, a.total_runs * ((run_len + R + 1) / 2 + 1), &a.nf_list); if (rc) return rc; }
  s.nf_list = a.nf_list; s.nf_frames_per_unit = 2;
  const size_t lds = (size_t)K * 4 + 256 * 8 + (size_t)R3 * 256 * 8 + (size_t)(K / 2) * 8 + (size_t)W * XCH * 8;
  dispatch_note("istft.packed");
  if (s.has_scale) hipLaunchKernelGGL((k_istft_packed<R, true, W>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
  else hipLaunchKernelGGL((k_istft_packed<R, false, W>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// s.onesided: z = packed c64[batch][M][K/2], y = f32[batch][out_len].  Fused for N = fft_length = 1024 and hop = 128 ... 1024.
int launch_istft_packed_wave(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled) {
  *handled = false;
  if (s.M == 0 || s.batch == 0 || window_host == nullptr || s.filt) return NXSIG_OK;
  if (tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  if (s.K != 1024 || s.N != 1024) return NXSIG_OK;
  if (s.hop != 128 && s.hop != 256 && s.hop != 512 && s.hop != 1024) return NXSIG_OK;
  if (s.M < 2 * (1024 / s.hop) - 1) return NXSIG_OK;
  int rc = ensure_wave_tables_1024(c);
  if (rc) return rc;
  *handled = true;
  switch (1024 / s.hop) {
    case 1: return launch_istft_packed_R<1>(c, s, window_host);
    case 2: return launch_istft_packed_R<2>(c, s, window_host);
    case 4: return launch_istft_packed_R<4>(c, s, window_host);
    default: return launch_istft_packed_R<8>(c, s, window_host);
  }
}

}  // namespace nxsig
