#include "wave_stft.hpp"
namespace nxsig {
struct IstftWaveArgs {
  const v2f* z;               // c64[batch][M][K]
  int64_t M;
  int32_t batch, hop;
  int64_t segs_per_row;       // M + R - 1  (out_len = segs_per_row * hop)
  int64_t run_len, runs_per_row, total_runs;
  const float* wtab;          // f32[K]
  const v2f* twB;
  const v2f* twC;
  const v2f* twH;             // w_K^k0, k0 < K/2 (two-frames-per-FFT variant only)
  float scale;
  const float* den;           // f32[2R-1][hop]: RECIPROCAL of the guarded OLA normaliser: head segments 0..R-2, interior, tail segments
  v2f* y;                     // c64[batch][segs_per_row * hop]
  v2f* dummy;
  const v2f* filt = nullptr;  // c64[K] spectrum filter (FILT variant of k_istft_wave only)
  const v2f* zeros = nullptr; // c64[K] of zeros: the spectrum the tail-flush frames m >= M of k_istft_wave read (their samples are then
                              // exactly zero and no per-sample `live` factor is needed)
  int* nf_list = nullptr;     // kernels that invert several frames per transform: units that hold a non-finite bin are reported here
                              // ({count, capacity, int64 (row << 40 | first frame) ...}) and redone frame by frame by k_istft_nf_fix
};
// ---- iSTFT for N = 2K (2048), round 5: decimation in TIME.  x[2n'] = IDFT_K(Z[k] + Z[k + K])[n'],
// x[2n' + 1] = IDFT_K((Z[k] - Z[k + K]) e^{+2 pi i k / 2K})[n']: the split sits on the INPUT side (lane-local: a lane loads Z[k] and
// Z[k + K] for k = lane + 64 s), so each core's output is final samples — the even ones, then the odd ones — and is windowed and folded
// into the pending overlap sums as soon as its core is done.  Against k_istft_wave_dbl (decimation in frequency: both cores' outputs
// alive at once for the x = E +- t O combine): the peak register demand drops (the next frame's loads are issued half before each
// core), the window travels as conflict-free 16-byte LDS reads (a lane owns samples 4 lane + {0..3} + 256 q: one quad per q) and the
// half-twiddles as conflict-free 8-byte reads — k_istft_wave_dbl's stride-2 table reads spent 15 % of its LDS cycles in bank
// conflicts (profiles/r05/istft2048_sq_counters.txt: 1.49 x the LDS pipeline cycles of the N = 1024 kernel for the same bytes).
// hop must be a multiple of 256 (R = N / hop in {1, 2, 4, 8}); hop = 128 keeps k_istft_wave_dbl.
template <int K, int R, bool SCALE, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_istft_wave_dit(IstftWaveArgs a) {
  constexpr int N2 = 2 * K;
  constexpr int P = K / 64;
  constexpr int R3 = K / 256;
  constexpr int NQ = K / 128;
  constexpr int QS = NQ / R;             // 256-sample slots per hop segment
  constexpr int XCH = K + K / 16 + 16;
  static_assert(NQ % R == 0 && QS >= 1, "hop must be a multiple of 256");
  v4f* s_w4 = reinterpret_cast<v4f*>(g_wave_smem);      // window quads: s_w4[i] = w[4 i .. 4 i + 3]
  v2f* s_twB = reinterpret_cast<v2f*>(s_w4 + N2 / 4);
  v2f* s_twC = s_twB + 256;
  v2f* s_twH = s_twC + R3 * 256;          // exp(+2 pi i k / 2K), k < K
  v2f* s_x = s_twH + K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < N2 / 4; i += 64 * W) s_w4[i] = reinterpret_cast<const v4f*>(a.wtab)[i];
  for (int i = tid; i < K; i += 64 * W) s_twH[i] = a.twH[i];
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < R3 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  const int64_t run = (int64_t)blockIdx.x * W + wave;
  if (run >= a.total_runs) return;
  const int64_t row = run / a.runs_per_row;
  const int64_t j0 = (run - row * a.runs_per_row) * a.run_len;
  int64_t j1 = j0 + a.run_len;
  if (j1 > a.segs_per_row) j1 = a.segs_per_row;
  const int64_t m_start = j0 >= (R - 1) ? j0 - (R - 1) : 0;
  const float invN = 1.0f / (float)N2;
  // pending overlap sums: [segment][core: even / odd samples][parity of the core's output][slot]
  v2f pend[R - 1 > 0 ? R - 1 : 1][2][2][QS];
#pragma unroll
  for (int i = 0; i < R - 1; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int qq = 0; qq < QS; ++qq) pend[i][c][e][qq] = v2f{0.f, 0.f};

  const v2f* zrow = a.z + (size_t)row * a.M * N2 + lane;
  v2f na[P], nb[P];   // Z[k], Z[k + K] of the frame in flight, k = lane + 64 s (non-temporal: the spectrogram is read once)
  auto issue_lo = [&](int64_t m) {
    const v2f* pz = zrow + (size_t)(m < a.M ? m : a.M - 1) * N2;
#pragma unroll
    for (int s = 0; s < P; ++s) na[s] = __builtin_nontemporal_load(pz + 64 * s);
  };
  auto issue_hi = [&](int64_t m) {
    const v2f* pz = zrow + (size_t)(m < a.M ? m : a.M - 1) * N2 + K;
#pragma unroll
    for (int s = 0; s < P; ++s) nb[s] = __builtin_nontemporal_load(pz + 64 * s);
  };
  issue_lo(m_start);
  issue_hi(m_start);
  for (int64_t m = m_start; m < j1; ++m) {
    // ---- input split (lane-local): d0 = Z[k] + Z[k + K] -> even samples, d1 = (Z[k] - Z[k + K]) t[k] -> odd samples
    v2f d0[P], d1[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
      d0[s] = na[s] + nb[s];
      d1[s] = wcmul(na[s] - nb[s], s_twH[lane + 64 * s]);
    }
    const int64_t mn = m + 1 < j1 ? m + 1 : m;   // unconditional prefetch keeps the loop branch-free
    __builtin_amdgcn_sched_barrier(0);
    const float live = m < a.M ? 1.0f : 0.0f;   // tail flush: frames m >= M do not exist
    const int64_t j = m;                        // segment j is complete once frame j has been folded in
    v2f oute[2][QS];                            // the finished segment's even samples wait for the odd ones
    // one core's samples ((IDFT / N) * scale) * window folded into the pending sums in ascending frame order (lib/nx_signal.ex:609-628)
    auto fold = [&](const int c, v2f (*zz)[NQ], v2f (*out)[QS]) {
#pragma unroll
      for (int qq = 0; qq < QS; ++qq) {
        v2f f[R][2];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const v4f wq = s_w4[lane + 64 * (i * QS + qq)] * live;   // w[4 lane + {0..3} + 256 q]: (even p0, odd p0, even p1, odd p1)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            v2f v = fft_eps0(zz[e][i * QS + qq] * invN);            // Nx.ifft's clean-up (:609) precedes scale and window
            if (SCALE) v = v * a.scale;
            f[i][e] = v * (c == 0 ? (e == 0 ? wq.x : wq.z) : (e == 0 ? wq.y : wq.w));
          }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (R == 1) { out[e][qq] = f[0][e]; }
          else {
            out[e][qq] = pend[0][c][e][qq] + f[0][e];
#pragma unroll
            for (int i = 0; i + 1 < R - 1; ++i) pend[i][c][e][qq] = pend[i + 1][c][e][qq] + f[i + 1][e];
            pend[R - 2][c][e][qq] = f[R - 1][e];
          }
        }
      }
    };
    {
      v2f ze[2][NQ];
      wave_fft_core<K, true>(d0, ze, xb, s_twB, s_twC, lane);
      __builtin_amdgcn_sched_barrier(0);
      fold(0, ze, oute);
    }
    issue_lo(mn);   // the next frame's 16 KB travel during the second core and the stores (both halves here: issued before the first
    issue_hi(mn);   // core they cost 32 registers more than two waves per SIMD leave: 124 B of scratch per lane)
    __builtin_amdgcn_sched_barrier(0);
    v2f outo[2][QS];
    {
      v2f zo[2][NQ];
      wave_fft_core<K, true>(d1, zo, xb, s_twB, s_twC, lane);
      __builtin_amdgcn_sched_barrier(0);
      fold(1, zo, outo);
    }
    // guarded normaliser of segment j (reciprocals; head rows 0..R-2, interior row R-1, tail rows R..2R-2), 16-byte stores: a lane
    // owns samples 4 lane + {0, 1} and + {2, 3} of every 256-sample slot
    const int64_t trow = j < R - 1 ? j : (j >= a.M ? R + (j - a.M) : R - 1);
    const float* dp = a.den + trow * a.hop + 4 * lane;
    v2f* yp = (j >= j0) ? a.y + (size_t)row * a.segs_per_row * a.hop + j * a.hop + 4 * lane : a.dummy + 4 * lane;
#pragma unroll
    for (int qq = 0; qq < QS; ++qq) {
      const v4f rd = *reinterpret_cast<const v4f*>(dp + 256 * qq);
      const v4f o0 = v4f{oute[0][qq].x * rd.x, oute[0][qq].y * rd.x, outo[0][qq].x * rd.y, outo[0][qq].y * rd.y};
      const v4f o1 = v4f{oute[1][qq].x * rd.z, oute[1][qq].y * rd.z, outo[1][qq].x * rd.w, outo[1][qq].y * rd.w};
      __builtin_nontemporal_store(o0, (gv4f*)(yp + 256 * qq));
      __builtin_nontemporal_store(o1, (gv4f*)(yp + 256 * qq + 2));
    }
  }
}


template __global__ void k_istft_wave_dit<1024, 4, false, 4>(IstftWaveArgs);
}
