// Wave-private FFT kernels, part 1 of 3: iSTFT and overlap-save FIR kernels, and the launchers of the plain STFT
// (complex spectrum sink).  The device code shared with the log-mel and magnitude translation units — the FFT cores,
// the STFT body with its sinks, the Bluestein kernel and the launch templates — lives in wave_stft.hpp.
#include "wave_stft.hpp"

namespace nxsig {

// ============================================================================================ iSTFT
// NxSignal.istft/3 fused in one launch (lib/nx_signal.ex:609-637): one wave walks a RUN of consecutive frames of one
// row.  Per frame: c64 load -> inverse FFT (the core with conjugated twiddles, x 1/K) -> x scale x window -> the frame's
// R = N/hop hop-sized segments are folded into R-1 pending accumulators held in registers, always in ascending
// frame order (deterministic: no atomics, run-to-run bit-stable) -> the finished segment is divided by the OLA
// normaliser sum |w|^2 (guard 1e-10 -> 1, :635) and stored as c64 with 16-byte stores.  The lane layout of the
// core (sample n = 2 lane + par + 128 q) keeps a lane's position inside every hop segment identical, so the
// overlap-add needs no data movement at all.  Runs start R-1 frames early to rebuild their pending sums (halo
// recompute) instead of exchanging partial sums between waves.
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
__device__ __forceinline__ void istft_report_nonfinite(int* list, int64_t row, int64_t first_frame) {
  const int i = atomicAdd(list, 1);
  if (i < list[1]) reinterpret_cast<int64_t*>(list + 2)[i] = (row << 40) | first_frame;
}

// c64 product the way Nx.multiply forms it on the BinaryBackend: in double, each component rounded once (same expression as
// k_spectrum_mul, so the fused and the two-step chain agree bit for bit)
__device__ __forceinline__ v2f cmul_c64_rounded(v2f a, v2f b) {
  const double re = (double)a.x * (double)b.x - (double)a.y * (double)b.y;
  const double im = (double)a.x * (double)b.y + (double)a.y * (double)b.x;
  return v2f{(float)re, (float)im};
}

// FILT: every frame's spectrum is multiplied by a.filt as it arrives from HBM (the z * H step of STFT-domain filtering,
// guides/filtering.livemd:141): a lane always loads the same 16 bins, so its 16 filter values sit in registers and the separate
// read-modify-write pass over the spectrogram (16 KB of HBM traffic per frame on top of this kernel's 10) disappears.  (Keeping the
// table in LDS to stay at 3 waves per SIMD was measured too: it spills 25 registers and runs 30 % slower than this form.)
// DEEP: the spectrum is prefetched TWO frames ahead into two register sets that trade roles every iteration (the loop is unrolled
// by two): 32 more registers, i.e. two waves per SIMD instead of three, for twice the bytes in flight per wave
template <int K, int R, bool SCALE, int W, bool FILT = false, bool NTL = false, bool DEEP = false>   // NTL: non-temporal loads of the spectrogram
__global__ __launch_bounds__(64 * W) void k_istft_wave(IstftWaveArgs a) {
  constexpr int P = K / 64;
  constexpr int R3 = K / 256;
  constexpr int NQ = K / 128;
  constexpr int QS = NQ / R;            // q values (of 128 samples each) per hop segment, per parity
  constexpr int XCH = K + K / 16 + 16;
  static_assert(NQ % R == 0, "hop must be a multiple of 128");
  float* s_w = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_twB = reinterpret_cast<v2f*>(s_w + K);
  v2f* s_twC = s_twB + 256;
  v2f* s_x = s_twC + R3 * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < K; i += 64 * W) s_w[i] = a.wtab[i];
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < R3 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  const int64_t run = (int64_t)blockIdx.x * W + wave;
  if (run >= a.total_runs) return;  // whole wave leaves; no barrier follows
  const int64_t row = run / a.runs_per_row;
  const int64_t j0 = (run - row * a.runs_per_row) * a.run_len;
  int64_t j1 = j0 + a.run_len;
  if (j1 > a.segs_per_row) j1 = a.segs_per_row;
  const int64_t m_start = j0 >= (R - 1) ? j0 - (R - 1) : 0;   // the run's R - 1 halo frames are recomputed (hand-over bound: +7 %)

  // this lane's window values wv[par][q] = w[2 lane + par + 128 q]
  float wv[2][NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const v2f w = *reinterpret_cast<const v2f*>(&s_w[2 * lane + 128 * q]);
    wv[0][q] = w.x; wv[1][q] = w.y;
  }
  const float invK = 1.0f / (float)K;
  v2f pend[R - 1 > 0 ? R - 1 : 1][2][QS];
#pragma unroll
  for (int i = 0; i < R - 1; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int qq = 0; qq < QS; ++qq) pend[i][e][qq] = v2f{0.f, 0.f};

  const v2f* zrow = a.z + (size_t)row * a.M * K + lane;
  v2f r[P], r2[DEEP ? P : 1];
  auto issue_into = [&](v2f* dst, int64_t m) {
    const v2f* pz = m < a.M ? zrow + (size_t)m * K : a.zeros + lane;  // frames past the end (tail flush) are a spectrum of zeros
#pragma unroll
    for (int s = 0; s < P; ++s) dst[s] = NTL ? __builtin_nontemporal_load(pz + 64 * s) : pz[64 * s];
  };
  v2f hv[FILT ? P : 1];
  if (FILT) {
#pragma unroll
    for (int s = 0; s < P; ++s) hv[FILT ? s : 0] = a.filt[lane + 64 * s];
  }
  v2f d[P];
  auto take_from = [&](const v2f* src) {   // the prefetched spectrum (times the filter) becomes the core's input
#pragma unroll
    for (int s = 0; s < P; ++s) {
      d[s] = FILT ? cmul_c64_rounded(src[s], hv[FILT ? s : 0]) : src[s];
    }
  };
  if (DEEP) {
    issue_into(r2, m_start);
    issue_into(r, m_start + 1 < j1 ? m_start + 1 : m_start);
    take_from(r2);
  } else {
    issue_into(r, m_start);
    take_from(r);
  }

  // one frame: rn holds (or is receiving) frame m + 1, rf is free and receives the frame after it
  auto body = [&](const int64_t m, v2f* rn, v2f* rf) {
    if (DEEP) issue_into(rf, m + 2 < j1 ? m + 2 : m);
    else issue_into(rf, m + 1 < j1 ? m + 1 : m);  // unconditional prefetch keeps the loop branch-free
    __builtin_amdgcn_sched_barrier(0);
    v2f zz[2][NQ];
    wave_fft_core<K, true>(d, zz, xb, s_twB, s_twC, lane);  // inverse direction (tables are conjugated)
    __builtin_amdgcn_sched_barrier(0);
    take_from(rn);
    __builtin_amdgcn_sched_barrier(0);

    // (The clean-up stays EAGER here — a compare and a select per component: the cold form of the other inverse kernels puts a
    // wave-uniform branch into this loop, and the two-frames-ahead schedule lost more to the split scheduling region than the 33
    // instructions per frame were worth: -3.5 % against the round-3 build side by side, profiles/r04/ab_libs_round3_vs_round4.jsonl.
    // The tail flush needs no factor: frames m >= M were loaded from a spectrum of zeros.)
    const int64_t j = m;                        // segment j is complete once frame j has been folded in
    // guarded normaliser of segment j from the host table (head rows 0..R-2, interior row R-1, tail rows R..2R-2)
    const int64_t trow = j < R - 1 ? j : (j >= a.M ? R + (j - a.M) : R - 1);
    const float* dp = a.den + trow * a.hop + 2 * lane;
    v2f den[QS];
#pragma unroll
    for (int qq = 0; qq < QS; ++qq) den[qq] = *reinterpret_cast<const v2f*>(dp + 128 * qq);
    // frame samples ((IDFT / K) * scale) * window (lib/nx_signal.ex:609-628, same rounding order) are folded
    // straight into the pending overlap sums, always in ascending frame order
    v2f out[2][QS];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int qq = 0; qq < QS; ++qq) {
        v2f f[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          v2f v = fft_eps0(zz[e][i * QS + qq] * invK);  // Nx.ifft's clean-up (:609) precedes scale and window
          if (SCALE) v = v * a.scale;
          f[i] = v * wv[e][i * QS + qq];
        }
        if (R == 1) { out[e][qq] = f[0]; }
        else {
          out[e][qq] = pend[0][e][qq] + f[0];
#pragma unroll
          for (int i = 0; i + 1 < R - 1; ++i) pend[i][e][qq] = pend[i + 1][e][qq] + f[i + 1];
          pend[R - 2][e][qq] = f[R - 1];
        }
      }
    // plain non-temporal stores here: the "sc1 nt" policy that helps the write-dominated STFT (+5 %) and the FIR (+3 %) costs
    // this read-dominated kernel 38 % (395 -> 546 us on config 3, measured)
    v2f* yp = (j >= j0) ? a.y + (size_t)row * a.segs_per_row * a.hop + j * a.hop + 2 * lane : a.dummy + 2 * lane;
#pragma unroll
    for (int qq = 0; qq < QS; ++qq) {
      // the table holds 1 / max-guarded normaliser (rounded from double): a multiply instead of 4 divisions, <= 1 ulp
      const v4f o = v4f{out[0][qq].x * den[qq].x, out[0][qq].y * den[qq].x, out[1][qq].x * den[qq].y, out[1][qq].y * den[qq].y};
      __builtin_nontemporal_store(o, (gv4f*)(yp + 128 * qq));
    }
  };
  if (DEEP) {
    for (int64_t m = m_start; m < j1; m += 2) {
      body(m, r, r2);
      if (m + 1 < j1) body(m + 1, r2, r);
    }
  } else {
    for (int64_t m = m_start; m < j1; ++m) body(m, r, r);
  }
}

// ---- iSTFT for N = 4096 (hop 512 / 1024 / 2048 / 4096): FOUR passes through the 1024-point inverse core per frame.
// x[n0 + 1024 m] = 1/4096 sum_r conj(w_4096^(r n0)) w_4^(-r m) Y_r[n0],  Y_r = IDFT_1024(Z[4 k' + r]) (unscaled): the lane reads
// Z[4 (l + 64 s) .. + 3] as two 16-byte loads (all four decimated sub-spectra at once: every byte of the frame is loaded
// exactly once), runs the core four times and combines lane-locally.  Sample n0 + 1024 m sits on the lane that owns n0 = 2 l +
// par + 128 q, so a lane's positions inside every hop segment are the same and the overlap-add stays in registers like in
// k_istft_wave.  The next frame's 32 loads are in flight during the four cores; the frame data (128), the prefetch (128) and the
// pending sums (up to 112) exceed 256 registers, so the kernel runs one wave per SIMD (4 waves per workgroup) and the
// compiler parks the surplus in the accumulation half of the unified register file.
template <int R, bool SCALE, int W>
__global__ __launch_bounds__(64 * W) void k_istft_wave_4k(IstftWaveArgs a) {
  constexpr int K = 1024, NQ = 8, P = 16, NF = 4096, XCH = K + K / 16 + 16;
  constexpr int HOPC = NF / R;           // hop (compile-time: 512 .. 4096)
  constexpr int SL = HOPC / 128;         // 128-sample slots per hop segment (per parity)
  v2f* s_wv = reinterpret_cast<v2f*>(g_wave_smem);   // window as adjacent pairs: s_wv[i] = (w[2 i], w[2 i + 1])
  v2f* s_twB = s_wv + NF / 2;
  v2f* s_twC = s_twB + 256;
  v2f* s_t4 = s_twC + 4 * 256;           // conj(w_4096^k), k < 1024
  v2f* s_x = s_t4 + K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NF / 2; i += 64 * W) s_wv[i] = reinterpret_cast<const v2f*>(a.wtab)[i];
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < 4 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  for (int i = tid; i < K; i += 64 * W) s_t4[i] = a.twH[i];
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  const int64_t run = (int64_t)blockIdx.x * W + wave;
  if (run >= a.total_runs) return;
  const int64_t row = run / a.runs_per_row;
  const int64_t j0 = (run - row * a.runs_per_row) * a.run_len;
  int64_t j1 = j0 + a.run_len;
  if (j1 > a.segs_per_row) j1 = a.segs_per_row;
  const int64_t m_start = j0 >= (R - 1) ? j0 - (R - 1) : 0;
  const float invK = 1.0f / (float)NF;
  v2f pend[R - 1 > 0 ? R - 1 : 1][2][SL];
#pragma unroll
  for (int i = 0; i < R - 1; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) pend[i][e][sl] = v2f{0.f, 0.f};

  const v4f* zrow = reinterpret_cast<const v4f*>(a.z + (size_t)row * a.M * NF) + 2 * lane;
  v4f ra[P], rb[P];   // Z[4 (l + 64 s) + 0, 1] and [+ 2, 3]
  auto issue_loads = [&](int64_t m) {
    const v4f* pz = zrow + (size_t)(m < a.M ? m : a.M - 1) * (NF / 2);
#pragma unroll
    for (int s = 0; s < P; ++s) { ra[s] = pz[128 * s]; rb[s] = pz[128 * s + 1]; }   // (non-temporal loads measured -3 % here)
  };
  issue_loads(m_start);
  v4f ca[P], cb[P];
#pragma unroll
  for (int s = 0; s < P; ++s) { ca[s] = ra[s]; cb[s] = rb[s]; }

  for (int64_t m = m_start; m < j1; ++m) {
    issue_loads(m + 1 < j1 ? m + 1 : m);  // unconditional prefetch keeps the loop branch-free
    __builtin_amdgcn_sched_barrier(0);
    v2f y[4][2][NQ];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v2f d[P];
#pragma unroll
      for (int s = 0; s < P; ++s) {
        const v4f t = r < 2 ? ca[s] : cb[s];
        d[s] = (r & 1) ? v2f{t.z, t.w} : v2f{t.x, t.y};
      }
      wave_fft_core<K, true>(d, y[r], xb, s_twB, s_twC, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < P; ++s) { ca[s] = ra[s]; cb[s] = rb[s]; }
    __builtin_amdgcn_sched_barrier(0);
    // ---- lane-local inverse radix-4: y[mm][par][q] <- 4096 x[n0 + 1024 mm]
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const v2f t1 = s_t4[2 * lane + e + 128 * q];
        const v2f t2 = wcmul(t1, t1), t3 = wcmul(t2, t1);
        v2f a0 = y[0][e][q], a1 = wcmul(y[1][e][q], t1), a2 = wcmul(y[2][e][q], t2), a3 = wcmul(y[3][e][q], t3);
        dft4<true>(a0, a1, a2, a3);
        y[0][e][q] = a0; y[1][e][q] = a1; y[2][e][q] = a2; y[3][e][q] = a3;
      }
    const float live = m < a.M ? 1.0f : 0.0f;  // tail flush: frames m >= M do not exist
    const int64_t j = m;
    const int64_t trow = j < R - 1 ? j : (j >= a.M ? R + (j - a.M) : R - 1);
    const float* dp = a.den + trow * a.hop + 2 * lane;
    v2f* yp = (j >= j0) ? a.y + (size_t)row * a.segs_per_row * a.hop + j * a.hop + 2 * lane : a.dummy + 2 * lane;
    // frame samples ((IDFT / K) * scale) * window folded into the pending overlap sums in ascending frame order;
    // sample n0 + 1024 mm lies in hop segment (128 q + 1024 mm) / HOPC at slot ((128 q + 1024 mm) % HOPC) / 128
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      v2f out[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        v2f f[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const int off = i * HOPC + 128 * sl;           // 128 q + 1024 mm of segment i, slot sl
          const int mm = off / 1024, q = (off % 1024) / 128;
          v2f v = fft_eps0(y[mm][e][q] * invK);
          if (SCALE) v = v * a.scale;
          const v2f wp = s_wv[lane + 64 * q + 512 * mm];   // (w[n], w[n + 1]) for n = 2 lane + 128 q + 1024 mm
          f[i] = v * ((e ? wp.y : wp.x) * live);
        }
        if (R == 1) { out[e] = f[0]; }
        else {
          out[e] = pend[0][e][sl] + f[0];
#pragma unroll
          for (int i = 0; i + 1 < R - 1; ++i) pend[i][e][sl] = pend[i + 1][e][sl] + f[i + 1];
          pend[R - 2][e][sl] = f[R - 1];
        }
      }
      const v2f den = *reinterpret_cast<const v2f*>(dp + 128 * sl);
      const v4f o = v4f{out[0].x * den.x, out[0].y * den.x, out[1].x * den.y, out[1].y * den.y};
      __builtin_nontemporal_store(o, (gv4f*)(yp + 128 * sl));
    }
  }
}

// ---- iSTFT for N = K/2 (512): TWO consecutive frames per 1024-point inverse FFT.  Z[k0] = C0 + w_K^k0 C1,
// Z[k0 + K/2] = C0 - w_K^k0 C1 is built lane-locally in the core's input layout (k0 = lane + 64 s'), and the inverse
// core returns sample n = lane + 64 q of frame 0 in zz[0][q] and of frame 1 in zz[1][q]: the overlap-add between the
// two frames and with the pending sums stays in registers for every hop that is a multiple of 64.
// DEEP: the next TWO pairs are in flight (see k_istft_wave): two waves per SIMD with 16 KB of loads each
template <int K, int R, bool SCALE, int W, bool DEEP>
__device__ __forceinline__ void istft_wave_half_body(const IstftWaveArgs& a) {
  constexpr int NH = K / 2;              // frame length (= fft_length)
  constexpr int R3 = K / 256;
  constexpr int NQ = K / 128;            // samples per lane per frame (n = lane + 64 q, q < NQ)
  constexpr int QS = NQ / R;             // samples per lane per hop segment
  constexpr int XCH = K + K / 16 + 16;
  static_assert(NQ % R == 0, "hop must be a multiple of 64");
  float* s_w = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_twB = reinterpret_cast<v2f*>(s_w + NH);
  v2f* s_twC = s_twB + 256;
  v2f* s_twH = s_twC + R3 * 256;          // w_K^k0, k0 < K/2 (forward)
  v2f* s_x = s_twH + NH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NH; i += 64 * W) { s_w[i] = a.wtab[i]; s_twH[i] = a.twH[i]; }
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

  float wv[NQ], tw_re[NQ], tw_im[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    wv[q] = s_w[lane + 64 * q];
    const v2f t = s_twH[lane + 64 * q];
    tw_re[q] = t.x; tw_im[q] = t.y;
  }
  const float invK = 1.0f / (float)K;
  v2f pend[R - 1 > 0 ? R - 1 : 1][QS];
#pragma unroll
  for (int i = 0; i < R - 1; ++i)
#pragma unroll
    for (int qq = 0; qq < QS; ++qq) pend[i][qq] = v2f{0.f, 0.f};

  const v2f* zrow = a.z + (size_t)row * a.M * NH + lane;
  v2f ra[2][NQ], rb[DEEP ? 2 : 1][DEEP ? NQ : 1];   // two register sets in the DEEP form: they trade roles every pair
  auto issue_into = [&](v2f (*dst)[DEEP ? NQ : NQ], int64_t m) {
    const int64_t last = a.M - 1;
    const v2f* p0 = zrow + (size_t)(m < last ? m : last) * NH;
    const v2f* p1 = zrow + (size_t)(m + 1 < last ? m + 1 : last) * NH;
#pragma unroll
    for (int q = 0; q < NQ; ++q) { dst[0][q] = __builtin_nontemporal_load(p0 + 64 * q); dst[1][q] = __builtin_nontemporal_load(p1 + 64 * q); }
  };
  v2f d[2 * NQ];
  // combine also tells whether the pair holds a non-finite bin (the sum of the bins is finite iff they all are; an overflowing
  // sum merely sends a finite pair down the solo route, which computes the same frames)
  bool nf_next = false;
  auto combine_from = [&](const v2f (*src)[NQ]) {
    v2f sum = v2f{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const v2f t = wcmul(src[1][q], v2f{tw_re[q], tw_im[q]});
      d[q] = src[0][q] + t;
      d[q + NQ] = src[0][q] - t;
      sum += d[q];   // C0 + w C1 is non-finite whenever C0 or C1 is
    }
    nf_next = wave_any_nonfinite(sum.x, sum.y);
  };
  v2f (*rbp)[NQ] = reinterpret_cast<v2f (*)[NQ]>(&rb[0][0]);   // only dereferenced when DEEP
  if (DEEP) {
    issue_into(rbp, m_start);
    issue_into(ra, m_start + 2 < j1 ? m_start + 2 : m_start);
    combine_from(rbp);
  } else {
    issue_into(ra, m_start);
    combine_from(ra);
  }

  // one pair: rn holds (or is receiving) the next pair, rf is free and receives the one after it
  auto body = [&](const int64_t m, v2f (*rn)[NQ], v2f (*rf)[NQ]) {
    // a pair that holds a non-finite bin shares it between its two frames here; the reference inverts frame by frame (:609): the
    // unit is reported and k_istft_nf_fix redoes its samples (an in-kernel solo route cost the third wave per SIMD: -12 %)
    if (__builtin_expect(nf_next, 0) && lane == 0) istft_report_nonfinite(a.nf_list, row, m);
    if (DEEP) issue_into(rf, m + 4 < j1 ? m + 4 : m);
    else issue_into(rf, m + 2 < j1 ? m + 2 : m);
    __builtin_amdgcn_sched_barrier(0);
    v2f zz[2][NQ];
    wave_fft_core<K, true>(d, zz, xb, s_twB, s_twC, lane);
    __builtin_amdgcn_sched_barrier(0);
    combine_from(rn);
    __builtin_amdgcn_sched_barrier(0);
    ifft_eps_cold<NQ>(zz, kFftEps * (float)K);   // Nx.ifft's clean-up (:609), cold form: wave_stft.hpp
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t j = m + e;              // frame index = index of the segment it completes
      const float live = j < a.M ? 1.0f : 0.0f;
      // (the phantom second half of an odd last pair, j == segs_per_row, is never stored: keep its table row inside the table —
      //  round 3: the read one row past the 2R - 1 rows faulted when the table ended its allocation)
      const int64_t jt = j < a.segs_per_row ? j : a.segs_per_row - 1;
      const int64_t trow = jt < R - 1 ? jt : (jt >= a.M ? R + (jt - a.M) : R - 1);
      const float* dp = a.den + trow * a.hop + lane;
      const bool store = (j >= j0) && (j < j1);
      v2f* yp = store ? a.y + (size_t)row * a.segs_per_row * a.hop + j * a.hop + lane : a.dummy + lane;
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
        const float rd = dp[64 * qq];
        __builtin_nontemporal_store(out * rd, (__attribute__((address_space(1))) v2f*)(yp + 64 * qq));
      }
    }
  };
  if (DEEP) {
    for (int64_t m = m_start; m < j1; m += 4) {
      body(m, ra, rbp);
      if (m + 2 < j1) body(m + 2, rbp, ra);
    }
  } else {
    for (int64_t m = m_start; m < j1; m += 2) body(m, ra, ra);
  }
}

template <int K, int R, bool SCALE, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(3, 3)))   // 168 VGPRs: the third wave per SIMD is worth 12 %
void k_istft_wave_half(IstftWaveArgs a) { istft_wave_half_body<K, R, SCALE, W, false>(a); }

template <int K, int R, bool SCALE, int W>
__global__ __launch_bounds__(64 * W) void k_istft_wave_half_deep(IstftWaveArgs a) { istft_wave_half_body<K, R, SCALE, W, true>(a); }

// ---- iSTFT for N = K/J (J = 4: N = 256, J = 8: N = 128): J consecutive frames per 1024-point inverse FFT.
// Y[k0 + N m] = 1/J sum_j (C_j[k0] w_K^(j k0)) w_J^(jm) is a lane-local forward radix-J butterfly in the core's input
// layout (k0 = lane + 64 s', m = s / (16/J)); the UNSCALED inverse transform then returns c_j[n] at index J n + j, i.e.
// zz[par][q] holds frame j = (2 lane + par) mod J, sample n = (2 lane + par + 128 q) / J.  Frames of one transform sit
// on different lanes, so the overlap-add goes through the wave's exchange buffer (idle after the core): the J windowed
// frames are parked there frame-major with plain writes, and every lane then gathers its output positions: carry of the
// earlier units + the frames that cover the position, summed in ascending frame order (deterministic, run-to-run
// bit-stable; 16-byte LDS reads, no read-modify-write chains).  The first J hop positions are normalised and stored
// with 16-byte stores; the following N - hop positions become the carry of the next unit (a small LDS strip per wave).
// DPP quad_perm control that makes lane `to` of every group of G lanes (G = 2: lanes {0,1}, {2,3} of a quad; G = 4: the quad) read
// lane `from`; the other lanes read themselves
constexpr int hop_ctrl(int G, int from, int to) {
  int sel[4] = {0, 1, 2, 3};
  if (from >= G || to >= G) return 0xE4;   // (a dead branch of an unrolled lane loop: identity)
  for (int base = 0; base < 4; base += G) sel[base + to] = base + from;
  return sel[0] | (sel[1] << 2) | (sel[2] << 4) | (sel[3] << 6);
}
template <int CTRL>
__device__ __forceinline__ v2f dpp_quad(v2f v) {
  return v2f{__int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.x), CTRL, 0xf, 0xf, true)),
             __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.y), CTRL, 0xf, 0xf, true))};
}

template <int K, int J, int R, bool SCALE, int W, bool REGOLA = true>
__global__ __launch_bounds__(64 * W) void k_istft_wave_quad(IstftWaveArgs a) {
  constexpr int NJ = K / J;              // frame length (= fft_length)
  constexpr int P = K / 64;
  constexpr int PJ = P / J;              // bins per lane per frame: k0 = lane + 64 s', s' < PJ
  constexpr int R3 = K / 256;
  constexpr int NQ = K / 128;
  constexpr int XCH = K + K / 16 + 16;
  constexpr int HOP = NJ / R;
  constexpr int CARRY = NJ - HOP;        // positions handed to the next unit
  constexpr int OUTN = J * HOP;          // positions finished per unit
  constexpr int CPAD = CARRY > 0 ? CARRY : 2;
  // Parked frames sit NJP = NJ + 16 / J complex values apart (round 4).  A park write of one instruction alternates between J / 2
  // frames at the same sample index (lane l, parity e: frame (2 l + e) mod J, sample (2 l + e) / J + ...): with the frames NJ
  // apart — a multiple of 32 banks — those lanes met in the same banks (2-way conflicts for N = 256, 4-way for N = 128: 11 % of the
  // kernel's LDS cycles, profiles/r03/istft256_sq_counters.txt).  2 NJP complex = 4 NJ + 64 / J dwords puts frame e + 2 sixteen
  // (J = 4) or eight (J = 8) banks after frame e: the 16 lanes of a ds_write_b64 group cover all 32 banks once.  The gather's
  // 16-byte reads stay aligned (NJP * 8 B is a multiple of 16) and contiguous across lanes.
  // REGOLA: overlap-add in registers (see the unit loop); false (NXSIG_ISTFT_REGOLA=0): the LDS park / gather form of rounds 2-3
  constexpr int G = J / 2;               // lanes that share a sample index
  constexpr int QW = 128 / J;            // samples per position step Q
  constexpr int HQ = NQ / R;             // position steps per hop
  constexpr int QF = J * HQ;             // finished steps per unit
  constexpr int NC = NQ - HQ;            // carried steps
  constexpr int NQT = QF + NC;
  static_assert(NQ % R == 0 && HQ >= 1, "hop must be a multiple of 128 / J samples");
  constexpr int NJP = NJ + 16 / J;
  static_assert(J * NJP <= XCH, "padded frames must fit the wave's exchange buffer");
  static_assert(OUTN % 128 == 0, "J hop must be a multiple of 128");
  static_assert(2 * HOP >= 128 / J, "lanes of one instruction would share an accumulator cell");
  float* s_w = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_twB = reinterpret_cast<v2f*>(s_w + NJ);
  v2f* s_twC = s_twB + 256;
  v2f* s_twQ = s_twC + R3 * 256;          // [j-1][k0] = conj(w_K^(j k0)), k0 < NJ
  v2f* s_x = s_twQ + (J - 1) * NJ;
  v2f* s_carry = s_x + W * XCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NJ; i += 64 * W) s_w[i] = a.wtab[i];
  for (int i = tid; i < (J - 1) * NJ; i += 64 * W) s_twQ[i] = a.twH[i];
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < R3 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  v2f* carry = s_carry + wave * CPAD;
  const int64_t run = (int64_t)blockIdx.x * W + wave;
  if (run >= a.total_runs) return;
  const int64_t row = run / a.runs_per_row;
  const int64_t u0 = (run - row * a.runs_per_row) * a.run_len;   // units of J frames / J output segments
  const int64_t units_per_row = (a.segs_per_row + J - 1) / J;
  int64_t u1 = u0 + a.run_len;
  if (u1 > units_per_row) u1 = units_per_row;
  constexpr int HALO = (R - 1 + J - 1) / J;                      // earlier units whose frames reach into this run
  const int64_t us = u0 >= HALO ? u0 - HALO : 0;
  for (int i = lane; i < CARRY; i += 64) carry[i] = v2f{0.f, 0.f};
  const float invK = 1.0f / (float)K;
  const int64_t out_len = a.segs_per_row * HOP;
  // REGOLA: this lane's window values w[(lane >> 1) + 32 q], the interior reciprocal normaliser of its output position, the carry
  v2f creg[REGOLA && NC > 0 ? NC : 1];
  float wq[REGOLA ? NQ : 1];
  float rd_int[REGOLA ? HQ : 1];
  if (REGOLA) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) wq[REGOLA ? q : 0] = s_w[(lane / G) + QW * q];
#pragma unroll
    for (int Q = 0; Q < NC; ++Q) creg[REGOLA ? Q : 0] = v2f{0.f, 0.f};
#pragma unroll
    for (int h = 0; h < HQ; ++h) rd_int[REGOLA ? h : 0] = a.den[(R - 1) * HOP + (lane / G) + QW * h];
  }

  // this lane's window values: a lane parks sample n = (2 lane + e) / J + (128 / J) q of frame (2 lane + e) mod J
  constexpr bool WREG = J == 4 && !REGOLA;   // N = 256 (LDS form): 16 registers, still three waves per SIMD; N = 128 re-reads the LDS table
  float wreg[WREG ? 2 : 1][WREG ? NQ : 1];
  if (WREG) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int q = 0; q < NQ; ++q) wreg[WREG ? e : 0][WREG ? q : 0] = s_w[(2 * lane + e + 128 * q) / J];
  }
  const v2f* zrow = a.z + (size_t)row * a.M * NJ + lane;
  v2f r[J][PJ];
  auto issue_loads = [&](int64_t u) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int64_t m = u * J + j;
      const v2f* pz = zrow + (size_t)(m < a.M ? m : a.M - 1) * NJ;  // clamped: frames past the end contribute zero
#pragma unroll
      for (int s = 0; s < PJ; ++s) r[j][s] = __builtin_nontemporal_load(pz + 64 * s);
    }
  };
  v2f d[P];
  bool nf_next = false;   // the unit just packed holds a non-finite bin (see k_istft_wave_half)
  auto pack = [&]() {   // d[s' + PJ m] = sum_j (C_j[k0] w_K^(j k0)) w_J^(jm)   (the 1/J rides in invK)
    v2f sum = v2f{0.f, 0.f};
#pragma unroll
    for (int s = 0; s < PJ; ++s) {
      v2f t[J];
      t[0] = r[0][s];
#pragma unroll
      for (int j = 1; j < J; ++j) {
        const v2f w = s_twQ[(j - 1) * NJ + lane + 64 * s];
        t[j] = wcmul(r[j][s], v2f{w.x, -w.y});
      }
      if (J == 4) dft4<false>(t[0], t[1], t[2], t[3]);
      else dft8<false>(t);
      sum += t[0];   // the butterfly's first output is the sum of the J (twiddled) bins: non-finite whenever one of them is
#pragma unroll
      for (int m = 0; m < J; ++m) d[s + PJ * m] = t[m];
    }
    nf_next = wave_any_nonfinite(sum.x, sum.y);
  };
  issue_loads(us);
  pack();

  for (int64_t u = us; u < u1; ++u) {
    if (__builtin_expect(nf_next, 0) && lane == 0) istft_report_nonfinite(a.nf_list, row, u * J);   // see k_istft_wave_half
    issue_loads(u + 1 < u1 ? u + 1 : u);  // unconditional prefetch keeps the loop branch-free
    __builtin_amdgcn_sched_barrier(0);
    v2f zz[2][NQ];
    wave_fft_core<K, true>(d, zz, xb, s_twB, s_twC, lane);
    __builtin_amdgcn_sched_barrier(0);
    pack();
    __builtin_amdgcn_sched_barrier(0);

    ifft_eps_cold<NQ>(zz, kFftEps * (float)K);   // Nx.ifft's clean-up (:609), cold form: wave_stft.hpp
    if constexpr (REGOLA) {
      // ---- the overlap-add stays in REGISTERS (round 4).  Element i = 2 lane + e + 128 q of the transform is sample
      // n = lane / G + QW q of frame j = 2 (lane % G) + e (G = J / 2 lanes share a sample index, QW = 128 / J): a group of G
      // adjacent lanes holds sample m' + QW q of all J frames of the unit.  Output position t' = m' + QW Q of the unit
      // (t' = hop j + n, hop = QW HQ) takes frame j's sample q = Q - HQ j, so every position is summed INSIDE its lane group: the
      // partial sum starts on the lane of the first frame that covers it (with the carry of the earlier units: always lane 0 of
      // the group), collects that lane's two frames, hops to the next lane (DPP quad_perm) and so on — ascending frame order, one
      // add per frame like the LDS form, whatever the unit alignment (sharded = unsharded bit for bit); the compiler fuses each
      // product into its add here, so the two forms differ by that rounding (<= 1.6e-7 of the signal).  Q < QF = J HQ are finished and end on the lane of their last frame
      // (lane g: Q in [2 g HQ, 2 (g + 1) HQ)); the NC = NQ - HQ positions beyond end on the last lane and hop back to lane 0 as
      // the next unit's carry.  No LDS traffic, no per-position address arithmetic: the park / gather form spent ~400 of its ~900
      // instructions per unit there (ISA histogram of N = 256: profiles/r04/README.md).
      const int g_lane = lane % G;
      // wave-uniform facts of the unit, as 32-bit numbers relative to its first segment / frame u J
      const int64_t rel_m = a.M - u * J, rel_s = a.segs_per_row - u * J;
      const int m_rel = rel_m > 64 ? 64 : (int)rel_m;        // frames j < m_rel of the unit exist
      const int s_rel = rel_s > 64 ? 64 : (int)rel_s;        // segments k < s_rel of the unit exist
      const bool interior = u * J >= R - 1 && m_rel >= J;    // every segment of the unit takes the interior normaliser row
      const bool stored = u >= u0;                           // (a run's halo unit only feeds the carry)
      const float live0 = 2 * g_lane < m_rel ? 1.0f : 0.0f, live1 = 2 * g_lane + 1 < m_rel ? 1.0f : 0.0f;
      // windowed sample (e, q) of this lane: ((IDFT / N) * scale) * window, the reference's rounding order; every (e, q) enters
      // exactly one position Q, so it is formed where it is consumed (no 32-register array of products)
      auto elem = [&](const int e, const int q) -> v2f {
        v2f v = zz[e][q] * invK;
        if (SCALE) v = v * a.scale;
        return v * (wq[q] * (e ? live1 : live0));
      };
      v2f S[NQT];
#pragma unroll
      for (int Q = 0; Q < NQT; ++Q) {
        // frames j with 0 <= Q - HQ j < NQ: jf .. jl, on lanes jf / 2 .. jl / 2 of the group
        const int jl = Q / HQ < J - 1 ? Q / HQ : J - 1;
        const int jf = Q - NQ + 1 > 0 ? (Q - NQ + 1 + HQ - 1) / HQ : 0;
        v2f acc = v2f{0.f, 0.f};
        if (Q < NC) acc = creg[Q < NC ? Q : 0];            // (jf = 0 there: the carry waits on lane 0)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g < jf / 2 || g > jl / 2) continue;
          if (g > jf / 2) acc = g == 1 ? dpp_quad<hop_ctrl(G, 0, 1)>(acc) : (g == 2 ? dpp_quad<hop_ctrl(G, 1, 2)>(acc) : dpp_quad<hop_ctrl(G, 2, 3)>(acc));
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int j = 2 * g + e, q = Q - HQ * j;
            if (j >= jf && j <= jl) acc = acc + elem(e, (q >= 0 && q < NQ) ? q : 0);
          }
        }
        S[Q] = acc;                                         // lives on lane jl / 2 of the group
      }
#pragma unroll
      for (int Q = 0; Q < NC; ++Q)                          // the tail beyond the unit ended on the last lane: back to lane 0
        creg[Q] = G == 2 ? dpp_quad<hop_ctrl(2, 1, 0)>(S[QF + Q]) : dpp_quad<hop_ctrl(4, 3, 0)>(S[QF + Q]);
      // finished positions: lane g of a group stores Q = 2 g HQ + r, r < 2 HQ: hop segment 2 g + r / HQ of the unit, position
      // m' + QW (r % HQ) inside it; store r of a wave writes G runs of QW consecutive samples
      v2f* yrow = a.y + (size_t)row * out_len + u * (int64_t)OUTN + (lane / G) + (int64_t)g_lane * 2 * HOP;
#pragma unroll
      for (int r = 0; r < 2 * HQ; ++r) {
        // this lane's value: S[2 g HQ + r] for its own g (one select per further lane of the group)
        v2f val = S[r];
#pragma unroll
        for (int g = 1; g < G; ++g) val = g_lane == g ? S[2 * g * HQ + r] : val;
        const int k = 2 * g_lane + r / HQ;                  // segment of the unit
        float rd = rd_int[r % HQ];
        if (!interior) {                                    // wave-uniform: the row's first unit(s) and its last ones
          const int64_t seg = u * J + k;
          const int64_t trow = seg < R - 1 ? seg : (seg >= a.M ? R + (seg - a.M) : R - 1);
          rd = k < s_rel ? a.den[trow * HOP + (lane / G) + QW * (r % HQ)] : 0.0f;
        }
        v2f* yp = (stored && k < s_rel) ? yrow + QW * r : a.dummy + lane;
        __builtin_nontemporal_store(val * rd, (gv2f*)yp);
      }
    } else {
    // ---- park the unit's J windowed frames in the (now idle) exchange buffer, frame-major: xb[j NJ + n]
      //      ((IDFT / N) * scale) * window, lib/nx_signal.ex:609-628, same rounding order; plain writes, no ordering issue
  #pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int je = (2 * lane + e) % J;   // the frame this lane holds at parity e (128 q is a multiple of J)
        const float live = (u * J + je) < a.M ? 1.0f : 0.0f;
  #pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int n = (2 * lane + e + 128 * q) / J;
          v2f v = zz[e][q] * invK;
          if (SCALE) v = v * a.scale;
          xb[je * NJP + n] = v * ((WREG ? wreg[WREG ? e : 0][WREG ? q : 0] : s_w[n]) * live);
        }
      }
      wave_lds_fence();
      // position t of the unit (t = 0 is sample u J hop of the row) = carry of earlier units + frames j with 0 <= t - j hop < NJ,
      // summed in ascending frame order; every lane takes adjacent pairs (16-byte LDS reads: hop is even)
      auto gather = [&](int t) -> v4f {
        v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
        if (CARRY > 0 && t < CARRY) acc = *reinterpret_cast<const v4f*>(&carry[t]);
  #pragma unroll
        for (int j = 0; j < J; ++j) {
          const int off = t - j * HOP;
          if (off >= 0 && off < NJ) acc += *reinterpret_cast<const v4f*>(&xb[j * NJP + off]);
        }
        return acc;
      };
      // ---- finished positions: x reciprocal of the guarded normaliser, 16-byte stores
      const int64_t t_unit = u * OUTN;
      v2f* yrow = a.y + (size_t)row * out_len;
  #pragma unroll
      for (int i = 0; i < OUTN / 128; ++i) {
        const int t = 2 * lane + 128 * i;
        const v4f acc = gather(t);
        const int64_t seg = u * J + t / HOP;           // absolute hop segment
        const int pos = t % HOP;
        const int64_t trow = seg < R - 1 ? seg : (seg >= a.M ? R + (seg - a.M) : R - 1);
        const bool inside = u >= u0 && t_unit + t < out_len;
        const v2f rd = inside ? *reinterpret_cast<const v2f*>(a.den + trow * HOP + pos) : v2f{0.f, 0.f};
        const v4f o = v4f{acc.x * rd.x, acc.y * rd.x, acc.z * rd.y, acc.w * rd.y};
        v2f* yp = inside ? yrow + t_unit + t : a.dummy + 2 * lane;
        __builtin_nontemporal_store(o, (gv4f*)yp);
      }
      // ---- carry for the next unit: positions OUTN .. OUTN + CARRY - 1 (all reads first, then the writes)
      constexpr int CI = (CARRY + 127) / 128;
      v4f nc[CI > 0 ? CI : 1];
  #pragma unroll
      for (int i = 0; i < CI; ++i) {
        const int t = 2 * lane + 128 * i;
        nc[i] = t < CARRY ? gather(OUTN + t) : v4f{0.f, 0.f, 0.f, 0.f};
      }
      wave_lds_fence();
  #pragma unroll
      for (int i = 0; i < CI; ++i) {
        const int t = 2 * lane + 128 * i;
        if (t < CARRY) *reinterpret_cast<v4f*>(&carry[t]) = nc[i];
      }
      wave_lds_fence();  // all reads of the buffer are done before the next pass A overwrites it
    }
  }
}

// ---- iSTFT for N = 2K (2048): ONE frame per TWO 1024-point inverse FFTs (decimation in frequency).  A 16-byte load
// yields Z[2k'] and Z[2k'+1] together; E = IDFT_K(even bins), O = IDFT_K(odd bins) come out of the inverse core in the
// adjacent-pair layout, and x[n] = E[n] + t O[n], x[n + K] = E[n] - t O[n], t = exp(+2 pi i n / 2K), is lane-local,
// as is the overlap-add for every hop that is a multiple of 128.
template <int K, int R, bool SCALE, int W>
__global__ __launch_bounds__(64 * W) void k_istft_wave_dbl(IstftWaveArgs a) {
  constexpr int N2 = 2 * K;              // frame length (= fft_length)
  constexpr int P = K / 64;
  constexpr int R3 = K / 256;
  constexpr int NQ = K / 128;
  constexpr int NQ2 = 2 * NQ;            // q' values of 128 samples per frame, per parity
  constexpr int QS = NQ2 / R;            // q' values per hop segment
  constexpr int XCH = K + K / 16 + 16;
  static_assert(NQ2 % R == 0, "hop must be a multiple of 128");
  float* s_w = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_twB = reinterpret_cast<v2f*>(s_w + N2);
  v2f* s_twC = s_twB + 256;
  v2f* s_twH = s_twC + R3 * 256;          // exp(+2 pi i n / 2K), n < K
  v2f* s_x = s_twH + K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < N2; i += 64 * W) s_w[i] = a.wtab[i];
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
  v2f pend[R - 1 > 0 ? R - 1 : 1][2][QS];
#pragma unroll
  for (int i = 0; i < R - 1; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int qq = 0; qq < QS; ++qq) pend[i][e][qq] = v2f{0.f, 0.f};

  const v4f* zrow = reinterpret_cast<const v4f*>(a.z + (size_t)row * a.M * N2) + lane;
  // even / odd bins (k' = lane + 64 s): one 16-byte load per point pair (non-temporal: the spectrogram is read once).  For
  // R <= 4 the next frame's loads are in flight during the two cores; R = 8 has no registers left for that (7 pending
  // segments) and loads at the top of the iteration.
  constexpr bool PF = R <= 4;
  v4f nv[P];
  auto issue_loads = [&](int64_t m) {
    const v4f* pz = zrow + (size_t)(m < a.M ? m : a.M - 1) * K;
#pragma unroll
    for (int s = 0; s < P; ++s) nv[s] = __builtin_nontemporal_load(pz + 64 * s);
  };
  if (PF) issue_loads(m_start);
  for (int64_t m = m_start; m < j1; ++m) {
    if (!PF) issue_loads(m);
    v2f de[P], dq[P];
#pragma unroll
    for (int s = 0; s < P; ++s) { de[s] = v2f{nv[s].x, nv[s].y}; dq[s] = v2f{nv[s].z, nv[s].w}; }
    if (PF) {
      issue_loads(m + 1 < j1 ? m + 1 : m);  // unconditional prefetch keeps the loop branch-free
      __builtin_amdgcn_sched_barrier(0);
    }
    v2f ze[2][NQ], zo[2][NQ];
    wave_fft_core<K, true>(de, ze, xb, s_twB, s_twC, lane);
    wave_fft_core<K, true>(dq, zo, xb, s_twB, s_twC, lane);
    const float live = m < a.M ? 1.0f : 0.0f;
    const int64_t j = m;
    const int64_t trow = j < R - 1 ? j : (j >= a.M ? R + (j - a.M) : R - 1);
    const float* dp = a.den + trow * a.hop + 2 * lane;
    v2f* yp = (j >= j0) ? a.y + (size_t)row * a.segs_per_row * a.hop + j * a.hop + 2 * lane : a.dummy + 2 * lane;
    // frame samples x[n'] for n' = 2 lane + e + 128 q', q' < NQ2 (q' >= NQ is the upper half n + K)
    auto sample = [&](int e, int qp) -> v2f {
      const int q = qp % NQ;
      const v2f t = s_twH[2 * lane + e + 128 * q];
      const v2f to = wcmul(zo[e][q], t);
      v2f v = fft_eps0((qp < NQ ? ze[e][q] + to : ze[e][q] - to) * invN);
      if (SCALE) v = v * a.scale;
      return v * (s_w[2 * lane + e + 128 * qp] * live);
    };
#pragma unroll
    for (int qq = 0; qq < QS; ++qq) {
      v2f out[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        v2f f[R];
#pragma unroll
        for (int i = 0; i < R; ++i) f[i] = sample(e, i * QS + qq);
        if (R == 1) { out[e] = f[0]; }
        else {
          out[e] = pend[0][e][qq] + f[0];
#pragma unroll
          for (int i = 0; i + 1 < R - 1; ++i) pend[i][e][qq] = pend[i + 1][e][qq] + f[i + 1];
          pend[R - 2][e][qq] = f[R - 1];
        }
      }
      const v2f rd = *reinterpret_cast<const v2f*>(dp + 128 * qq);
      const v4f o = v4f{out[0].x * rd.x, out[0].y * rd.x, out[1].x * rd.y, out[1].y * rd.y};
      __builtin_nontemporal_store(o, (gv4f*)(yp + 128 * qq));
    }
  }
}

// ============================================================================================ FIR (overlap-save)
// y = x * h by overlap-save block FFT convolution (the `Filters.fir` of BASELINE config 5; equals the reference's
// Convolution.convolve(x, h, method: :fft) of lib/nx_signal/convolution.ex:252-329 to fp32 rounding).  One wave
// filters TWO consecutive blocks of K samples packed as re / im of one complex FFT (h is real):
//   load (adjacent-pair layout) -> core_T (forward) -> x H/K, conj -> core (forward again = inverse up to conj)
//   -> the K - (taps-1) valid samples of both blocks leave with 8-byte stores.
// Block b covers full-convolution outputs [b V, (b+1) V), V = K - (taps-1), from x[b V - (taps-1) + n].
struct FirWaveArgs {
  const float* x;                      // row base shifted by the grid phase (see launch_fir_wave_W)
  int64_t L, batch_stride;
  int64_t xlo, xhi;                    // valid sample indices relative to x: [xlo, xhi) = [-phase, L - phase)
  int32_t batch, taps;
  int64_t V, nblocks, first_block;     // blocks (per row) covering the requested output slice
  int64_t pairs_per_row;               // block pairs per row
  int64_t pb_lo, pb_hi;                // interior pairs [pb_lo, pb_hi): both blocks fully inside input and output
  int64_t units_per_row, total_units, chunk;  // STREAM: interior pairs; EDGE: the other pairs of each row
  int64_t out_start, out_len;
  const v2f* H;                        // c64[K] natural order, pre-scaled by 1/K
  const v2f* twB;
  const v2f* twC;
  float* y;                            // f32[batch][out_len]
  int* row_flags;                      // FirLaunch::row_flags: a non-finite sample poisons its whole row, like the reference's one transform
  const v2f* coefA = nullptr;          // k_fir_r2k: c64[1024] each, W[k] = A[k] Z[k] + B[k] conj Z[(1024 - k) mod 1024]
  const v2f* coefB = nullptr;
  // per-row grid phase (round 5): rows whose length is not a multiple of 32 samples start off a 128-byte boundary, every row at a
  // different offset, and one common phase aligns the first row only (the other rows' streaming stores then hit partial lines: 2 x
  // slower).  row_mod = out_len mod 32 (0: off): row r shifts its block grid by fir_row_shift(r) = (-r out_len) mod 32 further
  // samples, i.e. x'' = x' + shift, [xlo, xhi) and out_start move down by shift, so that every block of every row starts a line of y.
  int32_t row_mod = 0;
};
__device__ __forceinline__ int fir_row_shift(const int32_t row_mod, const int64_t row) {
  return row_mod ? (int)((32 - (int)((row * row_mod) & 31)) & 31) : 0;
}

int launch_fir_wave32(Ctx* c, const float* x, int64_t batch_stride, int32_t batch, int32_t taps, int64_t first_block, int64_t pb_lo,
                      int64_t dp_per_row, int64_t out_start, int64_t out_len, const float2* H_dev, float* y, int* row_flags, int row_mod);  // kernels_wave_fir32.hip

// STREAM = true : interior pairs only, 8-byte vector access, branch-free and software-pipelined like k_stft_wave
//                 (requires (taps-1) % 128 == 0 and even offsets — checked by the launcher)
// STREAM = false: the few edge pairs of every row (and every pair when the fast conditions fail): bounds-checked
// TQ > 0: taps - 1 == 128 TQ at compile time (STREAM launches of the default kernels): which 128-sample slots of a block are valid
// folds away — with a run-time tap count every slot cost a scalar branch plus the VALU book-keeping of its condition, which ate what
// round 4's cheaper clean-up had saved (SQ_INSTS_VALU per pair 706 -> 688 instead of -> ~650).  TQ = 0: run-time tap count.
template <int K, bool STREAM, int W, bool HREG = false, int TQ = 0>
__global__ __launch_bounds__(64 * W) void k_fir_wave(FirWaveArgs a) {
  constexpr int P = K / 64;
  constexpr int R3 = K / 256;
  constexpr int NQ = K / 128;
  constexpr int XCH = K + K / 16 + 16;
  v2f* s_twB = reinterpret_cast<v2f*>(g_wave_smem);
  v2f* s_twC = s_twB + 256;
  v2f* s_H = s_twC + R3 * 256;   // the inverse core reads the same tables and conjugates inside its multiplies
  v2f* s_x = s_H + (HREG ? 0 : K);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < R3 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  if (!HREG)
    for (int i = tid; i < K; i += 64 * W) s_H[i] = a.H[i];
  v2f hv[HREG ? P : 1];   // HREG: the lane's P filter-spectrum values stay in registers
  if (HREG) {
#pragma unroll
    for (int s = 0; s < P; ++s) hv[HREG ? s : 0] = a.H[lane + 64 * s];
  }
  __syncthreads();
  v2f* xb = s_x + wave * XCH;

  const int64_t p_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t p_end = p_begin + a.chunk;
  if (p_end > a.total_units) p_end = a.total_units;
  const int tm1 = TQ > 0 ? 128 * TQ : a.taps - 1;

  if (STREAM) {
    int64_t row = (p_begin + wave) / a.units_per_row;
    int64_t pin = (p_begin + wave) - row * a.units_per_row;
    int64_t nrow = row, npin = pin;
    auto advance = [&](int64_t& r, int64_t& q) {
      q += W;
      while (q >= a.units_per_row) { q -= a.units_per_row; ++r; }
    };
    advance(nrow, npin);
    v2f r1[NQ], r2[NQ];
    auto issue_loads = [&](int64_t rw, int64_t pi) {
      const int64_t b1 = a.first_block + 2 * (a.pb_lo + pi);
      const float* p1 = a.x + (size_t)rw * a.batch_stride + (b1 * a.V - tm1 + fir_row_shift(a.row_mod, rw)) + 2 * lane;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        r1[q] = *reinterpret_cast<const v2f*>(p1 + 128 * q);   // default cache policy: the pair's blocks overlap and neighbours re-read
        r2[q] = *reinterpret_cast<const v2f*>(p1 + a.V + 128 * q);   // the halo (non-temporal loads measured 0…-3 %)
      }
    };
    v2f zz[2][NQ];  // zz[par][q] = (x1[n], x2[n]), n = 2 lane + par + 128 q
    auto pack = [&]() {
#pragma unroll
      for (int q = 0; q < NQ; ++q) { zz[0][q] = v2f{r1[q].x, r2[q].x}; zz[1][q] = v2f{r1[q].y, r2[q].y}; }
    };
    if (p_begin + wave < p_end) { issue_loads(row, pin); pack(); }
    for (int64_t pr = p_begin + wave; pr < p_end; pr += W) {
      const bool more = pr + W < p_end;
      issue_loads(more ? nrow : row, more ? npin : pin);  // unconditional prefetch: branch-free loop
      __builtin_amdgcn_sched_barrier(0);
      v2f d[P];
      wave_fft_core_T<K>(zz, d, xb, s_twB, s_twC, lane);
#pragma unroll
      for (int s = 0; s < P; ++s) d[s] = wcmul(d[s], HREG ? hv[HREG ? s : 0] : s_H[lane + 64 * s]);  // Z H / K
      v2f u[2][NQ];
      wave_fft_core<K, true, true>(d, u, xb, s_twB, s_twC, lane);  // inverse: u = (y1[n], y2[n])
      __builtin_amdgcn_sched_barrier(0);
      pack();  // next pair (its samples landed during the two transforms)
      __builtin_amdgcn_sched_barrier(0);
      // Non-finite samples (round 4: tested on the OUTPUTS the inverse core holds anyway instead of one packed add per input
      // point).  Every bin of a block's forward transform depends on every sample of the block and every output on every bin, so
      // an Inf / NaN sample leaves NO finite output in its block: one output value per block (u[..].x: block 1, .y: block 2) says
      // whether the block held one.  (Finite samples whose transform overflows poison a row whose outputs overflow anyway.)
      if (wave_any_nonfinite(u[0][NQ - 1].x, u[0][NQ - 1].y) && lane == 0) atomicOr(a.row_flags + row, 1);
      // Nx.ifft's clean-up (|y| <= 1e-10 -> +0, convolution.ex:282) concerns digital silence only: a lane first takes the minimum
      // magnitude of its outputs (one v_min3 per two values) and the wave applies the compare-and-select per value only when some
      // lane holds a value at or below the threshold (wave-uniform branch, cold) — 2 instructions per output became 1/2
      float amin = 3.0e38f;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (128 * q >= tm1) {
          amin = min3abs(u[0][q].x, u[0][q].y, amin);
          amin = min3abs(u[1][q].x, u[1][q].y, amin);
        }
      if (__builtin_amdgcn_ballot_w64(amin <= kFftEps) != 0) {   // cold
#pragma unroll
        for (int q = 0; q < NQ; ++q) { u[0][q] = fft_eps0(u[0][q]); u[1][q] = fft_eps0(u[1][q]); }
      }
      const int64_t b1 = a.first_block + 2 * (a.pb_lo + pin);
      // the pair's two valid parts are one contiguous run of 2 V outputs: streaming stores (sc1 nt) through a row descriptor
      const StreamRow ys(a.y + (size_t)row * a.out_len + (b1 * a.V - a.out_start + fir_row_shift(a.row_mod, row)), (uint32_t)(2 * a.V) * 4);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (128 * q >= tm1) {  // uniform: (taps-1) % 128 == 0
          ys.template st8<18>(v2f{u[0][q].x, u[1][q].x}, lane * 8 + 512 * q - tm1 * 4);
          ys.template st8<18>(v2f{u[0][q].y, u[1][q].y}, lane * 8 + 512 * q - tm1 * 4 + a.V * 4);
        }
      }
      row = nrow; pin = npin;
      advance(nrow, npin);
    }
  } else {
    const int64_t n_lo = a.pb_lo, n_hi = a.pairs_per_row - a.pb_hi;  // edge pairs per row: [0, pb_lo) and [pb_hi, pairs)
    for (int64_t pr = p_begin + wave; pr < p_end; pr += W) {
      const int64_t row = pr / a.units_per_row;
      const int64_t e = pr - row * a.units_per_row;
      const int64_t pb = e < n_lo ? e : a.pb_hi + (e - n_lo);
      (void)n_hi;
      const int64_t b1 = a.first_block + 2 * pb, b2 = b1 + 1;
      const bool have2 = (b2 - a.first_block) < a.nblocks;
      const int rsh = fir_row_shift(a.row_mod, row);
      const float* xr = a.x + (size_t)row * a.batch_stride + rsh;
      const int64_t xlo = a.xlo - rsh, xhi = a.xhi - rsh;
      float* yr = a.y + (size_t)row * a.out_len;
      const int64_t s1 = b1 * a.V - tm1, s2 = s1 + a.V;
      const int64_t o1 = b1 * a.V - a.out_start + rsh - tm1;  // y index of block-1 sample n is o1 + n (n >= taps-1)
      v2f zz[2][NQ];
      v2f nfs = v2f{0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          const int n = 2 * lane + e2 + 128 * q;
          const int64_t p1 = s1 + n, p2 = s2 + n;
          const float v1 = (p1 >= xlo && p1 < xhi) ? xr[p1] : 0.0f;
          const float v2 = (have2 && p2 >= xlo && p2 < xhi) ? xr[p2] : 0.0f;
          zz[e2][q] = v2f{v1, v2};
          nfs += zz[e2][q];
        }
      if (wave_any_nonfinite(nfs.x, nfs.y) && lane == 0) atomicOr(a.row_flags + row, 1);
      v2f d[P];
      wave_fft_core_T<K>(zz, d, xb, s_twB, s_twC, lane);
#pragma unroll
      for (int s = 0; s < P; ++s) d[s] = wcmul(d[s], HREG ? hv[HREG ? s : 0] : s_H[lane + 64 * s]);
      wave_fft_core<K, true, true>(d, zz, xb, s_twB, s_twC, lane);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          const int n = 2 * lane + e2 + 128 * q;
          if (n >= tm1) {
            const int64_t y1 = o1 + n, y2 = y1 + a.V;
            if (y1 >= 0 && y1 < a.out_len) yr[y1] = fft_eps0(zz[e2][q].x);
            if (have2 && y2 >= 0 && y2 < a.out_len) yr[y2] = fft_eps0(zz[e2][q].y);
          }
        }
    }
  }
}

// ---- round 5: ONE real block of 2048 samples per 1024-point complex transform.  z[m] = x[2m] + i x[2m + 1],
// Z = FFT_1024(z); the Hermitian untangle of the 2048-point real spectrum, the product with H and the re-tangle of the result fuse
// into  W[k] = A[k] Z[k] + B[k] conj Z[(1024 - k) mod 1024],  A = (S - D sin th) / 1024,  B = i D cos th / 1024,  S / D = (H[k] +-
// H[k + 1024]) / 2,  th = 2 pi k / 2048  (tables computed on the host in double), and IFFT_1024(W)[m] = y[2m] + i y[2m + 1].  Against
// the pair form (two 1024-sample blocks as re / im: k_fir_wave): 1 792 instead of 1 536 outputs per two transforms, 16-byte loads AND
// stores (a lane owns 4 consecutive samples per 256-sample slot), at the price of the partner fetch (32 ds_bpermute per block), a
// second complex multiply per bin and the B table (LDS; A rides in the registers H used).  STREAM blocks only (taps - 1 a multiple of
// 256 <= 1024, 16-byte aligned rows); the edge blocks of a row stay with k_fir_wave<2048, false> (same block grid: V = 2048 - taps + 1).
// TQ2: taps - 1 == 256 TQ2.
template <int W, int TQ2>
__global__ __launch_bounds__(64 * W) void k_fir_r2k(FirWaveArgs a) {
  constexpr int K = 1024, P = 16, R3 = 4, NQ = 8, XCH = K + K / 16 + 16, TM1 = 256 * TQ2;
  v2f* s_twB = reinterpret_cast<v2f*>(g_wave_smem);
  v2f* s_twC = s_twB + 256;
  v2f* s_B = s_twC + R3 * 256;
  v2f* s_x = s_B + K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < R3 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  for (int i = tid; i < K; i += 64 * W) s_B[i] = a.coefB[i];
  v2f av[P];   // this lane's A values (k = lane + 64 s)
#pragma unroll
  for (int s = 0; s < P; ++s) av[s] = a.coefA[lane + 64 * s];
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  const int64_t p_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t p_end = p_begin + a.chunk;
  if (p_end > a.total_units) p_end = a.total_units;
  // unit u of a row = block first_block + 2 pb_lo + u (the interior PAIRS of the 2048-block grid, one block per unit)
  int64_t row = (p_begin + wave) / a.units_per_row;
  int64_t uin = (p_begin + wave) - row * a.units_per_row;
  int64_t nrow = row, nuin = uin;
  auto advance = [&](int64_t& r, int64_t& q) {
    q += W;
    while (q >= a.units_per_row) { q -= a.units_per_row; ++r; }
  };
  advance(nrow, nuin);
  v4f r4[NQ];
  auto issue_loads = [&](int64_t rw, int64_t ui) {
    const int64_t b = a.first_block + 2 * a.pb_lo + ui;
    const v4f* p = reinterpret_cast<const v4f*>(a.x + (size_t)rw * a.batch_stride + (b * a.V - TM1 + fir_row_shift(a.row_mod, rw))) + lane;
#pragma unroll
    for (int q = 0; q < NQ; ++q) r4[q] = p[64 * q];   // samples 4 lane + 256 q .. + 3: z[2 lane + 128 q], z[2 lane + 1 + 128 q]
  };
  v2f zz[2][NQ];
  auto pack = [&]() {
#pragma unroll
    for (int q = 0; q < NQ; ++q) { zz[0][q] = v2f{r4[q].x, r4[q].y}; zz[1][q] = v2f{r4[q].z, r4[q].w}; }
  };
  if (p_begin + wave < p_end) { issue_loads(row, uin); pack(); }
  const int src = ((64 - lane) & 63) << 2;
  for (int64_t pr = p_begin + wave; pr < p_end; pr += W) {
    const bool more = pr + W < p_end;
    issue_loads(more ? nrow : row, more ? nuin : uin);  // unconditional prefetch: branch-free loop
    __builtin_amdgcn_sched_barrier(0);
    v2f d[P];
    wave_fft_core_T<K>(zz, d, xb, s_twB, s_twC, lane);   // d[s] = Z[lane + 64 s]
    // partner conj Z[(1024 - k) mod 1024], k = lane + 64 s: lane (64 - lane) & 63, register 15 - s (lane 0: its own (16 - s) mod 16)
    v2f wv[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
      v2f pz;
      pz.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d[P - 1 - s].x)));
      pz.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d[P - 1 - s].y)));
      if (lane == 0) pz = d[(P - s) % P];
      const v2f t = wcmul(d[s], av[s]);
      wv[s] = t + wcmul_conj(s_B[lane + 64 * s], pz);   // B conj(pz): wcmul_conj(a, b) = a conj(b)
    }
    v2f u[2][NQ];
    wave_fft_core<K, true, true>(wv, u, xb, s_twB, s_twC, lane);  // unscaled inverse: u[par][q] = (y[2m], y[2m + 1]), m = 2 lane + par + 128 q
    __builtin_amdgcn_sched_barrier(0);
    pack();  // next block (its samples landed during the two transforms)
    __builtin_amdgcn_sched_barrier(0);
    // non-finite samples: every output of a block depends on every sample of the block (see k_fir_wave)
    if (wave_any_nonfinite(u[0][NQ - 1].x, u[0][NQ - 1].y) && lane == 0) atomicOr(a.row_flags + row, 1);
    float amin = 3.0e38f;
#pragma unroll
    for (int q = TQ2; q < NQ; ++q) {
      amin = min3abs(u[0][q].x, u[0][q].y, amin);
      amin = min3abs(u[1][q].x, u[1][q].y, amin);
    }
    if (__builtin_amdgcn_ballot_w64(amin <= kFftEps) != 0) {   // cold: Nx.ifft's clean-up (convolution.ex:282)
#pragma unroll
      for (int q = 0; q < NQ; ++q) { u[0][q] = fft_eps0(u[0][q]); u[1][q] = fft_eps0(u[1][q]); }
    }
    const int64_t b = a.first_block + 2 * a.pb_lo + uin;
    const StreamRow ys(a.y + (size_t)row * a.out_len + (b * a.V - a.out_start + fir_row_shift(a.row_mod, row)), (uint32_t)a.V * 4);
#pragma unroll
    for (int q = TQ2; q < NQ; ++q)
      ys.st16(v4f{u[0][q].x, u[0][q].y, u[1][q].x, u[1][q].y}, lane * 16 + 1024 * q - TM1 * 4);
    row = nrow; uin = nuin;
    advance(nrow, nuin);
  }
}

int launch_stft_r20(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);  // kernels_wave_r20.hip
int launch_stft_rab(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);    // kernels_wave_rab.hip
int rab_length_part(int K);                                                               // >= 0: a native A x B kernel exists for fft length K
int launch_stft_wave_8k(Ctx* c, const StftLaunch& s, bool* handled);                      // kernels_wave_8k.hip

#ifdef NXSIG_TRACE
extern "C" int nxsig_diag_set_trace(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_trace), &p, sizeof(p));
}
#endif

int launch_stft_wave(Ctx* c, const StftLaunch& s, bool* handled) {
  *handled = false;
  if (s.fr.M == 0 || s.batch == 0) return NXSIG_OK;
  if (tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  if (s.window_padK == nullptr) return NXSIG_OK;
  // 4 waves per workgroup everywhere: 8 / 12 / 16 measured equal or slower (tables are re-read from L2 either way)
  switch (s.K) {
    case 1024: {                                                                   // two frames per 1024-point complex FFT
      *handled = true;
      // ONE-ROUND geometry (round 6): a launch whose workgroups all fit on the chip at once (BASELINE config 1, and config 2 as
      // written: one 60 s stream = 5 624 frame pairs) is one round of start-up latencies; 4-wave workgroups of 12 pairs land 2 + 2 +
      // ... + 1 on the CUs (469 on 256) and a 1 s stream fills 24 of them.  Here ONE 12-wave workgroup per CU (120 KB of LDS exclude
      // a second one; the dispatcher places them one per CU, tools/wg_census.hip) on 11/16 of the CUs, the chunk a multiple of four
      // pairs: measured optimum of interleaved sweeps over chunk sizes and stream lengths (tools/sweep_small.py,
      // profiles/r06/one_round_geometry.txt): 1 s ... 30 s streams 2.2 ... 1.2 x the many-round geometry, 60 s +1 ... 7 %
      // (box-dependent).  NXSIG_WAVE_SMALL_W=0 keeps the old geometry, NXSIG_WAVE_SMALL_CHUNK forces a chunk.
      const int64_t pairs = (int64_t)s.batch * ((s.fr.M + 1) / 2);
      int64_t chunk = tune(c, kT_WAVE_SMALL_CHUNK, 0);
      const bool forced = chunk > 0;
      if (tune(c, kT_WAVE_SMALL_W, 12) >= 12 && (forced || pairs <= (int64_t)c->num_cus * 24)) {
        if (!forced) {
          const int64_t wgs = ((int64_t)c->num_cus * 11) / 16;
          chunk = (pairs + wgs - 1) / wgs;
          if (chunk >= 16) chunk = (chunk + 3) & ~(int64_t)3;
        }
        return launch_wave<1024, kModePair, 12>(c, s, nullptr, chunk);
      }
      if (forced && pairs <= (int64_t)c->num_cus * 24) return launch_wave<1024, kModePair, 4>(c, s, nullptr, chunk);
      return launch_wave<1024, kModePair, 4>(c, s);
    }
    case 512: *handled = true; return launch_wave<1024, kModeQuad, 4, 2>(c, s);    // 4 frames interleaved into one transform
    case 256: *handled = true; return launch_wave<1024, kModeQuad, 4, 4>(c, s);    // 8 frames
    case 128: *handled = true; return launch_wave<1024, kModeQuad, 4, 8>(c, s);    // 16 frames
    case 2048: *handled = true; return launch_wave<1024, kModeReal2x, 4>(c, s);    // one frame as even/odd samples
    case 4096: *handled = true; return launch_wave<2048, kModeReal2x, 4>(c, s);    // same on the 2048-point core
    case 8192: {                                                                   // four passes through the 1024-point core
      int rc8 = launch_stft_wave_8k(c, s, handled);
      if (rc8 || *handled) return rc8;
    } break;
    default: break;
  }
  if (s.K == 400) {  // 20 x 20 native kernel (kernels_wave_r20.hip)
    int rc20 = launch_stft_r20(c, s, handled, nullptr);
    if (rc20 || *handled) return rc20;
  }
  if (rab_length_part(s.K) >= 0) {  // A x B native kernels (kernels_wave_rab*.hip): 100 ... 1600, see wave_rab.hpp
    int rcab = launch_stft_rab(c, s, handled, nullptr);
    if (rcab || *handled) return rcab;
  }
  if ((s.K & (s.K - 1)) != 0 && s.K > 16 && s.K <= 1024 && !tune(c, kT_DISABLE_BLUE_WAVE, 0)) {
    *handled = true;  // non-power-of-two: Bluestein through the 1024- (Kb <= 512) or 2048-point core
    return s.K <= 512 ? launch_blue_wave<1024>(c, s) : launch_blue_wave<2048>(c, s);
  }
  return NXSIG_OK;
}

// guarded normaliser rows (lib/nx_signal.ex:630-635) as RECIPROCALS: f32[2R-1][hop] = head segments 0..R-2, the interior
// segment, tail segments; double accumulation in ascending frame order, one rounding
int istft_den_table(Ctx* c, int R, int hop, const float* window_host, const float** out) {
  const uint64_t dkey = fnv1a(0xDE18ull ^ ((uint64_t)R << 32) ^ ((uint64_t)hop << 8), window_host, (size_t)R * hop * sizeof(float));
  auto hit = c->memo.find(dkey);
  if (hit != c->memo.end()) { *out = reinterpret_cast<const float*>(hit->second[0]); return NXSIG_OK; }
  std::vector<float> den((size_t)(2 * R - 1) * hop);
  auto w2 = [&](int idx) { const float w = std::fabs(window_host[idx]); return (double)(w * w); };
  for (int row = 0; row < 2 * R - 1; ++row)
    for (int pos = 0; pos < hop; ++pos) {
      double acc = 0.0;
      for (int rr = R - 1; rr >= 0; --rr) {  // frame j - rr contributes w2[rr*hop + pos]
        bool have;
        if (row < R - 1) have = rr <= row;             // head segment j = row: frames j - rr >= 0
        else if (row == R - 1) have = true;            // interior
        else have = rr >= row - R + 1;                 // tail segment j = M + (row - R): frames j - rr <= M - 1
        if (have) acc += w2(rr * hop + pos);
      }
      const float d = (float)acc;
      den[(size_t)row * hop + pos] = (float)(1.0 / (double)(d > 1.0e-10f ? d : 1.0f));  // reciprocal of the guarded normaliser
    }
  const void* dd = nullptr;
  int rc3 = ctx_table(c, 0xDE17ull ^ ((uint64_t)R << 32), den.data(), den.size() * sizeof(float), &dd);
  if (rc3) return rc3;
  *out = reinterpret_cast<const float*>(dd);
  c->memo[dkey] = {reinterpret_cast<uint64_t>(dd)};
  return NXSIG_OK;
}

template <int R, int W, bool HALF = false, bool DBL = false>
static int launch_istft_wave_R(Ctx* c, const IstftLaunch& s, const float* window_padK, const float* window_host) {
  constexpr int K = 1024, R3 = K / 256, XCH = K + K / 16 + 16;
  IstftWaveArgs a;
  a.z = reinterpret_cast<const v2f*>(s.z); a.M = s.M; a.batch = s.batch; a.hop = s.hop;
  a.segs_per_row = s.M + R - 1;
  a.wtab = window_padK;
  Ctx::WaveTables& wt = c->wave_tables[K];
  if (!wt.twB) return NXSIG_ERR_UNSUPPORTED;  // tables are created by ensure_wave_tables below
  a.twB = reinterpret_cast<const v2f*>(wt.twBi);  // conjugated tables: the kernel runs the core in inverse direction
  a.twC = reinterpret_cast<const v2f*>(wt.twCi);
  a.scale = s.scale_mul;
  { int rc3 = istft_den_table(c, R, s.hop, window_host, &a.den); if (rc3) return rc3; }
  a.y = reinterpret_cast<v2f*>(s.y);
  a.filt = reinterpret_cast<const v2f*>(s.filt);
  {
    const void* dz = nullptr;
    auto hitz = c->memo.find(0x2E2000000000ull ^ (uint64_t)K);   // looked up once per context (the table cache hashes the content)
    if (hitz != c->memo.end()) dz = reinterpret_cast<const void*>(hitz->second[0]);
    else {
      static const std::vector<float2> zero_row((size_t)K, make_float2(0.f, 0.f));
      int rcz = ctx_table(c, 0x2E20ull, zero_row.data(), zero_row.size() * sizeof(float2), &dz);
      if (rcz) return rcz;
      c->memo[0x2E2000000000ull ^ (uint64_t)K] = {reinterpret_cast<uint64_t>(dz)};
    }
    a.zeros = reinterpret_cast<const v2f*>(dz);
  }
  void* dummy = nullptr;
  { int rc2 = ctx_scratch(c, 3, (size_t)8192 * sizeof(float2), &dummy); if (rc2) return rc2; }
  a.dummy = reinterpret_cast<v2f*>(dummy);
  const int64_t total_segs = a.segs_per_row * s.batch;
  // Two-frames-ahead prefetch (DEEP, round 3): 8 resident waves per CU with 16 KB of loads in flight each instead of 12 with 8 KB.
  // Interleaved A/B sweeps (tools/sweep_istft.py NXSIG_ISTFT_DEEP 0 1; 1 / 2 / 4 / 8 / 16 / 32 streams of 60 s): +1.5 / +8 / +14 /
  // +10 / +2 / +0.4 %; hop 128 / 512: +1.3 / +3 %; bench.py's laps of config 3: +1.3 %.  NXSIG_ISTFT_DEEP=0 selects the one-ahead form.
  const bool deep = !DBL && !HALF && !s.filt && tune(c, kT_ISTFT_DEEP, 1);
  // the N = 512 pair kernel likewise at hop = N / 4 (2 / 8 / 16 streams of 60 s: +13 / +9 / +5 %); at hop N / 8 and N / 2 the
  // one-ahead form with three waves per SIMD stays ahead (-4 % / -4 ... -7 % for the deep form) and is kept there
  const bool half_deep = HALF && tune(c, kT_ISTFT_HALF_DEEP, R == 4 ? 1 : 0);
  const int waves_per_cu = tune(c, kT_ISTFT_RUNS_PER_CU, (DBL || s.filt || deep || half_deep) ? 8 : 12);  // the filtered variant holds 16 more
                                                                                         // complex values per lane: 2 waves per SIMD  // = resident waves per CU: one even round
  int64_t run_len = (total_segs + (int64_t)c->num_cus * waves_per_cu - 1) / ((int64_t)c->num_cus * waves_per_cu);
  // (a floor of 8 left a single 60 s stream with 1 406 runs of 8 + 3 frames for 2 048 wave slots: 2.58 TB/s against 2.96 at 4 ... 6)
  const int min_run = istft_min_run(c, total_segs, (int64_t)c->num_cus * waves_per_cu, (!HALF && !DBL) ? 4 : 8);   // (N = 512 / 2048: 8 stays 1 ... 3 % ahead)
  if (run_len < min_run) run_len = min_run;
  if (HALF) run_len = (run_len + 1) & ~(int64_t)1;  // frame pairs: runs start at even segments
  a.run_len = run_len;
  a.runs_per_row = (a.segs_per_row + run_len - 1) / run_len;
  a.total_runs = a.runs_per_row * s.batch;
  const int64_t blocks = (a.total_runs + W - 1) / W;
  if (HALF) {
    const void* dh = nullptr;
    {  // built once per context (the content-addressed table cache alone would recompute the sines on every call to look it up)
      auto hit = c->memo.find(0x774800000000ull ^ (uint64_t)K);
      if (hit != c->memo.end()) dh = reinterpret_cast<const void*>(hit->second[0]);
      else {
        std::vector<float2> twH((size_t)K / 2);
        for (int k0 = 0; k0 < K / 2; ++k0) {
          const double ang = -6.283185307179586476925286766559 * (double)k0 / (double)K;
          twH[k0] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
        int rc4 = ctx_table(c, 0x7748ull ^ (uint64_t)K, twH.data(), twH.size() * sizeof(float2), &dh);
        if (rc4) return rc4;
        c->memo[0x774800000000ull ^ (uint64_t)K] = {reinterpret_cast<uint64_t>(dh)};
      }
    }
    a.twH = reinterpret_cast<const v2f*>(dh);
    const size_t lds = (size_t)(K / 2) * 4 + 256 * 8 + (size_t)R3 * 256 * 8 + (size_t)(K / 2) * 8 + (size_t)W * XCH * 8;
    // every run also walks its halo units: capacity = units walked by all runs
    { int rcl = istft_nf_list(c, a.total_runs * ((run_len + R + 1) / 2 + 1), &a.nf_list); if (rcl) return rcl; }
    s.nf_list = a.nf_list; s.nf_frames_per_unit = 2;
    dispatch_note(half_deep ? "istft.half.deep" : "istft.half");
    if (half_deep) {
      if (s.has_scale) hipLaunchKernelGGL((k_istft_wave_half_deep<K, R, true, W>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
      else hipLaunchKernelGGL((k_istft_wave_half_deep<K, R, false, W>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    } else if (s.has_scale) hipLaunchKernelGGL((k_istft_wave_half<K, R, true, W>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    else hipLaunchKernelGGL((k_istft_wave_half<K, R, false, W>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
  } else if (DBL) {
    const void* dh = nullptr;
    {
      auto hit = c->memo.find(0x774900000000ull ^ (uint64_t)K);
      if (hit != c->memo.end()) dh = reinterpret_cast<const void*>(hit->second[0]);
      else {
        std::vector<float2> twH((size_t)K);
        for (int n = 0; n < K; ++n) {
          const double ang = 6.283185307179586476925286766559 * (double)n / (double)(2 * K);
          twH[n] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
        int rc4 = ctx_table(c, 0x7749ull ^ (uint64_t)K, twH.data(), twH.size() * sizeof(float2), &dh);
        if (rc4) return rc4;
        c->memo[0x774900000000ull ^ (uint64_t)K] = {reinterpret_cast<uint64_t>(dh)};
      }
    }
    a.twH = reinterpret_cast<const v2f*>(dh);
    const size_t lds = (size_t)(2 * K) * 4 + 256 * 8 + (size_t)R3 * 256 * 8 + (size_t)K * 8 + (size_t)W * XCH * 8;
    dispatch_note("istft.dbl");
    if (s.has_scale) hipLaunchKernelGGL((k_istft_wave_dbl<K, R, true, W>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    else hipLaunchKernelGGL((k_istft_wave_dbl<K, R, false, W>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
  } else {
    a.twH = nullptr;
    const size_t lds = (size_t)K * 4 + 256 * 8 + (size_t)R3 * 256 * 8 + (size_t)W * XCH * 8;
    // the spectrogram is read once (5 % run halos aside): non-temporal loads (+2…5 % in interleaved A/B runs, round 2; the
    // default-policy instantiation went with its switch in round 4)
    dispatch_note(s.filt ? "istft.wave.filt" : (deep ? "istft.wave.deep" : "istft.wave"));
    if (s.filt) {
      if (s.has_scale) hipLaunchKernelGGL((k_istft_wave<K, R, true, W, true, true>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
      else hipLaunchKernelGGL((k_istft_wave<K, R, false, W, true, true>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    } else if (deep) {
      if (s.has_scale) hipLaunchKernelGGL((k_istft_wave<K, R, true, W, false, true, true>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
      else hipLaunchKernelGGL((k_istft_wave<K, R, false, W, false, true, true>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    } else if (s.has_scale) hipLaunchKernelGGL((k_istft_wave<K, R, true, W, false, true>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    else hipLaunchKernelGGL((k_istft_wave<K, R, false, W, false, true>), dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
  }
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}


template <int J, int R>
static int launch_istft_wave_quad(Ctx* c, const IstftLaunch& s, const float* window_host) {
  constexpr int K = 1024, W = 6, R3 = K / 256, XCH = K + K / 16 + 16, NJ = K / J, HOP = NJ / R;
  constexpr int CPAD = (NJ - HOP) > 0 ? (NJ - HOP) : 2;
  IstftWaveArgs a;
  a.z = reinterpret_cast<const v2f*>(s.z); a.M = s.M; a.batch = s.batch; a.hop = s.hop;
  a.segs_per_row = s.M + R - 1;
  a.wtab = s.window;
  Ctx::WaveTables& wt = c->wave_tables[K];
  if (!wt.twB) return NXSIG_ERR_UNSUPPORTED;
  a.twB = reinterpret_cast<const v2f*>(wt.twBi);
  a.twC = reinterpret_cast<const v2f*>(wt.twCi);
  a.twH = reinterpret_cast<const v2f*>(wt.twQ[J == 4 ? 1 : 2]);  // conj(w_K^(j k0)); the kernel conjugates on load
  a.scale = s.scale_mul;
  { int rc3 = istft_den_table(c, R, s.hop, window_host, &a.den); if (rc3) return rc3; }
  a.y = reinterpret_cast<v2f*>(s.y);
  void* dummy = nullptr;
  { int rc2 = ctx_scratch(c, 3, (size_t)8192 * sizeof(float2), &dummy); if (rc2) return rc2; }
  a.dummy = reinterpret_cast<v2f*>(dummy);
  const int64_t units_per_row = (a.segs_per_row + J - 1) / J;
  const int64_t total_units = units_per_row * s.batch;
  const int waves_per_cu = tune(c, kT_ISTFT_RUNS_PER_CU, 12);  // = the resident waves per CU (round 3: 3.63 / 3.38 TB/s against 3.47 / 3.33 at 24
                                                                    // for N = 256 / 128, 8 x 60 s; a two-units-ahead prefetch like k_istft_wave's
                                                                    // DEEP form measured -3 ... +3 % here and was not kept)
  int64_t run_len = (total_units + (int64_t)c->num_cus * waves_per_cu - 1) / ((int64_t)c->num_cus * waves_per_cu);
  const int min_run = istft_min_run(c, total_units, (int64_t)c->num_cus * waves_per_cu, 8);
  if (run_len < min_run) run_len = min_run;
  a.run_len = run_len;
  a.runs_per_row = (units_per_row + run_len - 1) / run_len;
  a.total_runs = a.runs_per_row * s.batch;
  const int64_t blocks = (a.total_runs + W - 1) / W;
  const size_t lds = (size_t)NJ * 4 + 256 * 8 + (size_t)R3 * 256 * 8 + (size_t)(J - 1) * NJ * 8 + (size_t)W * XCH * 8 + (size_t)W * CPAD * 8;
  { int rcl = istft_nf_list(c, a.total_runs * (run_len + (R - 1 + J - 1) / J + 1), &a.nf_list); if (rcl) return rcl; }
  s.nf_list = a.nf_list; s.nf_frames_per_unit = J;
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note("istft.quad");
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  // register overlap-add by default (round 4, interleaved A/B on one box, 8 x 60 s: N = 256 hop 64 / 32: +9.5 / +8.8 %, N = 128 hop
  // 32 / 16 / 64: +17.6 / +9.5 / +5.4 %); N = 256 at hop 128 keeps the LDS form (-3 % for the register form)
  if (!tune(c, kT_ISTFT_REGOLA, (J == 4 && R == 2) ? 0 : 1))
    return s.has_scale ? go(k_istft_wave_quad<K, J, R, true, W, false>) : go(k_istft_wave_quad<K, J, R, false, W, false>);
  return s.has_scale ? go(k_istft_wave_quad<K, J, R, true, W>) : go(k_istft_wave_quad<K, J, R, false, W>);
}

template <int R>
static int launch_istft_wave_4k(Ctx* c, const IstftLaunch& s, const float* window_host) {
  constexpr int K = 1024, W = 4, XCH = K + K / 16 + 16, NF = 4096;
  IstftWaveArgs a;
  a.z = reinterpret_cast<const v2f*>(s.z); a.M = s.M; a.batch = s.batch; a.hop = s.hop;
  a.segs_per_row = s.M + R - 1;
  a.wtab = s.window;  // raw window, N == K == 4096
  Ctx::WaveTables& wt = c->wave_tables[K];
  if (!wt.twB) return NXSIG_ERR_UNSUPPORTED;
  a.twB = reinterpret_cast<const v2f*>(wt.twBi);
  a.twC = reinterpret_cast<const v2f*>(wt.twCi);
  a.scale = s.scale_mul;
  { int rc3 = istft_den_table(c, R, s.hop, window_host, &a.den); if (rc3) return rc3; }
  a.y = reinterpret_cast<v2f*>(s.y);
  void* dummy = nullptr;
  { int rc2 = ctx_scratch(c, 3, (size_t)8192 * sizeof(float2), &dummy); if (rc2) return rc2; }
  a.dummy = reinterpret_cast<v2f*>(dummy);
  {
    const void* d4 = nullptr;
    auto hit = c->memo.find(0x8B1400000000ull ^ (uint64_t)K);   // built once per context
    if (hit != c->memo.end()) d4 = reinterpret_cast<const void*>(hit->second[0]);
    else {
      std::vector<float2> t4(K);
      for (int k = 0; k < K; ++k) {
        const double ang = 6.283185307179586476925286766559 * (double)k / 4096.0;  // conj(w_4096^k)
        t4[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
      }
      int rc4 = ctx_table(c, 0x8B14ull, t4.data(), t4.size() * sizeof(float2), &d4);
      if (rc4) return rc4;
      c->memo[0x8B1400000000ull ^ (uint64_t)K] = {reinterpret_cast<uint64_t>(d4)};
    }
    a.twH = reinterpret_cast<const v2f*>(d4);
  }
  const int64_t total_segs = a.segs_per_row * s.batch;
  const int waves_per_cu = 4;  // one wave per SIMD
  int64_t run_len = (total_segs + (int64_t)c->num_cus * waves_per_cu - 1) / ((int64_t)c->num_cus * waves_per_cu);
  const int min_run = istft_min_run(c, total_segs, (int64_t)c->num_cus * waves_per_cu, 8);
  if (run_len < min_run) run_len = min_run;
  a.run_len = run_len;
  a.runs_per_row = (a.segs_per_row + run_len - 1) / run_len;
  a.total_runs = a.runs_per_row * s.batch;
  const int64_t blocks = (a.total_runs + W - 1) / W;
  const size_t lds = (size_t)NF * 4 + 256 * 8 + (size_t)4 * 256 * 8 + (size_t)K * 8 + (size_t)W * XCH * 8;
  auto go = [&](auto kernel) -> int {
    NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note("istft.4k");
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  return s.has_scale ? go(k_istft_wave_4k<R, true, W>) : go(k_istft_wave_4k<R, false, W>);
}

int launch_istft_r20(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled);  // kernels_wave_r20.hip
int launch_istft_rab(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled);  // kernels_wave_rab.hip
bool rab_inverse_only(int K);                                                                  // kernels_wave_rab.hip

static int launch_istft_wave_tuned(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled) {
  *handled = false;
  if (s.M == 0 || s.batch == 0) return NXSIG_OK;
  if (tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  if (window_host == nullptr) return NXSIG_OK;
  // a spectrum filter is fused into the N = 1024 kernel only; elsewhere the caller multiplies first (launch_istft)
  if (s.filt && (s.K != 1024 || tune(c, kT_DISABLE_FUSED_FILTER, 0))) return NXSIG_OK;
  if (s.K == 512 && s.N == 512) {  // two frames per 1024-point inverse FFT
    if (s.hop != 64 && s.hop != 128 && s.hop != 256 && s.hop != 512) return NXSIG_OK;
    if (s.M < 2 * (512 / s.hop) - 1) return NXSIG_OK;
    int rc5 = ensure_wave_tables_1024(c);
    if (rc5) return rc5;
    *handled = true;
    switch (512 / s.hop) {
      case 1: return launch_istft_wave_R<1, 4, true>(c, s, s.window, window_host);
      case 2: return launch_istft_wave_R<2, 4, true>(c, s, s.window, window_host);
      case 4: return launch_istft_wave_R<4, 4, true>(c, s, s.window, window_host);
      default: return launch_istft_wave_R<8, 4, true>(c, s, s.window, window_host);
    }
  }
  if (s.K == 400 && s.N == 400) {  // native 20 x 20 inverse (kernels_wave_r20.hip); declines odd hops / short inputs
    int rc20 = launch_istft_r20(c, s, window_host, handled);
    if (rc20 || *handled) return rc20;
  }
  if (s.K == s.N && rab_length_part(s.K) >= 0) {  // A x B inverses (kernels_wave_rab*.hip), same conditions
    int rcab = launch_istft_rab(c, s, window_host, handled);
    if (rcab || *handled) return rcab;
  }
  if ((s.K == 256 && s.N == 256) || (s.K == 128 && s.N == 128)) {  // 4 / 8 frames per 1024-point inverse FFT
    const int R = s.N / s.hop;
    if (s.hop * R != s.N || (R != 1 && R != 2 && R != 4 && R != 8)) return NXSIG_OK;
    if (s.M < 2 * R - 1) return NXSIG_OK;
    int rc5 = ensure_wave_tables_1024(c);
    if (rc5) return rc5;
    *handled = true;
    if (s.K == 256) {
      switch (R) {
        case 1: return launch_istft_wave_quad<4, 1>(c, s, window_host);
        case 2: return launch_istft_wave_quad<4, 2>(c, s, window_host);
        case 4: return launch_istft_wave_quad<4, 4>(c, s, window_host);
        default: return launch_istft_wave_quad<4, 8>(c, s, window_host);
      }
    }
    switch (R) {
      case 1: return launch_istft_wave_quad<8, 1>(c, s, window_host);
      case 2: return launch_istft_wave_quad<8, 2>(c, s, window_host);
      case 4: return launch_istft_wave_quad<8, 4>(c, s, window_host);
      default: return launch_istft_wave_quad<8, 8>(c, s, window_host);
    }
  }
  if (s.K == 2048 && s.N == 2048) {  // one frame per two 1024-point inverse FFTs
    if (s.hop != 256 && s.hop != 512 && s.hop != 1024 && s.hop != 2048) return NXSIG_OK;
    if (s.M < 2 * (2048 / s.hop) - 1) return NXSIG_OK;
    int rc5 = ensure_wave_tables_1024(c);
    if (rc5) return rc5;
    *handled = true;
    switch (2048 / s.hop) {
      case 1: return launch_istft_wave_R<1, 4, false, true>(c, s, s.window, window_host);
      case 2: return launch_istft_wave_R<2, 4, false, true>(c, s, s.window, window_host);
      case 4: return launch_istft_wave_R<4, 4, false, true>(c, s, s.window, window_host);
      default: return launch_istft_wave_R<8, 4, false, true>(c, s, s.window, window_host);
    }
  }
  if (s.K == 4096 && s.N == 4096 && !tune(c, kT_DISABLE_4K, 0)) {  // four passes through the 1024-point inverse core per frame
    if (s.hop != 512 && s.hop != 1024 && s.hop != 2048 && s.hop != 4096) return NXSIG_OK;
    if (s.M < 2 * (4096 / s.hop) - 1) return NXSIG_OK;
    int rc5 = ensure_wave_tables_1024(c);
    if (rc5) return rc5;
    *handled = true;
    switch (4096 / s.hop) {
      case 1: return launch_istft_wave_4k<1>(c, s, window_host);
      case 2: return launch_istft_wave_4k<2>(c, s, window_host);
      case 4: return launch_istft_wave_4k<4>(c, s, window_host);
      default: return launch_istft_wave_4k<8>(c, s, window_host);
    }
  }
  if (s.K != 1024 || s.N != 1024) return NXSIG_OK;       // other sizes: generic two-stage path
  if (s.hop != 128 && s.hop != 256 && s.hop != 512 && s.hop != 1024) return NXSIG_OK;
  if (s.M < 2 * (1024 / s.hop) - 1) return NXSIG_OK;  // head and tail rows must not overlap
  int rc = ensure_wave_tables_1024(c);
  if (rc) return rc;
  *handled = true;
  // the window table of the istft is the raw window (N == K)
  switch (1024 / s.hop) {
    case 1: return launch_istft_wave_R<1, 4>(c, s, s.window, window_host);
    case 2: return launch_istft_wave_R<2, 4>(c, s, s.window, window_host);
    case 4: return launch_istft_wave_R<4, 4>(c, s, s.window, window_host);
    default: return launch_istft_wave_R<8, 4>(c, s, s.window, window_host);
  }
}

// the tuned inverse kernels; power-of-two frame lengths whose hop they do not take (N / hop not in {1, 2, 4, 8}, e.g. 512-sample frames
// every 160 samples) go to the two-pass A x B inverses instead of the generic path
int launch_istft_wave(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled) {
  int rc = launch_istft_wave_tuned(c, s, window_host, handled);
  if (rc || *handled) return rc;
  if (s.M == 0 || s.batch == 0 || window_host == nullptr || s.filt) return NXSIG_OK;
  if (s.K == s.N && rab_inverse_only(s.K)) return launch_istft_rab(c, s, window_host, handled);
  return NXSIG_OK;
}

// spectrum of the zero-padded taps in double (radix-2, 1024 points, once per distinct filter)
static void host_fft1024_f64(std::vector<double>& re, std::vector<double>& im) {
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

template <int W, int K = 1024, bool R2K = false>
static int launch_fir_wave_W(Ctx* c, const FirLaunch& s_in, bool* handled) {
  *handled = false;
  constexpr int R3 = K / 256, XCH = K + K / 16 + 16;
  if (s_in.out_len <= 0 || s_in.batch == 0) return NXSIG_OK;
  if (tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  if (s_in.taps > K / 2 + 1) return NXSIG_OK;  // a block would be < 50 % efficient: the generic path uses bigger blocks
  // The block geometry may treat the filter as LONGER than it is: trailing zero taps change nothing (the spectrum H is that of
  // the taps zero-padded to K either way), only the valid part of a block shrinks.  Rounding taps - 1 up to a multiple of 32
  // (128 on the 2048-point blocks) gives EVERY filter length the streaming kernels instead of the bounds-checked edge path.
  const int taps_h = s_in.taps;
  FirLaunch s = s_in;
  // ... and LEADING zero taps delay the result by their number: the filter is treated as `lead` zeros + h and the requested slice
  // starts `lead` outputs later.  On the 2048-point blocks (no 4-byte kernel there) this makes out_start a multiple of 4 — the grid
  // phase then keeps x 16-byte aligned — for the tap counts whose :same / :valid offset is not: every even count (400, 512, 600, 900
  // taps ran on the bounds-checked kernel throughout: 0.19-0.26 of the roofline against 0.45-0.51 for 449 / 513 / 769)
  int lead = 0;
  if (K == 2048 && tune(c, kT_FIR_PAD_TAPS, 1)) {
    // (x itself may start off a 16-byte boundary — a sliced tensor: its offset joins the slice offset)
    const int64_t xoff = (int64_t)((reinterpret_cast<uintptr_t>(s_in.x) >> 2) & 3);
    lead = (int)((4 - (s_in.out_start + xoff) % 4) % 4);
    const int q0 = R2K ? 256 : 128;
    if (((s_in.taps + lead - 1 + q0 - 1) / q0) * q0 + 1 > K / 2 + 1) lead = 0;
  }
  s.taps = s_in.taps + lead;
  s.out_start = s_in.out_start + lead;
  {
    const int q = R2K ? 256 : (K == 1024 ? 32 : 128);   // (R2K: whole 256-sample slots of the real-block kernel)
    const int eff = ((s.taps - 1 + q - 1) / q) * q + 1;
    if (eff <= K / 2 + 1 && tune(c, kT_FIR_PAD_TAPS, 1)) s.taps = eff;
  }
  int rc = ensure_wave_tables(c, K);
  if (rc) return rc;
  *handled = true;
  const void* Hd = nullptr;
  const uint64_t hkey = fnv1a(0xF1B0ull ^ ((uint64_t)K << 32), s.h_host, (size_t)taps_h * sizeof(float)) ^ (uint64_t)taps_h ^ ((uint64_t)lead << 20);
  const void *Ad = nullptr, *Bd = nullptr;
  auto hit = c->memo.find(hkey);
  if (hit != c->memo.end() && (!R2K || hit->second.size() >= 3)) {
    Hd = reinterpret_cast<const void*>(hit->second[0]);  // same taps as an earlier call: no host FFT
    if (R2K) { Ad = reinterpret_cast<const void*>(hit->second[1]); Bd = reinterpret_cast<const void*>(hit->second[2]); }
  } else {
    std::vector<double> re(K, 0.0), im(K, 0.0);
    for (int i = 0; i < taps_h; ++i) re[lead + i] = (double)s.h_host[i];
    host_fft1024_f64(re, im);
    std::vector<float2> H(K);
    for (int i = 0; i < K; ++i) H[i] = make_float2((float)(re[i] / K), (float)(im[i] / K));
    rc = ctx_table(c, 0xF1A1ull ^ (uint64_t)K, H.data(), H.size() * sizeof(float2), &Hd);
    if (rc) return rc;
    c->memo[hkey] = {reinterpret_cast<uint64_t>(Hd)};
    if constexpr (R2K) {   // fused untangle x H x re-tangle coefficients of k_fir_r2k (K == 2048 here), in double
      std::vector<float2> A(1024), B(1024);
      for (int k = 0; k < 1024; ++k) {
        const double sr = 0.5 * (re[k] + re[k + 1024]), si = 0.5 * (im[k] + im[k + 1024]);
        const double dr = 0.5 * (re[k] - re[k + 1024]), di = 0.5 * (im[k] - im[k + 1024]);
        const double th = 6.283185307179586476925286766559 * (double)k / 2048.0, sn = std::sin(th), cs = std::cos(th);
        A[k] = make_float2((float)((sr - dr * sn) / 1024.0), (float)((si - di * sn) / 1024.0));
        B[k] = make_float2((float)(-di * cs / 1024.0), (float)(dr * cs / 1024.0));   // i D cos th
      }
      if ((rc = ctx_table(c, 0xF1A2ull, A.data(), A.size() * sizeof(float2), &Ad))) return rc;
      if ((rc = ctx_table(c, 0xF1A3ull, B.data(), B.size() * sizeof(float2), &Bd))) return rc;
      c->memo[hkey] = {reinterpret_cast<uint64_t>(Hd), reinterpret_cast<uint64_t>(Ad), reinterpret_cast<uint64_t>(Bd)};
    }
  }
  FirWaveArgs a;
  a.V = K - (s.taps - 1);
  // Grid phase: block b covers full-convolution outputs [b V + phase, (b + 1) V + phase) with phase = out_start mod 32, so every
  // block's first output is y[multiple of 32]: the streaming stores (128-byte runs per half wave / 512-byte runs per wave, "sc1 nt")
  // hit whole cache lines when the rows of y are 128-byte aligned.  With mode :same, out_start = div(taps - 1, 2) is rarely
  // aligned by itself, and partial-line streaming writes cost a factor 2 (measured: 255 taps :same 2.2 -> 4.4 TB/s).  The shift
  // is applied by moving the signal's origin: x' = x + phase, valid indices [-phase, L - phase), out_start' = out_start - phase.
  const int64_t phase = tune(c, kT_FIR_PHASE, 1) ? (s.out_start % 32) : 0;
  const int64_t out_start = s.out_start - phase;
  a.x = s.x + phase; a.xlo = -phase; a.xhi = s.L - phase;
  a.L = s.L; a.batch_stride = s.batch_stride; a.batch = s.batch; a.taps = s.taps;
  // rows that do not start on a 128-byte boundary of y (row length not a multiple of 32, more than one row): per-row grid phase, see
  // FirWaveArgs::row_mod.  A row's grid moves by up to 31 samples: the block range of the launch covers every shift (one more block in
  // front), and a pair is interior only if it is so for every shift
  const int row_mod = (tune(c, kT_FIR_PHASE, 1) && s.batch > 1) ? (int)(s.out_len % 32) : 0;
  const int64_t margin = row_mod ? 31 : 0;
  a.row_mod = row_mod;
  a.first_block = out_start - margin >= 0 ? (out_start - margin) / a.V : -1;
  const int64_t last_block = (out_start + s.out_len - 1) / a.V;
  a.nblocks = last_block - a.first_block + 1;
  a.pairs_per_row = (a.nblocks + 1) / 2;
  a.out_start = out_start; a.out_len = s.out_len;
  a.H = reinterpret_cast<const v2f*>(Hd);
  Ctx::WaveTables& wt = c->wave_tables[K];
  a.twB = reinterpret_cast<const v2f*>(wt.twB);
  a.twC = reinterpret_cast<const v2f*>(wt.twC);
  a.y = s.y; a.row_flags = s.row_flags;
  // 8-byte vector access needs every offset even: taps-1 multiple of 128 (=> V even), even strides, aligned bases
  // (with the per-row phase every row of y starts a line, and row r of x sits r (stride - out_len) samples off one)
  const bool rows2 = row_mod ? (s.batch_stride - s.out_len) % 2 == 0 : (s.batch_stride % 2 == 0 && s.out_len % 2 == 0);
  const bool rows4 = row_mod ? (s.batch_stride - s.out_len) % 4 == 0 : (s.batch_stride % 4 == 0 && s.out_len % 4 == 0);
  const bool fast8 = ((s.taps - 1) % 128 == 0) && rows2 && (out_start % 2 == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.x) & 7) == 0) && ((reinterpret_cast<uintptr_t>(s.y) & 7) == 0);
  // the 32 x 32 kernel (kernels_wave_fir32.hip, 4-byte accesses: no alignment conditions) takes what the 8-byte kernel cannot;
  // where both apply the 8-byte kernel is ~4 % faster (NXSIG_FIR32: 0 never, 1 when needed, 2 always)
  const int fir32_mode = tune(c, kT_FIR32, 1);
  const bool use32 = K == 1024 && (s.taps - 1) % 32 == 0 && (fir32_mode == 2 || (fir32_mode == 1 && !fast8));
  const bool fast = use32 || fast8;
  // interior pairs pb in [pb_lo, pb_hi): block pair (b1, b1+1), b1 = first_block + 2 pb, reads x'[b1 V - (taps-1) .. +V+K)
  // inside [xlo, xhi) and writes y[b1 V - out_start' .. + 2V) inside [0, out_len)
  a.pb_lo = 0; a.pb_hi = 0;
  if (fast) {
    const int64_t tm1 = s.taps - 1;
    int64_t lo = 0;
    while (lo < a.pairs_per_row) {
      const int64_t b1 = a.first_block + 2 * lo;
      if (b1 * a.V - tm1 >= a.xlo && b1 * a.V - a.out_start >= 0) break;
      ++lo;
    }
    int64_t hi = a.pairs_per_row;
    while (hi > lo) {
      const int64_t b1 = a.first_block + 2 * (hi - 1);
      const bool have2 = (b1 + 1 - a.first_block) < a.nblocks;
      if (have2 && b1 * a.V - tm1 + a.V + K <= a.xhi - margin && b1 * a.V - a.out_start + 2 * a.V + margin <= s.out_len) break;
      --hi;
    }
    a.pb_lo = lo; a.pb_hi = hi;
  }
  const bool hreg = K == 1024 && W <= 8 && tune(c, kT_FIR_HREG, 1);  // 156 VGPRs: three waves per SIMD
  const size_t lds = 256 * 8 + (size_t)R3 * 256 * 8 + (hreg ? 0 : (size_t)K * 8) + (size_t)W * XCH * 8;
  auto launch = [&](bool stream, int64_t units_per_row) -> int {
    if (units_per_row <= 0) return NXSIG_OK;
    a.units_per_row = units_per_row;
    a.total_units = units_per_row * s.batch;
    const int units_per_wave = c->tuning.set[kT_FIR_UNITS_PER_WAVE] ? tune(c, kT_FIR_UNITS_PER_WAVE, 8) : fill_units_per_wave(c, a.total_units, W, 8);  // short chunks, many workgroups (see launch_wave); small launches spread out
    a.chunk = (int64_t)W * (units_per_wave < 1 ? 1 : units_per_wave);
    // the edge launch has few, slow (bounds-checked) units: one per wave so that they all run concurrently, unless every pair
    // goes through it (filters whose taps - 1 is not a multiple of 128)
    // (many short rows: up to two such workgroups per CU before a wave takes a second unit — 4 096 edge pairs of 2 048 one-second rows
    // ran as 128 workgroups of 8 units per wave)
    if (!stream) {
      int64_t per_wave = (a.total_units + (int64_t)c->num_cus * W * 2 - 1) / ((int64_t)c->num_cus * W * 2);
      if (per_wave < 1) per_wave = 1;
      if (per_wave < units_per_wave) a.chunk = (int64_t)W * per_wave;
    }
    const int64_t blocks = (a.total_units + a.chunk - 1) / a.chunk;
    if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "fir: signal too long for one launch");
    hipError_t attr_rc = hipSuccess;
    auto fire = [&](auto kernel) {   // kernels of the 2048-point blocks need > 64 KB of dynamic LDS: opt in per instantiation
      if (lds > 64 * 1024) attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      dispatch_note(K == 1024 ? (stream ? "fir.pair" : "fir.pair.edge") : (stream ? "fir.pair2k" : "fir.pair2k.edge"));
      if (attr_rc == hipSuccess) hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    };
    // stream launches: taps - 1 is a multiple of 128 (fast8): K = 1024 serves 128 .. 512, K = 2048 (taps > 513) 640 .. 1024
    const int tq = stream ? (s.taps - 1) / 128 : 0;
    constexpr int TQ0 = K == 1024 ? 1 : 5;
    auto go_stream = [&](auto hreg_c) {
      constexpr bool HR = decltype(hreg_c)::value;
      switch (tq - TQ0) {
        case 0: fire(k_fir_wave<K, true, W, HR, TQ0>); break;
        case 1: fire(k_fir_wave<K, true, W, HR, TQ0 + 1>); break;
        case 2: fire(k_fir_wave<K, true, W, HR, TQ0 + 2>); break;
        case 3: fire(k_fir_wave<K, true, W, HR, TQ0 + 3>); break;
        default: fire(k_fir_wave<K, true, W, HR>); break;
      }
    };
    if (hreg) {
      if (stream) go_stream(std::integral_constant<bool, K == 1024>{});
      else fire(k_fir_wave<K, false, W, K == 1024>);
    } else if (stream) {
      if (K == 1024) fire(k_fir_wave<K, true, W>);   // (NXSIG_FIR_HREG=0: run-time tap count)
      else go_stream(std::false_type{});
    } else fire(k_fir_wave<K, false, W>);
    NXSIG_HIP_TRY(attr_rc);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  if constexpr (R2K) {
    // interior pairs of the 2048-block grid, ONE block per unit, on the real-block kernel (16-byte accesses: rows, offsets and V
    // multiples of 4 samples); anything else falls back to this grid's pair kernel below
    const bool al16 = fast8 && (s.taps - 1) % 256 == 0 && s.taps - 1 >= 256 && s.taps - 1 <= 1024 && rows4 && (out_start % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(s.y) & 15) == 0) && a.pb_hi > a.pb_lo;
    if (al16) {
      int rc1 = ensure_wave_tables(c, 1024);
      if (rc1) return rc1;
      Ctx::WaveTables& w1 = c->wave_tables[1024];
      FirWaveArgs b = a;
      b.twB = reinterpret_cast<const v2f*>(w1.twB); b.twC = reinterpret_cast<const v2f*>(w1.twC);
      b.coefA = reinterpret_cast<const v2f*>(Ad); b.coefB = reinterpret_cast<const v2f*>(Bd);
      constexpr int W4 = 4;
      b.units_per_row = 2 * (a.pb_hi - a.pb_lo);
      b.total_units = b.units_per_row * s.batch;
      b.chunk = (int64_t)W4 * (c->tuning.set[kT_FIR_UNITS_PER_WAVE] ? tune(c, kT_FIR_UNITS_PER_WAVE, 8) : fill_units_per_wave(c, b.total_units, W4, 8));
      const int64_t blocks = (b.total_units + b.chunk - 1) / b.chunk;
      if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "fir: signal too long for one launch");
      const size_t lds4 = 256 * 8 + (size_t)4 * 256 * 8 + (size_t)1024 * 8 + (size_t)W4 * (1024 + 64 + 16) * 8;
      dispatch_note("fir.r2k");
      switch ((s.taps - 1) / 256) {
        case 1: hipLaunchKernelGGL((k_fir_r2k<W4, 1>), dim3((unsigned)blocks), dim3(64 * W4), lds4, c->stream, b); break;
        case 2: hipLaunchKernelGGL((k_fir_r2k<W4, 2>), dim3((unsigned)blocks), dim3(64 * W4), lds4, c->stream, b); break;
        case 3: hipLaunchKernelGGL((k_fir_r2k<W4, 3>), dim3((unsigned)blocks), dim3(64 * W4), lds4, c->stream, b); break;
        default: hipLaunchKernelGGL((k_fir_r2k<W4, 4>), dim3((unsigned)blocks), dim3(64 * W4), lds4, c->stream, b); break;
      }
      NXSIG_HIP_TRY(hipGetLastError());
      return launch(false, a.pairs_per_row - (a.pb_hi - a.pb_lo));
    }
  }
  if (use32) {  // interior pairs two at a time on the 32 x 32 kernel (kernels_wave_fir32.hip); an odd leftover joins the edge pairs
    a.pb_hi -= (a.pb_hi - a.pb_lo) & 1;
    rc = launch_fir_wave32(c, a.x, s.batch_stride, s.batch, s.taps, a.first_block, a.pb_lo, (a.pb_hi - a.pb_lo) / 2, a.out_start, s.out_len,
                           reinterpret_cast<const float2*>(Hd), s.y, s.row_flags, row_mod);
  } else {
    rc = launch(true, a.pb_hi - a.pb_lo);
  }
  if (rc) return rc;
  return launch(false, a.pairs_per_row - (a.pb_hi - a.pb_lo));
}

int launch_fir_wave(Ctx* c, const FirLaunch& s, bool* handled) {
  // 2048-sample blocks (32 points per lane, six waves per workgroup) for 514..1025 taps; four-wave workgroups on 1024-sample
  // blocks otherwise (7- and 14-wave workgroups measured slower in round 2 and went with their switch in round 4)
  // real 2048-sample blocks on the 1024-point core (k_fir_r2k, round 5) from 386 taps on: interleaved A/B on config 5's shard
  // (tools/sweep_fir.py NXSIG_FIR_R2K 0 2 with SWEEP_TAPS): 129 / 193 / 257 taps -11 / -2 / -5 %, 289 / 321 / 385 +2 / +3 / -1 %,
  // 449 / 513 / 641 / 769 / 1025 taps +20 / +15 / +9 / +22 / +22 %.  NXSIG_FIR_R2K: 0 never, 1 (default) from 386 taps, 2 whenever it applies
  const int r2k = tune(c, kT_FIR_R2K, 1);
  if (s.taps <= 1025 && (r2k == 2 || (r2k == 1 && s.taps > 385))) return launch_fir_wave_W<6, 2048, true>(c, s, handled);
  if (s.taps > 513) return launch_fir_wave_W<6, 2048>(c, s, handled);
  return launch_fir_wave_W<4>(c, s, handled);
}

}  // namespace nxsig
