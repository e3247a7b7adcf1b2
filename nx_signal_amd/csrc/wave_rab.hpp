// Wave kernels, part 5 (round 5), templates: composite fft lengths K = A x B computed natively (NxSignal.stft / istft,
// lib/nx_signal.ex:94-102 and :609-637 with fft_length: K) — the 10 / 20 / 30 / 40 ms frames of 16 / 32 / 48 kHz audio (320, 480, 640,
// 960) first, then every length the codelets of small_dft.hpp reach (table below).  The Bluestein kernel pays two 1024- or 2048-point
// transforms per frame pair for such lengths (0.08-0.15 of the HBM roofline) and lengths above 1024 fell to the generic workgroup-per-
// frame kernels (0.02); this is kernels_wave_r20.hip's two-pass scheme with the two factors free:
//
//   one K = A B point complex FFT on max(A, B) lanes: Cooley-Tukey n = B n1 + n2, k = k1 + A k2
//   raw samples of the unit's 2 T frames (one contiguous span) -> LDS                       16-byte loads when aligned and inside
//   pass A   lane n2 < B: DFT_A over n1 of u[B n1 + n2] (frame slice x window fused, :94-101), x W_K^(n2 k1)
//   A x B transpose through LDS (row stride B + 1)
//   pass B   lane k1 < A: DFT_B over n2 -> U[k1 + A k2]
//   natural-order U in LDS -> untangle XA = (U + conj U') / 2, XB = -i (U - conj U') / 2 -> 16-byte stores (:129)
//
// T = 64 / max(A, B) transforms per wave, two real frames per transform as re / im, the small DFTs are register codelets (small_dft.hpp).
// Sinks as in kernels_wave_r20.hip: complex spectrum, fused log-mel, magnitude / power / dBFS / one-sided rows.
// Measured (tools/bench_configs.py gen<N>, 16 rows, 1.7 GB of output): 320 / 480 / 640 / 960 at 0.67 / 0.63 / 0.60 / 0.61 of 8 TB/s.
// The inverses: k_istft_rab below.  The kernels are instantiated by kernels_wave_rab.hip (part 0 + the dispatchers) and
// kernels_wave_rab_p1.hip ... _p3.hip (one list each, so that the translation units compile side by side).
#pragma once
#include "small_dft.hpp"

// (K, A, B) of every native composite length, one list per translation unit.  K % 4 == 0 (bin pairs of the one-sided sinks).
#define NXSIG_RAB_PART0(X) X(320, 16, 20) X(480, 24, 20) X(640, 32, 20) X(960, 32, 30)
#define NXSIG_RAB_PART1(X) X(32, 8, 4) X(64, 8, 8) X(100, 10, 10) X(120, 12, 10) X(160, 16, 10) X(200, 20, 10) X(240, 16, 15) X(300, 20, 15) X(360, 24, 15) X(384, 24, 16)
#define NXSIG_RAB_PART2(X) X(500, 25, 20) X(600, 30, 20) X(720, 30, 24) X(768, 32, 24) X(800, 32, 25) X(900, 30, 30)
#define NXSIG_RAB_PART3(X) X(1000, 40, 25) X(1200, 40, 30) X(1280, 40, 32) X(1600, 40, 40)
#define NXSIG_RAB_PART5(X) X(192, 16, 12) X(288, 24, 12) X(576, 24, 24) X(1152, 48, 24) X(1440, 48, 30) X(1536, 48, 32) X(1920, 48, 40)
// round 6: radix 7 (the 20 / 40 ms frames of 44.1 kHz audio) and the 50 / 60 / 80 ms frames of 48 kHz audio; complex-spectrum sink only
// (log-mel / magnitude sinks of these lengths take the two-step path).  882 % 4 == 2: fine for the spectrum sink's bin pairs; 441 is ODD:
// single bins and 8-byte accesses in the drains, 8-byte spectrum loads in the inverse.
#define NXSIG_RAB_PART6(X) X(441, 21, 21) X(882, 42, 21) X(1764, 42, 42) X(2205, 35, 63)
#define NXSIG_RAB_PART7(X) X(2400, 50, 48) X(2880, 60, 48) X(3840, 64, 60)
// inverse only: power-of-two frame lengths with a hop the N / hop in {1, 2, 4, 8} kernels of kernels_wave.hip do not take (e.g. 512 / 160)
#define NXSIG_RAB_INVERSE_ONLY(X) X(128, 16, 8) X(256, 16, 16) X(512, 32, 16) X(1024, 32, 32)

namespace nxsig {

// Row stride of the transposed block in complex cells: ODD, so that the lanes of pass B (lane k1 reads row k1) start in different banks.
// B + 1 for an even B; an odd B took B + 1 too until round 6 — 16 cells for B = 15 (240 / 300 / 360): every second lane of a half-wave on
// the same bank, 480 cycles for the 30 reads of pass B instead of 30 (tools/lds_bank_model.py) — now B + 2.
constexpr int rab_bp(int B) { return B + ((B & 1) ? 2 : 1); }

// LDS tables in front of the waves' buffers: window (floats) and twiddles (complex cells), each region padded to 16 bytes.  Until round
// 6's second pass they were K floats and K cells: for K = 441 the twiddles and EVERY wave buffer behind them started 4 bytes off an
// 8-byte boundary — every ds_read / write_b64 of the kernel unaligned, 52 LDS cycles per instruction instead of 5 — and for K = 882 the
// buffers sat 8 bytes off a 16-byte boundary (unaligned b128 accesses in the untangle: 9.6 cycles).
constexpr int rab_wcells(int K) { return (K + 3) & ~3; }
constexpr int rab_tcells(int K) { return (K + 1) & ~1; }
constexpr int rab_tab_bytes(int K) { return 4 * rab_wcells(K) + 8 * rab_tcells(K); }

// Forward kernels of the 50- / 60- / 64-point lengths: the window stays in global memory (L1 / L2 hits, read once per unit beside the
// staging loads) when the 4 K bytes it frees in LDS buy another wave: 3840 = 64 x 60 runs 4 instead of 3
#ifndef NXSIG_RAB_WG
#define NXSIG_RAB_WG 1
#endif
constexpr bool rab_wg(int A, int B) {
  const int KB = A * B, LT = A > B ? A : B, T = 64 / LT, TRS = A * rab_bp(B);
  const int BUF = ((T * (TRS > KB ? TRS : KB) + 15) & ~15) + 16;
  int w12 = (160 * 1024 - KB * 12) / (BUF * 8), w8 = (160 * 1024 - KB * 8) / (BUF * 8);
  if (w12 > 8) w12 = 8;
  if (w8 > 8) w8 = 8;
  return NXSIG_RAB_WG && (A > 48 || B > 48) && w8 > w12 && w8 <= 4;   // (<= 4 waves: one per SIMD, the whole register file — the window
                                                                        // loads of a unit are in flight together; at 7 waves 2400 spilled 228 B)
}

struct RabArgs {
  WaveArgs w;              // framing, window (f32[K], zero beyond N), div / has_scale, z; pairs_per_row = ceil(M / 2)
  const v2f* tw;           // c64[A][B]: W_K^(n2 k1) at [k1 * B + n2]: the lanes n2 of pass A read consecutive cells (at [n2 * A + k1]
                           // every lane of a group hit the same LDS bank: 67 % of the LDS cycles of the 960-point kernel were bank conflicts)
  int64_t units_per_row;   // ceil(pairs_per_row / T): a unit = T frame pairs = 2 T frames
  int64_t total_units;
  // sinks other than the complex spectrum (same fields as MelWaveArgs / R20Args)
  int32_t mel_bins = 0, nnz = 0;
  const float* csr_w = nullptr;
  const int* csr_off = nullptr;
  const int* csr_lo = nullptr;
  float* out = nullptr;
  int* gmax = nullptr;
  int32_t mag_kind = -1;
};

// SINK: kSinkSpectrum (c64 rows of K bins), kSinkMel (log-mel of the bins below K / 2), kSinkMag (|X| / |X|^2 / one-sided rows of them)
// Three waves per SIMD where the registers allow it without scratch (spectrum sink of 320 / 480 / 640, log-mel of 320 / 480): 0.56 / 0.56 /
// 0.51-0.53 of the roofline against 0.52 / 0.53 / 0.49 at two; 960 is held at two by its 75 KB of LDS per workgroup either way (0.48-0.49).
// What made the third wave possible: ONE inlined copy of the transform + sink (the solo route re-enters it with sel = 0, 1) and the
// untangle loop unrolled by 2 instead of fully — 216-256 registers became 145-167
// waves per SIMD the register allocator must reach: three wherever that costs no scratch (checked with tools/kernel_resources.py)
constexpr int rab_min_waves(int A, int B, int sink) {
  const int K = A * B, LT = A > B ? A : B;
  if (LT > 32 || K == 960) return 2;    // 960: 75 KB of LDS per workgroup hold it at two anyway
  return (sink == kSinkSpectrum || (sink == kSinkMel && K <= 480)) ? 3 : 2;
}

template <int A, int B, bool SCALE, int W, int SINK = kSinkSpectrum>
__global__ __launch_bounds__(64 * W)
__attribute__((amdgpu_waves_per_eu(rab_wg(A, B) ? 1 : rab_min_waves(A, B, SINK), rab_wg(A, B) ? 1 : 3))) void k_stft_rab(RabArgs b) {
  const WaveArgs& a = b.w;
  constexpr bool MEL = SINK == kSinkMel, MAG = SINK == kSinkMag;
  constexpr int KB = A * B, LT = A > B ? A : B, T = 64 / LT, NV = LT;
  constexpr int TRS = A * rab_bp(B);                        // one transform's transposed block (row stride B + 1, B + 2 for an odd B)
  constexpr int BUF = ((T * (TRS > KB ? TRS : KB) + 15) & ~15) + 16;   // complex cells per wave: staging (<= 2 BUF floats) / T x TRS / T x KB
  constexpr int NRS = 10;                                   // 16-byte loads per lane that prefetch a unit's span (<= 2560 floats)
  constexpr bool WG = rab_wg(A, B);
  float* s_w0 = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_tw = reinterpret_cast<v2f*>(s_w0 + (WG ? 0 : rab_wcells(KB)));
  v2f* s_x = s_tw + rab_tcells(KB);
  const float* s_w = WG ? a.wtab : s_w0;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: unit arithmetic on the scalar unit
  for (int i = tid; i < KB; i += 64 * W) { if (!WG) s_w0[i] = a.wtab[i]; s_tw[i] = b.tw[i]; }
  float* s_csr = reinterpret_cast<float*>(s_x + W * BUF);
  int* s_off = reinterpret_cast<int*>(s_csr + (MEL ? b.nnz : 0));
  int* s_lo = s_off + (MEL ? b.mel_bins + 1 : 0);
  if (MEL) {
    for (int i = tid; i < b.nnz; i += 64 * W) s_csr[i] = b.csr_w[i];
    for (int i = tid; i <= b.mel_bins; i += 64 * W) s_off[i] = b.csr_off[i];
    for (int i = tid; i < b.mel_bins; i += 64 * W) s_lo[i] = b.csr_lo[i];
  }
  __syncthreads();
  float vmax = -3.0e38f;
  bool melbad = false;
  constexpr int HALF = KB / 2;
  v2f* buf = s_x + wave * BUF;
  float* S = reinterpret_cast<float*>(buf);
  const int g = lane / LT, l = lane % LT;         // transform of the unit (g >= T: idle lanes), lane inside it
  const int nuse = a.N < KB ? a.N : KB;
  const int span = (2 * T - 1) * a.hop + nuse;
  const int span4 = (span + 3) & ~3;

  const int64_t p_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t p_end = p_begin + a.chunk;
  if (p_end > b.total_units) p_end = b.total_units;
  // the span of a unit that lies inside the stored row (and is 16-byte aligned) is fetched one unit ahead into registers
  v4f rs[NRS];
  auto prefetch = [&](int64_t row, int64_t u) -> bool {
    const int64_t start = 2 * T * u * (int64_t)a.hop - a.lo;
    const float* p = a.x + (size_t)row * a.batch_stride + start;
    const bool inside = start >= 0 && start + span4 <= a.L && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    if (inside) {
      const v4f* p4 = reinterpret_cast<const v4f*>(p) + lane;
#pragma unroll
      for (int c = 0; c < NRS; ++c)
        if (256 * c + 4 * lane < span4) rs[c] = p4[64 * c];
    }
    return inside;
  };
  auto stage_slow = [&](const float* xr, int64_t q0) {
    const int64_t start = q0 - a.lo;
    if (start >= 0 && start + span4 <= a.L && (reinterpret_cast<uintptr_t>(xr + start) & 15) == 0) {
      // inside the row and 16-byte aligned (the lengths without a register prefetch land here for every unit: round 6, 4-byte loads until then)
      const v4f* p4 = reinterpret_cast<const v4f*>(xr + start);
      for (int i0 = lane; 4 * i0 < span4; i0 += 256) {
        v4f t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = (4 * (i0 + 64 * k) < span4) ? p4[i0 + 64 * k] : v4f(0.0f);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (4 * (i0 + 64 * k) < span4) *reinterpret_cast<v4f*>(&S[4 * (i0 + 64 * k)]) = t[k];
      }
    } else if (start >= 0 && start + span <= a.L) {   // inside the row (whatever the padding mode) but not 16-byte aligned: 4-byte loads
      for (int i = lane; i < span; i += 64) S[i] = xr[start + i];
    } else {                                                      // padding / mirror / row end: per-sample bounds, eight loads in flight
      for (int i0 = lane; i0 < span; i0 += 512) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (i0 + 64 * k < span) ? fetch_any(xr, a, q0 + i0 + 64 * k) : 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (i0 + 64 * k < span) S[i0 + 64 * k] = t[k];
      }
    }
  };
  constexpr bool PF = LT < 30;   // register prefetch of the next unit's span (the 30- / 32-point codelets need the registers)
  // DIRECT (round 6; one frame pair per wave with A = B, window as long as the transform): a unit that lies inside
  // its row is loaded straight into pass A's registers — lane n2 takes x[B n1 + n2] of frame A and of frame B (4-byte loads, rows of
  // 4 B bytes; the overlap of the two frames comes from L1) — and the NEXT unit's loads are issued once pass B's results are parked in
  // LDS, so they travel under the untangle + store of this one.  Until then these lengths staged every unit through LDS with a loop
  // of 4-byte loads and waited for it (no prefetch: the registers were the codelets').  Units at row ends / in padding keep that route.
#ifndef NXSIG_RAB_DIRECT_FWD
#define NXSIG_RAB_DIRECT_FWD 1
#endif
  // The early issue keeps 2 A more registers alive across the untangle: only the square lengths (1600 = 40 x 40, 1764 = 42 x 42) afford it
  // (0.50 -> 0.565, 0.50 -> 0.53); the others spilled 60 ... 890 B to scratch and lost up to half (2400 0.44 -> 0.24, 3840 0.31 -> 0.13), and
  // loading them at the top of the unit instead (no early issue, no LDS staging) lost too (2400 0.37, 2880 0.25, 3840 0.19: 2 A 4-byte
  // loads of 4 B-byte rows against one pass of the span): they keep the staged route.
  constexpr bool DIRECT = T == 1 && A == B && NXSIG_RAB_DIRECT_FWD;
  constexpr bool DPF = DIRECT;
  v2f v[NV];
  auto load_direct = [&](int64_t row, int64_t u) -> bool {
    const int64_t start = 2 * u * (int64_t)a.hop - a.lo;
    const bool inside = nuse == KB && start >= 0 && start + a.hop + KB <= a.L;      // (wave-uniform)
    if (inside) {
      const float* p = a.x + (size_t)row * a.batch_stride + start + (lane < B ? lane : 0);
      const float* q = p + a.hop;
#pragma unroll
      for (int n1 = 0; n1 < A; ++n1) v[n1] = v2f{p[B * n1], q[B * n1]};
    }
    return inside;
  };
  // (row, unit inside the row) of the wave's units: one division per wave, then increments
  int64_t row = (p_begin + wave) / b.units_per_row;
  int64_t u = (p_begin + wave) - row * b.units_per_row;
  bool have = (PF && p_begin + wave < p_end) ? prefetch(row, u) : false;
  bool dcur = (DPF && p_begin + wave < p_end) ? load_direct(row, u) : false, dnext = false;
  for (int64_t ui = p_begin + wave; ui < p_end; ui += W) {
    int64_t nrow = row, nu = u + W;
    while (nu >= b.units_per_row) { nu -= b.units_per_row; ++nrow; }
    const float* xr = a.x + (size_t)row * a.batch_stride;
    const int64_t q0 = 2 * T * u * (int64_t)a.hop;    // padded-signal index of the unit's first sample
    // ---- the unit's raw samples -> LDS
    if constexpr (DIRECT && !DPF) dcur = load_direct(row, u);
    if (!dcur) {
      if (have) {
#pragma unroll
        for (int c = 0; c < NRS; ++c)
          if (256 * c + 4 * lane < span4) *reinterpret_cast<v4f*>(&S[256 * c + 4 * lane]) = rs[c];
      } else {
        stage_slow(xr, q0);
      }
      wave_lds_fence();
    }
    have = (PF && ui + W < p_end) ? prefetch(nrow, nu) : false;   // next unit's samples travel during this unit's transforms
    const int64_t pair = T * u + g;
    const bool active = g < T && pair < a.pairs_per_row;
    const int64_t mA = 2 * pair;
    const bool haveB = active && (mA + 1 < a.M);
    // pass A input of lane n2 = l: u[B n1 + n2], n1 < A.  sel < 0: the pair rides as frame A + i frame B; sel = 0 / 1: frame A / frame B
    // ALONE as the real part (solo route of a unit that holds a non-finite sample, see k_stft_r20)
    auto build = [&](int sel) {
      const float* fa = S + (2 * (g < T ? g : 0)) * a.hop + l;
      const float* fb = fa + a.hop;
      const bool on = active && l < B, onB = on && haveB;
      if (DIRECT && dcur && sel < 0) {     // the raw samples are in v already
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) {
          const float w = s_w[B * n1 + (l < B ? l : 0)];
          const float pa = v[n1].x * w, pb = v[n1].y * w;        // exact f32 products like the reference (:101)
          v[n1] = v2f{on ? pa : 0.0f, onB ? pb : 0.0f};
        }
        return;
      }
      // unconditional LDS reads + selects (a branch per element cost more than the selects; the reads of idle lanes and of n >= nuse stay
      // inside the wave's buffer, launch_rab checks it, and are discarded, never multiplied by zero: Inf x 0 would be NaN).  SHORT: the
      // window is shorter than the transform (wave-uniform): only then does an element need its own compare
      auto fill = [&](auto short_window) {
        constexpr bool SHORT = decltype(short_window)::value;
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) {
          const int n = B * n1 + l;
          const bool in = !SHORT || n < nuse;
          const float w = s_w[n];
          const float pa = fa[B * n1] * w, pb = fb[B * n1] * w;  // exact f32 products like the reference (:101)
          const float qa = (on && in) ? pa : 0.0f, qb = (onB && in) ? pb : 0.0f;
          v[n1] = sel < 0 ? v2f{qa, qb} : v2f{sel == 0 ? qa : qb, 0.0f};
        }
      };
      if (nuse == KB) fill(std::false_type{}); else fill(std::true_type{});
    };
    constexpr int NP = SINK == kSinkSpectrum ? KB / 2 : KB / 4;    // bin pairs per frame that reach the sink
    constexpr int NI = (NP + 63) / 64;
    v2f pw[T][2][NI];  // MEL: |XA|^2, |XB|^2 of the lane's bin pairs, parked in registers until every lane has read U
    auto xform_sink = [&](const int sel, const bool last) {
      dft_n<A>(v);
      if (l < B) {
#pragma unroll
        for (int k1 = 1; k1 < A; ++k1) {
          v[k1] = wcmul(v[k1], s_tw[k1 * B + l]);
          if (A > 16 && (k1 & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
      }
      wave_lds_fence();                               // every lane has read its samples: the buffer becomes the exchange
      if (g < T && l < B) {
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) buf[g * TRS + k1 * rab_bp(B) + l] = v[k1];
      }
      wave_lds_fence();
      // ---- pass B: lane k1 = l < A: DFT_B over n2
      if (g < T && l < A) {
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) v[n2] = buf[g * TRS + l * rab_bp(B) + n2];
      }
      dft_n<B>(v);
      wave_lds_fence();
      if (g < T && l < A) {
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) buf[g * KB + l + A * k2] = v[k2];   // U[k1 + A k2] in natural order
      }
      wave_lds_fence();
      if constexpr (DPF) {
        if (last) dnext = ui + W < p_end ? load_direct(nrow, nu) : false;   // the registers are free: the next unit travels under the stores
      }
      // ---- untangle + store.  All 64 lanes walk the T transforms one after the other: lane takes the bin pairs p = lane + 64 i
      //      (bins 2 p, 2 p + 1), so a wave instruction stores 1 KiB of one frame's row contiguously
#pragma unroll
      for (int gg = 0; gg < T; ++gg) {
        const int64_t pr = T * u + gg;
        const bool act = pr < a.pairs_per_row;                 // wave-uniform
        const int64_t m0 = 2 * pr;
        const bool hb = act && (m0 + 1 < a.M);
        const v2f* U = buf + gg * KB;
        v2f* zA = a.z + ((size_t)row * a.M + m0) * KB;
        v2f* zB = zA + KB;
        if constexpr ((KB & 1) != 0) {
          // odd fft length (441 = 21 x 21, round 6): rows of K x 8 bytes are 8-byte aligned only and bin K - 1 has no pair: single bins,
          // 8-byte LDS reads and stores (512 B per wave instruction); spectrum sink only
          constexpr int NB1 = (KB + 63) / 64;
#pragma unroll 2
          for (int i = 0; i < NB1; ++i) {
            const int k = lane + 64 * i;
            if (act && k < KB) {
              const v2f uu = U[k], pp = U[k == 0 ? 0 : KB - k];
              v2f xa = fft_eps0(v2f{uu.x + pp.x, uu.y - pp.y} * 0.5f);
              v2f xv = fft_eps0(v2f{uu.y + pp.y, pp.x - uu.x} * 0.5f);
              if (SCALE) { xa = xa / a.div; xv = xv / a.div; }
              const bool stA = sel <= 0, stB = hb && sel != 0;
              if (sel == 1) xv = xa;
              if (stA) __builtin_nontemporal_store(xa, (gv2f*)(zA + k));
              if (stB) __builtin_nontemporal_store(xv, (gv2f*)(zB + k));
            }
          }
          continue;
        }
#pragma unroll 2
        for (int i = 0; i < NI; ++i) {
          const int pi = lane + 64 * i;
          if (MEL) { pw[gg][0][i] = v2f{0.f, 0.f}; pw[gg][1][i] = v2f{0.f, 0.f}; }
          if (act && pi < NP) {
            const int k = 2 * pi;
            const v4f uu = *reinterpret_cast<const v4f*>(&U[k]);
            const v2f p0 = U[k == 0 ? 0 : KB - k], p1 = U[KB - 1 - k];
            v4f xa = fft_eps0(v4f{uu.x + p0.x, uu.y - p0.y, uu.z + p1.x, uu.w - p1.y} * 0.5f);  // Nx.fft's clean-up (:102)
            v4f xv = fft_eps0(v4f{uu.y + p0.y, p0.x - uu.x, uu.w + p1.y, p1.x - uu.z} * 0.5f);
            if (SCALE) { xa = xa / a.div; xv = xv / a.div; }
            const bool stA = sel <= 0, stB = hb && sel != 0;      // solo rounds: the transform's real part is frame A (sel 0) / B (sel 1)
            if (sel == 1) xv = xa;
            if (SINK == kSinkSpectrum) {
              if (stA) __builtin_nontemporal_store(xa, (gv4f*)(zA + k));
              if (stB) __builtin_nontemporal_store(xv, (gv4f*)(zB + k));
            } else {
              const v2f pa = v2f{xa.x * xa.x + xa.y * xa.y, xa.z * xa.z + xa.w * xa.w};
              const v2f pb = v2f{xv.x * xv.x + xv.y * xv.y, xv.z * xv.z + xv.w * xv.w};
              if (MEL) { pw[gg][0][i] = pa; pw[gg][1][i] = pb; }
              else if (b.mag_kind == 3) {   // one-sided complex rows: the same values the spectrum sink stores, bins below K / 2 only
                float* o = b.out + (((size_t)row * a.M + m0) * HALF + k) * 2;
                if (stA) __builtin_nontemporal_store(xa, (gv4f*)o);
                if (stB) __builtin_nontemporal_store(xv, (gv4f*)(o + 2 * HALF));
              } else {
                const v2f va = b.mag_kind == 1 ? pa : v2f{__builtin_sqrtf(pa.x), __builtin_sqrtf(pa.y)};
                const v2f vb = b.mag_kind == 1 ? pb : v2f{__builtin_sqrtf(pb.x), __builtin_sqrtf(pb.y)};
                float* o = b.out + ((size_t)row * a.M + m0) * HALF + k;
                float mx = -3.0e38f;
                if (stA) { __builtin_nontemporal_store(va, (gv2f*)o); mx = va.x > va.y ? va.x : va.y; }
                if (stB) { __builtin_nontemporal_store(vb, (gv2f*)(o + HALF)); mx = vb.x > mx ? vb.x : mx; mx = vb.y > mx ? vb.y : mx; }
                vmax = mx > vmax ? mx : vmax;
              }
            }
          }
        }
      }
    };
    // ---- non-finite samples: the reference transforms every frame alone (lib/nx_signal.ex:94-102), so an Inf / NaN reaches only the
    // frames that contain it.  A unit whose windowed samples are not all finite leaves the paired route: its frames A, then (samples
    // re-staged) its frames B, ride alone as real parts.
    build(-1);
    bool solo = false;
    {
      v2f t = v[0];
#pragma unroll
      for (int n1 = 1; n1 < A; ++n1) t += v[n1];
      const bool nf = ((__float_as_uint(t.x) & 0x7f800000u) == 0x7f800000u) || ((__float_as_uint(t.y) & 0x7f800000u) == 0x7f800000u);
      solo = __builtin_amdgcn_ballot_w64(nf) != 0;
      if (MEL) { melbad |= solo; solo = false; }   // log-mel: the whole tensor is poisoned instead (gmax[1], see stft_wave_body)
    }
    // one inlined copy of the transform + sink: the paired route is pass 0 with sel = -1; a non-finite unit takes two passes (sel 0, 1)
    const int npass = solo ? 2 : 1;
#pragma nounroll
    for (int ps = 0; ps < npass; ++ps) {
      const int sel = solo ? ps : -1;
      if (solo) {
        if (ps == 1 || dcur) {
          wave_lds_fence();      // round A's partner reads are done
          stage_slow(xr, q0);    // the exchange overwrote the samples (a DIRECT unit never staged them)
          wave_lds_fence();
        }
        build(sel);
      }
      xform_sink(sel, ps == npass - 1);
    }
    if (MEL) {   // (pw[][][] was filled by xform_sink(-1))
      wave_lds_fence();                          // every partner read of U is done: the buffer becomes the power spectra
      float* mags = reinterpret_cast<float*>(buf);  // frame f of the unit (f = 2 g + {0, 1}) at mags[f * HALF + k]
#pragma unroll
      for (int gg = 0; gg < T; ++gg)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int pi = lane + 64 * i;
          if (pi < NP) {
            *reinterpret_cast<v2f*>(&mags[(2 * gg) * HALF + 2 * pi]) = pw[gg][0][i];
            *reinterpret_cast<v2f*>(&mags[(2 * gg + 1) * HALF + 2 * pi]) = pw[gg][1][i];
          }
        }
      wave_lds_fence();
      // sparse filterbank + log10: one band per lane, the unit's 2 T frames share every weight
      const int rot = b.mel_bins & 63;   // the partial pass takes the narrowest bands (see stft_wave_body, wave_stft.hpp)
      for (int mb0 = lane; mb0 < b.mel_bins; mb0 += 64) {
        const int mb = mb0 + rot < b.mel_bins ? mb0 + rot : mb0 + rot - b.mel_bins;
        const int o0 = s_off[mb], o1 = s_off[mb + 1], k0 = s_lo[mb];
        float acc[2 * T];
#pragma unroll
        for (int f = 0; f < 2 * T; ++f) acc[f] = 0.0f;
        for (int jj = o0; jj < o1; ++jj) {
          const float wv = s_csr[jj];
#pragma unroll
          for (int f = 0; f < 2 * T; ++f) acc[f] = fmaf(mags[f * HALF + k0 + (jj - o0)], wv, acc[f]);
        }
#pragma unroll
        for (int f = 0; f < 2 * T; ++f) {
          const int64_t m = 2 * T * u + f;
          melbad |= (m < a.M) && !(acc[f] < INFINITY);
          const float av = acc[f] > 1.0e-10f ? acc[f] : 1.0e-10f;
          const float vv = __log2f(av) * 0.30102999566398120f;
          if (m < a.M) { b.out[((size_t)row * a.M + m) * b.mel_bins + mb] = vv; vmax = vv > vmax ? vv : vmax; }
        }
      }
    }
    wave_lds_fence();  // all reads of the buffer are done before the next unit's samples overwrite it
    row = nrow; u = nu;
    dcur = dnext;
  }
  if (MEL || (MAG && b.mag_kind == 2)) {  // one atomic per wave: running maximum in ordered-int encoding
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(vmax, off); vmax = o > vmax ? o : vmax; }
    if (lane == 0 && p_begin + wave < p_end) {
      const int i = __float_as_int(vmax);
      atomicMax(b.gmax, i >= 0 ? i : i ^ 0x7fffffff);
    }
    if (MEL && __builtin_amdgcn_ballot_w64(melbad) != 0 && lane == 0) atomicOr(b.gmax + 1, 1);
  }
}

template <int A, int B, bool ALL_SINKS = true>
inline int launch_rab(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel) {
  if (!ALL_SINKS && mel) return NXSIG_OK;   // spectrum sink only for this length: the caller's two-step path serves the other sinks
  // four waves per workgroup, short-lived workgroups.  (ONE workgroup per CU sized to fill the LDS — 6 ... 12 waves for 720 ... 960, what
  // the persistent inverse kernels gain 40-70 % from — measured 0.46 / 0.59 / 0.55 / 0.50 / 0.56 against 0.52 / 0.56 / 0.55 / 0.50 / 0.60
  // here for 720 / 768 / 800 / 900 / 960: the blocks retire together and the CU idles between them)
  constexpr int KB = A * B, LT = A > B ? A : B, T = 64 / LT;
  constexpr int TRS = A * rab_bp(B);
  constexpr int BUF = ((T * (TRS > KB ? TRS : KB) + 15) & ~15) + 16;
  // (1920: four waves would need 87 KB — one workgroup, four waves per CU; eight share the tables in 150 KB)
  // (round 6: lengths above 1920 cannot hold eight exchange buffers: as many waves as fit 160 KB beside the tables — 6 / 5 / 3 for 2400 / 2880 / 3840)
  constexpr int TABS = rab_wg(A, B) ? KB * 8 : rab_tab_bytes(KB);        // bytes of tables in LDS
  constexpr int W_FIT = (160 * 1024 - TABS) / (BUF * 8);
  constexpr int W = (KB * 12 + 4 * BUF * 8 > 80 * 1024) ? (W_FIT < 8 ? W_FIT : 8) : 4, WM = W;   // WM: the log-mel sink
  static_assert(W >= 1, "the tables and one exchange buffer must fit the LDS");
  const int nuse = s.fr.N < KB ? s.fr.N : KB;
  const int64_t span = (2 * T - 1) * (int64_t)s.fr.hop + nuse;
  if (LT < 30 && span + 3 > 2560) return NXSIG_OK;   // the unit's span must fit the prefetch registers (used below 30-point codelets only)
  if ((2 * T - 1) * (int64_t)s.fr.hop + KB + LT > 2 * BUF) return NXSIG_OK;   // ... and every lane's reads (idle lanes included) the wave's buffer
  RabArgs b;
  int sink = kSinkSpectrum;
  size_t lds_extra = 0;
  if (mel && mel->mag_kind >= 0) {
    sink = kSinkMag;
    b.out = mel->out; b.mag_kind = mel->mag_kind;
    int rcm = launch_mel_init(c, &b.gmax);
    if (rcm) return rcm;
  } else if (mel) {  // CSR of the triangular filter rows restricted to bins < K / 2
    sink = kSinkMel;
    std::vector<float> cw;
    std::vector<int> off(mel->mel_bins + 1, 0), lo(mel->mel_bins, 0);
    const int half = KB / 2;
    for (int mb = 0; mb < mel->mel_bins; ++mb) {
      const float* fr = mel->filters_host + (size_t)mb * KB;
      int l = half, h = 0;
      for (int k = 0; k < half; ++k)
        if (fr[k] != 0.0f) { if (k < l) l = k; h = k + 1; }
      if (h <= l) { l = 0; h = 0; }
      lo[mb] = l;
      for (int k = l; k < h; ++k) cw.push_back(fr[k]);
      off[mb + 1] = (int)cw.size();
    }
    if (cw.empty()) cw.push_back(0.0f);
    if (cw.size() > 6144 || mel->mel_bins > 1024) return NXSIG_OK;  // Bluestein / two-step path
    const void *dw = nullptr, *doff = nullptr, *dlo = nullptr;
    int rcm;
    if ((rcm = ctx_table(c, 0xC5A1ull, cw.data(), cw.size() * sizeof(float), &dw))) return rcm;
    if ((rcm = ctx_table(c, 0xC5A2ull, off.data(), off.size() * sizeof(int), &doff))) return rcm;
    if ((rcm = ctx_table(c, 0xC5A3ull, lo.data(), lo.size() * sizeof(int), &dlo))) return rcm;
    b.mel_bins = mel->mel_bins; b.nnz = (int)cw.size();
    b.csr_w = reinterpret_cast<const float*>(dw); b.csr_off = reinterpret_cast<const int*>(doff); b.csr_lo = reinterpret_cast<const int*>(dlo);
    b.out = mel->out;
    if ((rcm = launch_mel_init(c, &b.gmax))) return rcm;
    lds_extra = (size_t)b.nnz * 4 + (size_t)(2 * mel->mel_bins + 1) * 4;
  }
  {  // the launch must fit the LDS BEFORE the call is committed: a dense filterbank beside the 150 KB of the 1920-point kernel does not,
     // and the caller's two-step path takes it (it used to surface as a HIP launch error after *handled was set)
    const int wl = sink == kSinkMel ? WM : W;
    if ((size_t)TABS + (size_t)wl * BUF * 8 + lds_extra > (size_t)160 * 1024) return NXSIG_OK;
  }
  *handled = true;
  if (mel) *mel->handled = true;
  WaveArgs& a = b.w;
  a.x = s.x; a.batch_stride = s.batch_stride; a.L = s.fr.L; a.lo = s.fr.lo; a.M = s.fr.M;
  a.N = s.fr.N; a.hop = s.fr.hop; a.reflect = s.fr.reflect; a.batch = s.batch;
  a.pairs_per_row = (s.fr.M + 1) / 2;
  a.div = s.inv_scale_div; a.has_scale = s.has_scale; a.z = reinterpret_cast<v2f*>(s.z);
  a.twB = a.twC = a.twR = nullptr; a.dummy = nullptr; a.wtab = s.window_padK;
  a.units_per_row = 0; a.u_split = 0; a.u_add0 = 0; a.u_add1 = 0;
  b.units_per_row = (a.pairs_per_row + T - 1) / T;
  b.total_units = b.units_per_row * s.batch;
  a.total_pairs = b.total_units;
  const uint64_t key = 0x2AB000000000ull ^ ((uint64_t)A << 16) ^ (uint64_t)B;
  auto hit = c->memo.find(key);
  if (hit != c->memo.end()) b.tw = reinterpret_cast<const v2f*>(hit->second[0]);
  else {
    std::vector<float2> tw((size_t)KB);
    for (int n2 = 0; n2 < B; ++n2)
      for (int k1 = 0; k1 < A; ++k1) {
        const double ang = -6.283185307179586476925286766559 * (double)(n2 * k1) / (double)KB;
        tw[(size_t)k1 * B + n2] = make_float2((float)std::cos(ang), (float)std::sin(ang));
      }
    const void* dt = nullptr;
    int rc = ctx_table(c, 0x2AB0ull ^ ((uint64_t)A << 16) ^ (uint64_t)B, tw.data(), tw.size() * sizeof(float2), &dt);
    if (rc) return rc;
    c->memo[key] = {reinterpret_cast<uint64_t>(dt)};
    b.tw = reinterpret_cast<const v2f*>(dt);
  }
  const int w = sink == kSinkMel ? WM : W;
  a.chunk = (int64_t)w * fill_units_per_wave(c, b.total_units, w, sink == kSinkMel ? 8 : 4);   // four units per wave (two: -1 ... -3 %), short-lived workgroups; the mel sink amortises its CSR preload
  const int64_t blocks = (b.total_units + a.chunk - 1) / a.chunk;
  if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "stft: too many frames for one launch");
  const size_t lds = (size_t)TABS + (size_t)w * BUF * 8 + lds_extra;
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note(sink == kSinkSpectrum ? "stft.rab" : (sink == kSinkMel ? "mel.rab" : "mag.rab"));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * w), lds, c->stream, b);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  int rc = NXSIG_OK;
  if constexpr (ALL_SINKS) {
    if (sink == kSinkMel) rc = s.has_scale ? go(k_stft_rab<A, B, true, WM, kSinkMel>) : go(k_stft_rab<A, B, false, WM, kSinkMel>);
    else if (sink == kSinkMag) rc = s.has_scale ? go(k_stft_rab<A, B, true, W, kSinkMag>) : go(k_stft_rab<A, B, false, W, kSinkMag>);
    else rc = s.has_scale ? go(k_stft_rab<A, B, true, W>) : go(k_stft_rab<A, B, false, W>);
  } else {
    rc = s.has_scale ? go(k_stft_rab<A, B, true, W>) : go(k_stft_rab<A, B, false, W>);
  }
  if (rc) return rc;
  if (sink == kSinkMel) return launch_mel_finish(c, mel->out, (int64_t)s.batch * s.fr.M * mel->mel_bins, b.gmax);
  if (sink == kSinkMag && mel->mag_kind == 2) {
    const int64_t n = (int64_t)s.batch * s.fr.M * (KB / 2);
    hipLaunchKernelGGL(k_mag_db_pass2, dim3(mag_db_blocks(c, n)), dim3(256), 0, c->stream, mel->out, n, b.gmax);
    NXSIG_HIP_TRY(hipGetLastError());
  }
  return NXSIG_OK;
}

// ============================================================================================ STFT of COMPLEX samples, fft_length = A x B
// NxSignal.stft/3 on a c64 signal (lib/nx_signal.ex:94-102: the same slices x window -> Nx.fft(length: K), one complex frame per
// transform, nothing to untangle): the two-pass transform of k_stft_rab with ONE frame per max(A, B)-lane group, T frames per wave
// iteration.  The unit's c64 span is staged in LDS (8-byte loads), pass A multiplies by the real window, the natural-order result leaves
// with 16-byte stores after Nx.fft's clean-up and the scaling.  Frames are independent, so a non-finite sample needs no special route.
struct RabCArgs {
  const v2f* x;            // c64[batch][...], rows batch_stride cells apart
  int64_t batch_stride, L, lo, M;
  int32_t N, hop, reflect, batch;
  const float* wtab;       // f32[K], zero beyond N
  const v2f* tw;           // c64[A][B]: W_K^(n2 k1) at [k1 * B + n2]
  float div;
  int32_t has_scale;
  v2f* z;                  // c64[batch][M][K]
  int64_t units_per_row, total_units, chunk;   // a unit = T frames
};

__device__ __forceinline__ v2f fetch_any_c64(const v2f* __restrict__ x, const RabCArgs& a, int64_t q) {
  int64_t pos = q - a.lo;
  if (a.reflect) {
    if (a.L == 1) return x[0];
    const int64_t period = 2 * (a.L - 1);
    if (pos < 0) pos = -pos;
    if (pos >= a.L) pos = period - pos;
    if (pos < 0 || pos >= a.L) {
      pos %= period;
      if (pos < 0) pos += period;
      if (pos >= a.L) pos = period - pos;
    }
    return x[pos];
  }
  return (pos >= 0 && pos < a.L) ? x[pos] : v2f{0.f, 0.f};
}

template <int A, int B, bool SCALE, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(2, 3))) void k_stft_rab_c64(RabCArgs a) {
  constexpr int KB = A * B, LT = A > B ? A : B, T = 64 / LT, NV = LT;
  constexpr int TRS = A * rab_bp(B);
  constexpr int BUF = ((T * (TRS > KB ? TRS : KB) + 15) & ~15) + 16;
  float* s_w = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_tw = reinterpret_cast<v2f*>(s_w + rab_wcells(KB));
  v2f* s_x = s_tw + rab_tcells(KB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < KB; i += 64 * W) { s_w[i] = a.wtab[i]; s_tw[i] = a.tw[i]; }
  __syncthreads();
  v2f* buf = s_x + wave * BUF;
  const int g = lane / LT, l = lane % LT;
  const int nuse = a.N < KB ? a.N : KB;
  const int span = (T - 1) * a.hop + nuse;   // cells
  const int64_t p_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t p_end = p_begin + a.chunk;
  if (p_end > a.total_units) p_end = a.total_units;
  int64_t row = (p_begin + wave) / a.units_per_row;
  int64_t u = (p_begin + wave) - row * a.units_per_row;
  for (int64_t ui = p_begin + wave; ui < p_end; ui += W) {
    const v2f* xr = a.x + (size_t)row * a.batch_stride;
    const int64_t q0 = (int64_t)T * u * a.hop;          // padded-signal index of the unit's first sample
    const int64_t start = q0 - a.lo;
    // ---- the unit's samples -> LDS, eight loads in flight per lane
    const bool inside = start >= 0 && start + span <= a.L;   // wave-uniform
    for (int i0 = lane; i0 < span; i0 += 512) {
      v2f t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = i0 + 64 * k;
        t[k] = i < span ? (inside ? xr[start + i] : fetch_any_c64(xr, a, q0 + i)) : v2f{0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + 64 * k < span) buf[i0 + 64 * k] = t[k];
    }
    wave_lds_fence();
    const int64_t m = (int64_t)T * u + g;
    const bool active = g < T && m < a.M;
    v2f v[NV];
    {
      const v2f* fa = buf + (g < T ? g : 0) * a.hop + l;
      const bool on = active && l < B;
      auto fill = [&](auto short_window) {
        constexpr bool SHORT = decltype(short_window)::value;
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) {
          const int n = B * n1 + l;
          const bool in = on && (!SHORT || n < nuse);
          const float w = s_w[n];
          const v2f t = fa[B * n1];
          v[n1] = in ? v2f{t.x * w, t.y * w} : v2f{0.f, 0.f};   // (selected, never multiplied by zero: Inf x 0 would be NaN)
        }
      };
      if (nuse == KB) fill(std::false_type{}); else fill(std::true_type{});
    }
    dft_n<A>(v);
    if (l < B) {
#pragma unroll
      for (int k1 = 1; k1 < A; ++k1) {
        v[k1] = wcmul(v[k1], s_tw[k1 * B + l]);
        if (A > 16 && (k1 & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    }
    wave_lds_fence();
    if (g < T && l < B) {
#pragma unroll
      for (int k1 = 0; k1 < A; ++k1) buf[g * TRS + k1 * rab_bp(B) + l] = v[k1];
    }
    wave_lds_fence();
    if (g < T && l < A) {
#pragma unroll
      for (int n2 = 0; n2 < B; ++n2) v[n2] = buf[g * TRS + l * rab_bp(B) + n2];
    }
    dft_n<B>(v);
    wave_lds_fence();
    if (g < T && l < A) {
#pragma unroll
      for (int k2 = 0; k2 < B; ++k2) buf[g * KB + l + A * k2] = v[k2];   // X[k1 + A k2] in natural order
    }
    wave_lds_fence();
    // ---- clean-up (:102), scaling, 16-byte stores: all 64 lanes walk the T frames one after the other
    constexpr int NI = (KB / 2 + 63) / 64;
#pragma unroll
    for (int gg = 0; gg < T; ++gg) {
      const int64_t mm = (int64_t)T * u + gg;
      if (mm < a.M) {   // wave-uniform
        const v2f* U = buf + gg * KB;
        v2f* zr = a.z + ((size_t)row * a.M + mm) * KB;
        if constexpr ((KB & 1) != 0) {   // odd fft length: 8-byte stores (rows are 8-byte aligned only)
#pragma unroll 2
          for (int i = 0; i < (KB + 63) / 64; ++i) {
            const int k = lane + 64 * i;
            if (k < KB) {
              v2f xv = fft_eps0(U[k]);
              if (SCALE) xv = xv / a.div;
              __builtin_nontemporal_store(xv, (gv2f*)(zr + k));
            }
          }
          continue;
        }
#pragma unroll 2
        for (int i = 0; i < NI; ++i) {
          const int k = 2 * (lane + 64 * i);
          if (k < KB) {
            v4f xv = fft_eps0(*reinterpret_cast<const v4f*>(&U[k]));
            if (SCALE) xv = xv / a.div;
            __builtin_nontemporal_store(xv, (gv4f*)(zr + k));
          }
        }
      }
    }
    wave_lds_fence();
    u += W;
    while (u >= a.units_per_row) { u -= a.units_per_row; ++row; }
  }
}

template <int A, int B>
inline int launch_rab_c64(Ctx* c, const StftLaunch& s, bool* handled) {
  constexpr int KB = A * B, LT = A > B ? A : B, T = 64 / LT;
  constexpr int TRS = A * rab_bp(B);
  constexpr int BUF = ((T * (TRS > KB ? TRS : KB) + 15) & ~15) + 16;
  constexpr int W_FIT = (160 * 1024 - rab_tab_bytes(KB)) / (BUF * 8), W = W_FIT < 4 ? W_FIT : 4;   // (3840 = 64 x 60: three exchange buffers beside the tables)
  static_assert(W >= 1, "the tables and one exchange buffer must fit the LDS");
  if ((int64_t)(T - 1) * s.fr.hop + KB + LT > BUF) return NXSIG_OK;   // the unit's span (idle lanes' reads included) must fit the wave's buffer
  if ((reinterpret_cast<uintptr_t>(s.z) & 15) != 0) return NXSIG_OK;
  *handled = true;
  RabCArgs a;
  a.x = reinterpret_cast<const v2f*>(s.x); a.batch_stride = s.batch_stride; a.L = s.fr.L; a.lo = s.fr.lo; a.M = s.fr.M;
  a.N = s.fr.N; a.hop = s.fr.hop; a.reflect = s.fr.reflect; a.batch = s.batch;
  a.wtab = s.window_padK; a.div = s.inv_scale_div; a.has_scale = s.has_scale; a.z = reinterpret_cast<v2f*>(s.z);
  a.units_per_row = (s.fr.M + T - 1) / T;
  a.total_units = a.units_per_row * s.batch;
  const uint64_t key = 0x2AB000000000ull ^ ((uint64_t)A << 16) ^ (uint64_t)B;
  auto hit = c->memo.find(key);
  if (hit != c->memo.end()) a.tw = reinterpret_cast<const v2f*>(hit->second[0]);
  else {
    std::vector<float2> tw((size_t)KB);
    for (int n2 = 0; n2 < B; ++n2)
      for (int k1 = 0; k1 < A; ++k1) {
        const double ang = -6.283185307179586476925286766559 * (double)(n2 * k1) / (double)KB;
        tw[(size_t)k1 * B + n2] = make_float2((float)std::cos(ang), (float)std::sin(ang));
      }
    const void* dt = nullptr;
    int rc = ctx_table(c, 0x2AB0ull ^ ((uint64_t)A << 16) ^ (uint64_t)B, tw.data(), tw.size() * sizeof(float2), &dt);
    if (rc) return rc;
    c->memo[key] = {reinterpret_cast<uint64_t>(dt)};
    a.tw = reinterpret_cast<const v2f*>(dt);
  }
  a.chunk = (int64_t)W * fill_units_per_wave(c, a.total_units, W, 4);
  const int64_t blocks = (a.total_units + a.chunk - 1) / a.chunk;
  if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "stft: too many frames for one launch");
  const size_t lds = (size_t)rab_tab_bytes(KB) + (size_t)W * BUF * 8;
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note("stft_c64.rab");
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  return s.has_scale ? go(k_stft_rab_c64<A, B, true, W>) : go(k_stft_rab_c64<A, B, false, W>);
}

// ============================================================================================ iSTFT, N = fft_length = A x B
// NxSignal.istft/3 (lib/nx_signal.ex:609-637) for 320 / 480 / 640 / 960-point frames, any even hop: k_istft_r20's scheme with the two
// factors free.  The same two-pass transform in inverse direction (IDFT(z) = conj(DFT(conj z)) / K: the conjugations ride on the LDS
// reads and the epilogue), ONE complex frame per max(A, B)-lane group, T consecutive frames per wave iteration, a run of such units
// per wave.  The T spectra are one contiguous span (16-byte loads one unit ahead).  The windowed frames are parked in LDS in natural
// order and every lane gathers its output positions: carry of the earlier units + the frames that cover the position, in ascending
// frame order (deterministic; sharded = unsharded), x reciprocal of the guarded normaliser (:630-637), 16-byte stores; the following
// K - hop positions become the carry strip of the next unit.
struct IstftRabArgs {
  const v2f* z;               // c64[batch][M][K]
  int64_t M;
  int32_t batch, hop, RP;     // RP = ceil(K / hop): frames that cover one output sample
  int64_t out_len;            // (M - 1) hop + K
  int64_t units_per_row, run_len, runs_per_row, total_runs;
  const float* wtab;          // f32[K]
  const v2f* tw;              // c64[A][B] forward twiddles W_K^(n2 k1) at [k1 * B + n2]
  float scale;
  const float* den;           // f32[2 RP - 1][hop]: reciprocal of the guarded normaliser: head segments, interior, tail segments
  v2f* y;                     // c64[batch][out_len]
  v2f* dummy;
  int32_t cstride;            // cells of LDS per wave for the carry strip: K - hop rounded up to 16 (round 6: K until then)
};

// ODD: the hop is odd (8-byte LDS gathers and stores instead of 16-byte ones).  The scale factor is always multiplied in (1.0f when the
// call has none: exact), so the two instantiations per length are the two hop parities
// The 42- ... 64-point codelets run at most FOUR waves per workgroup, one per SIMD, with the whole 512-entry register file (256 + 256
// accumulation registers as spill space): at 256 they spilled 220 ... 720 B per lane to scratch
#ifndef NXSIG_RAB_TOUCH
#define NXSIG_RAB_TOUCH 1
#endif
#ifndef NXSIG_RAB_BIG_LT
#define NXSIG_RAB_BIG_LT 48
#endif
constexpr bool rab_big(int A, int B) { return A > NXSIG_RAB_BIG_LT || B > NXSIG_RAB_BIG_LT; }

// WMAX waves per workgroup at most; the launch picks W = blockDim.x / 64 <= WMAX from what the LDS holds for THIS hop (the carry strip is
// K - hop cells, not K).  TG: window and twiddles are read from global memory (L1 / L2 hits) instead of LDS copies — for the long lengths
// the 12 K bytes of tables cost a wave (2880 = 60 x 48: 4 waves instead of 2; 3840 = 64 x 60: 3 instead of 1).
template <int A, int B, bool ODD, int WMAX, bool TG>
__global__ __launch_bounds__(64 * WMAX) __attribute__((amdgpu_waves_per_eu(rab_big(A, B) ? 1 : 2, rab_big(A, B) ? 1 : 3))) void k_istft_rab(IstftRabArgs a) {
  constexpr int KB = A * B, LT = A > B ? A : B, T = 64 / LT, NV = LT, CMAX = KB;
  constexpr int TRS = A * rab_bp(B);
  constexpr int BUF = ((T * (TRS > KB ? TRS : KB) + 15) & ~15) + 16;
  constexpr bool KODD = (KB & 1) != 0;          // odd frame length (441): spectra rows are 8-byte aligned only -> 8-byte loads
  constexpr int N4 = KODD ? T * KB : T * KB / 2;   // pieces of a unit's T spectra: 16 bytes each (8 when KODD)
  constexpr int NRS = (N4 + 63) / 64;
  const int W = __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6));
  float* s_w0 = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_tw0 = reinterpret_cast<v2f*>(s_w0 + rab_wcells(KB));
  v2f* s_x = TG ? reinterpret_cast<v2f*>(g_wave_smem) : s_tw0 + rab_tcells(KB);
  v2f* s_carry = s_x + W * BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (!TG) {
    for (int i = tid; i < KB; i += 64 * W) { s_w0[i] = a.wtab[i]; s_tw0[i] = a.tw[i]; }
    __syncthreads();
  }
  const float* s_w = TG ? a.wtab : s_w0;
  const v2f* s_tw = TG ? a.tw : s_tw0;
  // the interior row of the normaliser (every unit but the first and last few of a row reads this one) sits in LDS: the global load in
  // each step of the overlap-add loop was a round trip to L2 per 128 samples, exposed with 2 - 8 waves per CU
  float* s_den = reinterpret_cast<float*>(s_carry + W * a.cstride);
  for (int i = tid; i < a.hop; i += 64 * W) s_den[i] = a.den[(size_t)(a.RP - 1) * a.hop + i];
  __syncthreads();
  v2f* buf = s_x + wave * BUF;
  v2f* carry = s_carry + wave * a.cstride;
  const int g = lane / LT, l = lane % LT;
  const int hop = a.hop;
  const int CARRY = KB - hop;             // positions handed to the next unit
  const int OUTN = T * hop;               // positions finished per unit
  const int64_t run = (int64_t)blockIdx.x * W + wave;
  if (run >= a.total_runs) return;
  const int64_t row = run / a.runs_per_row;
  const int64_t u0 = (run - row * a.runs_per_row) * a.run_len;
  int64_t u1 = u0 + a.run_len;
  if (u1 > a.units_per_row) u1 = a.units_per_row;
  const int halo = (a.RP - 1 + T - 1) / T;    // earlier units whose frames reach into this run
  const int64_t us = u0 >= halo ? u0 - halo : 0;
  for (int i = lane; i < CARRY; i += 64) carry[i] = v2f{0.f, 0.f};
  const float invK = 1.0f / (float)KB;
  const v2f* zrow = a.z + (size_t)row * a.M * KB;

  // DIRECT (one frame per wave, i.e. 33- ... 64-point codelets): the next unit's spectrum is loaded straight into pass A's registers
  // (lane n2 takes z[B n1 + n2]: rows of 8 B bytes, 8-byte loads) once pass B's results are parked in LDS, and arrives under the
  // overlap-add of this unit — no prefetch registers beside the transform's own, no staging pass through LDS
#ifndef NXSIG_RAB_DIRECT
#define NXSIG_RAB_DIRECT 1
#endif
  constexpr bool DIRECT = T == 1 && NXSIG_RAB_DIRECT;
  using pvec = std::conditional_t<KODD, v2f, v4f>;
  pvec rs[DIRECT ? 1 : NRS];
  auto prefetch = [&](int64_t u) {
    const int64_t m0 = T * u;
    const pvec* p4 = reinterpret_cast<const pvec*>(zrow + (size_t)m0 * KB) + lane;
    const int64_t avail4 = KODD ? (a.M - m0) * (int64_t)KB : (a.M - m0) * (KB / 2);   // pieces that exist from frame m0 on (frames past the end: zeros)
#pragma unroll
    for (int c = 0; c < (DIRECT ? 1 : NRS); ++c) {
      const int i4 = lane + 64 * c;
      rs[c] = (i4 < N4 && i4 < avail4) ? p4[64 * c] : pvec(0.0f);
    }
  };
  v2f v[NV];
  auto load_direct = [&](int64_t u) {
    const bool have = lane < B && u < a.M;
    const v2f* p = zrow + (size_t)(have ? u : 0) * KB + (have ? lane : 0);
#pragma unroll
    for (int n1 = 0; n1 < A; ++n1) v[n1] = have ? p[B * n1] : v2f{0.f, 0.f};
  };
  if constexpr (DIRECT) load_direct(us); else prefetch(us);
  for (int64_t u = us; u < u1; ++u) {
    if constexpr (!DIRECT) {
      // ---- the unit's T spectra -> LDS
#pragma unroll
      for (int c = 0; c < NRS; ++c) {
        const int i4 = lane + 64 * c;
        if (i4 < N4) *reinterpret_cast<pvec*>(&buf[KODD ? i4 : 2 * i4]) = rs[c];
      }
      wave_lds_fence();
      prefetch(u + 1 < u1 ? u + 1 : u);
    }
    // ---- pass A on conj(z): lane n2 = l < B of frame g takes conj z[B n1 + n2]
#pragma unroll
    for (int n1 = 0; n1 < A; ++n1) {
      if constexpr (DIRECT) v[n1].y = -v[n1].y;
      else {
        const v2f t = (g < T && l < B) ? buf[g * KB + B * n1 + l] : v2f{0.f, 0.f};
        v[n1] = v2f{t.x, -t.y};
        if (A > 16 && (n1 & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    }
    dft_n<A>(v);
    if (l < B) {
#pragma unroll
      for (int k1 = 1; k1 < A; ++k1) {
        v[k1] = wcmul(v[k1], s_tw[k1 * B + l]);
        if (A > 16 && (k1 & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    }
    wave_lds_fence();
    if (g < T && l < B) {
#pragma unroll
      for (int k1 = 0; k1 < A; ++k1) buf[g * TRS + k1 * rab_bp(B) + l] = v[k1];
    }
    wave_lds_fence();
    if (g < T && l < A) {
#pragma unroll
      for (int n2 = 0; n2 < B; ++n2) v[n2] = buf[g * TRS + l * rab_bp(B) + n2];
    }
    dft_n<B>(v);
    wave_lds_fence();
    // ---- x[n] = conj(T[n]) / K, x scale, x window (lib/nx_signal.ex:611-628), n = l + A k2; parked frame-major
    if (g < T && l < A) {
      const float live = (T * u + g) < a.M ? 1.0f : 0.0f;
#pragma unroll
      for (int k2 = 0; k2 < B; ++k2) {
        const int n = l + A * k2;
        v2f x = fft_eps0(v2f{v[k2].x, -v[k2].y} * invK);  // Nx.ifft's clean-up (:609) precedes scale and window
        x = x * a.scale;
        buf[g * KB + n] = x * (s_w[n] * live);
      }
    }
    wave_lds_fence();
    if constexpr (DIRECT) load_direct(u + 1 < u1 ? u + 1 : u);
    // position t of the unit (t = 0 is sample T u hop of the row): carry + covering frames in ascending order.  CW cells per lane and
    // step: 2 (16-byte LDS gathers and stores) for an even hop, 1 for an odd one (the cells of a frame then sit at odd offsets)
    const int64_t t_unit = u * OUTN;
    v2f* yrow = a.y + (size_t)row * a.out_len;
    const bool interior = (int64_t)T * u >= a.RP - 1 && (int64_t)T * u + T - 1 < a.M;   // every hop segment of the unit takes the interior row
    constexpr int NC = (CMAX + 127) / 128;
    auto finish = [&](auto cells) {
      constexpr int CW = decltype(cells)::value;
      using vec = std::conditional_t<CW == 2, v4f, v2f>;
      auto gather = [&](int t) -> vec {
        vec acc = vec(0.0f);
        if (t < CARRY) acc = *reinterpret_cast<const vec*>(&carry[t]);
#pragma unroll
        for (int f = 0; f < T; ++f) {
          const int off = t - f * hop;
          if (off >= 0 && off < KB) acc += *reinterpret_cast<const vec*>(&buf[f * KB + off]);
        }
        return acc;
      };
      for (int t = CW * lane; t < OUTN; t += 64 * CW) {
        const vec acc = gather(t);
        const int64_t tabs = t_unit + t;
        const bool inside = u >= u0 && tabs < a.out_len;
        v2f rd = v2f{0.f, 0.f};
        if (inside) {
          int f = 0;                        // hop segment of the unit that holds t (a 64-bit division per lane and iteration until round 5)
#pragma unroll
          for (int j = 1; j < T; ++j) f += t >= j * hop ? 1 : 0;
          const int64_t seg = (int64_t)T * u + f;
          const int pos = t - f * hop;
          if (interior) {                   // (wave-uniform)
            if constexpr (CW == 2) rd = *reinterpret_cast<const v2f*>(s_den + pos);
            else rd.x = s_den[pos];
          } else {
            const int64_t trow = seg < a.RP - 1 ? seg : (seg >= a.M ? a.RP + (seg - a.M) : a.RP - 1);
            if constexpr (CW == 2) rd = *reinterpret_cast<const v2f*>(a.den + trow * hop + pos);
            else rd.x = a.den[trow * hop + pos];
          }
        }
        v2f* yp = inside ? yrow + tabs : a.dummy + CW * lane;
        if constexpr (CW == 2) __builtin_nontemporal_store(v4f{acc.x * rd.x, acc.y * rd.x, acc.z * rd.y, acc.w * rd.y}, (gv4f*)yp);
        else __builtin_nontemporal_store(acc * rd.x, (gv2f*)yp);
      }
      // ---- carry for the next unit (all reads first, then the writes)
      vec nc[NC * 2 / CW];
#pragma unroll
      for (int i = 0; i < NC * 2 / CW; ++i) {
        const int t = CW * lane + 64 * CW * i;
        nc[i] = t < CARRY ? gather(OUTN + t) : vec(0.0f);
      }
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < NC * 2 / CW; ++i) {
        const int t = CW * lane + 64 * CW * i;
        if (t < CARRY) *reinterpret_cast<vec*>(&carry[t]) = nc[i];
      }
    };
    finish(std::integral_constant<int, (ODD || KODD) ? 1 : 2>{});
    wave_lds_fence();
  }
}

// Quarter-hop inverses of the one-frame-per-wave lengths with B a multiple of 4 (1152 ... 1920, 2400 / 2880 / 3840; hop K / 4 = A x B / 4): pass B leaves x[l + A k2] in lane l,
// so the four frames that overlap a sample sit in the SAME lane, B / 4 registers apart — the overlap-add runs in registers (acc[j] = what
// earlier frames left at l + A j), the finished quarter leaves as 8-byte stores straight from them.  No output staging, no carry strip,
// no gather pass: the LDS holds the tables and ONE transpose buffer per wave (3840: 3 waves instead of 2; 2400 / 2880: 4 instead of 3).
// Sums run in ascending frame order from +0 like k_istft_rab's (carry first, then the frame); the codelets are compiled per kernel, so
// against the LDS form a sample may differ in its last bit (fma contraction follows the surrounding code).
// the window of the quarter-hop kernel moves from LDS into registers when that buys a wave within WMAX (3840 = 64 x 60: 4 waves instead of 3)
constexpr bool rab_q_wr(int A, int B, int WMAX) {
  const int KB = A * B, TRS = A * rab_bp(B), BUF = ((TRS + 15) & ~15) + 16;
  int w12 = (160 * 1024 - KB * 12) / (BUF * 8), w8 = (160 * 1024 - KB * 8) / (BUF * 8);
  if (w12 > WMAX) w12 = WMAX;
  if (w8 > WMAX) w8 = WMAX;
  return WMAX <= 4 && w8 > w12;
}

template <int A, int B, int WMAX>
__global__ __launch_bounds__(64 * WMAX) __attribute__((amdgpu_waves_per_eu(WMAX <= 4 ? 1 : 2, WMAX <= 4 ? 1 : 2))) void k_istft_rab_q(IstftRabArgs a) {
  constexpr int KB = A * B, LT = A > B ? A : B, NV = LT, Q = B / 4, NA = B - Q;
  static_assert(64 / LT == 1 && B % 4 == 0, "one frame per wave, a quarter of B registers per hop");
  constexpr int TRS = A * rab_bp(B);
  constexpr int BUF = ((TRS + 15) & ~15) + 16;
  constexpr bool WR = rab_q_wr(A, B, WMAX);    // the window in B registers per lane (its positions never change) instead of 4 K bytes of LDS
  const int W = __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6));
  float* s_w = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_tw = reinterpret_cast<v2f*>(s_w + (WR ? 0 : KB));
  v2f* s_x = s_tw + KB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < KB; i += 64 * W) { if (!WR) s_w[i] = a.wtab[i]; s_tw[i] = a.tw[i]; }
  __syncthreads();
  v2f* buf = s_x + wave * BUF;
  const int hop = a.hop;                    // = A * Q
  const int64_t run = (int64_t)blockIdx.x * W + wave;
  if (run >= a.total_runs) return;
  const int64_t row = run / a.runs_per_row;
  const int64_t u0 = (run - row * a.runs_per_row) * a.run_len;
  int64_t u1 = u0 + a.run_len;
  if (u1 > a.units_per_row) u1 = a.units_per_row;
  const int halo = a.RP - 1;                // earlier frames that reach into this run
  const int64_t us = u0 >= halo ? u0 - halo : 0;
  const float invK = 1.0f / (float)KB;
  const v2f* zrow = a.z + (size_t)row * a.M * KB;
  v2f* yrow = a.y + (size_t)row * a.out_len;
  const int lA = lane < A ? lane : 0;
  float rdv[Q];                             // the interior row of the normaliser at this lane's positions
#pragma unroll
  for (int k2 = 0; k2 < Q; ++k2) rdv[k2] = a.den[(size_t)(a.RP - 1) * hop + lA + A * k2];
  float wv[WR ? B : 1];
  if constexpr (WR) {
#pragma unroll
    for (int k2 = 0; k2 < B; ++k2) wv[k2] = a.wtab[lA + A * k2];
  }
  v2f acc[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) acc[j] = v2f{0.f, 0.f};
  float junk = 0.0f;
  v2f v[NV];
  {
    const bool have = lane < B && us < a.M;
    const v2f* p = zrow + (size_t)(have ? us : 0) * KB + (have ? lane : 0);
#pragma unroll
    for (int n1 = 0; n1 < A; ++n1) v[n1] = have ? p[B * n1] : v2f{0.f, 0.f};
  }
  for (int64_t u = us; u < u1; ++u) {
    // the frame after this one is touched a line per lane NOW (its loads are issued at the end of the unit, into the registers the
    // overlap-add frees, and pass A needs them at once: with one wave per SIMD nothing else covers the trip to HBM); the touched values
    // are summed into `junk` at the END of the unit (behind a scheduling barrier) so that the wait for them lands there
    constexpr int NTOUCH = NXSIG_RAB_TOUCH ? (KB * 8 + 8191) / 8192 : 0;
    float touch[NTOUCH > 0 ? NTOUCH : 1];
    const bool touching = NTOUCH > 0 && u + 1 < u1 && u + 1 < a.M;
    if (touching) {
      const char* nb = reinterpret_cast<const char*>(zrow + (size_t)(u + 1) * KB);
#pragma unroll
      for (int j = 0; j < NTOUCH; ++j) {
        const int off = 128 * (lane + 64 * j);
        touch[j] = *reinterpret_cast<const float*>(nb + (off < KB * 8 ? off : 0));
      }
    }
    // ---- pass A on conj(z): lane n2 < B takes conj z[B n1 + n2]
#pragma unroll
    for (int n1 = 0; n1 < A; ++n1) v[n1].y = -v[n1].y;
    dft_n<A>(v);
    if (lane < B) {
#pragma unroll
      for (int k1 = 1; k1 < A; ++k1) {
        v[k1] = wcmul(v[k1], s_tw[k1 * B + lane]);
        if ((k1 & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    }
    wave_lds_fence();
    if (lane < B) {
#pragma unroll
      for (int k1 = 0; k1 < A; ++k1) buf[k1 * rab_bp(B) + lane] = v[k1];
    }
    wave_lds_fence();
    if (lane < A) {
#pragma unroll
      for (int n2 = 0; n2 < B; ++n2) v[n2] = buf[lane * rab_bp(B) + n2];
    }
    dft_n<B>(v);
    // ---- x[n] = conj(T[n]) / K, x scale, x window (lib/nx_signal.ex:611-628), n = lane + A k2; + what the three frames before left there
    const float live = u < a.M ? 1.0f : 0.0f;
    const bool interior = u >= a.RP - 1 && u < a.M;
    const int64_t trow = u < a.RP - 1 ? u : (u >= a.M ? a.RP + (u - a.M) : a.RP - 1);
    const int64_t t_unit = u * (int64_t)hop;
    const int64_t un = u + 1 < u1 ? u + 1 : u;                 // the next frame arrives in the registers this loop frees
    const bool have = lane < B && un < a.M;
    const v2f* pn = zrow + (size_t)(have ? un : 0) * KB + (have ? lane : 0);
#pragma unroll
    for (int k2 = 0; k2 < B; ++k2) {
      const int n = lA + A * k2;
      v2f x = fft_eps0(v2f{v[k2].x, -v[k2].y} * invK);  // Nx.ifft's clean-up (:609) precedes scale and window
      x = x * a.scale;
      x = x * ((WR ? wv[WR ? k2 : 0] : s_w[n]) * live);
      asm("" : "+v"(x));                                   // the windowed sample is ROUNDED before it is added (the LDS form stores it): no fma with the sum
      const v2f sum = (k2 < NA ? acc[k2 < NA ? k2 : 0] : v2f{0.f, 0.f}) + x;
      if (k2 < Q) {
        const float rd = interior ? rdv[k2 < Q ? k2 : 0] : a.den[trow * hop + n];
        if (lane < A && u >= u0 && t_unit + n < a.out_len) __builtin_nontemporal_store(sum * rd, (gv2f*)(yrow + t_unit + n));
      } else {
        acc[k2 - Q >= 0 ? k2 - Q : 0] = sum;
      }
      if (k2 < A) v[k2] = have ? pn[B * k2] : v2f{0.f, 0.f};
    }
    if constexpr (A > B) {
#pragma unroll
      for (int n1 = B; n1 < A; ++n1) v[n1] = have ? pn[B * n1] : v2f{0.f, 0.f};
    }
    if (touching) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NTOUCH; ++j) junk += touch[j];
    }
  }
  if (junk == 1.2345e-38f && a.hop < 0) a.dummy[lane] = v2f{junk, junk};   // (never: keeps the touches alive)
}

template <int A, int B>
inline int launch_istft_rab_AB(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled) {
  constexpr int KB = A * B, LT = A > B ? A : B, T = 64 / LT;
  constexpr int TRS = A * rab_bp(B);
  constexpr int BUF = ((T * (TRS > KB ? TRS : KB) + 15) & ~15) + 16;
  // ONE workgroup per CU with as many waves as the LDS (exchange + carry strip of K - hop cells per wave, the tables once unless they
  // stay in global memory) and the registers allow: 12 / 12 / 8 / 6 waves for 320 / 480 / 640 / 960 at hop K / 4.  (Round 5, until then
  // 2 workgroups of 4 / 4 / 2 / 2 waves: 0.25 -> 0.42 for 640.  Round 6: the strip follows the hop and the long lengths read their
  // tables from global memory: 1764 / 2400 / 2880 / 3840 run 6 / 4 / 4 / 3 waves instead of 4 / 3 / 2 / 1)
  constexpr bool BIG = rab_big(A, B);
  // one-frame-per-wave lengths whose LDS holds more than eight waves at a quarter hop (882 = 42 x 21: 11, 1000 = 40 x 25: 10) may run
  // three per SIMD as well: since they load straight into pass A's registers they need 140 VGPRs (tools/kernel_resources.py)
  constexpr int W_QUARTER = (160 * 1024 - rab_tab_bytes(KB) - KB) / ((BUF + ((3 * KB / 4 + 15) & ~15)) * 8);
  constexpr int WMAX = BIG ? 4 : ((LT <= 24 || (T == 1 && W_QUARTER > 8)) ? 12 : 8);
#ifndef NXSIG_RAB_TG_BYTES
#define NXSIG_RAB_TG_BYTES (1 << 30)
#endif
  constexpr bool TG = KB * 12 >= NXSIG_RAB_TG_BYTES;
  const int hop = s.hop;
  if (hop < 1 || hop > KB) return NXSIG_OK;                         // (an odd hop takes the kernel's 8-byte gathers)
  const int RP = (KB + hop - 1) / hop;
  if (s.M < 2 * RP - 1) return NXSIG_OK;                            // head and tail rows of the normaliser must not overlap
  if ((reinterpret_cast<uintptr_t>(s.z) & 15) || (reinterpret_cast<uintptr_t>(s.y) & 15)) return NXSIG_OK;
  *handled = true;
  IstftRabArgs a;
  a.z = reinterpret_cast<const v2f*>(s.z); a.M = s.M; a.batch = s.batch; a.hop = hop; a.RP = RP;
  a.out_len = (s.M - 1) * (int64_t)hop + KB;
  a.wtab = s.window; a.scale = s.scale_mul; a.y = reinterpret_cast<v2f*>(s.y);
  {  // forward twiddles (shared with the stft kernel of the same factors)
    const uint64_t key = 0x2AB000000000ull ^ ((uint64_t)A << 16) ^ (uint64_t)B;
    auto hit = c->memo.find(key);
    if (hit != c->memo.end()) a.tw = reinterpret_cast<const v2f*>(hit->second[0]);
    else {
      std::vector<float2> tw((size_t)KB);
      for (int n2 = 0; n2 < B; ++n2)
        for (int k1 = 0; k1 < A; ++k1) {
          const double ang = -6.283185307179586476925286766559 * (double)(n2 * k1) / (double)KB;
          tw[(size_t)k1 * B + n2] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
      const void* dt = nullptr;
      int rc = ctx_table(c, 0x2AB0ull ^ ((uint64_t)A << 16) ^ (uint64_t)B, tw.data(), tw.size() * sizeof(float2), &dt);
      if (rc) return rc;
      c->memo[key] = {reinterpret_cast<uint64_t>(dt)};
      a.tw = reinterpret_cast<const v2f*>(dt);
    }
  }
  {  // reciprocal of the guarded normaliser (:630-635): head segments j = 0..RP-2, the interior, tail segments j = M..M+RP-2
    const uint64_t dkey = fnv1a(0xDE2Bull ^ ((uint64_t)hop << 8) ^ ((uint64_t)KB << 40), window_host, (size_t)KB * sizeof(float));
    auto hit = c->memo.find(dkey);
    if (hit != c->memo.end()) {
      a.den = reinterpret_cast<const float*>(hit->second[0]);
    } else {
      std::vector<float> den((size_t)(2 * RP - 1) * hop);
      auto w2 = [&](int idx) { const float w = std::fabs(window_host[idx]); return (double)(w * w); };
      for (int rowi = 0; rowi < 2 * RP - 1; ++rowi)
        for (int pos = 0; pos < hop; ++pos) {
          double acc = 0.0;
          for (int rr = RP - 1; rr >= 0; --rr) {   // ascending frame order
            if (rr * hop + pos >= KB) continue;
            bool have;
            if (rowi < RP - 1) have = rr <= rowi;
            else if (rowi == RP - 1) have = true;
            else have = rr >= rowi - RP + 1;
            if (have) acc += w2(rr * hop + pos);
          }
          const float d = (float)acc;
          den[(size_t)rowi * hop + pos] = (float)(1.0 / (double)(d > 1.0e-10f ? d : 1.0f));
        }
      const void* dd = nullptr;
      int rc = ctx_table(c, 0xDE2Cull ^ ((uint64_t)hop << 8) ^ ((uint64_t)KB << 40), den.data(), den.size() * sizeof(float), &dd);
      if (rc) return rc;
      a.den = reinterpret_cast<const float*>(dd);
      c->memo[dkey] = {reinterpret_cast<uint64_t>(dd)};
    }
  }
  void* dummy = nullptr;
  { int rc2 = ctx_scratch(c, 3, (size_t)8192 * sizeof(float2), &dummy); if (rc2) return rc2; }
  a.dummy = reinterpret_cast<v2f*>(dummy);
  const int64_t segs = (a.out_len + hop - 1) / hop;              // hop segments of the output (the last may be partial)
  a.units_per_row = (segs + T - 1) / T;
  if constexpr (T == 1 && B % 4 == 0) {
    if (hop * 4 == KB && tune(c, kT_ISTFT_REGOLA, 1)) {   // overlap-add in registers
      constexpr int BUFQ = ((TRS + 15) & ~15) + 16;
      constexpr int WQ = BIG ? 4 : 8;                     // one / two waves per SIMD: 512 / 256 registers
      constexpr size_t TABQ = rab_q_wr(A, B, WQ) ? 8 : 12;
      int W = (int)((160 * 1024 - (size_t)KB * TABQ) / ((size_t)BUFQ * 8));
      if (W > WQ) W = WQ;
      a.cstride = 0;
      const int waves_per_cu = tune(c, kT_ISTFT_RUNS_PER_CU, W);
      const int64_t run_len = istft_balanced_run_len(a.units_per_row, s.batch, (int64_t)c->num_cus * waves_per_cu, RP - 1,
                                                     istft_min_run(c, a.units_per_row * s.batch, (int64_t)c->num_cus * waves_per_cu, 8));
      a.run_len = run_len;
      a.runs_per_row = (a.units_per_row + run_len - 1) / run_len;
      a.total_runs = a.runs_per_row * s.batch;
      const int64_t blocks = (a.total_runs + W - 1) / W;
      const size_t lds = (size_t)KB * TABQ + (size_t)W * BUFQ * 8;
      auto kernel = k_istft_rab_q<A, B, WQ>;
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      dispatch_note("istft.rab.q");
      hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
      NXSIG_HIP_TRY(hipGetLastError());
      return NXSIG_OK;
    }
  }
  a.cstride = ((KB - hop) + 15) & ~15;
  if (a.cstride < 16) a.cstride = 16;
  const size_t tables = TG ? 0 : (size_t)rab_tab_bytes(KB);
  const size_t den_lds = ((size_t)hop * 4 + 15) & ~(size_t)15;
  int W = (int)((160 * 1024 - tables - den_lds) / ((size_t)(BUF + a.cstride) * 8));
  if (W > WMAX) W = WMAX;
  if (W < 1) { *handled = false; return NXSIG_OK; }
  const int waves_per_cu = tune(c, kT_ISTFT_RUNS_PER_CU, W);  // = resident waves per CU
  const int64_t run_len = istft_balanced_run_len(a.units_per_row, s.batch, (int64_t)c->num_cus * waves_per_cu, (RP - 1 + T - 1) / T,
                                                 istft_min_run(c, a.units_per_row * s.batch, (int64_t)c->num_cus * waves_per_cu, 8));
  a.run_len = run_len;
  a.runs_per_row = (a.units_per_row + run_len - 1) / run_len;
  a.total_runs = a.runs_per_row * s.batch;
  const int64_t blocks = (a.total_runs + W - 1) / W;
  const size_t lds = tables + (size_t)W * BUF * 8 + (size_t)W * a.cstride * 8 + den_lds;
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note("istft.rab");
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  return (hop & 1) ? go(k_istft_rab<A, B, true, WMAX, TG>) : go(k_istft_rab<A, B, false, WMAX, TG>);
}

}  // namespace nxsig
