// Wave kernels, part 4: fft_length 400 = 20 x 20 computed natively (25 ms speech frames at 16 kHz, n_fft = 400).
//
// The Bluestein kernel pays two 1024-point transforms (about 102 kflop) per frame pair for a 400-point DFT (about
// 17 kflop) and is VALU-bound.  Here one 400-point complex FFT runs on 20 lanes x 20 points as two radix-20 passes
// (Cooley-Tukey n = 20 n1 + n2, k = k1 + 20 k2; each 20-point DFT is a prime-factor 4 x 5 butterfly without internal
// twiddles), three transforms per wave (lanes 60..63 idle).  As everywhere in this library two real frames ride as
// re / im of one transform and are separated afterwards through the Hermitian partner U[(400 - k) mod 400].
//
//   raw samples of the unit's six frames (one contiguous span) -> LDS            16-byte loads when aligned and inside
//   pass A   lane n2: DFT20 over n1 of u[20 n1 + n2] (frame slice x window fused, lib/nx_signal.ex:94-101), x W400^(n2 k1)
//   20 x 20 transpose through LDS (row stride 21: conflict-free both ways)
//   pass B   lane k1: DFT20 over n2 -> U[k1 + 20 k2]
//   natural-order U in LDS -> untangle XA = (U + conj U')/2, XB = -i (U - conj U')/2 -> 16-byte stores (:129)
#include "small_dft.hpp"   // dft5 / dft20 codelets (shared with kernels_wave_rab.hip)

namespace nxsig {

struct R20Args {
  WaveArgs w;              // framing, window (f32[400], zero beyond N), div / has_scale, z; pairs_per_row = ceil(M / 2)
  const v2f* tw;           // c64[20][20]: W_400^(n2 k1) at [k1 * 20 + n2] (symmetric; read with the lane index last: no LDS bank conflict)
  int64_t units_per_row;   // ceil(pairs_per_row / 3): a unit = three frame pairs = six frames
  int64_t total_units;
  int32_t fast_ok;         // prefetch aligned, in-bounds spans with 16-byte loads one unit ahead
  // sinks other than the complex spectrum (same fields as MelWaveArgs)
  int32_t mel_bins, nnz;
  const float* csr_w;
  const int* csr_off;
  const int* csr_lo;
  float* out;
  int* gmax;
  int32_t mag_kind;
};

// SINK: kSinkSpectrum (c64 rows of 400 bins), kSinkMel (log-mel of the bins below 200), kSinkMag (|X| / |X|^2 of them)
template <bool SCALE, int W, int SINK>
__global__ __launch_bounds__(64 * W) void k_stft_r20(R20Args b) {
  const WaveArgs& a = b.w;
  constexpr bool MEL = SINK == kSinkMel, MAG = SINK == kSinkMag;
  constexpr int KB = 400, HALF = 200;
  constexpr int BUF = 1280;                      // complex cells per wave: staging (<= 2560 floats) / 3 x 20 x 21 / 3 x 400
  float* s_w = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_tw = reinterpret_cast<v2f*>(s_w + KB);
  v2f* s_x = s_tw + KB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: unit arithmetic on the scalar unit
  for (int i = tid; i < KB; i += 64 * W) { s_w[i] = a.wtab[i]; s_tw[i] = b.tw[i]; }
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
  v2f* buf = s_x + wave * BUF;
  float* S = reinterpret_cast<float*>(buf);
  const int g = lane / 20, l20 = lane % 20;       // transform of the unit (g == 3: idle lanes), lane inside it
  const int nuse = a.N < KB ? a.N : KB;
  const int span = 5 * a.hop + nuse;
  const int span4 = (span + 3) & ~3;

  const int64_t p_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t p_end = p_begin + a.chunk;
  if (p_end > b.total_units) p_end = b.total_units;
  // the span of a unit that lies inside the stored row (and is 16-byte aligned) is fetched one unit ahead into registers
  v4f rs[10];
  auto prefetch = [&](int64_t row, int64_t u) -> bool {
    const int64_t start = 6 * u * (int64_t)a.hop - a.lo;
    const float* p = a.x + (size_t)row * a.batch_stride + start;
    const bool inside = b.fast_ok && start >= 0 && start + span4 <= a.L && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    if (inside) {
      const v4f* p4 = reinterpret_cast<const v4f*>(p) + lane;
#pragma unroll
      for (int c = 0; c < 10; ++c)
        if (256 * c + 4 * lane < span4) rs[c] = p4[64 * c];
    }
    return inside;
  };
  auto stage_slow = [&](const float* xr, int64_t q0) {
    const int64_t start = q0 - a.lo;
    if (start >= 0 && start + span <= a.L) {   // inside the row (whatever the padding mode) but not 16-byte aligned: 4-byte loads
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
  // (row, unit inside the row) of the wave's units: one division per wave, then increments
  int64_t row = (p_begin + wave) / b.units_per_row;
  int64_t u = (p_begin + wave) - row * b.units_per_row;
  bool have = (p_begin + wave < p_end) ? prefetch(row, u) : false;
  for (int64_t ui = p_begin + wave; ui < p_end; ui += W) {
    int64_t nrow = row, nu = u + W;
    while (nu >= b.units_per_row) { nu -= b.units_per_row; ++nrow; }
    const float* xr = a.x + (size_t)row * a.batch_stride;
    const int64_t q0 = 6 * u * (int64_t)a.hop;    // padded-signal index of the unit's first sample
    // ---- the unit's raw samples -> LDS
    if (have) {
#pragma unroll
      for (int c = 0; c < 10; ++c)
        if (256 * c + 4 * lane < span4) *reinterpret_cast<v4f*>(&S[256 * c + 4 * lane]) = rs[c];
    } else {
      stage_slow(xr, q0);
    }
    wave_lds_fence();
    have = (ui + W < p_end) ? prefetch(nrow, nu) : false;   // next unit's samples travel during this unit's transforms
    // ---- pass A: lane n2 = l20 of transform g: u[20 n1 + n2] = (frame A + i frame B) x window
    const int64_t pair = 3 * u + g;
    const bool active = g < 3 && pair < a.pairs_per_row;
    const int64_t mA = 2 * pair;
    const bool haveB = active && (mA + 1 < a.M);
    v2f v[20];
    // sel < 0: the pair rides as frame A + i frame B; sel = 0 / 1: frame A / frame B ALONE as the real part (solo route)
    auto build = [&](int sel) {
      const float* fa = S + (2 * (g < 3 ? g : 0)) * a.hop + l20;
      const float* fb = fa + a.hop;
      const bool onB = active && haveB;
      // unconditional LDS reads + selects instead of a branch per element (the reads stay inside the wave's buffer, launch_stft_r20
      // checks it; what is not wanted is discarded, never multiplied by zero: Inf x 0 would be NaN).  SHORT: the window is shorter
      // than the transform (wave-uniform): only then does an element need its own compare
      auto fill = [&](auto short_window) {
        constexpr bool SHORT = decltype(short_window)::value;
#pragma unroll
        for (int n1 = 0; n1 < 20; ++n1) {
          const int n = 20 * n1 + l20;
          const bool in = !SHORT || n < nuse;
          const float w = s_w[n];
          const float pa = fa[20 * n1] * w, pb = fb[20 * n1] * w;  // exact f32 products like the reference (:101)
          const float qa = (active && in) ? pa : 0.0f, qb = (onB && in) ? pb : 0.0f;
          v[n1] = sel < 0 ? v2f{qa, qb} : v2f{sel == 0 ? qa : qb, 0.0f};
        }
      };
      if (nuse == KB) fill(std::false_type{}); else fill(std::true_type{});
    };
    constexpr int NP = SINK == kSinkSpectrum ? 200 : 100;    // bin pairs per frame that reach the sink
    constexpr int NI = (NP + 63) / 64;
    v2f pw[3][2][NI];  // MEL: |XA|^2, |XB|^2 of the lane's bin pairs, parked in registers until every lane has read U
    // transforms + untangle + sink of the unit; sel as above (sel >= 0: only that frame of every pair is stored)
    auto xform_sink = [&](const int sel) {
    dft20(v);
#pragma unroll
    for (int k1 = 1; k1 < 20; ++k1) v[k1] = wcmul(v[k1], s_tw[k1 * 20 + l20]);
    wave_lds_fence();                               // every lane has read its samples: the buffer becomes the exchange
    if (g < 3) {
#pragma unroll
      for (int k1 = 0; k1 < 20; ++k1) buf[g * 420 + k1 * 21 + l20] = v[k1];
    }
    wave_lds_fence();
    // ---- pass B: lane k1 = l20: DFT20 over n2
    if (g < 3) {
#pragma unroll
      for (int n2 = 0; n2 < 20; ++n2) v[n2] = buf[g * 420 + l20 * 21 + n2];
    }
    dft20(v);
    wave_lds_fence();
    if (g < 3) {
#pragma unroll
      for (int k2 = 0; k2 < 20; ++k2) buf[g * KB + l20 + 20 * k2] = v[k2];   // U[k1 + 20 k2] in natural order
    }
    wave_lds_fence();
    // ---- untangle + sink.  All 64 lanes walk the three transforms one after the other: lane takes the bin pairs
    //      p = lane + 64 i (bins 2p, 2p + 1), so a wave instruction stores 1 KiB of one frame's row contiguously
    //      (only the bins below 200 for the mel / magnitude sinks)
#pragma unroll
    for (int gg = 0; gg < 3; ++gg) {
      const int64_t pr = 3 * u + gg;
      const bool act = pr < a.pairs_per_row;                 // wave-uniform
      const int64_t m0 = 2 * pr;
      const bool hb = act && (m0 + 1 < a.M);
      const v2f* U = buf + gg * KB;
      v2f* zA = a.z + ((size_t)row * a.M + m0) * KB;
      v2f* zB = zA + KB;
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
            else if (b.mag_kind == 3) {   // one-sided complex rows: the same values the spectrum sink stores, bins below 200 only
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
    };  // xform_sink
    // ---- non-finite samples: the reference transforms every frame alone (lib/nx_signal.ex:94-102), so an Inf / NaN reaches only
    // the frames that contain it.  A unit whose windowed samples are not all finite leaves the paired route: its frames A, then
    // (samples re-staged) its frames B, ride alone as real parts.  The log-mel sink needs none of this: there a non-finite
    // |z|^2 poisons the whole tensor through reduce_max (:511).
    build(-1);
    bool solo = false;
    {
      v2f t = v[0];
#pragma unroll
      for (int n1 = 1; n1 < 20; ++n1) t += v[n1];
      const bool nf = ((__float_as_uint(t.x) & 0x7f800000u) == 0x7f800000u) || ((__float_as_uint(t.y) & 0x7f800000u) == 0x7f800000u);
      solo = __builtin_amdgcn_ballot_w64(nf) != 0;
      if (MEL) { melbad |= solo; solo = false; }   // log-mel: the whole tensor is poisoned instead (gmax[1], see stft_wave_body)
    }
    // one inlined copy of the transform + sink (round 5: two copies — the paired call and the solo loop — cost the spectrum sink 70
    // registers and its third wave per SIMD): the paired route is pass 0 with sel = -1; a non-finite unit takes two passes (sel 0, 1)
    const int npass = solo ? 2 : 1;
#pragma nounroll
    for (int ps = 0; ps < npass; ++ps) {
      const int sel = solo ? ps : -1;
      if (solo) {
        if (ps == 1) {
          wave_lds_fence();      // round A's partner reads are done
          stage_slow(xr, q0);    // the exchange overwrote the samples
          wave_lds_fence();
        }
        build(sel);
      }
      xform_sink(sel);
    }
    // (MEL: pw[][][] is filled by xform_sink(-1))
    if (MEL) {
      wave_lds_fence();                          // every partner read of U is done: the buffer becomes the power spectra
      float* mags = reinterpret_cast<float*>(buf);  // frame f of the unit (f = 2 g + {0, 1}) at mags[f * 200 + k]
#pragma unroll
      for (int gg = 0; gg < 3; ++gg)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int pi = lane + 64 * i;
          if (pi < NP) {
            *reinterpret_cast<v2f*>(&mags[(2 * gg) * HALF + 2 * pi]) = pw[gg][0][i];
            *reinterpret_cast<v2f*>(&mags[(2 * gg + 1) * HALF + 2 * pi]) = pw[gg][1][i];
          }
        }
      wave_lds_fence();
      // sparse filterbank + log10: one band per lane, the unit's six frames share every weight (six independent sums)
      const int rot = b.mel_bins & 63;   // the partial pass takes the narrowest bands (see stft_wave_body, wave_stft.hpp)
      for (int mb0 = lane; mb0 < b.mel_bins; mb0 += 64) {
        const int mb = mb0 + rot < b.mel_bins ? mb0 + rot : mb0 + rot - b.mel_bins;
        const int o0 = s_off[mb], o1 = s_off[mb + 1], k0 = s_lo[mb];
        float acc[6];
#pragma unroll
        for (int f = 0; f < 6; ++f) acc[f] = 0.0f;
        for (int jj = o0; jj < o1; ++jj) {
          const float wv = s_csr[jj];
#pragma unroll
          for (int f = 0; f < 6; ++f) acc[f] = fmaf(mags[f * HALF + k0 + (jj - o0)], wv, acc[f]);
        }
#pragma unroll
        for (int f = 0; f < 6; ++f) {
          const int64_t m = 6 * u + f;
          melbad |= (m < a.M) && !(acc[f] < INFINITY);
          const float av = acc[f] > 1.0e-10f ? acc[f] : 1.0e-10f;
          const float vv = __log2f(av) * 0.30102999566398120f;
          if (m < a.M) { b.out[((size_t)row * a.M + m) * b.mel_bins + mb] = vv; vmax = vv > vmax ? vv : vmax; }
        }
      }
    }
    wave_lds_fence();  // all reads of the buffer are done before the next unit's samples overwrite it
    row = nrow; u = nu;
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

int launch_stft_r20(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel) {
  *handled = false;
  constexpr int W = 4, KB = 400, BUF = 1280;
  if (s.K != KB || s.fr.M == 0 || s.batch == 0 || s.window_padK == nullptr) return NXSIG_OK;
  if (tune(c, kT_DISABLE_R20, 0) || tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  if (5 * (int64_t)s.fr.hop + KB + 4 > 2 * BUF) return NXSIG_OK;  // every lane's reads of the unit's span (idle lanes included) must fit the wave's buffer
  R20Args b;
  b.mel_bins = 0; b.nnz = 0; b.csr_w = nullptr; b.csr_off = nullptr; b.csr_lo = nullptr; b.out = nullptr; b.gmax = nullptr;
  b.mag_kind = -1;
  int sink = kSinkSpectrum;
  size_t lds_extra = 0;
  if (mel && mel->mag_kind >= 0) {
    sink = kSinkMag;
    b.out = mel->out; b.mag_kind = mel->mag_kind;
    int rcm = launch_mel_init(c, &b.gmax);
    if (rcm) return rcm;
  } else if (mel) {  // CSR of the triangular filter rows restricted to bins < 200
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
  *handled = true;
  if (mel) *mel->handled = true;
  WaveArgs& a = b.w;
  a.x = s.x; a.batch_stride = s.batch_stride; a.L = s.fr.L; a.lo = s.fr.lo; a.M = s.fr.M;
  a.N = s.fr.N; a.hop = s.fr.hop; a.reflect = s.fr.reflect; a.batch = s.batch;
  a.pairs_per_row = (s.fr.M + 1) / 2;
  a.div = s.inv_scale_div; a.has_scale = s.has_scale; a.z = reinterpret_cast<v2f*>(s.z);
  a.twB = a.twC = a.twR = nullptr; a.dummy = nullptr; a.wtab = s.window_padK;
  a.units_per_row = 0; a.u_split = 0; a.u_add0 = 0; a.u_add1 = 0;
  b.units_per_row = (a.pairs_per_row + 2) / 3;
  b.total_units = b.units_per_row * s.batch;
  a.total_pairs = b.total_units;
  b.fast_ok = 1;  // 16-byte register prefetch of aligned spans (measured: 5.35 vs 4.1-4.9 TB/s for loads at the point of use)
  std::vector<float2> tw((size_t)KB);
  for (int n2 = 0; n2 < 20; ++n2)
    for (int k1 = 0; k1 < 20; ++k1) {
      const double ang = -6.283185307179586476925286766559 * (double)(n2 * k1) / (double)KB;
      tw[(size_t)k1 * 20 + n2] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
  const void* dt = nullptr;
  int rc = ctx_table(c, 0x20A20ull, tw.data(), tw.size() * sizeof(float2), &dt);
  if (rc) return rc;
  b.tw = reinterpret_cast<const v2f*>(dt);
  const int units_per_wave = fill_units_per_wave(c, b.total_units, W, sink == kSinkMel ? 8 : 2);  // the mel sink amortises its CSR preload
  a.chunk = (int64_t)W * (units_per_wave < 1 ? 1 : units_per_wave);
  const int64_t blocks = (b.total_units + a.chunk - 1) / a.chunk;
  if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "stft: too many frames for one launch");
  const size_t lds = (size_t)KB * 4 + (size_t)KB * 8 + (size_t)W * BUF * 8 + lds_extra;
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note(sink == kSinkSpectrum ? "stft.r20" : (sink == kSinkMel ? "mel.r20" : "mag.r20"));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, b);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  if (sink == kSinkMel) rc = s.has_scale ? go(k_stft_r20<true, W, kSinkMel>) : go(k_stft_r20<false, W, kSinkMel>);
  else if (sink == kSinkMag) rc = s.has_scale ? go(k_stft_r20<true, W, kSinkMag>) : go(k_stft_r20<false, W, kSinkMag>);
  else rc = s.has_scale ? go(k_stft_r20<true, W, kSinkSpectrum>) : go(k_stft_r20<false, W, kSinkSpectrum>);
  if (rc) return rc;
  if (sink == kSinkMel) return launch_mel_finish(c, mel->out, (int64_t)s.batch * s.fr.M * mel->mel_bins, b.gmax);
  if (sink == kSinkMag && mel->mag_kind == 2) {
    const int64_t n = (int64_t)s.batch * s.fr.M * (KB / 2);
    hipLaunchKernelGGL(k_mag_db_pass2, dim3(mag_db_blocks(c, n)), dim3(256), 0, c->stream, mel->out, n, b.gmax);
    NXSIG_HIP_TRY(hipGetLastError());
  }
  return NXSIG_OK;
}

// ============================================================================================ iSTFT, N = fft_length = 400
// NxSignal.istft/3 (lib/nx_signal.ex:609-637) for 400-point frames, any even hop: the same 20 x 20 transform in inverse
// direction (IDFT(z) = conj(DFT(conj z)) / 400: the conjugations ride on the LDS reads / the epilogue), ONE complex frame
// per 20-lane group, three consecutive frames per wave iteration, a run of such units per wave.  The three frames'
// spectra are one contiguous 9 600-byte span (16-byte loads one unit ahead).  The windowed frames are parked in LDS in
// natural order and every lane gathers its output positions: carry of the earlier units + the frames that cover the
// position, in ascending frame order (deterministic), x reciprocal of the guarded normaliser, 16-byte stores, 1 KiB per
// wave instruction; the following 400 - hop positions become the carry strip of the next unit (k_istft_wave_quad's scheme,
// here for a hop that need not divide the frame length).
struct IstftR20Args {
  const v2f* z;               // c64[batch][M][400]
  int64_t M;
  int32_t batch, hop, RP;     // RP = ceil(400 / hop): frames that cover one output sample
  int64_t out_len;            // (M - 1) hop + 400
  int64_t units_per_row, run_len, runs_per_row, total_runs;
  const float* wtab;          // f32[400]
  const v2f* tw;              // c64[20][20] forward twiddles W_400^(n2 k1)
  float scale;
  const float* den;           // f32[2 RP - 1][hop]: reciprocal of the guarded normaliser: head segments, interior, tail segments
  v2f* y;                     // c64[batch][out_len]
  v2f* dummy;
};

template <bool SCALE, int W>
__global__ __launch_bounds__(64 * W) void k_istft_r20(IstftR20Args a) {
  constexpr int KB = 400, BUF = 1280, CMAX = 400;
  float* s_w = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_tw = reinterpret_cast<v2f*>(s_w + KB);
  v2f* s_x = s_tw + KB;
  v2f* s_carry = s_x + W * BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < KB; i += 64 * W) { s_w[i] = a.wtab[i]; s_tw[i] = a.tw[i]; }
  __syncthreads();
  v2f* buf = s_x + wave * BUF;
  v2f* carry = s_carry + wave * CMAX;
  const int g = lane / 20, l20 = lane % 20;
  const int hop = a.hop;
  const int CARRY = KB - hop;             // positions handed to the next unit
  const int OUTN = 3 * hop;               // positions finished per unit
  const int64_t run = (int64_t)blockIdx.x * W + wave;
  if (run >= a.total_runs) return;
  const int64_t row = run / a.runs_per_row;
  const int64_t u0 = (run - row * a.runs_per_row) * a.run_len;
  int64_t u1 = u0 + a.run_len;
  if (u1 > a.units_per_row) u1 = a.units_per_row;
  const int halo = (a.RP - 1 + 2) / 3;    // earlier units whose frames reach into this run
  const int64_t us = u0 >= halo ? u0 - halo : 0;
  for (int i = lane; i < CARRY; i += 64) carry[i] = v2f{0.f, 0.f};
  const float invK = 1.0f / (float)KB;
  const v2f* zrow = a.z + (size_t)row * a.M * KB;

  // three frames = 1200 c64 = 600 float4: lane takes float4 number lane + 64 c
  v4f rs[10];
  auto prefetch = [&](int64_t u) {
    const int64_t m0 = 3 * u;
    const v4f* p4 = reinterpret_cast<const v4f*>(zrow + (size_t)m0 * KB) + lane;
    const int64_t avail4 = (a.M - m0) * (KB / 2);   // float4s that exist from frame m0 on (frames past the end: zeros)
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const int i4 = lane + 64 * c;
      rs[c] = (i4 < 600 && i4 < avail4) ? p4[64 * c] : v4f{0.f, 0.f, 0.f, 0.f};
    }
  };
  prefetch(us);
  for (int64_t u = us; u < u1; ++u) {
    // ---- the unit's three spectra -> LDS
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const int i4 = lane + 64 * c;
      if (i4 < 600) *reinterpret_cast<v4f*>(&buf[2 * i4]) = rs[c];
    }
    wave_lds_fence();
    prefetch(u + 1 < u1 ? u + 1 : u);
    // ---- pass A on conj(z): lane n2 = l20 of frame g takes conj z[20 n1 + n2]
    v2f v[20];
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) {
      const v2f t = g < 3 ? buf[g * KB + 20 * n1 + l20] : v2f{0.f, 0.f};
      v[n1] = v2f{t.x, -t.y};
    }
    dft20(v);
#pragma unroll
    for (int k1 = 1; k1 < 20; ++k1) v[k1] = wcmul(v[k1], s_tw[k1 * 20 + l20]);
    wave_lds_fence();
    if (g < 3) {
#pragma unroll
      for (int k1 = 0; k1 < 20; ++k1) buf[g * 420 + k1 * 21 + l20] = v[k1];
    }
    wave_lds_fence();
    if (g < 3) {
#pragma unroll
      for (int n2 = 0; n2 < 20; ++n2) v[n2] = buf[g * 420 + l20 * 21 + n2];
    }
    dft20(v);
    wave_lds_fence();
    // ---- x[n] = conj(T[n]) / 400, x scale, x window (lib/nx_signal.ex:611-628), n = l20 + 20 k2; parked frame-major
    if (g < 3) {
      const float live = (3 * u + g) < a.M ? 1.0f : 0.0f;
#pragma unroll
      for (int k2 = 0; k2 < 20; ++k2) {
        const int n = l20 + 20 * k2;
        v2f x = fft_eps0(v2f{v[k2].x, -v[k2].y} * invK);  // Nx.ifft's clean-up (:609) precedes scale and window
        if (SCALE) x = x * a.scale;
        buf[g * KB + n] = x * (s_w[n] * live);
      }
    }
    wave_lds_fence();
    // position t of the unit (t = 0 is sample 3 u hop of the row): carry + covering frames in ascending order
    auto gather = [&](int t) -> v4f {
      v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
      if (t < CARRY) acc = *reinterpret_cast<const v4f*>(&carry[t]);
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const int off = t - f * hop;
        if (off >= 0 && off < KB) acc += *reinterpret_cast<const v4f*>(&buf[f * KB + off]);
      }
      return acc;
    };
    const int64_t t_unit = u * OUTN;
    v2f* yrow = a.y + (size_t)row * a.out_len;
    for (int t = 2 * lane; t < OUTN; t += 128) {
      const v4f acc = gather(t);
      const int64_t tabs = t_unit + t;
      const bool inside = u >= u0 && tabs < a.out_len;
      v2f rd = v2f{0.f, 0.f};
      if (inside) {
        int f = 0;                        // hop segment of the unit that holds t (a 64-bit division per lane and iteration until round 5)
#pragma unroll
        for (int j = 1; j < 3; ++j) f += t >= j * hop ? 1 : 0;
        const int64_t seg = (int64_t)3 * u + f;
        const int pos = t - f * hop;
        const int64_t trow = seg < a.RP - 1 ? seg : (seg >= a.M ? a.RP + (seg - a.M) : a.RP - 1);
        rd = *reinterpret_cast<const v2f*>(a.den + trow * hop + pos);
      }
      const v4f o = v4f{acc.x * rd.x, acc.y * rd.x, acc.z * rd.y, acc.w * rd.y};
      v2f* yp = inside ? yrow + tabs : a.dummy + 2 * lane;
      __builtin_nontemporal_store(o, (gv4f*)yp);
    }
    // ---- carry for the next unit (all reads first, then the writes)
    v4f nc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = 2 * lane + 128 * i;
      nc[i] = t < CARRY ? gather(OUTN + t) : v4f{0.f, 0.f, 0.f, 0.f};
    }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = 2 * lane + 128 * i;
      if (t < CARRY) *reinterpret_cast<v4f*>(&carry[t]) = nc[i];
    }
    wave_lds_fence();
  }
}

int launch_istft_r20(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled) {
  *handled = false;
  constexpr int W = 11, KB = 400, BUF = 1280, CMAX = 400;   // one workgroup of 11 waves per CU: 153 KB of LDS (13.4 KB per wave + the tables once)
  if (s.K != KB || s.N != KB || s.M == 0 || s.batch == 0 || window_host == nullptr) return NXSIG_OK;
  if (tune(c, kT_DISABLE_R20, 0) || tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  const int hop = s.hop;
  if (hop < 2 || (hop & 1) || hop > KB) return NXSIG_OK;            // 16-byte LDS gathers need an even hop
  const int RP = (KB + hop - 1) / hop;
  if (s.M < 2 * RP - 1) return NXSIG_OK;                            // head and tail rows of the normaliser must not overlap
  if ((reinterpret_cast<uintptr_t>(s.z) & 15) || (reinterpret_cast<uintptr_t>(s.y) & 15)) return NXSIG_OK;
  *handled = true;
  IstftR20Args a;
  a.z = reinterpret_cast<const v2f*>(s.z); a.M = s.M; a.batch = s.batch; a.hop = hop; a.RP = RP;
  a.out_len = (s.M - 1) * (int64_t)hop + KB;
  a.wtab = s.window; a.scale = s.scale_mul; a.y = reinterpret_cast<v2f*>(s.y);
  {  // forward twiddles (shared with the stft kernel)
    std::vector<float2> tw((size_t)KB);
    for (int n2 = 0; n2 < 20; ++n2)
      for (int k1 = 0; k1 < 20; ++k1) {
        const double ang = -6.283185307179586476925286766559 * (double)(n2 * k1) / (double)KB;
        tw[(size_t)k1 * 20 + n2] = make_float2((float)std::cos(ang), (float)std::sin(ang));
      }
    const void* dt = nullptr;
    int rc = ctx_table(c, 0x20A20ull, tw.data(), tw.size() * sizeof(float2), &dt);
    if (rc) return rc;
    a.tw = reinterpret_cast<const v2f*>(dt);
  }
  {  // reciprocal of the guarded normaliser (:630-635): head segments j = 0..RP-2, the interior, tail segments j = M..M+RP-2;
     // segment j gets w2[rr hop + pos] from frame j - rr whenever that frame exists and the index lies inside the window
    const uint64_t dkey = fnv1a(0xDE19ull ^ ((uint64_t)hop << 8), window_host, (size_t)KB * sizeof(float));
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
      int rc = ctx_table(c, 0xDE1Aull ^ ((uint64_t)hop << 8), den.data(), den.size() * sizeof(float), &dd);
      if (rc) return rc;
      a.den = reinterpret_cast<const float*>(dd);
      c->memo[dkey] = {reinterpret_cast<uint64_t>(dd)};
    }
  }
  void* dummy = nullptr;
  { int rc2 = ctx_scratch(c, 3, (size_t)8192 * sizeof(float2), &dummy); if (rc2) return rc2; }
  a.dummy = reinterpret_cast<v2f*>(dummy);
  const int64_t segs = (a.out_len + hop - 1) / hop;              // hop segments of the output (the last may be partial)
  a.units_per_row = (segs + 2) / 3;
  const int waves_per_cu = tune(c, kT_ISTFT_RUNS_PER_CU, W);  // = resident waves per CU
  const int64_t run_len = istft_balanced_run_len(a.units_per_row, s.batch, (int64_t)c->num_cus * waves_per_cu, (RP - 1 + 2) / 3,
                                                 istft_min_run(c, a.units_per_row * s.batch, (int64_t)c->num_cus * waves_per_cu, 8));
  a.run_len = run_len;
  a.runs_per_row = (a.units_per_row + run_len - 1) / run_len;
  a.total_runs = a.runs_per_row * s.batch;
  const int64_t blocks = (a.total_runs + W - 1) / W;
  const size_t lds = (size_t)KB * 4 + (size_t)KB * 8 + (size_t)W * BUF * 8 + (size_t)W * CMAX * 8;
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note("istft.r20");
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  return s.has_scale ? go(k_istft_r20<true, W>) : go(k_istft_r20<false, W>);
}

}  // namespace nxsig
