// The f64 / c128 tier of the STFT / iSTFT / FIR path.  The reference computes in the type of its operands: f64 samples or an
// f64 window make Nx.multiply promote to f64 and Nx.fft return c128 (lib/nx_signal.ex:101-102), a c128 spectrum is inverted in
// c128 (:609), f64 operands of Convolution.fftconvolve are transformed in c128 (lib/nx_signal/convolution.ex:276-284).  The
// north star of this repository is the f32 path (the tuned wave kernels); this unit gives the double-precision callers the same
// entry points with workgroup-per-frame kernels in double arithmetic — correct for every shape within the stated limits,
// HBM traffic exactly the algorithmic bytes, no tuning beyond that (measured figures: DESIGN.md "f64 tier").
//
//   k_stft_d<KIND>        frame slice x window -> K-point transform in LDS -> eps clean-up -> scale -> c128 store
//   k_fft_rows_d<INV,KIND> Nx.fft / Nx.ifft(length:) over rows (+ the x scale x window epilogue of istft :611-628)
//   k_ola_d<COMPS,NORM>   overlap_and_add in fixed frame order (+ the |w|^2 normaliser with the 1e-10 guard, :630-637)
//   k_fir_os_d            overlap-save block convolution, two real blocks as re / im of one complex transform
//   k_as_windowed_d       framing gather of 8-byte words
// KIND: 0 = power-of-two length <= 8192 (in-place radix-2^2 decimation in time on a bit-reversed load: ONE LDS buffer, so
// 8192 points of c128 = 128 KiB fit), 1 = Bluestein chirp-z for other lengths 65 .. 4096 (through the same transform of
// P = 2^ceil(log2(2K-1)) points), 2 = direct DFT from a table (the remaining lengths up to 65536).
// Twiddles, chirps and filter spectra are generated on the host in long double and rounded once to double.
// Each frame is its own transform here, so a non-finite sample reaches only the frames that contain it (rule 1 of DESIGN §3.0)
// by construction; the FIR follows the f32 path: a row that holds one comes out NaN from end to end.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "nxsig_internal.h"

namespace nxsig {

namespace {

constexpr int kT = 256;
constexpr double kEpsD = 1.0e-10;  // Nx.fft / Nx.ifft :eps default, applied to c128 results as well (SURVEY App. A rule 7)

__device__ __forceinline__ double eps0d(double x) { return fabs(x) <= kEpsD ? 0.0 : x; }
__device__ __forceinline__ double2 eps0d(double2 v) { return make_double2(eps0d(v.x), eps0d(v.y)); }
__device__ __forceinline__ double2 cmuld(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 caddd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csubd(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
template <bool INV>
__device__ __forceinline__ double2 mul_mi_d(double2 a) { return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x); }
template <bool INV>
__device__ __forceinline__ double2 twd(const double2* __restrict__ tw, int idx) {
  double2 w = tw[idx];
  if (INV) w.y = -w.y;
  return w;
}
__device__ __forceinline__ int brev(int n, int bits) { return bits ? (int)(__brev((unsigned)n) >> (32 - bits)) : 0; }

struct GeomD {
  int64_t L, lo, M;
  int32_t N, hop, reflect;
};

// sample of the (virtually) padded signal at padded index q — lib/nx_signal.ex:338 (Nx.pad, zeros) / :349 (Nx.reflect)
__device__ __forceinline__ double fetch_padded_d(const double* __restrict__ x, const GeomD& g, int64_t q) {
  int64_t pos = q - g.lo;
  if (g.reflect) {
    if (g.L == 1) return x[0];
    const int64_t period = 2 * (g.L - 1);
    pos %= period;
    if (pos < 0) pos += period;
    if (pos >= g.L) pos = period - pos;
    return x[pos];
  }
  return (pos >= 0 && pos < g.L) ? x[pos] : 0.0;
}

// the same for a COMPLEX signal (c128 samples, round 6): one double2 per sample
__device__ __forceinline__ double2 fetch_padded_cd(const double2* __restrict__ x, const GeomD& g, int64_t q) {
  int64_t pos = q - g.lo;
  if (g.reflect) {
    if (g.L == 1) return x[0];
    const int64_t period = 2 * (g.L - 1);
    pos %= period;
    if (pos < 0) pos += period;
    if (pos >= g.L) pos = period - pos;
    return x[pos];
  }
  return (pos >= 0 && pos < g.L) ? x[pos] : make_double2(0.0, 0.0);
}

extern __shared__ __attribute__((aligned(16))) unsigned char g_smem_d[];

// F rows of K = 2^logK points each, stored in BIT-REVERSED order, transformed in place to natural order: radix-2 decimation in
// time with two consecutive stages fused per pass (spans p and 2p: elements i, i+p, i+2p, i+3p), one leading radix-2 pass when
// logK is odd.  tw[j] = exp(-2 pi i j / K), j < K/2.  Unnormalised in both directions.
template <bool INV>
__device__ void lds_fft_d(double2* s, int K, int logK, int F, const double2* __restrict__ tw) {
  const int tid = threadIdx.x;
  int logp = 0;
  if (logK & 1) {
    const int total = F << (logK - 1);
    for (int w = tid; w < total; w += kT) {
      double2* r = s + 2 * (size_t)w;
      const double2 u0 = r[0], u1 = r[1];
      r[0] = caddd(u0, u1);
      r[1] = csubd(u0, u1);
    }
    __syncthreads();
    logp = 1;
  }
  while (logp < logK) {
    const int p = 1 << logp, q = K >> 2, logq = logK - 2, total = F * q;
    const int s1 = K >> (logp + 1), s2 = K >> (logp + 2);
    for (int w = tid; w < total; w += kT) {
      const int f = w >> logq, i = w & (q - 1);
      const int k = i & (p - 1);
      double2* r = s + (size_t)f * K + ((i >> logp) << (logp + 2)) + k;
      double2 a0 = r[0], a1 = r[p], a2 = r[2 * p], a3 = r[3 * p];
      if (logp > 0) {
        const double2 w1 = twd<INV>(tw, k * s1);
        a1 = cmuld(a1, w1);
        a3 = cmuld(a3, w1);
      }
      const double2 b0 = caddd(a0, a1), b1 = csubd(a0, a1);
      double2 b2 = caddd(a2, a3), b3 = csubd(a2, a3);
      if (logp > 0) {
        const double2 w2 = twd<INV>(tw, k * s2);
        b2 = cmuld(b2, w2);
        b3 = cmuld(b3, w2);
      }
      b3 = mul_mi_d<INV>(b3);   // W_{4p}^{k+p} = W_{4p}^k * (-i)  (forward)
      r[0] = caddd(b0, b2);
      r[2 * p] = csubd(b0, b2);
      r[p] = caddd(b1, b3);
      r[3 * p] = csubd(b1, b3);
    }
    __syncthreads();
    logp += 2;
  }
}

// Bluestein tables of one length K (see the header): chirp[n] = exp(-i pi n^2 / K); Bf = FFT_P of the wrapped conjugate chirp,
// pre-scaled by 1 / P; twP the forward twiddles of the P-point transform
struct BlueD {
  int32_t K, P, logP;
  const double2* chirp;
  const double2* Bf;
  const double2* twP;
};

// S holds x[n] * chirp[n] (n < nuse, zero beyond) in bit-reversed order of P points; on return S[k], k < K, is the DFT of x
__device__ void bluestein_core_d(double2* S, const BlueD& t) {
  const int tid = threadIdx.x;
  lds_fft_d<false>(S, t.P, t.logP, 1, t.twP);
  // x Bf, and back into bit-reversed order for the second transform: positions i and rev(i) trade places
  for (int i = tid; i < t.P; i += kT) {
    const int r = brev(i, t.logP);
    if (i < r) {
      const double2 a = cmuld(S[i], t.Bf[i]), b = cmuld(S[r], t.Bf[r]);
      S[i] = b;
      S[r] = a;
    } else if (i == r) {
      S[i] = cmuld(S[i], t.Bf[i]);
    }
  }
  __syncthreads();
  lds_fft_d<true>(S, t.P, t.logP, 1, t.twP);
  for (int k = tid; k < t.K; k += kT) S[k] = cmuld(S[k], t.chirp[k]);
  __syncthreads();
}

// ------------------------------------------------------------------------------------------ STFT
struct StftArgsD {
  const double* x;        // f64[batch][L]; CX: c128[batch][L] (batch_stride in COMPLEX elements)
  int64_t batch_stride;
  GeomD g;
  int32_t K, logK, F;
  const double* window;   // f64[N]
  const double2* tw;      // KIND 0: K/2 twiddles; KIND 2: K twiddles
  BlueD blue;             // KIND 1
  double div;
  int32_t has_scale;
  double2* z;             // c128[batch][M][K]
};

// CX: complex samples — c128 x f64 window componentwise (lib/nx_signal.ex:101 on a complex tensor), one transform per frame as ever
template <int KIND, bool CX = false>
__global__ __launch_bounds__(kT) void k_stft_d(StftArgsD a) {
  double2* S = reinterpret_cast<double2*>(g_smem_d);
  const int tid = threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.x * a.F;
  const double* x = a.x + (size_t)blockIdx.y * a.batch_stride * (CX ? 2 : 1);
  const double2* xc = reinterpret_cast<const double2*>(x);
  // windowed sample n of the frame that starts at padded index q0
  auto sample = [&](int64_t q, int n) -> double2 {
    const double w = a.window[n];
    if (CX) { const double2 v = fetch_padded_cd(xc, a.g, q); return make_double2(v.x * w, v.y * w); }
    return make_double2(fetch_padded_d(x, a.g, q) * w, 0.0);
  };
  const int nuse = a.g.N < a.K ? a.g.N : a.K;   // Nx.fft(length: K): rows zero-padded or truncated to K
  double2* z = a.z + ((size_t)blockIdx.y * a.g.M + m0) * a.K;
  if (KIND == 0) {
    const int total = a.F * a.K;
    for (int idx = tid; idx < total; idx += kT) {
      const int f = idx >> a.logK, n = idx & (a.K - 1);
      const int64_t m = m0 + f;
      double2 v = make_double2(0.0, 0.0);
      if (m < a.g.M && n < nuse) v = sample(m * a.g.hop + n, n);   // :101
      S[(size_t)f * a.K + brev(n, a.logK)] = v;
    }
    __syncthreads();
    lds_fft_d<false>(S, a.K, a.logK, a.F, a.tw);
    for (int idx = tid; idx < total; idx += kT) {
      const int f = idx >> a.logK;
      if (m0 + f >= a.g.M) break;
      double2 v = eps0d(S[idx]);   // the clean-up comes before the scaling (:102, :113)
      if (a.has_scale) { v.x = v.x / a.div; v.y = v.y / a.div; }
      z[idx] = v;
    }
  } else if (KIND == 1) {
    const BlueD& t = a.blue;
    for (int n = tid; n < t.P; n += kT) {
      double2 u = make_double2(0.0, 0.0);
      if (n < nuse) {
        const double2 v = sample(m0 * a.g.hop + n, n), c = t.chirp[n];
        u = CX ? cmuld(v, c) : make_double2(v.x * c.x, v.x * c.y);
      }
      S[brev(n, t.logP)] = u;
    }
    __syncthreads();
    bluestein_core_d(S, t);
    for (int k = tid; k < a.K; k += kT) {
      double2 v = eps0d(S[k]);
      if (a.has_scale) { v.x = v.x / a.div; v.y = v.y / a.div; }
      z[k] = v;
    }
  } else {
    double* s = reinterpret_cast<double*>(g_smem_d);
    for (int n = tid; n < nuse; n += kT) {
      const double2 v = sample(m0 * a.g.hop + n, n);
      if (CX) S[n] = v; else s[n] = v.x;
    }
    __syncthreads();
    for (int k = tid; k < a.K; k += kT) {
      double re = 0.0, im = 0.0;
      int idx = 0;
      for (int n = 0; n < nuse; ++n) {
        const double2 w = a.tw[idx];
        if (CX) {
          const double2 v = S[n];
          re += v.x * w.x - v.y * w.y;
          im += v.x * w.y + v.y * w.x;
        } else {
          re += s[n] * w.x;
          im += s[n] * w.y;
        }
        idx += k;
        if (idx >= a.K) idx -= a.K;
      }
      re = eps0d(re); im = eps0d(im);
      if (a.has_scale) { re = re / a.div; im = im / a.div; }
      z[k] = make_double2(re, im);
    }
  }
}

// ------------------------------------------------------------------------------------------ row transforms
struct FftRowsArgsD {
  const void* in;          // f64[rows][n_in] or c128[rows][n_in]
  int32_t in_is_real;
  int64_t rows;
  int32_t n_in, K, logK, F;
  const double2* tw;
  BlueD blue;
  // epilogue (istft :611-628): out = ((v / K for the inverse) -> eps) * scale * window[k]
  const double* post_window;
  double post_scale;
  int32_t has_post_scale;
  double2* out;            // c128[rows][K]
};

__device__ __forceinline__ double2 load_row_d(const FftRowsArgsD& a, int64_t r, int n) {
  if (a.in_is_real) return make_double2(reinterpret_cast<const double*>(a.in)[(size_t)r * a.n_in + n], 0.0);
  return reinterpret_cast<const double2*>(a.in)[(size_t)r * a.n_in + n];
}
template <bool INV>
__device__ __forceinline__ double2 rows_epilogue_d(const FftRowsArgsD& a, double2 v, int k, bool pow2) {
  if (INV) {
    if (pow2) { const double inv = 1.0 / (double)a.K; v.x *= inv; v.y *= inv; }   // exact for powers of two
    else { v.x = v.x / (double)a.K; v.y = v.y / (double)a.K; }
  }
  v = eps0d(v);
  if (a.has_post_scale) { v.x *= a.post_scale; v.y *= a.post_scale; }
  if (a.post_window) { const double w = a.post_window[k]; v.x *= w; v.y *= w; }
  return v;
}

template <bool INV, int KIND>
__global__ __launch_bounds__(kT) void k_fft_rows_d(FftRowsArgsD a) {
  double2* S = reinterpret_cast<double2*>(g_smem_d);
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * a.F;
  const int nuse = a.n_in < a.K ? a.n_in : a.K;
  double2* out = a.out + (size_t)r0 * a.K;
  if (KIND == 0) {
    const int total = a.F * a.K;
    for (int idx = tid; idx < total; idx += kT) {
      const int f = idx >> a.logK, n = idx & (a.K - 1);
      const int64_t r = r0 + f;
      double2 v = make_double2(0.0, 0.0);
      if (r < a.rows && n < nuse) v = load_row_d(a, r, n);
      S[(size_t)f * a.K + brev(n, a.logK)] = v;
    }
    __syncthreads();
    lds_fft_d<INV>(S, a.K, a.logK, a.F, a.tw);
    for (int idx = tid; idx < total; idx += kT) {
      const int f = idx >> a.logK;
      if (r0 + f >= a.rows) break;
      out[idx] = rows_epilogue_d<INV>(a, S[idx], idx & (a.K - 1), true);
    }
  } else if (KIND == 1) {
    const BlueD& t = a.blue;
    for (int n = tid; n < t.P; n += kT) {
      double2 u = make_double2(0.0, 0.0);
      if (n < nuse) {
        double2 v = load_row_d(a, r0, n);
        if (INV) v.y = -v.y;   // IDFT(z) = conj(DFT(conj z)) / K
        u = cmuld(v, t.chirp[n]);
      }
      S[brev(n, t.logP)] = u;
    }
    __syncthreads();
    bluestein_core_d(S, t);
    for (int k = tid; k < a.K; k += kT) {
      double2 v = S[k];
      if (INV) v.y = -v.y;
      out[k] = rows_epilogue_d<INV>(a, v, k, false);
    }
  } else {
    for (int n = tid; n < nuse; n += kT) S[n] = load_row_d(a, r0, n);
    __syncthreads();
    for (int k = tid; k < a.K; k += kT) {
      double re = 0.0, im = 0.0;
      int idx = 0;
      for (int n = 0; n < nuse; ++n) {
        const double2 w = twd<INV>(a.tw, idx);
        const double2 v = S[n];
        re += v.x * w.x - v.y * w.y;
        im += v.x * w.y + v.y * w.x;
        idx += k;
        if (idx >= a.K) idx -= a.K;
      }
      out[k] = rows_epilogue_d<INV>(a, make_double2(re, im), k, false);
    }
  }
}

// ------------------------------------------------------------------------------------------ overlap-add
// out[n] = sum over frames m ASCENDING of frames[m][n - m hop] (Nx.indexed_add's order on the BinaryBackend, :724);
// NORM: den[n] = sum_m |w|^2[n - m hop], out /= (den > 1e-10 ? den : 1)   (:630-637)
// WF32: the caller's window is f32 (widened exactly): |w|^2 is the f32 product, the sum is rounded to f32 and compared in f32
template <int COMPS, bool NORM, bool WF32>
__global__ __launch_bounds__(kT) void k_ola_d(const double* __restrict__ frames, int64_t M, int32_t N, int32_t hop,
                                             const double* __restrict__ window, double* __restrict__ out, int64_t out_len) {
  const int64_t n = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (n >= out_len) return;
  const double* fr = frames + (size_t)blockIdx.y * M * N * COMPS;
  int64_t m_hi = n / hop;
  if (m_hi > M - 1) m_hi = M - 1;
  const int64_t m_lo = (n - N + 1 <= 0) ? 0 : (n - N + hop) / hop;
  double acc[COMPS];
#pragma unroll
  for (int c = 0; c < COMPS; ++c) acc[c] = 0.0;
  double den = 0.0;
  for (int64_t m = m_lo; m <= m_hi; ++m) {
    const int32_t j = (int32_t)(n - m * hop);
    const double* p = fr + ((size_t)m * N + j) * COMPS;
#pragma unroll
    for (int c = 0; c < COMPS; ++c) acc[c] += p[c];
    if (NORM) {
      if (WF32) { const float w = fabsf((float)window[j]); den += (double)(w * w); }
      else { const double w = fabs(window[j]); den += w * w; }
    }
  }
  double* o = out + ((size_t)blockIdx.y * out_len + n) * COMPS;
  double d = 1.0;
  if (NORM) {
    if (WF32) { const float df = (float)den; d = df > 1.0e-10f ? (double)df : 1.0; }
    else d = den > 1.0e-10 ? den : 1.0;
  }
#pragma unroll
  for (int c = 0; c < COMPS; ++c) o[c] = NORM ? acc[c] / d : acc[c];
}

__global__ __launch_bounds__(kT) void k_as_windowed_d(const double* __restrict__ x, int64_t batch_stride, GeomD g, double* __restrict__ out) {
  const int64_t total = g.M * g.N;
  const double* xr = x + (size_t)blockIdx.y * batch_stride;
  double* o = out + (size_t)blockIdx.y * total;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int64_t m = i / g.N;
    const int n = (int)(i - m * g.N);
    o[i] = fetch_padded_d(xr, g, m * g.hop + n);
  }
}

// ------------------------------------------------------------------------------------------ FIR (overlap-save)
struct FirArgsD {
  const double* x;
  int64_t L, batch_stride;
  int32_t B, logB, taps;
  int64_t nblocks, first_block, out_start, out_len;
  const double2* H;     // c128[B] = FFT_B(h zero-padded)
  const double2* tw;
  double* y;
  int* row_flags;
};

__global__ __launch_bounds__(kT) void k_fir_os_d(FirArgsD a) {
  double2* S = reinterpret_cast<double2*>(g_smem_d);
  const int tid = threadIdx.x;
  const int64_t V = a.B - (a.taps - 1);
  const int64_t b1 = a.first_block + 2 * (int64_t)blockIdx.x, b2 = b1 + 1;
  const bool have2 = (b2 - a.first_block) < a.nblocks;
  const double* x = a.x + (size_t)blockIdx.y * a.batch_stride;
  const int64_t s1 = b1 * V - (a.taps - 1), s2 = b2 * V - (a.taps - 1);
  double nfsum = 0.0;
  for (int t = tid; t < a.B; t += kT) {
    const int64_t p1 = s1 + t, p2 = s2 + t;
    const double v1 = (p1 >= 0 && p1 < a.L) ? x[p1] : 0.0;
    const double v2 = (have2 && p2 >= 0 && p2 < a.L) ? x[p2] : 0.0;
    S[brev(t, a.logB)] = make_double2(v1, v2);
    nfsum += v1 * 0.0 + v2 * 0.0;   // 0 for finite samples, NaN otherwise
  }
  if (nfsum != nfsum) atomicOr(a.row_flags + blockIdx.y, 1);   // the row comes out NaN: see FirLaunch::row_flags
  __syncthreads();
  lds_fft_d<false>(S, a.B, a.logB, 1, a.tw);
  for (int i = tid; i < a.B; i += kT) {
    const int r = brev(i, a.logB);
    if (i < r) {
      const double2 u = cmuld(S[i], a.H[i]), v = cmuld(S[r], a.H[r]);
      S[i] = v;
      S[r] = u;
    } else if (i == r) {
      S[i] = cmuld(S[i], a.H[i]);
    }
  }
  __syncthreads();
  lds_fft_d<true>(S, a.B, a.logB, 1, a.tw);
  const double invB = 1.0 / (double)a.B;
  double* y = a.y + (size_t)blockIdx.y * a.out_len;
  for (int t = a.taps - 1 + tid; t < a.B; t += kT) {
    const double2 v = S[t];
    const int64_t n1 = b1 * V + (t - (a.taps - 1)) - a.out_start;
    if (n1 >= 0 && n1 < a.out_len) y[n1] = eps0d(v.x * invB);   // the Nx.ifft clean-up of fftconvolve (convolution.ex:282)
    const int64_t n2 = n1 + V;
    if (have2 && n2 >= 0 && n2 < a.out_len) y[n2] = eps0d(v.y * invB);
  }
}

__global__ __launch_bounds__(kT) void k_fir_poison_d(int* __restrict__ flags, double* __restrict__ y, int64_t out_len) {
  const int64_t row = blockIdx.x;
  if (flags[row] == 0) return;
  double* yr = y + (size_t)row * out_len;
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  for (int64_t i = threadIdx.x; i < out_len; i += kT) yr[i] = qnan;
  __syncthreads();
  if (threadIdx.x == 0) flags[row] = 0;
}

// ------------------------------------------------------------------------------------------ host side
int ilog2i(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
bool is_pow2i(int v) { return v > 0 && (v & (v - 1)) == 0; }

template <typename KernelT>
int ensure_lds_d(KernelT kernel, size_t bytes) {
  if (bytes > 160 * 1024) return set_error(NXSIG_ERR_UNSUPPORTED, "f64 tier: the transform does not fit the 160 KiB of LDS");
  if (bytes > 64 * 1024)
    NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return NXSIG_OK;
}

const long double kPiL = 3.141592653589793238462643383279502884L;

// exp(-2 pi i j / K) for j < count, in long double, rounded once
void twiddles_host(int K, int count, std::vector<double2>& out) {
  out.resize(count);
  for (int j = 0; j < count; ++j) {
    const long double ang = -2.0L * kPiL * (long double)j / (long double)K;
    out[j] = make_double2((double)cosl(ang), (double)sinl(ang));
  }
}

// in-place radix-2 transform of P = 2^logP points in long double (host: Bluestein kernel spectra, filter spectra)
void host_fft_ld(std::vector<long double>& re, std::vector<long double>& im) {
  const size_t n = re.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    for (size_t k = 0; k < len / 2; ++k) {
      const long double ang = -2.0L * kPiL * (long double)k / (long double)len;
      const long double wr = cosl(ang), wi = sinl(ang);
      for (size_t i = k; i < n; i += len) {
        const size_t j = i + len / 2;
        const long double tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;
        re[j] = re[i] - tr; im[j] = im[i] - ti;
        re[i] += tr; im[i] += ti;
      }
    }
  }
}

// cached device table of one (kind, K): kinds 1 = K/2 twiddles, 2 = K twiddles, 3 = chirp, 4 = Bluestein kernel spectrum
int table_d(Ctx* c, int kind, int K, const double2** out) {
  const int64_t key = ((int64_t)kind << 32) | (uint32_t)K;
  auto it = c->f64_tables.find(key);
  if (it != c->f64_tables.end()) { *out = reinterpret_cast<const double2*>(it->second); return NXSIG_OK; }
  std::vector<double2> h;
  if (kind == 1) twiddles_host(K, K / 2 > 0 ? K / 2 : 1, h);
  else if (kind == 2) twiddles_host(K, K, h);
  else {
    int P = 1;
    while (P < 2 * K - 1) P <<= 1;
    std::vector<long double> cr(K), ci(K);
    for (int n = 0; n < K; ++n) {   // n^2 mod 2K in integers keeps the angle small
      const int64_t q = ((int64_t)n * n) % (2 * (int64_t)K);
      const long double ang = -kPiL * (long double)q / (long double)K;
      cr[n] = cosl(ang); ci[n] = sinl(ang);
    }
    if (kind == 3) {
      h.resize(K);
      for (int n = 0; n < K; ++n) h[n] = make_double2((double)cr[n], (double)ci[n]);
    } else {
      std::vector<long double> br(P, 0.0L), bi(P, 0.0L);
      for (int n = 0; n < K; ++n) {
        br[n] = cr[n]; bi[n] = -ci[n];
        if (n) { br[P - n] = cr[n]; bi[P - n] = -ci[n]; }
      }
      host_fft_ld(br, bi);
      h.resize(P);
      for (int i = 0; i < P; ++i) h[i] = make_double2((double)(br[i] / (long double)P), (double)(bi[i] / (long double)P));
    }
  }
  const void* d = nullptr;
  int rc = ctx_table(c, 0xD64ull ^ ((uint64_t)kind << 40) ^ ((uint64_t)K << 8), h.data(), h.size() * sizeof(double2), &d);
  if (rc) return rc;
  c->f64_tables[key] = d;
  *out = reinterpret_cast<const double2*>(d);
  return NXSIG_OK;
}

// which kernel a transform length takes (0 / 1 / 2 as in the header; -1 = outside the tier)
int kind_of(int K, int nuse) {
  if (is_pow2i(K) && K <= 8192) return 0;
  if (!is_pow2i(K) && K > 64 && K <= 4096) return 1;
  if (K <= 65536 && nuse <= 8192) return 2;
  return -1;
}

int blue_of(Ctx* c, int K, BlueD* t) {
  t->K = K;
  t->P = 1;
  while (t->P < 2 * K - 1) t->P <<= 1;
  t->logP = ilog2i(t->P);
  int rc;
  if ((rc = table_d(c, 3, K, &t->chirp))) return rc;
  if ((rc = table_d(c, 4, K, &t->Bf))) return rc;
  return table_d(c, 1, t->P, &t->twP);
}

int rows_per_block(int K) {
  int F = 1;
  while (F * K < 1024 && F < 256) F <<= 1;
  return F;
}

}  // namespace

template <int KIND>
static void launch_stft_kind_d(bool cx, dim3 grid, size_t lds, hipStream_t stream, const StftArgsD& a) {
  if (cx) hipLaunchKernelGGL((k_stft_d<KIND, true>), grid, dim3(kT), lds, stream, a);
  else hipLaunchKernelGGL((k_stft_d<KIND, false>), grid, dim3(kT), lds, stream, a);
}

int launch_stft_f64(Ctx* c, const StftLaunchD& s) {
  if (s.fr.M == 0 || s.batch == 0) return NXSIG_OK;
  const bool cx = s.x_is_complex != 0;
  StftArgsD a{};
  a.x = s.x; a.batch_stride = s.batch_stride;
  a.g.L = s.fr.L; a.g.lo = s.fr.lo; a.g.M = s.fr.M; a.g.N = s.fr.N; a.g.hop = s.fr.hop; a.g.reflect = s.fr.reflect;
  a.K = s.K; a.window = s.window; a.div = s.div; a.has_scale = s.has_scale; a.z = s.z;
  const int nuse = s.fr.N < s.K ? s.fr.N : s.K;
  const int kind = kind_of(s.K, nuse);
  int rc;
  if (kind == 0) {
    a.logK = ilog2i(s.K); a.F = rows_per_block(s.K);
    if ((rc = table_d(c, 1, s.K, &a.tw))) return rc;
    const size_t lds = (size_t)a.F * s.K * sizeof(double2);
    if ((rc = cx ? ensure_lds_d(k_stft_d<0, true>, lds) : ensure_lds_d(k_stft_d<0, false>, lds))) return rc;
    dim3 grid((unsigned)((s.fr.M + a.F - 1) / a.F), (unsigned)s.batch);
    launch_stft_kind_d<0>(cx, grid, lds, c->stream, a);
  } else if (kind == 1) {
    a.F = 1;
    if ((rc = blue_of(c, s.K, &a.blue))) return rc;
    const size_t lds = (size_t)a.blue.P * sizeof(double2);
    if ((rc = cx ? ensure_lds_d(k_stft_d<1, true>, lds) : ensure_lds_d(k_stft_d<1, false>, lds))) return rc;
    launch_stft_kind_d<1>(cx, dim3((unsigned)s.fr.M, (unsigned)s.batch), lds, c->stream, a);
  } else if (kind == 2) {
    a.F = 1;
    if ((rc = table_d(c, 2, s.K, &a.tw))) return rc;
    const size_t lds = (size_t)(nuse > 0 ? nuse : 1) * (cx ? sizeof(double2) : sizeof(double));
    if ((rc = cx ? ensure_lds_d(k_stft_d<2, true>, lds) : ensure_lds_d(k_stft_d<2, false>, lds))) return rc;
    launch_stft_kind_d<2>(cx, dim3((unsigned)s.fr.M, (unsigned)s.batch), lds, c->stream, a);
  } else {
    return set_error(NXSIG_ERR_UNSUPPORTED, "stft (f64): fft_length above 65536 (or a non-power-of-two one with more than 8192 samples per frame) "
                                            "is outside the f64 tier; the f32 path transforms up to 2^26 points");
  }
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

int launch_fft_f64(Ctx* c, const void* in, bool in_is_real, int64_t rows, int32_t n_in, int32_t K, bool inverse, double2* out,
                   const double* post_window, double post_scale, bool has_post_scale) {  // defaults: nxsig_internal.h
  if (rows == 0) return NXSIG_OK;
  FftRowsArgsD a{};
  a.in = in; a.in_is_real = in_is_real; a.rows = rows; a.n_in = n_in; a.K = K; a.out = out;
  a.post_window = post_window; a.post_scale = post_scale; a.has_post_scale = has_post_scale;
  const int nuse = n_in < K ? n_in : K;
  const int kind = kind_of(K, nuse);
  int rc;
  auto go = [&](auto kf, auto ki, unsigned blocks, size_t lds) -> int {
    if ((rc = ensure_lds_d(kf, lds))) return rc;
    if ((rc = ensure_lds_d(ki, lds))) return rc;
    if (inverse) hipLaunchKernelGGL(ki, dim3(blocks), dim3(kT), lds, c->stream, a);
    else hipLaunchKernelGGL(kf, dim3(blocks), dim3(kT), lds, c->stream, a);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  if (rows > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "fft (f64): too many rows for one launch");
  if (kind == 0) {
    a.logK = ilog2i(K); a.F = rows_per_block(K);
    if ((rc = table_d(c, 1, K, &a.tw))) return rc;
    return go(k_fft_rows_d<false, 0>, k_fft_rows_d<true, 0>, (unsigned)((rows + a.F - 1) / a.F), (size_t)a.F * K * sizeof(double2));
  }
  if (kind == 1) {
    a.F = 1;
    if ((rc = blue_of(c, K, &a.blue))) return rc;
    return go(k_fft_rows_d<false, 1>, k_fft_rows_d<true, 1>, (unsigned)rows, (size_t)a.blue.P * sizeof(double2));
  }
  if (kind == 2) {
    a.F = 1;
    if ((rc = table_d(c, 2, K, &a.tw))) return rc;
    return go(k_fft_rows_d<false, 2>, k_fft_rows_d<true, 2>, (unsigned)rows, (size_t)(nuse > 0 ? nuse : 1) * sizeof(double2));
  }
  return set_error(NXSIG_ERR_UNSUPPORTED, "fft (f64): lengths above 65536 (or non-power-of-two rows of more than 8192 elements) are outside "
                                          "the f64 tier; the f32 path transforms up to 2^26 points");
}

int launch_ola_f64(Ctx* c, const double* frames, int64_t M, int32_t batch, int32_t N, int32_t hop, int32_t comps, const double* window,
                   bool norm, bool window_f32, double* out) {
  const int64_t out_len = M * hop + (N - hop);
  if (out_len <= 0 || batch == 0) return NXSIG_OK;
  dim3 grid((unsigned)((out_len + kT - 1) / kT), (unsigned)batch);
  if (comps == 1) {
    if (!norm) hipLaunchKernelGGL((k_ola_d<1, false, false>), grid, dim3(kT), 0, c->stream, frames, M, N, hop, window, out, out_len);
    else if (window_f32) hipLaunchKernelGGL((k_ola_d<1, true, true>), grid, dim3(kT), 0, c->stream, frames, M, N, hop, window, out, out_len);
    else hipLaunchKernelGGL((k_ola_d<1, true, false>), grid, dim3(kT), 0, c->stream, frames, M, N, hop, window, out, out_len);
  } else {
    if (!norm) hipLaunchKernelGGL((k_ola_d<2, false, false>), grid, dim3(kT), 0, c->stream, frames, M, N, hop, window, out, out_len);
    else if (window_f32) hipLaunchKernelGGL((k_ola_d<2, true, true>), grid, dim3(kT), 0, c->stream, frames, M, N, hop, window, out, out_len);
    else hipLaunchKernelGGL((k_ola_d<2, true, false>), grid, dim3(kT), 0, c->stream, frames, M, N, hop, window, out, out_len);
  }
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// istft :609-637 in c128: Nx.ifft rows -> x scale -> x window (into scratch) -> overlap-add with the |w|^2 normaliser
int launch_istft_f64(Ctx* c, const IstftLaunchD& s) {
  if (s.M == 0 || s.batch == 0) return NXSIG_OK;
  void* fr = nullptr;
  const int64_t rows = (int64_t)s.batch * s.M;
  int rc = ctx_scratch(c, 27, (size_t)rows * s.N * sizeof(double2), &fr);
  if (rc) return rc;
  if ((rc = launch_fft_f64(c, s.z, false, rows, s.K, s.K, true, reinterpret_cast<double2*>(fr), s.window, s.scale_mul, s.has_scale != 0)))
    return rc;
  return launch_ola_f64(c, reinterpret_cast<const double*>(fr), s.M, s.batch, s.N, s.hop, 2, s.window, true, s.window_f32 != 0,
                        reinterpret_cast<double*>(s.y));
}

int launch_as_windowed_f64(Ctx* c, const double* x, int64_t batch_stride, int32_t batch, const Framing& f, double* out) {
  if (f.M == 0 || batch == 0) return NXSIG_OK;
  GeomD g;
  g.L = f.L; g.lo = f.lo; g.M = f.M; g.N = f.N; g.hop = f.hop; g.reflect = f.reflect;
  const int64_t total = f.M * f.N;
  int64_t blocks = (total + kT - 1) / kT;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_as_windowed_d, dim3((unsigned)blocks, (unsigned)batch), dim3(kT), 0, c->stream, x, batch_stride, g, out);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

int fir_row_flags(Ctx* c, int32_t batch, int** out);   // kernels_generic.hip

int launch_fir_f64(Ctx* c, const FirLaunchD& s) {
  if (s.out_len <= 0 || s.batch == 0) return NXSIG_OK;
  if (s.taps > 4097)
    return set_error(NXSIG_ERR_UNSUPPORTED, "fir (f64): more than 4097 taps is outside the f64 tier (overlap-save blocks of 8192 samples)");
  int B = 1024;
  while (B < 4 * s.taps && B < 8192) B <<= 1;
  while (B < 2 * (s.taps - 1)) B <<= 1;
  std::vector<long double> re(B, 0.0L), im(B, 0.0L);
  for (int i = 0; i < s.taps; ++i) re[i] = (long double)s.h_host[i];
  host_fft_ld(re, im);
  std::vector<double2> H(B);
  for (int i = 0; i < B; ++i) H[i] = make_double2((double)re[i], (double)im[i]);
  FirArgsD a{};
  const void* Hd = nullptr;
  int rc = ctx_table(c, 0xF1AD000000000000ull ^ (uint64_t)B, H.data(), H.size() * sizeof(double2), &Hd);
  if (rc) return rc;
  a.H = reinterpret_cast<const double2*>(Hd);
  if ((rc = table_d(c, 1, B, &a.tw))) return rc;
  if ((rc = fir_row_flags(c, s.batch, &a.row_flags))) return rc;
  a.x = s.x; a.L = s.L; a.batch_stride = s.batch_stride; a.B = B; a.logB = ilog2i(B); a.taps = s.taps;
  const int64_t V = B - (s.taps - 1);
  a.first_block = s.out_start / V;
  const int64_t last_block = (s.out_start + s.out_len - 1) / V;
  a.nblocks = last_block - a.first_block + 1;
  a.out_start = s.out_start; a.out_len = s.out_len; a.y = s.y;
  const size_t lds = (size_t)B * sizeof(double2);
  if ((rc = ensure_lds_d(k_fir_os_d, lds))) return rc;
  dim3 grid((unsigned)((a.nblocks + 1) / 2), (unsigned)s.batch);
  hipLaunchKernelGGL(k_fir_os_d, grid, dim3(kT), lds, c->stream, a);
  NXSIG_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_fir_poison_d, dim3((unsigned)s.batch), dim3(kT), 0, c->stream, a.row_flags, s.y, s.out_len);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

}  // namespace nxsig
