// Generic gfx950 kernels of the STFT / iSTFT / FIR path: correct for EVERY shape the reference accepts
// (any frame length, hop, fft_length, padding mode, batch).  The tuned wave kernels in kernels_wave.hip take over for
// fft_length 128..2048 (stft), N = 512 / 1024 (istft) and <= 513 taps (fir).
//
//   k_stft_pow2       frame slice x window -> workgroup Stockham radix-4/2 FFT in LDS -> scale -> c64 store
//   k_stft_blue       non-power-of-two fft_length 65..4096: Bluestein chirp-z through the same LDS FFT
//   k_stft_dft        remaining lengths: direct DFT, table twiddles, double accumulation
//   k_fft_rows_*      Nx.fft / Nx.ifft(length:) over rows (+ optional x scale x window epilogue for istft)
//   k_ola             deterministic overlap-add (+ |w|^2 normaliser with the 1e-10 guard) in fixed frame order
//   k_istft_edge_fix  the ONE pass after any istft main kernel: f64 recomputation of the few ill-conditioned OLA samples and, for the
//                     kernels that invert several frames per transform, of the units they reported as holding a non-finite bin
//   k_as_windowed     framing gather
//   k_fir_os          overlap-save block convolution, two real blocks packed as re/im of one complex FFT
//   k_cmul_inplace    pointwise product of complex fftconvolve
//   k_mel_pass1/2     stft_to_mel: sparse band sums + log10, global max, clamp
//
// Twiddles are generated on the host in double and read from a table (never __sinf/__cosf).
// Reference lines: lib/nx_signal.ex:94-102 (frame/window/fft), :113-127 (scaling), :486-513 (stft_to_mel),
// :609-637 (istft), :684-736 (overlap_and_add), lib/nx_signal/convolution.ex:252-329 (fftconvolve).
#include <hip/hip_runtime.h>

#include <cstring>

#include "nxsig_internal.h"

namespace nxsig {

static constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 mul_mi(float2 a) {
  return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
template <bool INV>
__device__ __forceinline__ float2 twid(const float2* __restrict__ tw, int idx) {
  float2 w = tw[idx];
  if (INV) w.y = -w.y;
  return w;
}

struct FrameGeom {
  int64_t L, lo, M;
  int32_t N, hop, reflect;
};

// sample of the (virtually) padded signal at padded index q — lib/nx_signal.ex:338 (Nx.pad, zeros) / :349 (Nx.reflect)
__device__ __forceinline__ float fetch_padded(const float* __restrict__ x, const FrameGeom& g, int64_t q) {
  int64_t pos = q - g.lo;
  if (g.reflect) {
    if (g.L == 1) return x[0];
    const int64_t period = 2 * (g.L - 1);
    pos %= period;
    if (pos < 0) pos += period;
    if (pos >= g.L) pos = period - pos;
    return x[pos];
  }
  return (pos >= 0 && pos < g.L) ? x[pos] : 0.0f;
}

// Workgroup-wide Stockham autosort FFT of F rows of K points (K = 2^logK), ping-ponging between LDS
// buffers a (input) and b.  Pass with radix R and current sub-length p: for i in [0, K/R):
//   k = i mod p ; u_t = a[i + t K/R] * w^(t k K/(pR)) ; v = DFT_R(u) ; b[(i-k) R + k + r p] = v_r.
// Returns the buffer that holds the natural-order result.
template <bool INV>
__device__ float2* lds_fft_pow2(float2* a, float2* b, int K, int logK, int F, const float2* __restrict__ tw) {
  const int tid = threadIdx.x;
  int p = 1, logp = 0;
  while (logK - logp >= 2) {
    const int q = K >> 2;
    const int logq = logK - 2;
    const int total = F * q;
    const int step = K >> (logp + 2);
    for (int w = tid; w < total; w += kThreads) {
      const int f = w >> logq, i = w & (q - 1);
      const int k = i & (p - 1);
      const float2* src = a + (size_t)f * K;
      float2 u0 = src[i], u1 = src[i + q], u2 = src[i + 2 * q], u3 = src[i + 3 * q];
      if (p > 1) {
        u1 = cmul(u1, twid<INV>(tw, k * step));
        u2 = cmul(u2, twid<INV>(tw, 2 * k * step));
        u3 = cmul(u3, twid<INV>(tw, 3 * k * step));
      }
      const float2 s02 = cadd(u0, u2), d02 = csub(u0, u2);
      const float2 s13 = cadd(u1, u3), d13 = mul_mi<INV>(csub(u1, u3));
      float2* dst = b + (size_t)f * K + ((i - k) << 2) + k;
      dst[0] = cadd(s02, s13);
      dst[p] = cadd(d02, d13);
      dst[2 * p] = csub(s02, s13);
      dst[3 * p] = csub(d02, d13);
    }
    __syncthreads();
    float2* t = a; a = b; b = t;
    p <<= 2;
    logp += 2;
  }
  if (logK - logp == 1) {
    const int q = K >> 1;
    const int logq = logK - 1;
    const int total = F * q;
    const int step = K >> (logp + 1);
    for (int w = tid; w < total; w += kThreads) {
      const int f = w >> logq, i = w & (q - 1);
      const int k = i & (p - 1);
      const float2* src = a + (size_t)f * K;
      float2 u0 = src[i], u1 = src[i + q];
      if (p > 1) u1 = cmul(u1, twid<INV>(tw, k * step));
      float2* dst = b + (size_t)f * K + ((i - k) << 1) + k;
      dst[0] = cadd(u0, u1);
      dst[p] = csub(u0, u1);
    }
    __syncthreads();
    float2* t = a; a = b; b = t;
  }
  return a;
}

// ------------------------------------------------------------------------------------------ STFT
struct StftArgs {
  const float* x;
  int64_t batch_stride;
  FrameGeom g;
  int32_t K, logK, F;       // F rows per workgroup
  const float* window;
  const float2* tw;
  float div;                // spectrum / div
  int32_t has_scale;
  float2* z;
};

extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];

__global__ __launch_bounds__(kThreads) void k_stft_pow2(StftArgs a) {
  float2* A = reinterpret_cast<float2*>(g_smem);
  float2* B = A + (size_t)a.F * a.K;
  const int tid = threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.x * a.F;
  const float* x = a.x + (size_t)blockIdx.y * a.batch_stride;
  const int nuse = a.g.N < a.K ? a.g.N : a.K;  // Nx.fft(length: K): rows zero-padded or truncated to K
  const int total = a.F * a.K;
  for (int idx = tid; idx < total; idx += kThreads) {
    const int f = idx >> a.logK, n = idx & (a.K - 1);
    const int64_t m = m0 + f;
    float v = 0.0f;
    if (m < a.g.M && n < nuse) v = fetch_padded(x, a.g, m * a.g.hop + n) * a.window[n];  // :101 exact f32 product
    A[idx] = make_float2(v, 0.0f);
  }
  __syncthreads();
  float2* R = lds_fft_pow2<false>(A, B, a.K, a.logK, a.F, a.tw);
  float2* z = a.z + ((size_t)blockIdx.y * a.g.M + m0) * a.K;
  for (int idx = tid; idx < total; idx += kThreads) {
    const int f = idx >> a.logK;
    if (m0 + f >= a.g.M) break;
    float2 v = fft_eps0(R[idx]);  // Nx.fft's clean-up comes before the scaling (:102, :113)
    if (a.has_scale) { v.x = v.x / a.div; v.y = v.y / a.div; }  // :116/:119 true division, like the reference
    z[idx] = v;
  }
}

// non-power-of-two fft_length: one workgroup per frame, direct DFT with table twiddles
__global__ __launch_bounds__(kThreads) void k_stft_dft(StftArgs a) {
  float* s = reinterpret_cast<float*>(g_smem);
  const int tid = threadIdx.x;
  const int64_t m = blockIdx.x;
  const float* x = a.x + (size_t)blockIdx.y * a.batch_stride;
  const int nuse = a.g.N < a.K ? a.g.N : a.K;
  for (int n = tid; n < nuse; n += kThreads) s[n] = fetch_padded(x, a.g, m * a.g.hop + n) * a.window[n];
  __syncthreads();
  float2* z = a.z + ((size_t)blockIdx.y * a.g.M + m) * a.K;
  for (int k = tid; k < a.K; k += kThreads) {
    double dre = 0.0, dim = 0.0;  // long sums (this path serves K > 4096): accumulate in double
    int idx = 0;
    for (int n = 0; n < nuse; ++n) {
      const float2 w = a.tw[idx];
      dre += (double)s[n] * (double)w.x;
      dim += (double)s[n] * (double)w.y;
      idx += k;
      if (idx >= a.K) idx -= a.K;
    }
    float re = fft_eps0((float)dre), im = fft_eps0((float)dim);
    if (a.has_scale) { re = re / a.div; im = im / a.div; }
    z[k] = make_float2(re, im);
  }
}

// ------------------------------------------------------------------------------------------ row FFTs
struct FftRowsArgs {
  const void* in;           // f32[rows][n_in] or c64[rows][n_in]
  int32_t in_is_real;
  int64_t rows;
  int32_t n_in, K, logK, F;
  const float2* tw;
  // epilogue (istft :611-628): out = (v * scale) * window[k]; inverse transforms also apply 1/K first
  const float* post_window;
  float post_scale;
  int32_t has_post_scale;
  int32_t clean;            // 1: Nx.fft / Nx.ifft eps clean-up of the finished transform (0: sub-step of a longer transform)
  float2* out;              // c64[rows][K]
};

template <bool INV>
__device__ __forceinline__ float2 fft_epilogue(const FftRowsArgs& a, float2 v, int k) {
  if (INV) {
    const float invK = 1.0f / (float)a.K;  // exact for powers of two; the DFT path divides instead
    v.x *= invK; v.y *= invK;
  }
  if (a.clean) v = fft_eps0(v);  // Nx.fft / Nx.ifft clean-up, before whatever the caller multiplies in
  if (a.has_post_scale) { v.x *= a.post_scale; v.y *= a.post_scale; }
  if (a.post_window) { const float w = a.post_window[k]; v.x *= w; v.y *= w; }
  return v;
}

template <bool INV>
__global__ __launch_bounds__(kThreads) void k_fft_rows_pow2(FftRowsArgs a) {
  float2* A = reinterpret_cast<float2*>(g_smem);
  float2* B = A + (size_t)a.F * a.K;
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * a.F;
  const int nuse = a.n_in < a.K ? a.n_in : a.K;
  const int total = a.F * a.K;
  for (int idx = tid; idx < total; idx += kThreads) {
    const int f = idx >> a.logK, n = idx & (a.K - 1);
    const int64_t r = r0 + f;
    float2 v = make_float2(0.0f, 0.0f);
    if (r < a.rows && n < nuse) {
      if (a.in_is_real) v.x = reinterpret_cast<const float*>(a.in)[(size_t)r * a.n_in + n];
      else v = reinterpret_cast<const float2*>(a.in)[(size_t)r * a.n_in + n];
    }
    A[idx] = v;
  }
  __syncthreads();
  float2* R = lds_fft_pow2<INV>(A, B, a.K, a.logK, a.F, a.tw);
  float2* out = a.out + (size_t)r0 * a.K;
  for (int idx = tid; idx < total; idx += kThreads) {
    const int f = idx >> a.logK;
    if (r0 + f >= a.rows) break;
    out[idx] = fft_epilogue<INV>(a, R[idx], idx & (a.K - 1));
  }
}

template <bool INV>
__global__ __launch_bounds__(kThreads) void k_fft_rows_dft(FftRowsArgs a) {
  float2* s = reinterpret_cast<float2*>(g_smem);
  const int tid = threadIdx.x;
  const int64_t r = blockIdx.x;
  const int nuse = a.n_in < a.K ? a.n_in : a.K;
  for (int n = tid; n < nuse; n += kThreads) {
    float2 v = make_float2(0.0f, 0.0f);
    if (a.in_is_real) v.x = reinterpret_cast<const float*>(a.in)[(size_t)r * a.n_in + n];
    else v = reinterpret_cast<const float2*>(a.in)[(size_t)r * a.n_in + n];
    s[n] = v;
  }
  __syncthreads();
  float2* out = a.out + (size_t)r * a.K;
  for (int k = tid; k < a.K; k += kThreads) {
    double dre = 0.0, dim = 0.0;
    int idx = 0;
    for (int n = 0; n < nuse; ++n) {
      const float2 w = twid<INV>(a.tw, idx);
      const float2 v = s[n];
      dre += (double)v.x * (double)w.x - (double)v.y * (double)w.y;
      dim += (double)v.x * (double)w.y + (double)v.y * (double)w.x;
      idx += k;
      if (idx >= a.K) idx -= a.K;
    }
    float2 v = make_float2((float)dre, (float)dim);
    if (INV) { v.x = v.x / (float)a.K; v.y = v.y / (float)a.K; }
    if (a.clean) v = fft_eps0(v);
    if (a.has_post_scale) { v.x *= a.post_scale; v.y *= a.post_scale; }
    if (a.post_window) { const float w = a.post_window[k]; v.x *= w; v.y *= w; }
    out[k] = v;
  }
}

// ------------------------------------------------------------------------------------------ Bluestein (chirp-z)
// Non-power-of-two lengths K in (64, 4096]: X[k] = c[k] * sum_n (x[n] c[n]) conj(c)[k - n], c[n] = exp(-i pi n^2 / K),
// i.e. one circular convolution of length P = 2^ceil(log2(2K-1)) done with the LDS Stockham FFT:
// FFT_P -> x Bf (spectrum of the conj-chirp kernel, host-computed in double, pre-scaled by 1/P) -> IFFT_P -> x c[k].
// n^2 mod 2K is reduced in integers on the host, so the chirp tables are accurate to f32 rounding.
struct BlueTables {
  int32_t K, P, logP;
  const float2* chirp;   // [K]
  const float2* Bf;      // [P]
  const float2* twP;     // [P] forward twiddles of the P-point FFT
};

// A holds the row (complex, natural order) in [0, nuse), zeros up to P; returns the buffer whose first K entries are X
template <bool INV>
__device__ float2* bluestein_lds(float2* A, float2* B, const BlueTables& t, int nuse) {
  const int tid = threadIdx.x;
  for (int n = tid; n < nuse; n += kThreads) {
    float2 v = A[n];
    if (INV) v.y = -v.y;  // IDFT(z) = conj(DFT(conj z)) / K
    A[n] = cmul(v, t.chirp[n]);
  }
  __syncthreads();
  float2* R = lds_fft_pow2<false>(A, B, t.P, t.logP, 1, t.twP);
  float2* O = (R == A) ? B : A;
  for (int i = tid; i < t.P; i += kThreads) R[i] = cmul(R[i], t.Bf[i]);
  __syncthreads();
  float2* Y = lds_fft_pow2<true>(R, O, t.P, t.logP, 1, t.twP);
  for (int k = tid; k < t.K; k += kThreads) {
    float2 v = cmul(Y[k], t.chirp[k]);
    if (INV) { v.y = -v.y; v.x = v.x / (float)t.K; v.y = v.y / (float)t.K; }
    Y[k] = v;
  }
  __syncthreads();
  return Y;
}

struct StftBlueArgs {
  StftArgs s;
  BlueTables t;
};

__global__ __launch_bounds__(kThreads) void k_stft_blue(StftBlueArgs b) {
  const StftArgs& a = b.s;
  float2* A = reinterpret_cast<float2*>(g_smem);
  float2* Bb = A + b.t.P;
  const int tid = threadIdx.x;
  const int64_t m = blockIdx.x;
  const float* x = a.x + (size_t)blockIdx.y * a.batch_stride;
  const int nuse = a.g.N < a.K ? a.g.N : a.K;
  for (int n = tid; n < b.t.P; n += kThreads) {
    float v = 0.0f;
    if (n < nuse) v = fetch_padded(x, a.g, m * a.g.hop + n) * a.window[n];
    A[n] = make_float2(v, 0.0f);
  }
  __syncthreads();
  float2* Y = bluestein_lds<false>(A, Bb, b.t, nuse);
  float2* z = a.z + ((size_t)blockIdx.y * a.g.M + m) * a.K;
  for (int k = tid; k < a.K; k += kThreads) {
    float2 v = fft_eps0(Y[k]);
    if (a.has_scale) { v.x = v.x / a.div; v.y = v.y / a.div; }
    z[k] = v;
  }
}

struct FftBlueArgs {
  FftRowsArgs f;
  BlueTables t;
};

template <bool INV>
__global__ __launch_bounds__(kThreads) void k_fft_rows_blue(FftBlueArgs b) {
  const FftRowsArgs& a = b.f;
  float2* A = reinterpret_cast<float2*>(g_smem);
  float2* Bb = A + b.t.P;
  const int tid = threadIdx.x;
  const int64_t r = blockIdx.x;
  const int nuse = a.n_in < a.K ? a.n_in : a.K;
  for (int n = tid; n < b.t.P; n += kThreads) {
    float2 v = make_float2(0.0f, 0.0f);
    if (n < nuse) {
      if (a.in_is_real) v.x = reinterpret_cast<const float*>(a.in)[(size_t)r * a.n_in + n];
      else v = reinterpret_cast<const float2*>(a.in)[(size_t)r * a.n_in + n];
    }
    A[n] = v;
  }
  __syncthreads();
  float2* Y = bluestein_lds<INV>(A, Bb, b.t, nuse);
  float2* out = a.out + (size_t)r * a.K;
  for (int k = tid; k < a.K; k += kThreads) {
    float2 v = a.clean ? fft_eps0(Y[k]) : Y[k];
    if (a.has_post_scale) { v.x *= a.post_scale; v.y *= a.post_scale; }
    if (a.post_window) { const float w = a.post_window[k]; v.x *= w; v.y *= w; }
    out[k] = v;
  }
}

// ------------------------------------------------------------------------------------------ OLA
// out[n] = sum over frames m (ascending) of frames[m][n - m hop]; contributions are accumulated in double
// and rounded once (what Nx.indexed_add does on BinaryBackend, SURVEY App. A rule 8) -> deterministic.
// NORM: also den[n] = sum_m w2[n - m hop] (w2 = f32(|w|^2)), out /= (den > 1e-10 ? den : 1)  (:630-637)
template <int COMPS, bool NORM>
__global__ __launch_bounds__(kThreads) void k_ola(const float* __restrict__ frames, int64_t M, int32_t N, int32_t hop,
                                                 const float* __restrict__ window, float* __restrict__ out,
                                                 int64_t out_len) {
  const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (n >= out_len) return;
  const float* fr = frames + (size_t)blockIdx.y * M * N * COMPS;
  int64_t m_hi = n / hop;
  if (m_hi > M - 1) m_hi = M - 1;
  int64_t m_lo = (n - N + hop) / hop;  // ceil((n - N + 1) / hop) for n-N+1 > 0
  if (n - N + 1 <= 0) m_lo = 0;
  double acc[COMPS];
#pragma unroll
  for (int c = 0; c < COMPS; ++c) acc[c] = 0.0;
  double den = 0.0;
  for (int64_t m = m_lo; m <= m_hi; ++m) {
    const int32_t j = (int32_t)(n - m * hop);
    const float* p = fr + ((size_t)m * N + j) * COMPS;
#pragma unroll
    for (int c = 0; c < COMPS; ++c) acc[c] += (double)p[c];
    if (NORM) { const float w = fabsf(window[j]); den += (double)(w * w); }
  }
  float* o = out + ((size_t)blockIdx.y * out_len + n) * COMPS;
  if (NORM) {
    float d = (float)den;
    d = d > 1.0e-10f ? d : 1.0f;
#pragma unroll
    for (int c = 0; c < COMPS; ++c) o[c] = (float)acc[c] / d;
  } else {
#pragma unroll
    for (int c = 0; c < COMPS; ++c) o[c] = (float)acc[c];
  }
}

// ------------------------------------------------------------------------------------------ iSTFT edge fix-up
// Where the OLA normaliser den[n] is tiny (signal edges under a tapered window, or every frame boundary when
// hop == N) the division amplifies the fp32 round-off of the IFFT by 1/sqrt(den): the reference does not suffer
// from it because Nx.ifft works in double and rounds each (tiny) sample RELATIVELY (SURVEY App. A rule 7).  Those
// few samples are recomputed here exactly along the reference's chain: frame sample by direct evaluation of the
// inverse DFT in double -> round to f32 -> x scale -> x window (f32 roundings, lib/nx_signal.ex:611-628) ->
// summed over frames in double, rounded once (:724) -> / den (:637).  One workgroup per candidate sample.
struct EdgeFixArgs {
  const float2* z;          // c64[batch][M][K]
  int64_t M;
  int32_t N, hop;           // K == N
  const float* window;
  float scale;
  int32_t has_scale;
  float2* y;                // c64[batch][out_len]
  int64_t out_len;
  const int64_t* idx;       // flagged samples, host-computed (1e-10 < den[n] < tau): {n, m_lo, m_hi, bits of (float)den} each; nullptr = scan all n
  int64_t n_idx;
  int32_t batch = 1;        // rows (edge role of k_istft_edge_fix)
  float tau;
  const double2* tw;        // w_N^j = exp(+2 pi i j / N) in double, j in [0, N)
  const float2* filt;       // optional c64[N]: the frames are z * filt rounded to c64 (IstftLaunch::filt), or nullptr
  int32_t onesided = 0;     // 1: z is the packed half spectrum c64[batch][M][N / 2] (Re X[N/2] in the imaginary part of bin 0)
                            //    and y is REAL f32[batch][out_len] (nxsig_istft_packed_f32)
};
// bin k of frame row zr in either layout
__device__ __forceinline__ float2 istft_bin(const EdgeFixArgs& a, const float2* __restrict__ zr, int k) {
  if (!a.onesided) return zr[k];
  const int half = a.N >> 1;
  if (k == 0) return make_float2(zr[0].x, 0.0f);
  if (k < half) return zr[k];
  if (k == half) return make_float2(zr[0].y, 0.0f);
  const float2 c = zr[a.N - k];
  return make_float2(c.x, -c.y);
}

// one output sample of one row, along the reference's chain in double, by ONE WAVE (round 4: a whole workgroup per sample spent its
// time in an 8-level LDS reduction with barriers; a wave keeps N / 64 terms per lane in flight and reduces with six shuffles, and four
// times as many samples are resident per CU — the pass after config 3's kernel went from 15.6 to 10.6 us in rocprofv3, together with the twiddle recurrence below).
// the sum itself: frames m_lo .. m_hi cover sample n, d = the guarded normaliser
__device__ __forceinline__ void istft_sample_body(const EdgeFixArgs& a, const int64_t row, const int64_t n, const int64_t m_lo, const int64_t m_hi,
                                                  const float d) {
  const int lane = threadIdx.x & 63;
  const int rowlen = a.onesided ? (a.N >> 1) : a.N;
  const float2* zb = a.z + (size_t)row * a.M * rowlen;
  double acc_re = 0.0, acc_im = 0.0;
  for (int64_t m = m_lo; m <= m_hi; ++m) {
    const int j = (int)(n - m * a.hop);
    const float2* zr = zb + (size_t)m * rowlen;
    double sr = 0.0, si = 0.0;
    // twiddle index j k mod N, advanced without a division per term (32-bit: j < N, and N < 2^24 for every caller)
    int tix = a.N < (1 << 24) ? (int)((uint32_t)(j * lane) % (uint32_t)a.N) : (int)(((int64_t)j * lane) % a.N);
    const int tstep = a.N < (1 << 24) ? (int)((uint32_t)(j * 64) % (uint32_t)a.N) : (int)(((int64_t)j * 64) % a.N);
    auto term = [&](const double2 t, float2 v, const int k) {
      if (a.filt) {
        const float2 h = a.filt[k];
        v = make_float2((float)((double)v.x * (double)h.x - (double)v.y * (double)h.y),
                        (float)((double)v.x * (double)h.y + (double)v.y * (double)h.x));
      }
      sr += (double)v.x * t.x - (double)v.y * t.y;
      si += (double)v.x * t.y + (double)v.y * t.x;
    };
    // 16 spectrum loads in flight per lane (the sum is a chain of N / 64 dependent round trips otherwise), ONE table look-up per 16
    // terms: w^(j k) is a 64-way gather of 16-byte entries over up to 64 cache lines, and 16 of those per wave kept the CU's texture
    // path busy for most of this pass (4 416 waves after config 3's kernel: the pass took 20 us under the profiler with the gather, 13.5 without).  The other 15 twiddles follow by
    // w^(j (k + 64)) = w^(j k) w^(64 j) in double: at most 15 roundings of 1.1e-16 on values that are then rounded to f32.
    constexpr int U = 16;
    const double2 wstep = a.tw[tstep];
    const int cstep = a.N < (1 << 24) ? (int)((uint32_t)(tstep * U) % (uint32_t)a.N) : (int)(((int64_t)tstep * U) % a.N);
    int k = lane;
    for (; k + 64 * (U - 1) < a.N; k += 64 * U) {
      double2 t = a.tw[tix];
      float2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = istft_bin(a, zr, k + 64 * u);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        term(t, v[u], k + 64 * u);
        t = make_double2(t.x * wstep.x - t.y * wstep.y, t.x * wstep.y + t.y * wstep.x);
      }
      tix += cstep;
      if (tix >= a.N) tix -= a.N;
    }
    for (; k < a.N; k += 64) {
      term(a.tw[tix], istft_bin(a, zr, k), k);
      tix += tstep;
      if (tix >= a.N) tix -= a.N;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sr += __shfl_xor(sr, off); si += __shfl_xor(si, off); }   // every lane ends with the same sums
    double dr = sr / (double)a.N, di = si / (double)a.N;
    if (fabs(dr) <= 1.0e-10) dr = 0.0;   // Nx.ifft's eps clean-up, on the double result like the reference (App. A rule 7)
    if (fabs(di) <= 1.0e-10) di = 0.0;
    float fr = (float)dr, fi = (float)di;  // Nx.ifft rounds to c64
    if (a.has_scale) { fr *= a.scale; fi *= a.scale; }
    const float w = a.window[j];
    fr *= w; fi *= w;
    acc_re += (double)fr; acc_im += (double)fi;
  }
  if (lane == 0) {
    if (a.onesided) reinterpret_cast<float*>(a.y)[(size_t)row * a.out_len + n] = (float)acc_re / d;
    else a.y[(size_t)row * a.out_len + n] = make_float2((float)acc_re / d, (float)acc_im / d);
  }
}

// ALL = false: only ill-conditioned samples (1e-10 < den < tau) are recomputed; true: any sample (den <= 1e-10 divides by 1, :635)
template <bool ALL>
__device__ __forceinline__ void istft_sample_f64(const EdgeFixArgs& a, const int64_t row, const int64_t n) {
  int64_t m_hi = n / a.hop;
  if (m_hi > a.M - 1) m_hi = a.M - 1;
  const int64_t m_lo = (n - a.N + 1 <= 0) ? 0 : (n - a.N + a.hop) / a.hop;
  double den = 0.0;
  for (int64_t m = m_lo; m <= m_hi; ++m) {
    const float w = fabsf(a.window[n - m * a.hop]);
    den += (double)(w * w);
  }
  float d = (float)den;
  if (ALL) { if (!(d > 1.0e-10f)) d = 1.0f; }
  else if (!(d > 1.0e-10f) || d >= a.tau) return;  // uniform across the wave
  istft_sample_body(a, row, n, m_lo, m_hi, d);
}

constexpr int kFixWaves = kThreads / 64;   // samples per workgroup of the two fix-up kernels

// ONE pass after any istft main kernel, two roles (round 4: they were three launches — a memset of the list's count, the edge fix-up
// and the non-finite fix-up — on every call of the frame-packing kernels):
//  * workgroups [0, edge_blocks * batch): the edge fix-up.  A wave takes one candidate sample of one row; with `idx` the candidates and
//    their frame range / normaliser come from the host ({n, m_lo, m_hi, bits of d} per entry), without it every sample is a candidate.
//  * workgroups beyond: non-finite bins under kernels that invert SEVERAL frames with one transform (k_istft_wave_half: 2, _quad:
//    4 / 8).  The reference inverts every frame on its own (Nx.ifft row by row, lib/nx_signal.ex:609), so an Inf / NaN bin reaches only
//    the samples of ITS frame; inside a shared transform it reaches the partner frames' samples too.  Those kernels therefore report
//    every unit that holds a non-finite bin ((row << 40) | first frame, appended to a device list), and these workgroups recompute the
//    output samples the unit's frames touch with the per-sample chain above — frame by frame, exactly as the reference does.  Done out
//    of line on purpose: an in-kernel "solo" route cost the streaming kernels their third wave per SIMD (-12 %).
//    list[0] = number of entries appended (may exceed the capacity list[1]); entries follow as int64 from list + 2; list[-2] is a
//    ticket: the workgroup that finishes last puts the count back to zero, so an empty list (every call but the poisoned ones) costs
//    one load per workgroup and nothing is cleared between calls.
__global__ __launch_bounds__(kThreads) void k_istft_edge_fix(EdgeFixArgs a, int64_t edge_blocks, int* __restrict__ list, int32_t frames_per_unit,
                                                             int32_t nf_blocks) {
  const int wave = threadIdx.x >> 6;
  const int64_t b = blockIdx.x;
  if (b < edge_blocks * a.batch) {
    const int64_t row = b / edge_blocks;
    const int64_t i = (b - row * edge_blocks) * kFixWaves + wave;
    if (a.idx) {
      if (i >= a.n_idx) return;
      const int64_t* e = a.idx + 4 * i;
      istft_sample_body(a, row, e[0], e[1], e[2], __int_as_float((int)e[3]));
    } else if (i < a.out_len) {
      istft_sample_f64<false>(a, row, i);
    }
    return;
  }
  int cnt = list[0];
  if (cnt <= 0) return;                // the usual case; nothing to put back either
  if (cnt > list[1]) cnt = list[1];
  const int64_t* ent = reinterpret_cast<const int64_t*>(list + 2);
  const int64_t span = (int64_t)(frames_per_unit - 1) * a.hop + a.N;   // samples the unit's frames touch
  const int64_t total = (int64_t)cnt * span;
  const int64_t nb = b - edge_blocks * a.batch;
  for (int64_t wk = nb * kFixWaves + wave; wk < total; wk += (int64_t)nf_blocks * kFixWaves) {
    const int64_t e = wk / span, si = wk - e * span;
    const int64_t row = ent[e] >> 40, m0 = ent[e] & (((int64_t)1 << 40) - 1);
    const int64_t n = m0 * a.hop + si;
    if (row < a.batch && n < a.out_len) istft_sample_f64<true>(a, row, n);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(list - 2, 1) == nf_blocks - 1) {   // every workgroup of this role has read the count and done its share
      atomicExch(list, 0);
      atomicExch(list - 2, 0);
    }
  }
}

// ---- the edge role for N = A SP, SP in {2, 4, 8, 16, 32}, A <= 64 lanes (round 5; the text below says 64 for A).  k_istft_edge_fix spends N / 64 complex multiply-adds plus a
// twiddle recurrence per lane on EVERY candidate sample; with thousands of short rows (2 048 rows of 184 frames: 280 candidates per
// row) that pass took longer than the istft kernel itself (0.84 against 0.70 ms).  Here one wave takes a CHUNK of SP consecutive
// output positions n0 .. n0 + SP - 1 (n0 a multiple of SP) that share their frame range:
//   x_m[j] = 1/N sum_lane w_N^(j lane) sum_s Z_m[lane + 64 s] e^(2 pi i j s / SP),   j = n - m hop
// the inner sum has period SP in j: ONE lane-local SP-point transform per frame (radix-2 Stockham in double, static register indices)
// yields it for all SP positions of the chunk (a frame whose j0 = n0 - m hop is not a multiple of SP is rotated first), the outer sums
// of the SP positions are reduced together (butterfly reduction: SP - 1 + 6 - log2 SP exchanges instead of 6 SP), and the lane group
// that ends up with position g carries it through the reference's chain (divide by N, clean-up, round to f32, x scale, x window, sum
// over the frames in double, divide by the normaliser: lib/nx_signal.ex:609-637).  Same roundings as istft_sample_body except for the
// order of the double sums.  Chunk entry (20 x int64, host-built): n0, m_lo, m_hi, mask of the flagged positions, then the bits of the
// f32 normalisers, two per int64.
template <int SP>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(SP >= 32 ? 1 : 2, 8))) void k_istft_edge_chunks(EdgeFixArgs a, int64_t chunk_blocks) {
  constexpr int LOG = SP == 2 ? 1 : SP == 4 ? 2 : SP == 8 ? 3 : SP == 16 ? 4 : 5;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = blockIdx.x;
  const int64_t row = b / chunk_blocks;
  const int64_t ci = (b - row * chunk_blocks) * kFixWaves + wave;
  if (ci >= a.n_idx) return;
  const int64_t* e = a.idx + 20 * ci;
  const int64_t n0 = e[0], m_lo = e[1], m_hi = e[2];
  const uint32_t mask = (uint32_t)e[3];
  const int g = lane >> (6 - LOG);                       // the position this lane group finishes
  const bool mine = ((mask >> g) & 1u) != 0;
  const int N = a.N, rowlen = a.onesided ? (N >> 1) : N;
  const int A = N / SP;                                  // active lanes: bin k = lane + A s
  const float2* zb = a.z + (size_t)row * a.M * rowlen;
  double acc_re = 0.0, acc_im = 0.0;
  for (int64_t m = m_lo; m <= m_hi; ++m) {
    const int64_t j0 = n0 - m * a.hop;                   // may be negative for positions outside the mask
    const int rot = (int)(((j0 % SP) + SP) % SP);        // uniform
    const float2* zr = zb + (size_t)m * rowlen;
    double2 x[SP];
#pragma unroll
    for (int s = 0; s < SP; ++s) {
      float2 v = lane < A ? istft_bin(a, zr, lane + A * s) : make_float2(0.0f, 0.0f);
      if (a.filt && lane < A) {
        const float2 h = a.filt[lane + A * s];
        v = make_float2((float)((double)v.x * (double)h.x - (double)v.y * (double)h.y),
                        (float)((double)v.x * (double)h.y + (double)v.y * (double)h.x));
      }
      x[s] = make_double2((double)v.x, (double)v.y);
    }
    if (rot != 0) {   // x_s *= e^(2 pi i rot s / SP) = w_N^(64 rot s)
#pragma unroll
      for (int s = 1; s < SP; ++s) {
        const double2 t = a.tw[A * ((rot * s) % SP)];
        x[s] = make_double2(x[s].x * t.x - x[s].y * t.y, x[s].x * t.y + x[s].y * t.x);
      }
    }
    // SP-point transform with the + sign IN PLACE: radix-2 decimation in frequency, every index a compile-time constant; the result is
    // in bit-reversed order, X[c] sits in x[rev(c)] (a natural-order Stockham form with a second array held 242 VGPRs for SP = 16)
#pragma unroll
    for (int half = SP / 2; half >= 1; half /= 2) {
#pragma unroll
      for (int blk = 0; blk < SP; blk += 2 * half) {
#pragma unroll
        for (int j = 0; j < half; ++j) {
          const double2 u = x[blk + j], v = x[blk + j + half];
          x[blk + j] = make_double2(u.x + v.x, u.y + v.y);
          double2 d = make_double2(u.x - v.x, u.y - v.y);
          if (j != 0) {
            const double2 t = a.tw[A * (j * (SP / (2 * half)))];   // e^(2 pi i j / (2 half))
            d = make_double2(d.x * t.x - d.y * t.y, d.x * t.y + d.y * t.x);
          }
          x[blk + j + half] = d;
        }
      }
    }
    // position gg <-> inner sum with index gg (after the rotation above) = x[rev(gg)]; outer products with w_N^(j lane), j = j0 + gg, in place
    if (SP >= 16) __builtin_amdgcn_sched_barrier(0);
    {
      int tix = (int)((uint32_t)((int)(((j0 % N) + N) % N) * lane) % (uint32_t)N);   // (j0 lane) mod N, then + lane per position (lane < N)
#pragma unroll
      for (int gg = 0; gg < SP; ++gg) {
        int rv = 0;
#pragma unroll
        for (int bit = 0; bit < LOG; ++bit) rv |= ((gg >> bit) & 1) << (LOG - 1 - bit);
        const double2 t = a.tw[tix];
        const double2 xv = x[rv];
        x[rv] = make_double2(xv.x * t.x - xv.y * t.y, xv.x * t.y + xv.y * t.x);
        tix += lane;
        if (tix >= N) tix -= N;
        if (SP >= 16 && (gg & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // eight table gathers in flight, not SP (registers)
      }
    }
    // butterfly reduction on the bit-reversed storage: step q folds position bit LOG - 1 - q = storage bit q (stride 2^q) and lane bit
    // 5 - q; after it the lane holds SP >> (q + 1) partial sums over 2^(q + 1) lanes, the last one in x[0]
#pragma unroll
    for (int q = 0; q < LOG; ++q) {
      const int stride = 1 << q, off = 32 >> q;
      const bool up = (lane & off) != 0;
#pragma unroll
      for (int i = 0; i < SP; i += 2 * stride) {
        const double2 keep = up ? x[i + stride] : x[i];
        const double2 send = up ? x[i] : x[i + stride];
        x[i] = make_double2(keep.x + __shfl_xor(send.x, off), keep.y + __shfl_xor(send.y, off));
      }
    }
    double sr = x[0].x, si = x[0].y;
#pragma unroll
    for (int off = 32 >> LOG; off > 0; off >>= 1) { sr += __shfl_xor(sr, off); si += __shfl_xor(si, off); }
    const int64_t j = j0 + g;
    if (mine && j >= 0 && j < N) {
      double dr = sr / (double)N, di = si / (double)N;
      if (fabs(dr) <= 1.0e-10) dr = 0.0;   // Nx.ifft's eps clean-up, on the double result like the reference
      if (fabs(di) <= 1.0e-10) di = 0.0;
      float fr = (float)dr, fi = (float)di;
      if (a.has_scale) { fr *= a.scale; fi *= a.scale; }
      const float w = a.window[j];
      fr *= w; fi *= w;
      acc_re += (double)fr; acc_im += (double)fi;
    }
  }
  if (mine && (lane & ((64 >> LOG) - 1)) == 0) {
    const float d = __int_as_float((int)(((uint64_t)e[4 + (g >> 1)] >> (32 * (g & 1))) & 0xffffffffu));
    const int64_t n = n0 + g;
    if (a.onesided) reinterpret_cast<float*>(a.y)[(size_t)row * a.out_len + n] = (float)acc_re / d;
    else a.y[(size_t)row * a.out_len + n] = make_float2((float)acc_re / d, (float)acc_im / d);
  }
}

// ------------------------------------------------------------------------------------------ framing
__global__ __launch_bounds__(kThreads) void k_as_windowed(const float* __restrict__ x, int64_t batch_stride, FrameGeom g,
                                                         float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t total = g.M * g.N;
  if (idx >= total) return;
  const int64_t m = idx / g.N;
  const int32_t n = (int32_t)(idx - m * g.N);
  out[(size_t)blockIdx.y * total + idx] = fetch_padded(x + (size_t)blockIdx.y * batch_stride, g, m * g.hop + n);
}

// ------------------------------------------------------------------------------------------ FIR (generic)
// Overlap-save: block b produces full-convolution outputs [b V, (b+1) V), V = B - (taps-1), from the B input
// samples x[b V - (taps-1) + t].  Two consecutive blocks ride in the real and imaginary lanes of ONE complex
// FFT (h is real, so IFFT(FFT(x1 + i x2) H) = x1*h + i x2*h): no Hermitian split pass is needed.
struct FirArgs {
  const float* x;
  int64_t L, batch_stride;
  int32_t B, logB, taps;
  int64_t nblocks;          // number of V-sized output blocks covering the requested slice
  int64_t first_block;      // index of the first block
  int64_t out_start, out_len;
  const float2* H;          // device c64[B] = FFT_B(h zero-padded), computed on the host in double
  const float2* tw;
  float* y;
  int* row_flags;           // FirLaunch::row_flags
};

// Last pass of every overlap-save FIR call (see FirLaunch::row_flags): one workgroup per row; a flagged row becomes NaN from
// end to end, as the reference's single whole-row transform leaves it, and its flag is cleared for the next call.  Unflagged
// rows cost one load.  (A poisoned row is written by one workgroup: the cold path favours simplicity.)
__global__ __launch_bounds__(kThreads) void k_fir_poison(int* __restrict__ flags, float* __restrict__ y, int64_t out_len) {
  const int64_t row = blockIdx.x;
  if (flags[row] == 0) return;   // uniform across the workgroup
  float* yr = y + (size_t)row * out_len;
  const float qnan = __int_as_float(0x7fc00000);
  for (int64_t i = threadIdx.x; i < out_len; i += kThreads) yr[i] = qnan;
  __syncthreads();
  if (threadIdx.x == 0) flags[row] = 0;
}

int fir_row_flags(Ctx* c, int32_t batch, int** out) {
  const size_t need = (size_t)batch * sizeof(int);
  if (c->scratch_bytes[22] < need) {
    void* p = nullptr;
    const size_t bytes = need < 4096 ? 4096 : need * 2;
    int rc = ctx_scratch(c, 22, bytes, &p);
    if (rc) return rc;
    NXSIG_HIP_TRY(hipMemsetAsync(p, 0, bytes, c->stream));   // zero between calls: the poison pass clears what it consumes
  }
  *out = reinterpret_cast<int*>(c->scratch[22]);
  return NXSIG_OK;
}

// sample-sharded FIR (group.cpp): a member's slice came out NaN from end to end when ITS samples held an Inf / NaN (the poison pass
// above); flags[r] = "row r of this slice is poisoned", read off the slice's first output, is what the members all-reduce so that
// every slice of such a row ends up NaN like the reference's one transform leaves the whole row
__global__ __launch_bounds__(kThreads) void k_fir_flags_from_output(const float* __restrict__ y, int32_t rows, int64_t out_len, int* __restrict__ flags) {
  const int r = blockIdx.x * kThreads + threadIdx.x;
  if (r >= rows) return;
  const float v = y[(size_t)r * out_len];
  flags[r] = (v - v == 0.0f) ? 0 : 1;
}
int launch_fir_flags_from_output(Ctx* c, const float* y, int32_t rows, int64_t out_len, int* flags) {
  if (rows <= 0 || out_len <= 0) return NXSIG_OK;   // an empty slice holds nothing: its flags stay zero
  hipLaunchKernelGGL(k_fir_flags_from_output, dim3((unsigned)((rows + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream, y, rows, out_len, flags);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// ---- filters longer than one overlap-save block takes (launch_fir_partitioned, api.cpp): y[row][i] = sum over the partitions p that
// reach output i of part_p[row][i - i0_p].  Partition p is x * h[p S .. p S + taps_p) — a plain FIR call of the tuned kernels —
// and enters the full convolution p S samples late.  Up to eight partitions per pass; later passes add to y.
struct FirSumArgs {
  const float* src[8];
  int64_t i0[8], len[8];     // output indices [i0, i0 + len) of y that partition p reaches; its rows are len floats apart
  int32_t n, accumulate, clean;   // clean: last pass — Nx.ifft's clean-up of fftconvolve's result (convolution.ex:282) on the finished sums
  float scale;                    // the partitions were computed with h x 2^20 (see launch_fir_partitioned): x 2^-20 here, exact
  float* y;
  int64_t out_len;
};

__global__ __launch_bounds__(256) void k_fir_partition_sum(FirSumArgs a) {
  const int64_t row = blockIdx.y;
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  float* __restrict__ yr = a.y + (size_t)row * a.out_len;
  // 4-byte accesses, a wave on 256 consecutive bytes (the partitions' rows start at arbitrary offsets); all loads of a thread's four
  // outputs are issued before the sums
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    if (p < a.n) {   // uniform
      const float* __restrict__ sp = a.src[p] + (size_t)row * a.len[p] - a.i0[p];
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t i = base + 256 * e, j = i - a.i0[p];
        v[e] = (j >= 0 && j < a.len[p]) ? __builtin_nontemporal_load(sp + i) : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t i = base + 256 * e;
    if (i < a.out_len) {
      float v = acc[e] * a.scale;
      if (a.accumulate) v += yr[i];
      yr[i] = a.clean ? fft_eps0(v) : v;
    }
  }
}

int launch_fir_partition_sum(Ctx* c, int n, const float* const* src, const int64_t* i0, const int64_t* len, bool accumulate, bool clean, float scale,
                             float* y, int32_t batch, int64_t out_len) {
  FirSumArgs a;
  for (int p = 0; p < 8; ++p) { a.src[p] = p < n ? src[p] : nullptr; a.i0[p] = p < n ? i0[p] : 0; a.len[p] = p < n ? len[p] : 0; }
  a.n = n; a.accumulate = accumulate ? 1 : 0; a.clean = clean ? 1 : 0; a.scale = scale; a.y = y; a.out_len = out_len;
  const int64_t bx = (out_len + 1023) / 1024;
  if (bx > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "fir: output too long for one launch");
  dispatch_note("fir.partition_sum");
  hipLaunchKernelGGL(k_fir_partition_sum, dim3((unsigned)bx, (unsigned)batch), dim3(256), 0, c->stream, a);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

int launch_fir_poison(Ctx* c, const FirLaunch& s) {
  if (!s.row_flags || s.batch == 0 || s.out_len <= 0) return NXSIG_OK;
  hipLaunchKernelGGL(k_fir_poison, dim3((unsigned)s.batch), dim3(kThreads), 0, c->stream, s.row_flags, s.y, s.out_len);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

__global__ __launch_bounds__(kThreads) void k_fir_os(FirArgs a) {
  float2* A = reinterpret_cast<float2*>(g_smem);
  float2* Bf = A + a.B;
  const int tid = threadIdx.x;
  const int64_t V = a.B - (a.taps - 1);
  const int64_t b1 = a.first_block + 2 * (int64_t)blockIdx.x, b2 = b1 + 1;
  const bool have2 = (b2 - a.first_block) < a.nblocks;
  const float* x = a.x + (size_t)blockIdx.y * a.batch_stride;
  const int64_t s1 = b1 * V - (a.taps - 1), s2 = b2 * V - (a.taps - 1);
  float nfsum = 0.0f;
  for (int t = tid; t < a.B; t += kThreads) {
    const int64_t p1 = s1 + t, p2 = s2 + t;
    const float v1 = (p1 >= 0 && p1 < a.L) ? x[p1] : 0.0f;
    const float v2 = (have2 && p2 >= 0 && p2 < a.L) ? x[p2] : 0.0f;
    A[t] = make_float2(v1, v2);
    nfsum += v1 * 0.0f + v2 * 0.0f;   // 0 for finite samples, NaN otherwise
  }
  if (nfsum != nfsum) atomicOr(a.row_flags + blockIdx.y, 1);   // see FirLaunch::row_flags
  __syncthreads();
  float2* R = lds_fft_pow2<false>(A, Bf, a.B, a.logB, 1, a.tw);
  float2* O = (R == A) ? Bf : A;
  for (int t = tid; t < a.B; t += kThreads) R[t] = cmul(R[t], a.H[t]);
  __syncthreads();
  float2* Y = lds_fft_pow2<true>(R, O, a.B, a.logB, 1, a.tw);
  const float invB = 1.0f / (float)a.B;
  float* y = a.y + (size_t)blockIdx.y * a.out_len;
  for (int t = a.taps - 1 + tid; t < a.B; t += kThreads) {
    const float2 v = Y[t];
    const int64_t n1 = b1 * V + (t - (a.taps - 1)) - a.out_start;
    if (n1 >= 0 && n1 < a.out_len) y[n1] = fft_eps0(v.x * invB);  // the Nx.ifft clean-up of fftconvolve (convolution.ex:282)
    const int64_t n2 = n1 + V;
    if (have2 && n2 >= 0 && n2 < a.out_len) y[n2] = fft_eps0(v.y * invB);
  }
}

// pointwise complex product a[i] *= b[i] (the `Nx.multiply(sp1, sp2)` of convolution.ex:282)
__global__ __launch_bounds__(kThreads) void k_cmul_inplace(float2* __restrict__ a, const float2* __restrict__ b, int n) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < n) a[i] = cmul(a[i], b[i]);
}

// out[r][k] = z[r][k] * h[k] (guides/filtering.livemd:141: Nx.multiply(z, hfft)); BinaryBackend multiplies complex
// numbers in double and rounds each component once, so do the same (the kernel is HBM-bound either way)
__global__ __launch_bounds__(kThreads) void k_spectrum_mul(const float2* __restrict__ z, const float2* __restrict__ h,
                                                           float2* __restrict__ out, int64_t total, int32_t K) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const float2 a = z[i], b = h[i % K];
    const double re = (double)a.x * (double)b.x - (double)a.y * (double)b.y;
    const double im = (double)a.x * (double)b.y + (double)a.y * (double)b.x;
    out[i] = make_float2((float)re, (float)im);
  }
}

// |z| (kind 0 / 2) or |z|^2 (kind 1) of the bins below K/2: the two-step form of the fused magnitude sink
__global__ __launch_bounds__(kThreads) void k_mag_from_spectrum(const float2* __restrict__ z, int64_t rows, int32_t K,
                                                                int32_t kind, float* __restrict__ out, int* gmax) {
  const int half = K / 2;
  const int64_t total = rows * half;
  float vmax = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int64_t r = i / half;
    const int k = (int)(i - r * half);
    const float2 v = z[(size_t)r * K + k];
    const float p2 = v.x * v.x + v.y * v.y;
    const float o = kind == 1 ? p2 : sqrtf(p2);
    out[i] = o;
    vmax = o > vmax ? o : vmax;
  }
  if (kind == 2) {
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(vmax, off); vmax = o > vmax ? o : vmax; }
    if ((threadIdx.x & 63) == 0) atomicMax(gmax, __float_as_int(vmax));  // magnitudes are >= 0: plain int order
  }
}
__global__ __launch_bounds__(kThreads) void k_mag_db_pass2_g(float* __restrict__ out, int64_t n, const int* __restrict__ gmax) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const float mx = __int_as_float(*gmax);  // magnitudes are >= 0: stored in plain int order
  const float r = out[i] / mx;
  const float l = logf(r);  // <= 1 ulp from the correctly rounded double log the reference takes
  out[i] = (20.0f * l) / 2.3025851f;
}

// ------------------------------------------------------------------------------------------ stft_to_mel
// lib/nx_signal.ex:486-513.  The Slaney filters are triangles: band b is non-zero on a short bin range, so the
// `Nx.dot` over frequencies is a sparse band sum (per band a [lo, hi) range into the dense filter row).
// Pass 1: per frame |z|^2 of bins < K/2 into LDS, per (frame, band) the band sum in double (the reference's dot
//         accumulates in double), log10 as log(x) / log(10) in f32 steps, block max -> ordered-int atomicMax.
// Pass 2: max(x, gmax - 8), (x + 4) / 4.
struct MelArgs {
  const float2* z;
  int64_t rows;
  int32_t K, half, mel_bins;
  const float* filt;        // device f32[mel_bins][K]
  const int2* band;         // device [mel_bins]: non-zero bin range [lo, hi) within [0, K/2)
  float* out;               // f32[rows][mel_bins]
  int* gmax;                // ordered-int encoding of the running maximum
  float ln10;               // f32(log(10))
};

__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

static constexpr int kMelFramesPerBlock = 16;

// 16 frames per workgroup: |z|^2 of the bins below K / 2 into LDS, then every thread owns one mel band for a subset of the frames
// and walks the band ONCE, each filter weight loaded once and applied to all its frames (the first version re-read the weights
// from L2 for every frame: 1.2 ms on 32 x 60 s, now 0.87; what remains is double-precision arithmetic — the band sums and one
// libm-grade log per output — which the reference's rounding rule asks for; the fused sink of wave_stft.hpp is the fast path).  Rounding as before: Nx.abs in double -> f32 -> squared in f32; the band
// sum in double in ascending bin order, rounded once.
__global__ __launch_bounds__(kThreads) void k_mel_pass1(MelArgs a) {
  constexpr int FB = kMelFramesPerBlock;
  float* mags = reinterpret_cast<float*>(g_smem);  // [FB][half]
  __shared__ int s_max;
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * FB;
  if (tid == 0) s_max = f2ord(-3.0e38f);
  // half is a multiple of kThreads for every power-of-two K >= 512 and the loop is uniform otherwise too: eight independent
  // 8-byte loads in flight per thread
  const int total = FB * a.half;
#pragma unroll 8
  for (int idx = tid; idx < total; idx += kThreads) {
    const int f = idx / a.half, k = idx - f * a.half;
    float m = 0.0f;
    if (r0 + f < a.rows) {
      typedef float v2f_t __attribute__((ext_vector_type(2)));
      const v2f_t v = __builtin_nontemporal_load(reinterpret_cast<const v2f_t*>(a.z + (size_t)(r0 + f) * a.K + k));
      const float ab = (float)sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y);  // Nx.abs(c64) -> f32
      m = ab * ab;                                                                          // ** 2
      if (!(m < INFINITY)) atomicOr(a.gmax + 1, 1);                                         // see k_mel_pass2
    }
    mags[idx] = m;
  }
  __syncthreads();
  int lmax = f2ord(-3.0e38f);
  const int G = a.mel_bins >= kThreads ? 1 : kThreads / a.mel_bins;   // frame groups sharing the workgroup's threads
  const int per = (FB + G - 1) / G;                                    // frames per thread (<= 16)
  for (int t = tid; t < a.mel_bins * G; t += kThreads) {
    const int b = t % a.mel_bins, g = t / a.mel_bins;
    const int2 rg = a.band[b];
    const float* fr = a.filt + (size_t)b * a.K;
    double acc[FB];
#pragma unroll
    for (int i = 0; i < FB; ++i) acc[i] = 0.0;
    for (int k = rg.x; k < rg.y; ++k) {
      const double w = (double)fr[k];
#pragma unroll
      for (int i = 0; i < FB; ++i)
        if (i < per) acc[i] += (double)mags[(g + i * G) % FB * a.half + k] * w;   // frames g, g + G, ...: (index wraps only when unused)
    }
#pragma unroll
    for (int i = 0; i < FB; ++i) {
      const int f = g + i * G;
      if (i >= per || f >= FB || r0 + f >= a.rows) continue;
      float v = (float)acc[i];
      v = v > 1.0e-10f ? v : 1.0e-10f;                       // Nx.clip(mel_spec, 1.0e-10, :infinity)
      v = (float)log((double)v) / a.ln10;                    // Nx.log(.) / Nx.log(10)
      a.out[(size_t)(r0 + f) * a.mel_bins + b] = v;
      const int o = f2ord(v);
      lmax = o > lmax ? o : lmax;
    }
  }
  atomicMax(&s_max, lmax);
  __syncthreads();
  if (tid == 0) atomicMax(a.gmax, s_max);
}

// The tiled form of pass 1 (every fft_length whose |z|^2 tile fits the LDS): FB = 64 / 32 / 16 / 8 frames per workgroup.
//   phase 1  waves stream whole rows: lanes along the bins (512 B per load instruction), four rows in flight per wave;
//            |z|^2 goes to LDS with an odd row stride
//   phase 2  lane = frame: a wave walks one band (64 / FB bands when the tile has fewer than 64 frames) with the loop bounds
//            uniform over the frames — no divergence between narrow and wide bands, the weight read once per 64 frames from a
//            compact LDS copy of the non-zero filter entries, conflict-free |z|^2 reads (odd stride)
//   phase 3  the [FB][mel_bins] result tile leaves as one contiguous, coalesced block
// Same arithmetic and rounding as k_mel_pass1 (Nx.abs in double -> f32 -> squared in f32; band sum in double in ascending bin
// order, rounded once; log in double) with cheaper instruction sequences for the square root and the logarithm (below): on
// 768 256 frames of 400 bins -> 80 bands the pass takes 0.50 ms where k_mel_pass1 took 1.09 (it is bound by VALU issue, ~67 %
// busy, not by its 1.5 GB of traffic), bit-identical to the oracle on every value the tests compare.
struct MelTileArgs {
  MelArgs m;
  const float* cw;       // compact weights: band b's bins [lo, hi) at cw[woff + k - lo]
  const float* wpad;     // the same, [mel_bins][maxw] zero-padded (staged into LDS when it fits)
  const int4* bw;        // [mel_bins]: {lo, hi, woff, widest band of the wave's group (set on the group's first band)}
  const double2* logtab; // [128]: {1 / c_i, log(c_i)}, c_i = 1 + (i + 0.5) / 128
  int32_t nnz;           // entries of cw
  int32_t maxw;          // widest band
  int32_t fb_log2;       // frames per workgroup = 1 << fb_log2 (<= 64)
  int32_t hs;            // odd row stride of the |z|^2 tile
};

// Nx.abs(c64) -> f32 -> ** 2.  The reference takes the square root in double and rounds it to f32; here: re^2 + im^2 in double,
// a 1-ulp f32 square root of it, and one Newton correction whose residual s - r^2 is formed in double — the corrected value
// carries ~5e-15 relative error before the single rounding to f32, so it equals the rounded double square root except when that
// lies within 1e-7 ulp of a rounding boundary (v_rsq_f64 and its refinement cost three times as much).  Values whose f32 square
// would leave the normal range take the plain double path.
__device__ __forceinline__ double mel_norm2(float re, float im) {
  return fma((double)re, (double)re, (double)im * (double)im);
}
// 1e-30 < (float)s < 1e30 as one unsigned compare on the bit pattern (s >= 0)
__device__ __forceinline__ bool mel_abs_fast_ok(double s) { return __float_as_uint((float)s) - 0x0DA24261u < 0x7149F2CAu - 0x0DA24261u; }
__device__ __forceinline__ float mel_abs_fast(double s) {
  const float r = __builtin_amdgcn_sqrtf((float)s);
  const double rd = (double)r;
  const float e = (float)fma(-rd, rd, s);
  return fmaf(e, 0.5f * __builtin_amdgcn_rcpf(r), r);
}
__device__ __forceinline__ float mel_abs2(float re, float im) {
  const double s = mel_norm2(re, im);
  const float ab = mel_abs_fast_ok(s) ? mel_abs_fast(s) : (float)sqrt(s);
  return ab * ab;
}

// log(x) in double for a positive, normal x (the clipped band energies: 1e-10 <= x < 2^128): 7-bit table of {1 / c, log c},
// r = x_mantissa / c - 1 by one fma (|r| <= 2^-8), log1p(r) to the r^7 term; |x - 1| < 2^-8 takes r = x - 1 without the table.
// Relative error ~3e-16: after the rounding to f32 the result equals that of a correctly rounded log except within ~5e-9 ulp of a
// rounding boundary.  A quarter of the instructions of the library's double log (which works in double-double).
__device__ __forceinline__ double mel_log(double x, const double2* tab) {
  const long long bits = __double_as_longlong(x);
  const int hi = (int)(bits >> 32);
  const double2 t = tab[(hi >> 13) & 127];
  const double m = __longlong_as_double((bits & 0x000fffffffffffffll) | 0x3ff0000000000000ll);
  double r = fma(m, t.x, -1.0);
  double lc = t.y;
  double ef = (double)(((hi >> 20) & 0x7ff) - 1023);
  const double d = x - 1.0;
  if (fabs(d) < 0.00390625) { r = d; lc = 0.0; ef = 0.0; }
  double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
  p = fma(p, r, 0.2);
  p = fma(p, r, -0.25);
  p = fma(p, r, 1.0 / 3.0);
  p = fma(p, r, -0.5);
  p = fma(p * r, r, r);
  const double y = fma(ef, 0.69314718055994530942, lc + p);
  return hi >= 0x7ff00000 ? x : y;   // +inf (an overflowed band energy) and NaN pass through like log()
}

template <bool WLDS, int NT>
__global__ __launch_bounds__(NT) void k_mel_tile(MelTileArgs t) {
  typedef float v2f_t __attribute__((ext_vector_type(2)));
  constexpr int NW = NT / 64, RW = 4;
  const MelArgs& a = t.m;
  const int FB = 1 << t.fb_log2;
  float* mags = reinterpret_cast<float*>(g_smem);   // [FB][hs]
  float* res = mags + FB * t.hs + t.maxw;           // [FB][mel_bins], behind a zeroed pad of maxw floats
  float* wl = res + FB * a.mel_bins;                // [mel_bins][maxw], zero-padded band weights (WLDS only)
  __shared__ int s_max;
  __shared__ double2 s_log[128];
  __shared__ int4 s_bw[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)blockIdx.x * FB;
  const int nvalid = a.rows - r0 < FB ? (int)(a.rows - r0) : FB;
  if (tid == 0) s_max = f2ord(-3.0e38f);
  if (WLDS)
    for (int i = tid; i < a.mel_bins * t.maxw; i += NT) wl[i] = t.wpad[i];
  for (int i = tid; i < t.maxw; i += NT) mags[FB * t.hs + i] = 0.0f;
  if (t.hs > a.half)   // the odd row stride leaves one column nobody fills: a narrower band of a group reads it (times a zero weight)
    for (int f = tid; f < FB; f += NT) mags[f * t.hs + a.half] = 0.0f;
  for (int i = tid; i < 128; i += NT) s_log[i] = t.logtab[i];
  for (int i = tid; i < a.mel_bins; i += NT) s_bw[i] = t.bw[i];   // the launcher keeps mel_bins <= 256 on this kernel
  // loads are unconditional (row and bin indices clamped into the tile) so that all eight of an iteration are in flight
  // together; only the LDS stores are predicated
  for (int fb = wave * RW; fb < FB; fb += NW * RW) {
    const float2* zr[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int fr = fb + r < nvalid ? fb + r : nvalid - 1;
      zr[r] = a.z + (size_t)(r0 + fr) * a.K;
    }
    for (int k0 = 0; k0 < a.half; k0 += 128) {
      const int ka = k0 + lane, kb = ka + 64;
      const int ca = ka < a.half ? ka : a.half - 1, cb = kb < a.half ? kb : a.half - 1;
      v2f_t va[RW], vb[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        va[r] = __builtin_nontemporal_load(reinterpret_cast<const v2f_t*>(zr[r] + ca));
        vb[r] = __builtin_nontemporal_load(reinterpret_cast<const v2f_t*>(zr[r] + cb));
      }
      // rows past the end of the input were clamped to the last one: their tile rows hold finite values nobody reads.
      // Straight-line arithmetic for all eight values; the out-of-range square roots (one branch per eight) are redone in double
      double sa[RW], sb[RW];
      float aa[RW], ab[RW];
      bool odd = false;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        sa[r] = mel_norm2(va[r].x, va[r].y);
        sb[r] = mel_norm2(vb[r].x, vb[r].y);
        aa[r] = mel_abs_fast(sa[r]);
        ab[r] = mel_abs_fast(sb[r]);
        odd |= !(mel_abs_fast_ok(sa[r]) && mel_abs_fast_ok(sb[r]));
      }
      if (__builtin_expect(odd, 0)) {
        asm volatile("; out-of-range magnitudes: double square root" ::: "memory");   // keeps this block a branch (not selects)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          if (!mel_abs_fast_ok(sa[r])) aa[r] = (float)sqrt(sa[r]);
          if (!mel_abs_fast_ok(sb[r])) ab[r] = (float)sqrt(sb[r]);
          if (!(aa[r] * aa[r] < INFINITY) || !(ab[r] * ab[r] < INFINITY)) atomicOr(a.gmax + 1, 1);
        }
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        if (fb + r < FB && ka < a.half) mags[(fb + r) * t.hs + ka] = aa[r] * aa[r];
        if (fb + r < FB && kb < a.half) mags[(fb + r) * t.hs + kb] = ab[r] * ab[r];
      }
    }
  }
  __syncthreads();
  const int f = lane & (FB - 1), sub = lane >> t.fb_log2, bpw = 64 >> t.fb_log2;
  const float* mg = mags + f * t.hs;
  int lmax = f2ord(-3.0e38f);
  for (int b0 = wave * bpw; b0 < a.mel_bins; b0 += NW * bpw) {
    const bool live = b0 + sub < a.mel_bins;
    const int b = live ? b0 + sub : a.mel_bins - 1;
    const int4 rg = s_bw[b];
    double acc = 0.0;
    if (WLDS) {
      // the wave's bands walk the same number of bins (the widest of the group, .w of its first band): scalar loop control, the
      // narrower band multiplies zero-padded weights with the bins that follow it (finite x 0 adds nothing; past the last row
      // lies a zeroed pad)
      const int nmax = __builtin_amdgcn_readfirstlane(s_bw[b0].w);
      const float* mk = mg + rg.x;
      const float* wk = wl + b * t.maxw;
#pragma unroll 2
      for (int j = 0; j < nmax; ++j) acc += (double)mk[j] * (double)wk[j];
    } else {
      const float* w = t.cw + rg.z - rg.x;
      for (int k = rg.x; k < rg.y; ++k) acc += (double)mg[k] * (double)w[k];
    }
    float v = (float)acc;
    v = v > 1.0e-10f ? v : 1.0e-10f;                       // Nx.clip(mel_spec, 1.0e-10, :infinity)
    v = (float)mel_log((double)v, s_log) / a.ln10;         // Nx.log(.) / Nx.log(10)
    if (live) res[f * a.mel_bins + b] = v;
    if (live && f < nvalid) {
      const int o = f2ord(v);
      lmax = o > lmax ? o : lmax;
    }
  }
  atomicMax(&s_max, lmax);
  __syncthreads();
  const int n = nvalid * a.mel_bins;
  float* o = a.out + (size_t)r0 * a.mel_bins;
  if (((FB * a.mel_bins) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {   // every tile starts 16-byte aligned
    const int n4 = n >> 2;
    for (int i = tid; i < n4; i += NT) reinterpret_cast<float4*>(o)[i] = reinterpret_cast<const float4*>(res)[i];
    for (int i = (n4 << 2) + tid; i < n; i += NT) o[i] = res[i];
  } else {
    for (int i = tid; i < n; i += NT) o[i] = res[i];
  }
  if (tid == 0) atomicMax(a.gmax, s_max);
}

__global__ __launch_bounds__(kThreads) void k_mel_pass2(float* __restrict__ out, int64_t n, const int* __restrict__ gmax) {
  // gmax[1] != 0: some |z|^2 was inf / NaN.  The reference's dense Nx.dot multiplies it with the zero weights of every band
  // (inf x 0), the NaN reaches reduce_max and from there every element; the sparse band sums here never form that product
  const float floor_v = gmax[1] ? __int_as_float(0x7fc00000) : ord2f(*gmax) - 8.0f;   // Nx.reduce_max(log_spec) - 8
  const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4;
  if (i + 4 <= n && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    float4 v = *reinterpret_cast<float4*>(out + i);
    v.x = ((v.x > floor_v ? v.x : floor_v) + 4.0f) / 4.0f;   // NaN floor: the comparison fails, the NaN is taken
    v.y = ((v.y > floor_v ? v.y : floor_v) + 4.0f) / 4.0f;
    v.z = ((v.z > floor_v ? v.z : floor_v) + 4.0f) / 4.0f;
    v.w = ((v.w > floor_v ? v.w : floor_v) + 4.0f) / 4.0f;
    *reinterpret_cast<float4*>(out + i) = v;
    return;
  }
  for (int64_t j = i; j < n && j < i + 4; ++j) {
    const float v = out[j];
    out[j] = ((v > floor_v ? v : floor_v) + 4.0f) / 4.0f;
  }
}

// ========================================================================================== launchers
int launch_mel_init(Ctx* c, int** gmax);
static int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static constexpr int kMaxLdsPow2 = 8192;  // 2 x 8192 x 8 B = 128 KiB of the CU's 160 KiB LDS

template <typename KernelT>
static int ensure_lds(KernelT kernel, size_t bytes) {
  if (bytes > 64 * 1024) NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return NXSIG_OK;
}

static void host_fft_f64(std::vector<double>& re, std::vector<double>& im);

// chirp / kernel-spectrum tables of the Bluestein path, cached per context by content
static int blue_tables(Ctx* c, int K, BlueTables* t, int forceP = 0) {
  int P = 1, logP = 0;
  while (P < 2 * K - 1 || P < forceP) { P <<= 1; ++logP; }
  const uint64_t bkey = 0xB10E5000000000ull ^ ((uint64_t)K << 24) ^ (uint64_t)P ^ (forceP ? 0x800000000000ull : 0ull);
  {
    auto hit = c->memo.find(bkey);
    if (hit != c->memo.end()) {  // the tables depend on (K, P) only
      t->K = K; t->P = P; t->logP = logP;
      t->chirp = reinterpret_cast<const float2*>(hit->second[0]);
      t->Bf = reinterpret_cast<const float2*>(hit->second[1]);
      t->twP = reinterpret_cast<const float2*>(hit->second[2]);
      return NXSIG_OK;
    }
  }
  std::vector<float2> chirp((size_t)K), Bf((size_t)P);
  std::vector<double> cre((size_t)K), cim((size_t)K), bre((size_t)P, 0.0), bim((size_t)P, 0.0);
  for (int n = 0; n < K; ++n) {
    const int64_t q = ((int64_t)n * n) % (2 * (int64_t)K);  // exact phase index
    const double ang = -3.14159265358979323846 * (double)q / (double)K;
    cre[n] = std::cos(ang); cim[n] = std::sin(ang);
    chirp[n] = make_float2((float)cre[n], (float)cim[n]);
  }
  bre[0] = cre[0]; bim[0] = -cim[0];
  for (int n = 1; n < K; ++n) { bre[n] = cre[n]; bim[n] = -cim[n]; bre[P - n] = cre[n]; bim[P - n] = -cim[n]; }
  host_fft_f64(bre, bim);
  for (int i = 0; i < P; ++i) Bf[i] = make_float2((float)(bre[i] / P), (float)(bim[i] / P));
  const void *dc = nullptr, *db = nullptr;
  int rc = ctx_table(c, 0xB10E0ull ^ (uint64_t)K, chirp.data(), chirp.size() * sizeof(float2), &dc);
  if (rc) return rc;
  rc = ctx_table(c, 0xB10E1ull ^ (uint64_t)K ^ ((uint64_t)P << 20), Bf.data(), Bf.size() * sizeof(float2), &db);
  if (rc) return rc;
  t->K = K; t->P = P; t->logP = logP;
  t->chirp = reinterpret_cast<const float2*>(dc);
  t->Bf = reinterpret_cast<const float2*>(db);
  if (forceP) {
    t->twP = nullptr;
  } else {
    rc = ctx_twiddles(c, P, &t->twP);
    if (rc) return rc;
  }
  c->memo[bkey] = {reinterpret_cast<uint64_t>(t->chirp), reinterpret_cast<uint64_t>(t->Bf), reinterpret_cast<uint64_t>(t->twP)};
  return NXSIG_OK;
}
// tables for the wave-core Bluestein kernel: convolution length fixed to the core size P
int blue_tables_dev(Ctx* c, int K, int P, const float2** chirp, const float2** Bf) {
  BlueTables t;
  int rc = blue_tables(c, K, &t, P);
  if (rc) return rc;
  if (t.P != P) return set_error(NXSIG_ERR_UNSUPPORTED, "bluestein: fft_length does not fit the core");
  *chirp = t.chirp; *Bf = t.Bf;
  return NXSIG_OK;
}
static bool use_bluestein(int K) { return !is_pow2(K) && K > 64 && K <= 4096; }

static FrameGeom to_geom(const Framing& fr) {
  FrameGeom g;
  g.L = fr.L; g.lo = fr.lo; g.M = fr.M; g.N = fr.N; g.hop = fr.hop; g.reflect = fr.reflect;
  return g;
}

int launch_stft_generic(Ctx* c, const StftLaunch& s) {
  StftArgs a;
  a.x = s.x; a.batch_stride = s.batch_stride; a.g = to_geom(s.fr);
  a.K = s.K; a.window = s.window; a.div = s.inv_scale_div; a.has_scale = s.has_scale; a.z = s.z;
  if (s.fr.M == 0 || s.batch == 0) return NXSIG_OK;
  if ((is_pow2(s.K) && s.K > kMaxLdsPow2) || (!is_pow2(s.K) && s.K > 4096)) return launch_stft_big(c, s);  // kernels_nd.hip
  int rc = ctx_twiddles(c, s.K, &a.tw);
  if (rc) return rc;
  if (is_pow2(s.K) && s.K <= kMaxLdsPow2) {
    a.logK = ilog2(s.K);
    a.F = s.K >= 1024 ? 1 : 1024 / s.K;
    const size_t lds = (size_t)2 * a.F * a.K * sizeof(float2);
    rc = ensure_lds(k_stft_pow2, lds);
    if (rc) return rc;
    dim3 grid((unsigned)((s.fr.M + a.F - 1) / a.F), (unsigned)s.batch);
    dispatch_note("stft.generic.pow2");
    hipLaunchKernelGGL(k_stft_pow2, grid, dim3(kThreads), lds, c->stream, a);
  } else if (use_bluestein(s.K)) {
    StftBlueArgs b;
    b.s = a; b.s.logK = 0; b.s.F = 1;
    rc = blue_tables(c, s.K, &b.t);
    if (rc) return rc;
    const size_t lds = (size_t)2 * b.t.P * sizeof(float2);
    rc = ensure_lds(k_stft_blue, lds);
    if (rc) return rc;
    dim3 grid((unsigned)s.fr.M, (unsigned)s.batch);
    dispatch_note("stft.generic.blue");
    hipLaunchKernelGGL(k_stft_blue, grid, dim3(kThreads), lds, c->stream, b);
  } else {
    a.logK = 0; a.F = 1;
    const int nuse = s.fr.N < s.K ? s.fr.N : s.K;
    const size_t lds = (size_t)nuse * sizeof(float);
    dim3 grid((unsigned)s.fr.M, (unsigned)s.batch);
    dispatch_note("stft.generic.dft");
    hipLaunchKernelGGL(k_stft_dft, grid, dim3(kThreads), lds, c->stream, a);
  }
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

int launch_fft_rows_wave(Ctx* c, const void* in, bool in_is_real, int64_t rows, int32_t n_in, int32_t K, bool inverse,
                         const float* post_window, float post_scale, bool has_post_scale, float2* out, bool* handled, bool clean);

static int launch_fft_rows(Ctx* c, const void* in, bool in_is_real, int64_t rows, int32_t n_in, int32_t K, bool inverse,
                           const float* post_window, float post_scale, bool has_post_scale, float2* out, bool clean = true) {
  if (rows == 0) return NXSIG_OK;
  FftRowsArgs a;
  a.in = in; a.in_is_real = in_is_real ? 1 : 0; a.rows = rows; a.n_in = n_in; a.K = K;
  a.post_window = post_window; a.post_scale = post_scale; a.has_post_scale = has_post_scale ? 1 : 0; a.out = out;
  a.clean = clean ? 1 : 0;
  {  // K = 1024 / 2048 / 4096: one wave per row on the wave-private cores (kernels_wave_rows.hip)
    bool handled = false;
    int rcw = launch_fft_rows_wave(c, in, in_is_real, rows, n_in, K, inverse, post_window, post_scale, has_post_scale, out, &handled, clean);
    if (rcw || handled) return rcw;
  }
  if ((is_pow2(K) && (K > kMaxLdsPow2 || K >= fft_tiled_min(c))) || (!is_pow2(K) && K > 4096)) {
    // beyond the LDS-resident kernels: four-step / Bluestein rows in HBM (kernels_nd.hip), then the istft epilogue if any
    int rcb = launch_fft_big(c, in, in_is_real, rows, n_in, K, inverse, out, clean);
    if (rcb) return rcb;
    return launch_rows_post(c, out, rows, K, post_window, post_scale, has_post_scale, 1.0f, false);
  }
  int rc = ctx_twiddles(c, K, &a.tw);
  if (rc) return rc;
  if (is_pow2(K) && K <= kMaxLdsPow2) {
    a.logK = ilog2(K);
    a.F = K >= 1024 ? 1 : 1024 / K;
    const size_t lds = (size_t)2 * a.F * K * sizeof(float2);
    dim3 grid((unsigned)((rows + a.F - 1) / a.F));
    if (inverse) {
      rc = ensure_lds(k_fft_rows_pow2<true>, lds);
      if (rc) return rc;
      dispatch_note("fft.rows_generic.pow2");
      hipLaunchKernelGGL(k_fft_rows_pow2<true>, grid, dim3(kThreads), lds, c->stream, a);
    } else {
      rc = ensure_lds(k_fft_rows_pow2<false>, lds);
      if (rc) return rc;
      dispatch_note("fft.rows_generic.pow2");
      hipLaunchKernelGGL(k_fft_rows_pow2<false>, grid, dim3(kThreads), lds, c->stream, a);
    }
  } else if (use_bluestein(K)) {
    FftBlueArgs b;
    b.f = a; b.f.logK = 0; b.f.F = 1;
    rc = blue_tables(c, K, &b.t);
    if (rc) return rc;
    const size_t lds = (size_t)2 * b.t.P * sizeof(float2);
    dim3 grid((unsigned)rows);
    if (inverse) {
      rc = ensure_lds(k_fft_rows_blue<true>, lds);
      if (rc) return rc;
      dispatch_note("fft.rows_generic.blue");
      hipLaunchKernelGGL(k_fft_rows_blue<true>, grid, dim3(kThreads), lds, c->stream, b);
    } else {
      rc = ensure_lds(k_fft_rows_blue<false>, lds);
      if (rc) return rc;
      dispatch_note("fft.rows_generic.blue");
      hipLaunchKernelGGL(k_fft_rows_blue<false>, grid, dim3(kThreads), lds, c->stream, b);
    }
  } else {
    a.logK = 0; a.F = 1;
    const int nuse = n_in < K ? n_in : K;
    const size_t lds = (size_t)nuse * sizeof(float2);
    dim3 grid((unsigned)rows);
    dispatch_note("fft.rows_generic.dft");
    if (inverse) hipLaunchKernelGGL(k_fft_rows_dft<true>, grid, dim3(kThreads), lds, c->stream, a);
    else hipLaunchKernelGGL(k_fft_rows_dft<false>, grid, dim3(kThreads), lds, c->stream, a);
  }
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

int launch_fft(Ctx* c, const void* in, bool in_is_real, int64_t rows, int32_t n_in, int32_t K, bool inverse, float2* out, bool clean) {
  return launch_fft_rows(c, in, in_is_real, rows, n_in, K, inverse, nullptr, 1.0f, false, out, clean);
}

// ---- NxSignal.stft of COMPLEX samples (c64 IQ data; the reference frames, multiplies and transforms whatever tensor it is given,
// lib/nx_signal.ex:94-102).  fft_length 1024 / 2048 / 4096: one launch (launch_stft_c64_wave, kernels_wave_rows.hip: frame slice x
// window fused into the row kernels' loads).  Every other length: the windowed frames (truncated to fft_length, :102) go to a scratch
// tensor and the row transforms of Nx.fft take it from there — the same kernels, Bluestein and four-step paths as nxsig_fft —
// followed by the :spectrum / :psd division.  c64 x f32 is componentwise (SURVEY App. A rule 9): two exact f32 products per sample.
__global__ __launch_bounds__(kThreads) void k_frames_c64(const float2* __restrict__ x, int64_t batch_stride, FrameGeom g, int32_t n_use,
                                                        const float* __restrict__ window, float2* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t total = g.M * n_use;
  if (idx >= total) return;
  const int64_t m = idx / n_use;
  const int32_t n = (int32_t)(idx - m * n_use);
  const float2* xr = x + (size_t)blockIdx.y * batch_stride;
  int64_t pos = m * g.hop + n - g.lo;
  float2 v = make_float2(0.f, 0.f);
  if (g.reflect) {   // lib/nx_signal.ex:349 (Nx.reflect)
    if (g.L == 1) pos = 0;
    else {
      const int64_t period = 2 * (g.L - 1);
      pos %= period;
      if (pos < 0) pos += period;
      if (pos >= g.L) pos = period - pos;
    }
    v = xr[pos];
  } else if (pos >= 0 && pos < g.L) v = xr[pos];   // :338 (Nx.pad, zeros)
  const float w = window[n];
  out[(size_t)blockIdx.y * total + idx] = make_float2(v.x * w, v.y * w);
}

int launch_stft_c64_wave(Ctx* c, const StftLaunch& s, bool* handled);   // kernels_wave_rows.hip
int launch_stft_rab_c64(Ctx* c, const StftLaunch& s, bool* handled);    // kernels_wave_rab.hip: composite and power-of-two lengths to 1600

int launch_stft_c64(Ctx* c, const StftLaunch& s) {
  if (s.fr.M == 0 || s.batch == 0) return NXSIG_OK;
  if (s.batch > 65504) {   // rows are independent: slabs of 65 504 rows like launch_stft (the frame-gather kernel puts the row on gridDim.y)
    for (int32_t r0 = 0; r0 < s.batch; r0 += 65504) {
      StftLaunch b = s;
      b.batch = s.batch - r0 < 65504 ? s.batch - r0 : 65504;
      b.x = reinterpret_cast<const float*>(reinterpret_cast<const float2*>(s.x) + (size_t)r0 * s.batch_stride);
      b.z = s.z + (size_t)r0 * s.fr.M * s.K;
      int rcs = launch_stft_c64(c, b);
      if (rcs) return rcs;
    }
    return NXSIG_OK;
  }
  bool handled = false;
  int rc = launch_stft_rab_c64(c, s, &handled);   // 100 ... 1600 (1024 as 32 x 32: 0.42 against 0.39 on the framed row kernel)
  if (rc || handled) return rc;
  rc = launch_stft_c64_wave(c, s, &handled);       // 2048, 4096 (and 1024 when the kernels above decline)
  if (rc || handled) return rc;
  const int n_use = s.fr.N < s.K ? s.fr.N : s.K;
  const int64_t rows = (int64_t)s.batch * s.fr.M;
  void* frames = nullptr;
  if ((rc = ctx_scratch(c, 26, (size_t)rows * n_use * sizeof(float2), &frames))) return rc;
  const int64_t blocks = (s.fr.M * n_use + kThreads - 1) / kThreads;
  if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "stft: too many frames for one launch");
  dispatch_note("stft_c64.frames");
  hipLaunchKernelGGL(k_frames_c64, dim3((unsigned)blocks, (unsigned)s.batch), dim3(kThreads), 0, c->stream,
                     reinterpret_cast<const float2*>(s.x), s.batch_stride, to_geom(s.fr), n_use, s.window, reinterpret_cast<float2*>(frames));
  NXSIG_HIP_TRY(hipGetLastError());
  if ((rc = launch_fft_rows(c, frames, false, rows, n_use, s.K, false, nullptr, 1.0f, false, s.z))) return rc;
  if (s.has_scale) return launch_rows_post(c, s.z, rows, s.K, nullptr, 1.0f, false, s.inv_scale_div, true);
  return NXSIG_OK;
}

// generic istft: rows IFFT (x scale x window) into a scratch frames tensor, then the deterministic OLA + normaliser
int launch_istft_generic(Ctx* c, const IstftLaunch& s) {
  if (s.M == 0 || s.batch == 0) return NXSIG_OK;
  void* frames = nullptr;
  const size_t fbytes = (size_t)s.batch * s.M * s.N * sizeof(float2);
  int rc = ctx_scratch(c, 0, fbytes, &frames);
  if (rc) return rc;
  rc = launch_fft_rows(c, s.z, false, (int64_t)s.batch * s.M, s.K, s.K, true, s.window, s.scale_mul, s.has_scale != 0,
                       reinterpret_cast<float2*>(frames));
  if (rc) return rc;
  const int64_t out_len = s.M * s.hop + (s.N - s.hop);
  dim3 grid((unsigned)((out_len + kThreads - 1) / kThreads), (unsigned)s.batch);
  dispatch_note("istft.generic");
  hipLaunchKernelGGL((k_ola<2, true>), grid, dim3(kThreads), 0, c->stream, reinterpret_cast<const float*>(frames), s.M,
                     s.N, s.hop, s.window, reinterpret_cast<float*>(s.y), out_len);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// window lengths that are multiples of 4: a wave writes one frame at a time with 16-byte streaming stores (no index division:
// lanes walk the frame in steps of 64 quads); frames whose samples all lie inside the signal read them directly
__global__ __launch_bounds__(kThreads) void k_as_windowed_v4(const float* __restrict__ x, int64_t batch_stride, FrameGeom g, float* __restrict__ out) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) f4 gf4;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (kThreads / 64);
  const float* xr = x + (size_t)blockIdx.y * batch_stride;
  float* orow = out + (size_t)blockIdx.y * g.M * g.N;
  const int n4 = g.N / 4;
  for (int64_t m = wave; m < g.M; m += nwaves) {
    const int64_t q0 = m * g.hop - g.lo;                       // signal index of the frame's first sample
    const bool inside = q0 >= 0 && q0 + g.N <= g.L;            // uniform per wave
    f4* dst = reinterpret_cast<f4*>(orow + (size_t)m * g.N);
    for (int j = lane; j < n4; j += 64) {
      f4 v;
      if (inside) { const float* p = xr + q0 + 4 * j; v = f4{p[0], p[1], p[2], p[3]}; }
      else {
        const int64_t q = m * g.hop + 4 * j;
        v = f4{fetch_padded(xr, g, q), fetch_padded(xr, g, q + 1), fetch_padded(xr, g, q + 2), fetch_padded(xr, g, q + 3)};
      }
      __builtin_nontemporal_store(v, (gf4*)(dst + j));
    }
  }
}

int launch_as_windowed(Ctx* c, const float* x, int64_t batch_stride, int32_t batch, const Framing& fr, float* out) {
  if (fr.M == 0 || batch == 0) return NXSIG_OK;
  if (fr.N % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    int64_t blocks = (fr.M + kThreads / 64 - 1) / (kThreads / 64);
    const int64_t cap = (int64_t)c->num_cus * 16;
    if (blocks > cap) blocks = cap;
    dispatch_note("as_windowed.v4");
    hipLaunchKernelGGL(k_as_windowed_v4, dim3((unsigned)blocks, (unsigned)batch), dim3(kThreads), 0, c->stream, x, batch_stride, to_geom(fr), out);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  }
  const int64_t total = fr.M * fr.N;
  dim3 grid((unsigned)((total + kThreads - 1) / kThreads), (unsigned)batch);
  dispatch_note("as_windowed");
  hipLaunchKernelGGL(k_as_windowed, grid, dim3(kThreads), 0, c->stream, x, batch_stride, to_geom(fr), out);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// real frames, hop and N multiples of 4: four consecutive outputs per thread (they share their covering frames), 16-byte loads
// and stores, the same double accumulation in ascending frame order
__global__ __launch_bounds__(kThreads) void k_ola_v4(const float* __restrict__ frames, int64_t M, int32_t N, int32_t hop,
                                                    float* __restrict__ out, int64_t out_len) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) f4 gf4;
  const int64_t n = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4;
  if (n >= out_len) return;   // out_len = M hop + N - hop is a multiple of 4
  const float* fr = frames + (size_t)blockIdx.y * M * N;
  int64_t m_hi = n / hop;
  if (m_hi > M - 1) m_hi = M - 1;
  int64_t m_lo = (n - N + hop) / hop;
  if (n - N + 1 <= 0) m_lo = 0;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (int64_t m = m_lo; m <= m_hi; ++m) {
    const f4 v = *reinterpret_cast<const f4*>(fr + (size_t)m * N + (n - m * hop));
    a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
  }
  __builtin_nontemporal_store(f4{(float)a0, (float)a1, (float)a2, (float)a3}, (gf4*)(out + (size_t)blockIdx.y * out_len + n));
}

int launch_overlap_and_add(Ctx* c, const float* frames, int64_t M, int32_t batch, int32_t N, int32_t hop, int32_t comps,
                           float* out) {
  const int64_t out_len = M * hop + (N - hop);
  if (out_len == 0 || batch == 0) return NXSIG_OK;
  if (comps == 1 && N % 4 == 0 && hop % 4 == 0 && (reinterpret_cast<uintptr_t>(frames) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    // (a thread's 4 outputs n .. n+3 lie in the same hop segment and below the same frame ends, so they share m_lo .. m_hi)
    dim3 grid((unsigned)((out_len / 4 + kThreads - 1) / kThreads), (unsigned)batch);
    dispatch_note("ola.v4");
    hipLaunchKernelGGL(k_ola_v4, grid, dim3(kThreads), 0, c->stream, frames, M, N, hop, out, out_len);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  }
  dim3 grid((unsigned)((out_len + kThreads - 1) / kThreads), (unsigned)batch);
  dispatch_note("ola");
  if (comps == 1)
    hipLaunchKernelGGL((k_ola<1, false>), grid, dim3(kThreads), 0, c->stream, frames, M, N, hop, nullptr, out, out_len);
  else
    hipLaunchKernelGGL((k_ola<2, false>), grid, dim3(kThreads), 0, c->stream, frames, M, N, hop, nullptr, out, out_len);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// host FFT in double for the filter spectrum H (tiny: B <= 8192 points, once per distinct filter)
static void host_fft_f64(std::vector<double>& re, std::vector<double>& im) {
  const size_t n = re.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = -2.0 * 3.14159265358979323846 / (double)len;
    for (size_t i = 0; i < n; i += len) {
      for (size_t k = 0; k < len / 2; ++k) {
        const double wr = cos(ang * (double)k), wi = sin(ang * (double)k);
        const size_t u = i + k, v = i + k + len / 2;
        const double tr = re[v] * wr - im[v] * wi, ti = re[v] * wi + im[v] * wr;
        re[v] = re[u] - tr; im[v] = im[u] - ti;
        re[u] += tr; im[u] += ti;
      }
    }
  }
}

int launch_fir_generic(Ctx* c, const FirLaunch& s) {
  if (s.out_len <= 0 || s.batch == 0) return NXSIG_OK;
  if (s.taps > 4096)
    return set_error(NXSIG_ERR_UNSUPPORTED, "fir: more than 4096 taps is not supported yet (overlap-save block <= 8192)");
  int B = 1024;
  while (B < 8 * s.taps && B < 8192) B <<= 1;
  while (B < 2 * s.taps) B <<= 1;
  // filter spectrum, computed in double on the host, cached in HBM by content
  std::vector<double> re(B, 0.0), im(B, 0.0);
  for (int i = 0; i < s.taps; ++i) re[i] = (double)s.h_host[i];
  host_fft_f64(re, im);
  std::vector<float2> H(B);
  for (int i = 0; i < B; ++i) H[i] = make_float2((float)re[i], (float)im[i]);
  FirArgs a;
  const void* Hd = nullptr;
  int rc = ctx_table(c, 0xF1A0000000000000ull ^ (uint64_t)B, H.data(), H.size() * sizeof(float2), &Hd);
  if (rc) return rc;
  a.H = reinterpret_cast<const float2*>(Hd);
  rc = ctx_twiddles(c, B, &a.tw);
  if (rc) return rc;
  a.x = s.x; a.L = s.L; a.batch_stride = s.batch_stride; a.B = B; a.logB = ilog2(B); a.taps = s.taps;
  const int64_t V = B - (s.taps - 1);
  a.first_block = s.out_start / V;
  const int64_t last_block = (s.out_start + s.out_len - 1) / V;
  a.nblocks = last_block - a.first_block + 1;
  a.out_start = s.out_start; a.out_len = s.out_len; a.y = s.y; a.row_flags = s.row_flags;
  const size_t lds = (size_t)2 * B * sizeof(float2);
  rc = ensure_lds(k_fir_os, lds);
  if (rc) return rc;
  dim3 grid((unsigned)((a.nblocks + 1) / 2), (unsigned)s.batch);
  dispatch_note("fir.generic");
  hipLaunchKernelGGL(k_fir_os, grid, dim3(kThreads), lds, c->stream, a);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// the list a frame-packing istft kernel reports its non-finite units to (see k_istft_edge_fix): {ticket, pad | count, capacity, int64
// entries}; the pointer handed out is the count's.  Count and ticket are zero between calls: the fix-up pass that consumes a
// non-empty list puts them back (no per-call memset).
int istft_nf_list(Ctx* c, int64_t capacity, int** list) {
  if (capacity > 0x7fffffffLL) capacity = 0x7fffffffLL;
  if (c->scratch_bytes[23] < (size_t)(capacity + 2) * 8) {
    // (re)allocation: the header is written once, synchronously (it describes the buffer, not the call)
    const int64_t cap = capacity < 4096 ? 4096 : capacity * 2;
    void* p = nullptr;
    int rc = ctx_scratch(c, 23, (size_t)(cap + 2) * 8, &p);
    if (rc) return rc;
    const int hdr[4] = {0, 0, 0, (int)(cap > 0x7fffffffLL ? 0x7fffffffLL : cap)};
    NXSIG_HIP_TRY(hipMemcpy(p, hdr, sizeof(hdr), hipMemcpyHostToDevice));
  }
  *list = reinterpret_cast<int*>(c->scratch[23]) + 2;
  return NXSIG_OK;
}

// w_N^j = exp(+2 pi i j / N) in double (host libm): built once per context and N (the content-addressed table cache would
// otherwise make every call recompute N sine / cosine pairs just to look the table up)
static int edge_fix_twiddles(Ctx* c, int N, const double2** out) {
  const uint64_t key = 0xED6E00000000ull ^ (uint64_t)(uint32_t)N;
  auto hit = c->memo.find(key);
  if (hit != c->memo.end()) { *out = reinterpret_cast<const double2*>(hit->second[0]); return NXSIG_OK; }
  std::vector<double2> tw((size_t)N);
  for (int j = 0; j < N; ++j) {
    const double ang = 6.283185307179586476925286766559 * (double)j / (double)N;
    tw[j] = make_double2(std::cos(ang), std::sin(ang));
  }
  const void* d = nullptr;
  int rc = ctx_table(c, 0xED6Eull, tw.data(), tw.size() * sizeof(double2), &d);
  if (rc) return rc;
  c->memo[key] = {reinterpret_cast<uint64_t>(d)};
  *out = reinterpret_cast<const double2*>(d);
  return NXSIG_OK;
}

// The pass after ANY istft main kernel (generic or tuned): k_istft_edge_fix, both roles in one launch — the ill-conditioned edge
// samples recomputed in double and, for the kernels that invert several frames per transform (s.nf_list), the units they reported.
int launch_istft_fix(Ctx* c, const IstftLaunch& s, const float* window_host) {
  if (s.M == 0 || s.batch == 0) return NXSIG_OK;
  const int N = s.N, hop = s.hop;
  const int64_t out_len = s.M * hop + (N - hop);
  EdgeFixArgs a;
  a.z = s.z; a.filt = s.filt; a.M = s.M; a.N = N; a.hop = hop; a.window = s.window; a.scale = s.scale_mul; a.has_scale = s.has_scale;
  a.y = s.y; a.out_len = out_len; a.onesided = s.onesided ? 1 : 0; a.batch = s.batch;
  a.idx = nullptr; a.n_idx = 0; a.tau = 0.0f; a.tw = nullptr;
  // which samples are candidates is a pure function of (window, hop, M): memoised per context as {mode, tau bits, idx, n_idx}
  // key = hash of (hop, M, N) AS DATA followed by the window's content (shifting them into the seed aliased for large M / hop)
  uint64_t mode = 0;
  const int64_t ekey_geom[3] = {(int64_t)hop, (int64_t)s.M, (int64_t)N};
  const uint64_t ekey = fnv1a(fnv1a(0xED70ull, ekey_geom, sizeof(ekey_geom)), window_host, (size_t)N * sizeof(float));
  auto hit = c->memo.find(ekey);
  if (hit != c->memo.end()) {
    const std::vector<uint64_t>& v = hit->second;
    mode = v[0];
    uint32_t tb = (uint32_t)v[1];
    std::memcpy(&a.tau, &tb, 4);
    a.idx = reinterpret_cast<const int64_t*>(v[2]);
    a.n_idx = (int64_t)v[3];
  } else {
    // interior normaliser is periodic in n with period hop: den_mid[r] = sum_{j = r (mod hop)} |w[j]|^2
    const int period = hop < N ? hop : N;
    double dmin = 1e300, dmax = 0.0;
    for (int r = 0; r < period; ++r) {
      double d = 0.0;
      for (int j = r; j < N; j += hop) { const float w = std::fabs(window_host[j]); d += (double)(w * w); }
      dmin = d < dmin ? d : dmin;
      dmax = d > dmax ? d : dmax;
    }
    if (dmax > 0.0) {
      a.tau = (float)(0.02 * dmax);
      if (hop < N && dmin >= (double)a.tau) {
        // well-conditioned interior: only samples of the partial-overlap head / tail can be flagged, and which ones is a
        // pure function of the window -> list them on the host (a few hundred) with their frame range and normaliser
        const int64_t head_cnt = (N - hop) < out_len ? (N - hop) : out_len;
        const int64_t tail_start = s.M * hop > head_cnt ? s.M * hop : head_cnt;
        std::vector<int64_t> idx;
        auto consider = [&](int64_t n) {
          int64_t m_hi = n / hop;
          if (m_hi > s.M - 1) m_hi = s.M - 1;
          const int64_t m_lo = (n - N + 1 <= 0) ? 0 : (n - N + hop) / hop;
          double den = 0.0;
          for (int64_t m = m_lo; m <= m_hi; ++m) { const float w = std::fabs(window_host[n - m * hop]); den += (double)(w * w); }
          const float d = (float)den;
          if (d > 1.0e-10f && d < a.tau) {
            uint32_t db;
            std::memcpy(&db, &d, 4);
            idx.push_back(n); idx.push_back(m_lo); idx.push_back(m_hi); idx.push_back((int64_t)db);
          }
        };
        for (int64_t n = 0; n < head_cnt; ++n) consider(n);
        for (int64_t n = tail_start; n < out_len; ++n) consider(n);
        int SP = 0;   // smallest power of two <= 32 that divides N with N / SP <= 64 lanes
        for (int q = 2; q <= 32 && !SP; q *= 2)
          if (N % q == 0 && N / q <= 64) SP = q;
        const bool chunked = SP != 0 && N < (1 << 20) && !tune(c, kT_DISABLE_WAVE, 0);
        if (!idx.empty() && chunked) {
          // k_istft_edge_chunks: candidates grouped by (n / SP, frame range) -> {n0, m_lo, m_hi, mask, SP normalisers (f32 bits, two per word)}
          std::vector<int64_t> ch;
          for (size_t i = 0; i < idx.size(); i += 4) {
            const int64_t n = idx[i], n0 = n - n % SP;   // (n >= 0)
            const size_t last = ch.size() >= 20 ? ch.size() - 20 : 0;
            if (ch.empty() || ch[last] != n0 || ch[last + 1] != idx[i + 1] || ch[last + 2] != idx[i + 2]) {
              ch.resize(ch.size() + 20, 0);
              const size_t q = ch.size() - 20;
              ch[q] = n0; ch[q + 1] = idx[i + 1]; ch[q + 2] = idx[i + 2];
            }
            const size_t q = ch.size() - 20;
            const int g = (int)(n - n0);
            ch[q + 3] |= (int64_t)1 << g;
            ch[q + 4 + (g >> 1)] |= (int64_t)((uint64_t)(uint32_t)idx[i + 3] << (32 * (g & 1)));
          }
          const void* d = nullptr;
          int rc = ctx_table(c, 0x1DAull, ch.data(), ch.size() * sizeof(int64_t), &d);
          if (rc) return rc;
          a.idx = reinterpret_cast<const int64_t*>(d);
          a.n_idx = (int64_t)ch.size() / 20;
          mode = 3 + (uint64_t)SP * 16;   // 3 | SP << 4
        } else if (!idx.empty()) {
          const void* d = nullptr;
          int rc = ctx_table(c, 0x1D9ull, idx.data(), idx.size() * sizeof(int64_t), &d);
          if (rc) return rc;
          a.idx = reinterpret_cast<const int64_t*>(d);
          a.n_idx = (int64_t)idx.size() / 4;
          mode = 1;
        }
      } else {
        mode = 2;  // ill-conditioned interior (e.g. hop == N under a tapered window): every sample is a candidate
      }
    }
    uint32_t tb;
    std::memcpy(&tb, &a.tau, 4);
    c->memo[ekey] = {mode, (uint64_t)tb, reinterpret_cast<uint64_t>(a.idx), (uint64_t)a.n_idx};
  }
  if (mode == 0 && !s.nf_list) return NXSIG_OK;
  { int rc = edge_fix_twiddles(c, N, &a.tw); if (rc) return rc; }
  if ((mode & 15) == 3) {   // chunks of SP positions per wave (k_istft_edge_chunks); the non-finite role, if any, follows on its own
    const int64_t chunk_blocks = (a.n_idx + kFixWaves - 1) / kFixWaves;
    const int64_t blocks = chunk_blocks * s.batch;
    if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "istft: signal too long for the edge fix-up grid");
    switch ((int)(mode >> 4)) {
      case 2: dispatch_note("istft.edge_chunks"); hipLaunchKernelGGL(k_istft_edge_chunks<2>, dim3((unsigned)blocks), dim3(kThreads), 0, c->stream, a, chunk_blocks); break;
      case 4: dispatch_note("istft.edge_chunks"); hipLaunchKernelGGL(k_istft_edge_chunks<4>, dim3((unsigned)blocks), dim3(kThreads), 0, c->stream, a, chunk_blocks); break;
      case 8: dispatch_note("istft.edge_chunks"); hipLaunchKernelGGL(k_istft_edge_chunks<8>, dim3((unsigned)blocks), dim3(kThreads), 0, c->stream, a, chunk_blocks); break;
      case 16: dispatch_note("istft.edge_chunks"); hipLaunchKernelGGL(k_istft_edge_chunks<16>, dim3((unsigned)blocks), dim3(kThreads), 0, c->stream, a, chunk_blocks); break;
      default: dispatch_note("istft.edge_chunks"); hipLaunchKernelGGL(k_istft_edge_chunks<32>, dim3((unsigned)blocks), dim3(kThreads), 0, c->stream, a, chunk_blocks); break;
    }
    NXSIG_HIP_TRY(hipGetLastError());
    if (!s.nf_list) return NXSIG_OK;
    mode = 0;   // what is left for k_istft_edge_fix: the non-finite units
    a.idx = nullptr; a.n_idx = 0;
  }
  const int64_t edge_blocks = mode == 0 ? 0 : ((mode == 1 ? a.n_idx : out_len) + kFixWaves - 1) / kFixWaves;
  const int32_t nf_blocks = s.nf_list ? c->num_cus * 2 : 0;
  const int64_t blocks = edge_blocks * s.batch + nf_blocks;
  if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "istft: signal too long for the edge fix-up grid");
  dispatch_note("istft.edge_fix");
  hipLaunchKernelGGL(k_istft_edge_fix, dim3((unsigned)blocks), dim3(kThreads), 0, c->stream, a, edge_blocks, s.nf_list,
                     (int32_t)s.nf_frames_per_unit, nf_blocks);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// complex 1-D fftconvolve through the row-FFT kernels; a, b, out are device pointers
int launch_fftconvolve_c64(Ctx* c, const float2* a, int64_t n1, const float2* b, int64_t n2, int64_t start, int64_t len,
                           float2* out) {
  const int64_t full = n1 + n2 - 1;
  int P = 1;
  while (P < full) P <<= 1;
  if (full > ((int64_t)1 << 26)) return set_error(NXSIG_ERR_UNSUPPORTED, "fftconvolve (complex): n1 + n2 - 1 > 2^26 is not supported");
  void* sc = nullptr;
  int rc = ctx_scratch(c, 0, (size_t)3 * P * sizeof(float2), &sc);
  if (rc) return rc;
  float2* A = reinterpret_cast<float2*>(sc);
  float2* B = A + P;
  float2* C = B + P;
  if ((rc = launch_fft_big(c, a, false, 1, n1, P, false, A))) return rc;  // four-step rows beyond 8192 points
  if ((rc = launch_fft_big(c, b, false, 1, n2, P, false, B))) return rc;
  hipLaunchKernelGGL(k_cmul_inplace, dim3((P + kThreads - 1) / kThreads), dim3(kThreads), 0, c->stream, A, B, P);
  NXSIG_HIP_TRY(hipGetLastError());
  if ((rc = launch_fft_big(c, A, false, 1, P, P, true, C))) return rc;
  NXSIG_HIP_TRY(hipMemcpyAsync(out, C + start, (size_t)len * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
  return NXSIG_OK;
}

int launch_spectrum_mul(Ctx* c, const float2* z, int64_t rows, int32_t K, const float2* h_dev, float2* out) {
  const int64_t total = rows * K;
  if (total == 0) return NXSIG_OK;
  int64_t blocks = (total + kThreads - 1) / kThreads;
  const int64_t cap = (int64_t)c->num_cus * 32;
  if (blocks > cap) blocks = cap;
  dispatch_note("spectrum_mul");
  hipLaunchKernelGGL(k_spectrum_mul, dim3((unsigned)blocks), dim3(kThreads), 0, c->stream, z, h_dev, out, total, K);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

int launch_stft_to_mel(Ctx* c, const float2* z, int64_t rows, int32_t K, int32_t mel_bins, const float* filters_host,
                       float* out) {
  if (rows == 0 || mel_bins == 0) return NXSIG_OK;
  const int half = K / 2;
  std::vector<int2> band(mel_bins);
  for (int b = 0; b < mel_bins; ++b) {
    int lo = half, hi = 0;
    for (int k = 0; k < half; ++k)
      if (filters_host[(size_t)b * K + k] != 0.0f) { if (k < lo) lo = k; hi = k + 1; }
    if (hi <= lo) { lo = 0; hi = 0; }
    band[b] = make_int2(lo, hi);
  }
  MelArgs a;
  const void *fd = nullptr, *bd = nullptr;
  int rc = ctx_table(c, 0x3E1F11ull, filters_host, (size_t)mel_bins * K * sizeof(float), &fd);
  if (rc) return rc;
  rc = ctx_table(c, 0x3E1BA2Dull, band.data(), band.size() * sizeof(int2), &bd);
  if (rc) return rc;
  void* gm = nullptr;
  rc = ctx_scratch(c, 5, 256, &gm);
  if (rc) return rc;
  static const int init[2] = {(int)0x80000000, 0};  // {below every ordered-int value, no non-finite |z|^2 seen}
  NXSIG_HIP_TRY(hipMemcpyAsync(gm, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  a.z = z; a.rows = rows; a.K = K; a.half = half; a.mel_bins = mel_bins;
  a.filt = reinterpret_cast<const float*>(fd); a.band = reinterpret_cast<const int2*>(bd);
  a.out = out; a.gmax = reinterpret_cast<int*>(gm);
  a.ln10 = (float)std::log(10.0);
  // tiled kernel whenever one frame's bins fit the LDS budget (64 frames per tile for fft_length <= 512, fewer above), else
  // the first form
  std::vector<float> cw;
  std::vector<int4> bw(mel_bins);
  for (int b = 0; b < mel_bins; ++b) {
    bw[b] = make_int4(band[b].x, band[b].y, (int)cw.size(), 0);
    for (int k = band[b].x; k < band[b].y; ++k) cw.push_back(filters_host[(size_t)b * K + k]);
  }
  if (cw.empty()) cw.push_back(0.0f);
  const int hs = half | 1;
  int maxw = 1;
  for (int b = 0; b < mel_bins; ++b) maxw = std::max(maxw, band[b].y - band[b].x);
  const bool wlds = (size_t)mel_bins * maxw <= 4096;
  const bool tile_off = tune(c, kT_MEL_TILE, 1) == 0;
  int fb_log2 = -1;
  size_t lds = 0;
  const size_t budget = (size_t)tune(c, kT_MEL_LDS_KB, 56) * 1024;
  const int nt_knob = 0;
  for (int l = 6; l >= 0 && !tile_off && mel_bins <= 256; --l) {
    const size_t need = ((size_t)(1 << l) * (hs + mel_bins) + maxw + (wlds ? (size_t)mel_bins * maxw : 0)) * sizeof(float);
    if (need + 6200 <= budget) { fb_log2 = l; lds = need; break; }   // + the kernel's static tables
  }
  if (fb_log2 >= 0) {
    MelTileArgs t;
    const void *cd = nullptr, *od = nullptr;
    rc = ctx_table(c, 0x3E1C0Aull, cw.data(), cw.size() * sizeof(float), &cd);
    if (rc) return rc;
    const int bpw = 64 >> fb_log2;
    for (int b = 0; b < mel_bins; b += bpw) {
      int gw = 0;
      for (int j = b; j < std::min(b + bpw, mel_bins); ++j) gw = std::max(gw, bw[j].y - bw[j].x);
      bw[b].w = gw;
    }
    std::vector<float> wpad(wlds ? (size_t)mel_bins * maxw : 1, 0.0f);
    if (wlds)
      for (int b = 0; b < mel_bins; ++b)
        for (int k = band[b].x; k < band[b].y; ++k) wpad[(size_t)b * maxw + (k - band[b].x)] = filters_host[(size_t)b * K + k];
    const void* pd = nullptr;
    rc = ctx_table(c, 0x3E1C0Cull, wpad.data(), wpad.size() * sizeof(float), &pd);
    if (rc) return rc;
    rc = ctx_table(c, 0x3E1C0Bull ^ ((uint64_t)fb_log2 << 32), bw.data(), bw.size() * sizeof(int4), &od);
    if (rc) return rc;
    static const std::vector<double> logtab = [] {
      std::vector<double> v(256);
      for (int i = 0; i < 128; ++i) {
        const double ci = 1.0 + (i + 0.5) / 128.0;
        v[2 * i] = 1.0 / ci;
        v[2 * i + 1] = std::log(ci);
      }
      return v;
    }();
    const void* ld = nullptr;
    rc = ctx_table(c, 0x3E1106ull, logtab.data(), logtab.size() * sizeof(double), &ld);
    if (rc) return rc;
    t.m = a; t.cw = reinterpret_cast<const float*>(cd); t.bw = reinterpret_cast<const int4*>(od);
    t.wpad = reinterpret_cast<const float*>(pd); t.maxw = maxw;
    t.logtab = reinterpret_cast<const double2*>(ld);
    t.nnz = (int)cw.size(); t.fb_log2 = fb_log2; t.hs = hs;
    const int FB = 1 << fb_log2;
    const dim3 grid((unsigned)((rows + FB - 1) / FB));
    // 16 waves per tile when it has 32 or 64 frames: phase 2 is a chain of dependent LDS reads and double-precision
    // operations per band, and it is the number of resident waves that hides their latency
    // one wave per four frames of the tile (phase 1 streams four rows per wave), at least four waves
    int nt = fb_log2 >= 6 ? 1024 : fb_log2 == 5 ? 512 : 256;
    if (nt_knob) nt = nt_knob;
#define NXSIG_MEL_TILE_LAUNCH(W, NT)                                                \
    do {                                                                            \
      rc = ensure_lds(k_mel_tile<W, NT>, lds);                                      \
      if (rc) return rc;                                                            \
      dispatch_note("mel.tile");                                                   \
      hipLaunchKernelGGL((k_mel_tile<W, NT>), grid, dim3(NT), lds, c->stream, t);   \
    } while (0)
    if (wlds && nt == 1024) NXSIG_MEL_TILE_LAUNCH(true, 1024);
    else if (wlds && nt == 512) NXSIG_MEL_TILE_LAUNCH(true, 512);
    else if (wlds) NXSIG_MEL_TILE_LAUNCH(true, 256);
    else if (nt == 1024) NXSIG_MEL_TILE_LAUNCH(false, 1024);
    else if (nt == 512) NXSIG_MEL_TILE_LAUNCH(false, 512);
    else NXSIG_MEL_TILE_LAUNCH(false, 256);
#undef NXSIG_MEL_TILE_LAUNCH
  } else {
    const size_t lds1 = (size_t)kMelFramesPerBlock * half * sizeof(float);
    if (lds1 > 160 * 1024) return set_error(NXSIG_ERR_UNSUPPORTED, "stft_to_mel: fft_length beyond the LDS-resident kernels");
    int rc2 = ensure_lds(k_mel_pass1, lds1);
    if (rc2) return rc2;
    dispatch_note("mel.pass1");
    hipLaunchKernelGGL(k_mel_pass1, dim3((unsigned)((rows + kMelFramesPerBlock - 1) / kMelFramesPerBlock)), dim3(kThreads), lds1,
                       c->stream, a);
  }
  NXSIG_HIP_TRY(hipGetLastError());
  return launch_mel_finish(c, out, rows * mel_bins, reinterpret_cast<int*>(gm));
}

// bins below K / 2 of every row (two-step form of the one-sided spectrum sink)
// packed: the imaginary part of bin 0 carries Re X[K/2] (the layout nxsig_istft_packed_f32 inverts)
__global__ __launch_bounds__(kThreads) void k_half_from_spectrum(const float2* __restrict__ z, int64_t rows, int32_t K, float2* __restrict__ out,
                                                                 int packed) {
  const int half = K / 2;
  const int64_t total = rows * half;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int64_t r = i / half;
    const int k = (int)(i - r * half);
    float2 v = z[(size_t)r * K + k];
    if (packed && k == 0) v.y = z[(size_t)r * K + half].x;
    out[i] = v;
  }
}
// the inverse re-layout: full Hermitian rows c64[rows][K] from packed rows c64[rows][K / 2] (shapes without a fused inverse)
__global__ __launch_bounds__(kThreads) void k_full_from_packed(const float2* __restrict__ zp, int64_t rows, int32_t K, float2* __restrict__ out) {
  const int half = K / 2;
  const int64_t total = rows * K;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int64_t r = i / K;
    const int k = (int)(i - r * K);
    const float2* row = zp + (size_t)r * half;
    float2 v;
    if (k == 0) v = make_float2(row[0].x, 0.0f);
    else if (k < half) v = row[k];
    else if (k == half) v = make_float2(row[0].y, 0.0f);
    else { const float2 c = row[K - k]; v = make_float2(c.x, -c.y); }
    out[i] = v;
  }
}
__global__ __launch_bounds__(kThreads) void k_real_from_c64(const float2* __restrict__ in, int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) out[i] = in[i].x;
}
static unsigned capped_blocks(const Ctx* c, int64_t total) {
  int64_t blocks = (total + kThreads - 1) / kThreads;
  const int64_t cap = (int64_t)c->num_cus * 32;
  return (unsigned)(blocks > cap ? cap : (blocks < 1 ? 1 : blocks));
}
int launch_full_from_packed(Ctx* c, const float2* zp, int64_t rows, int32_t K, float2* out) {
  if (rows * K == 0) return NXSIG_OK;
  hipLaunchKernelGGL(k_full_from_packed, dim3(capped_blocks(c, rows * K)), dim3(kThreads), 0, c->stream, zp, rows, K, out);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}
int launch_real_from_c64(Ctx* c, const float2* in, int64_t n, float* out) {
  if (n == 0) return NXSIG_OK;
  hipLaunchKernelGGL(k_real_from_c64, dim3(capped_blocks(c, n)), dim3(kThreads), 0, c->stream, in, n, out);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}
int launch_half_from_spectrum(Ctx* c, const float2* z, int64_t rows, int32_t K, float2* out, bool packed) {
  const int64_t total = rows * (K / 2);
  if (total == 0) return NXSIG_OK;
  int64_t blocks = (total + kThreads - 1) / kThreads;
  const int64_t cap = (int64_t)c->num_cus * 32;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_half_from_spectrum, dim3((unsigned)blocks), dim3(kThreads), 0, c->stream, z, rows, K, out, packed ? 1 : 0);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

int launch_mag_from_spectrum(Ctx* c, const float2* z, int64_t rows, int32_t K, int kind, float* out) {
  const int64_t total = rows * (K / 2);
  if (total == 0) return NXSIG_OK;
  int* gm = nullptr;
  int rc = launch_mel_init(c, &gm);
  if (rc) return rc;
  int64_t blocks = (total + kThreads - 1) / kThreads;
  const int64_t cap = (int64_t)c->num_cus * 32;
  if (blocks > cap) blocks = cap;
  dispatch_note("mag.from_spectrum");
  hipLaunchKernelGGL(k_mag_from_spectrum, dim3((unsigned)blocks), dim3(kThreads), 0, c->stream, z, rows, K, kind, out, gm);
  NXSIG_HIP_TRY(hipGetLastError());
  if (kind == 2) {
    hipLaunchKernelGGL(k_mag_db_pass2_g, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream, out, total, gm);
    NXSIG_HIP_TRY(hipGetLastError());
  }
  return NXSIG_OK;
}

// shared by the fused stft->mel kernel: running-maximum cell and the clamp pass
int launch_mel_init(Ctx* c, int** gmax) {
  void* gm = nullptr;
  int rc = ctx_scratch(c, 5, 256, &gm);
  if (rc) return rc;
  static const int init[2] = {(int)0x80000000, 0};
  NXSIG_HIP_TRY(hipMemcpyAsync(gm, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  *gmax = reinterpret_cast<int*>(gm);
  return NXSIG_OK;
}
int launch_mel_finish(Ctx* c, float* out, int64_t n, int* gmax) {
  if (n <= 0 || mel_deferred(c)) return NXSIG_OK;
  hipLaunchKernelGGL(k_mel_pass2, dim3((unsigned)((n + 4 * kThreads - 1) / (4 * kThreads))), dim3(kThreads), 0, c->stream, out, n, gmax);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

}  // namespace nxsig
