/*
 * CPU ORACLE (C leg) — TEST / BASELINE INFRASTRUCTURE ONLY, never linked into the product.
 *
 * Plain-C restatement of how the reference's NxSignal.stft executes on Nx.BinaryBackend
 * (lib/nx_signal.ex:94-102): materialised framing (as_windowed :354-364) -> f32 window product
 * (Nx.multiply :101) -> per-row Nx.fft in double (:102) -> round to c64.  The per-row transform follows
 * the BinaryBackend algorithm as recalled in SURVEY.md §3.1/App. A rule 7: recursive even/odd radix-2
 * Cooley-Tukey with the twiddle exp(-2 pi i k / n) evaluated per butterfly (libm cos/sin, as
 * Complex.exp does), a naive O(n^2) DFT for odd lengths, components with |x| <= 1e-10 zeroed, one
 * rounding to f32 at the end.  nx 0.11.0 is not vendored in /root/reference, so this is a restatement
 * ("port"), validated against oracle/nx_oracle.py (numpy, pinned on the reference's doctest vectors)
 * in tests/test_oracle_baseline.py.
 *
 * Used by bench.py's cpu_baseline leg (timed on the GPU box's host cores) and by tests.
 * Build: oracle/build_baseline.py  (gcc -O2 -fopenmp -ffp-contract=off -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double re, im; } cplx;

static const double BB_PI = 3.14159265358979323846;

/* out[0..n) = DFT(in[0..n) with stride `stride`); scratch has room for n elements per recursion level */
static void bb_fft_rec(const cplx* in, int stride, int n, cplx* out, cplx* scratch) {
  if (n == 1) { out[0] = in[0]; return; }
  if (n % 2 == 1) { /* odd length: naive DFT */
    for (int k = 0; k < n; ++k) {
      double sr = 0.0, si = 0.0;
      for (int j = 0; j < n; ++j) {
        const double ang = -2.0 * BB_PI * (double)j * (double)k / (double)n;
        const double c = cos(ang), s = sin(ang);
        const cplx v = in[(size_t)j * stride];
        sr += v.re * c - v.im * s;
        si += v.re * s + v.im * c;
      }
      out[k].re = sr; out[k].im = si;
    }
    return;
  }
  const int h = n / 2;
  cplx* E = scratch;
  cplx* Od = scratch + h;
  bb_fft_rec(in, stride * 2, h, E, scratch + n);
  bb_fft_rec(in + stride, stride * 2, h, Od, scratch + n);
  for (int k = 0; k < h; ++k) {
    const double ang = -2.0 * BB_PI * (double)k / (double)n;
    const double c = cos(ang), s = sin(ang);
    const double tr = Od[k].re * c - Od[k].im * s, ti = Od[k].re * s + Od[k].im * c;
    out[k].re = E[k].re + tr; out[k].im = E[k].im + ti;
    out[k + h].re = E[k].re - tr; out[k + h].im = E[k].im - ti;
  }
}

static inline float bb_clean(double v, double eps) { return (fabs(v) <= eps) ? 0.0f : (float)v; }

/* x f32[L] -> z c64[M][K] (interleaved re,im), :valid padding. Returns M, or -1 on bad arguments. */
int64_t bb_stft_f32(const float* x, int64_t L, const float* w, int32_t N, int32_t hop, int32_t K, double eps,
                    float* z, int32_t threads) {
  if (L < N || N < 1 || hop < 1 || K < 1) return -1;
  const int64_t M = (L - N) / hop + 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
  /* materialised framing + window product, as the reference does before the FFT */
  float* frames = (float*)malloc((size_t)M * N * sizeof(float));
  if (!frames) return -1;
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) frames[(size_t)m * N + n] = x[m * hop + n] * w[n];
  const int nuse = N < K ? N : K;
#pragma omp parallel
  {
    cplx* in = (cplx*)malloc((size_t)K * sizeof(cplx));
    cplx* out = (cplx*)malloc((size_t)K * sizeof(cplx));
    cplx* scratch = (cplx*)malloc((size_t)2 * K * sizeof(cplx) + 64);
#pragma omp for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
      for (int n = 0; n < nuse; ++n) { in[n].re = (double)frames[(size_t)m * N + n]; in[n].im = 0.0; }
      for (int n = nuse; n < K; ++n) { in[n].re = 0.0; in[n].im = 0.0; }
      bb_fft_rec(in, 1, K, out, scratch);
      float* zr = z + (size_t)m * K * 2;
      for (int k = 0; k < K; ++k) { zr[2 * k] = bb_clean(out[k].re, eps); zr[2 * k + 1] = bb_clean(out[k].im, eps); }
    }
    free(in); free(out); free(scratch);
  }
  free(frames);
  return M;
}

/* Throughput leg for the all-cores baseline: the same per-frame work as bb_stft_f32 over `reps` passes of the stream
 * (reps * M frames shared out statically over the threads), every thread with its own persistent scratch and, after the
 * first pass, its own output row, so that thread start-up, first-touch page faults of a fresh result buffer and the
 * serial framing pass do not dominate a short sample.  z c64[M][K] receives pass 0.  Returns reps * M. */
int64_t bb_stft_f32_repeat(const float* x, int64_t L, const float* w, int32_t N, int32_t hop, int32_t K, double eps,
                           float* z, int32_t reps, int32_t threads) {
  if (L < N || N < 1 || hop < 1 || K < 1 || reps < 1) return -1;
  const int64_t M = (L - N) / hop + 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
  const int nuse = N < K ? N : K;
  const int64_t total = (int64_t)reps * M;
#pragma omp parallel
  {
    cplx* in = (cplx*)malloc((size_t)K * sizeof(cplx));
    cplx* out = (cplx*)malloc((size_t)K * sizeof(cplx));
    cplx* scratch = (cplx*)malloc((size_t)2 * K * sizeof(cplx) + 64);
    float* frame = (float*)malloc((size_t)N * sizeof(float));
    float* zt = (float*)malloc((size_t)K * 2 * sizeof(float));
#pragma omp for schedule(static)
    for (int64_t i = 0; i < total; ++i) {
      const int64_t m = i % M;
      for (int n = 0; n < N; ++n) frame[n] = x[m * hop + n] * w[n]; /* framing + f32 window product (:94-101) */
      for (int n = 0; n < nuse; ++n) { in[n].re = (double)frame[n]; in[n].im = 0.0; }
      for (int n = nuse; n < K; ++n) { in[n].re = 0.0; in[n].im = 0.0; }
      bb_fft_rec(in, 1, K, out, scratch);
      float* zr = i < M ? z + (size_t)m * K * 2 : zt;
      for (int k = 0; k < K; ++k) { zr[2 * k] = bb_clean(out[k].re, eps); zr[2 * k + 1] = bb_clean(out[k].im, eps); }
    }
    free(in); free(out); free(scratch); free(frame); free(zt);
  }
  return total;
}

int bb_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
