// Host-side generators of the small parameter tensors of the hot path: windows, sinc, firwin,
// fft_frequencies, stft times.  They are O(N) on a few KB and must reproduce Nx.BinaryBackend's
// rounding bit-for-bit (every elementwise op rounded to f32, transcendental functions in double
// on the f32 operand — SURVEY.md Appendix A), which a device `cosf` would not.  Compile this file
// with -ffp-contract=off: a fused multiply-add would skip a rounding the reference performs.
//
// Reference: lib/nx_signal/windows.ex, lib/nx_signal/waveforms.ex:451-457,
//            lib/nx_signal/filters.ex:147-279, lib/nx_signal.ex:108-111, :154-166.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "nxsig_internal.h"

namespace nxsig {

static const double kPi = 3.14159265358979323846;

// T = float: the f32 generators of the reference's defaults; T = double: `type: {:f, 64}` (every op rounds to double)
template <typename T> static inline T cosT(T x) { return (T)std::cos((double)x); }
template <typename T> static inline T sinT(T x) { return (T)std::sin((double)x); }
template <typename T> static inline T sqrtT(T x) { return (T)std::sqrt((double)x); }
template <typename T> static inline T expT(T x) { return (T)std::exp((double)x); }
template <typename T> static inline T powT(T x, double p) { return (T)std::pow((double)x, p); }

// Nx.linspace(start, stop, n:, endpoint:) in f32: iota * step + start
template <typename T>
static void linspace32(T start, T stop, int64_t n, bool endpoint, T* out) {
  const T div = (T)(endpoint ? n - 1 : n);
  const T step = (stop - start) / div;
  for (int64_t i = 0; i < n; ++i) out[i] = (T)i * step + start;
}

// Nx.cos(mult * @pi * n / (l - 1)): `mult * @pi` folds in double, becomes an f32 constant, then
// (c * n) and the division are each rounded to f32 (windows.ex:185-186, :244, :298).
template <typename T>
static inline T cos_term(T k, double mult, int lm1) {
  const T c = (T)(mult * kPi);
  const T ang = (c * k) / (T)lm1;
  return cosT<T>(ang);
}

template <typename T>
static void win_bartlett(int n, T* w) {  // windows.ex:57-78
  const int n2 = n / 2, left = n2 + n % 2;
  const T nf = (T)n;
  for (int i = 0; i < left; ++i) w[i] = ((T)i * (T)2.0) / nf;
  for (int i = 0; i < n2; ++i) {
    const T idx = (T)i + (T)left;
    w[left + i] = (T)2.0 - (idx * (T)2.0) / nf;
  }
}

template <typename T>
static void win_triangular(int n, T* w) {  // windows.ex:98-126
  const int h = (n + 1) / 2;
  std::vector<T> left(h);
  if (n % 2 == 1) {
    for (int i = 0; i < h; ++i) left[i] = (((T)i + (T)1.0) * (T)2.0) / (T)(n + 1);
    for (int i = 0; i < h; ++i) w[i] = left[i];
    for (int i = 1; i < h; ++i) w[h + i - 1] = left[h - 1 - i];
  } else {
    for (int i = 0; i < h; ++i) left[i] = ((T)2.0 * ((T)i + (T)1.0) - (T)1.0) / (T)n;
    for (int i = 0; i < h; ++i) w[i] = left[i];
    for (int i = 0; i < h; ++i) w[h + i] = left[h - 1 - i];
  }
}

template <typename T>
static void win_blackman(int n, bool periodic, T* w) {  // windows.ex:160-202
  const int l = periodic ? n + 1 : n;
  const int m = (l + 1) / 2;
  std::vector<T> left(m), full;
  for (int i = 0; i < m; ++i) {
    const T k = (T)i;
    const T a = (T)0.42 - (T)0.5 * cos_term(k, 2.0, l - 1);
    left[i] = a + (T)0.08 * cos_term(k, 4.0, l - 1);
  }
  full = left;
  if (l % 2 == 0) {
    for (int i = m - 1; i >= 0; --i) full.push_back(left[i]);
  } else {
    for (int i = m - 2; i >= 0; --i) full.push_back(left[i]);
  }
  for (int i = 0; i < n; ++i) w[i] = full[i];  // periodic: drop the last sample
}

template <typename T>
static void win_cosine2(int n, bool periodic, T a0, T a1, bool hann, T* w) {
  // hamming (windows.ex:225-250): 0.54 - 0.46*cos(.) ; hann (:278-305): 0.5*(1 - cos(.))
  const int l = periodic ? n + 1 : n;
  for (int i = 0; i < n; ++i) {
    const T c = cos_term((T)i, 2.0, l - 1);
    w[i] = hann ? (T)0.5 * ((T)1.0 - c) : a0 - a1 * c;
  }
}

template <typename T>
static T kaiser_i0(T x) {  // windows.ex:371-386
  const T ax = std::fabs(x);
  if (ax < (T)3.75) {
    T s = (T)1.0 + powT<T>(ax, 2) / (T)4.0;
    s = s + powT<T>(ax, 4) / (T)64.0;
    s = s + powT<T>(ax, 6) / (T)2304.0;
    s = s + powT<T>(ax, 8) / (T)147456.0;
    return s;
  }
  const T two_pi = (T)(2.0f * (float)kPi);   // 2 * Nx.Constants.pi(): an f32 tensor whatever the window type
  // The bracket reproduces the reference's kaiser doctests (windows.ex:322-338) bit-for-bit only when
  // evaluated in double and rounded once (empirical; see oracle/nx_oracle.py:_kaiser_i0).
  const double a = (double)ax;
  const T bracket = (T)(1.0 + 1.0 / (8.0 * a) + 9.0 / (128.0 * a * a));
  return expT<T>(ax) / sqrtT<T>(two_pi * ax) * bracket;
}

template <typename T>
static void win_kaiser(int n, bool periodic, double beta, double eps, T* w) {  // windows.ex:341-369
  const int wl = periodic ? n + 1 : n;
  std::vector<T> ratio(wl);
  linspace32(-(T)1.0, (T)1.0, wl, true, ratio.data());
  const T den = (T)kaiser_i0((float)beta);   // kaiser_bessel_i0(beta) on the NUMBER beta: an f32 tensor (windows.ex:362)
  for (int i = 0; i < n; ++i) {
    T arg = (T)1.0 - powT<T>(ratio[i], 2);
    arg = std::max(arg, (T)eps);
    const T r = (T)beta * sqrtT<T>(arg);
    w[i] = kaiser_i0(r) / den;
  }
}

template <typename T>
static int window_t(int kind, int n, bool periodic, double beta, double eps, T* out) {
  if (n < 0 || !out) return set_error(NXSIG_ERR_INVALID_ARG, "window: n must be >= 0 and out non-null");
  switch (kind) {
    case NXSIG_WIN_RECTANGULAR:
      for (int i = 0; i < n; ++i) out[i] = (T)1.0;
      return NXSIG_OK;
    case NXSIG_WIN_BARTLETT: win_bartlett(n, out); return NXSIG_OK;
    case NXSIG_WIN_TRIANGULAR: win_triangular(n, out); return NXSIG_OK;
    case NXSIG_WIN_BLACKMAN: win_blackman(n, periodic, out); return NXSIG_OK;
    case NXSIG_WIN_HAMMING: win_cosine2(n, periodic, (T)0.54, (T)0.46, false, out); return NXSIG_OK;
    case NXSIG_WIN_HANN: win_cosine2(n, periodic, (T)0., (T)0., true, out); return NXSIG_OK;
    case NXSIG_WIN_KAISER: win_kaiser(n, periodic, beta, eps, out); return NXSIG_OK;
    default: return set_error(NXSIG_ERR_INVALID_ARG, "unknown window kind " + std::to_string(kind));
  }
}

template <typename T>
static void sinc_t(const T* t, int64_t n, T* out) {  // waveforms.ex:451-457
  const T pi32 = (T)(float)kPi;   // pi() of Nx.Constants: f32 whatever the type of t (waveforms.ex:452)
  for (int64_t i = 0; i < n; ++i) {
    const T x = t[i] * pi32;
    out[i] = (x == (T)0.0) ? (T)1.0 : sinT<T>(x) / x;
  }
}

template <typename T>
static int firwin_t(int num_taps, const double* cutoff, int n_cutoff, int window_kind, double beta, bool pass_zero,
               bool scale, double sampling_rate, T* out) {  // filters.ex:147-252
  if (num_taps < 1 || n_cutoff < 1 || !cutoff || !out)
    return set_error(NXSIG_ERR_INVALID_ARG, "firwin: num_taps >= 1 and a non-empty cutoff list are required");
  const double nyq = sampling_rate / 2.0;
  std::vector<double> cl(cutoff, cutoff + n_cutoff);
  for (auto& c : cl) c = c / nyq;
  std::sort(cl.begin(), cl.end());
  if (cl.front() <= 0.0)
    return set_error(NXSIG_ERR_INVALID_ARG, "cutoff must be strictly between 0 and Nyquist (exclusive), got: " +
                                                std::to_string(cl.front() * nyq));
  if (cl.back() >= 1.0)
    return set_error(NXSIG_ERR_INVALID_ARG, "cutoff must be strictly between 0 and Nyquist (exclusive), got: " +
                                                std::to_string(cl.back() * nyq));
  const bool even_n_cuts = (n_cutoff % 2) == 0;
  const bool nyquist_gain = (pass_zero && even_n_cuts) || (!pass_zero && !even_n_cuts);
  if (nyquist_gain && num_taps % 2 == 0)
    return set_error(NXSIG_ERR_INVALID_ARG,
                     "a filter with non-zero gain at Nyquist (e.g. highpass) requires an odd number of taps, got: " +
                         std::to_string(num_taps));
  switch (window_kind) {
    case NXSIG_WIN_HAMMING: case NXSIG_WIN_HANN: case NXSIG_WIN_BLACKMAN: case NXSIG_WIN_BARTLETT:
    case NXSIG_WIN_RECTANGULAR: case NXSIG_WIN_KAISER: break;
    default:
      return set_error(NXSIG_ERR_INVALID_ARG,
                       "unknown window, supported: :hamming, :hann, :blackman, :bartlett, :rectangular, {:kaiser, beta}");
  }
  const T m = (T)((num_taps - 1) / 2.0);
  std::vector<T> alpha(num_taps), h(num_taps, (T)0.0), tmp(num_taps), sa(num_taps), sb(num_taps);
  for (int i = 0; i < num_taps; ++i) alpha[i] = (T)i - m;
  std::vector<double> freqs;
  freqs.push_back(0.0);
  for (double c : cl) freqs.push_back(c);
  freqs.push_back(1.0);
  for (size_t i = 0; i + 1 < freqs.size(); ++i) {
    const bool use = pass_zero ? (i % 2 == 0) : (i % 2 == 1);
    if (!use) continue;
    const T a = (T)(float)freqs[i], b = (T)(float)freqs[i + 1];  // defnp args: floats become f32 tensors (whatever `type`)
    for (int k = 0; k < num_taps; ++k) tmp[k] = a * alpha[k];
    sinc_t<T>(tmp.data(), num_taps, sa.data());
    for (int k = 0; k < num_taps; ++k) tmp[k] = b * alpha[k];
    sinc_t<T>(tmp.data(), num_taps, sb.data());
    for (int k = 0; k < num_taps; ++k) {  // acc + b*sinc(b*alpha) - a*sinc(a*alpha)  (:223-227)
      const T ca = a * sa[k], cb = b * sb[k];
      h[k] = (h[k] + cb) - ca;
    }
  }
  std::vector<T> w(num_taps);
  int rc = window_t<T>(window_kind, num_taps, /*periodic=*/false, beta, 1.0e-7, w.data());  // :254-279
  if (rc != NXSIG_OK) return rc;
  for (int k = 0; k < num_taps; ++k) h[k] = h[k] * w[k];
  if (scale) {  // firwin_scale :229-252
    double sf;
    if (pass_zero) sf = 0.0;
    else if (n_cutoff == 1) sf = 1.0;
    else sf = (cl[0] + cl[1]) / 2.0;
    const T c = (T)(kPi * sf);
    double acc = 0.0;  // Nx.dot: accumulate in double, round once
    for (int k = 0; k < num_taps; ++k) acc += (double)h[k] * (double)cosT<T>(alpha[k] * c);
    const T s = std::fabs((T)acc);
    for (int k = 0; k < num_taps; ++k) h[k] = h[k] / s;
  }
  std::copy(h.begin(), h.end(), out);
  return NXSIG_OK;
}

template <typename T>
static void fft_frequencies_t(double fs, int K, bool endpoint, T* out) {  // nx_signal.ex:154-166
  const float step = (float)fs / (float)K;   // sampling_rate enters the defn as an f32 tensor
  linspace32((T)0.0, (T)(step * (float)K), K, endpoint, out);
}

int window_f32(int kind, int n, bool periodic, double beta, double eps, float* out) { return window_t<float>(kind, n, periodic, beta, eps, out); }
int window_f64(int kind, int n, bool periodic, double beta, double eps, double* out) { return window_t<double>(kind, n, periodic, beta, eps, out); }
void sinc_f32(const float* t, int64_t n, float* out) { sinc_t<float>(t, n, out); }
void sinc_f64(const double* t, int64_t n, double* out) { sinc_t<double>(t, n, out); }
int firwin_f32(int num_taps, const double* cutoff, int n_cutoff, int window_kind, double beta, bool pass_zero, bool scale,
               double sampling_rate, float* out) {
  return firwin_t<float>(num_taps, cutoff, n_cutoff, window_kind, beta, pass_zero, scale, sampling_rate, out);
}
int firwin_f64(int num_taps, const double* cutoff, int n_cutoff, int window_kind, double beta, bool pass_zero, bool scale,
               double sampling_rate, double* out) {
  return firwin_t<double>(num_taps, cutoff, n_cutoff, window_kind, beta, pass_zero, scale, sampling_rate, out);
}
void fft_frequencies_f32(double fs, int K, bool endpoint, float* out) { fft_frequencies_t<float>(fs, K, endpoint, out); }
void fft_frequencies_f64(double fs, int K, bool endpoint, double* out) { fft_frequencies_t<double>(fs, K, endpoint, out); }


void stft_times_f32(int N, double fs, int64_t M, float* out) {  // nx_signal.ex:108-111
  const float two_fs = 2.0f * (float)fs;
  const float time_step = (float)((double)N / (double)two_fs);
  const float last = time_step * (float)M;
  linspace32(time_step, last, M, true, out);
}

// NxSignal.mel_filters/4 — lib/nx_signal.ex:412-445, op by op in f32 with double transcendentals
void mel_filters_f32(int K, int mel_bins, double fs, double max_mel, double f_sp, float* out) {
  std::vector<float> fftfreqs(K), mels(mel_bins + 2), mel_f(mel_bins + 2);
  fft_frequencies_f32(fs, K, false, fftfreqs.data());
  linspace32(0.0f, (float)max_mel / (float)f_sp, mel_bins + 2, true, mels.data());
  const float fsp = (float)f_sp;
  const float min_log_hz = 1000.0f;
  const float min_log_mel = min_log_hz / fsp;
  const float logstep = (float)std::log((double)6.4f) / 27.0f;  // Nx.log(6.4) / 27
  for (int i = 0; i < mel_bins + 2; ++i) {
    const float lin = fsp * mels[i];
    const float arg = logstep * (mels[i] - min_log_mel);
    const float lg = min_log_hz * expT<float>(arg);
    mel_f[i] = (mels[i] >= min_log_mel) ? lg : lin;
  }
  for (int b = 0; b < mel_bins; ++b) {
    const float fd_lo = mel_f[b + 1] - mel_f[b], fd_hi = mel_f[b + 2] - mel_f[b + 1];
    const float enorm = 2.0f / (mel_f[b + 2] - mel_f[b]);
    for (int k = 0; k < K; ++k) {
      const float lower = -(mel_f[b] - fftfreqs[k]) / fd_lo;       // -ramps[b] / fdiff[b]
      const float upper = (mel_f[b + 2] - fftfreqs[k]) / fd_hi;    // ramps[b+2] / fdiff[b+1]
      float w = lower < upper ? lower : upper;
      w = (w > 0.0f) ? w : 0.0f;                                    // Nx.max(0, .): +0.0 for w <= 0
      out[(size_t)b * K + k] = w * enorm;
    }
  }
}

// f32 scalar the spectrum is divided by (stft :116/:119) or multiplied by (istft :614/:617).
// Nx.sum accumulates in double and rounds once; window ** 2 is an exact f32 product.
// f64 window: Nx.sum in f64 (sequential double accumulation), sampling_rate an f32 tensor, the square root in double
double scaling_factor_f64(const double* w, int N, int scaling, double fs) {
  double acc = 0.0;
  if (scaling == NXSIG_SCALE_SPECTRUM) {
    for (int i = 0; i < N; ++i) acc += w[i];
    return acc;
  }
  for (int i = 0; i < N; ++i) acc += w[i] * w[i];
  return std::sqrt((double)(float)fs * acc);
}

float scaling_factor(const float* w, int N, int scaling, double fs) {
  double acc = 0.0;
  if (scaling == NXSIG_SCALE_SPECTRUM) {
    for (int i = 0; i < N; ++i) acc += (double)w[i];
    return (float)acc;
  }
  for (int i = 0; i < N; ++i) acc += (double)(w[i] * w[i]);
  const float s2 = (float)acc;
  const float prod = (float)fs * s2;
  return sqrtT<float>(prod);
}

}  // namespace nxsig
