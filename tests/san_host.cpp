// TEST INFRASTRUCTURE: AddressSanitizer / UndefinedBehaviorSanitizer run of the host-side numerics
// (nx_signal_amd/csrc/host_numerics.cpp: windows, sinc, firwin, mel_filters, fft_frequencies, stft times, scaling factors),
// of the framing geometry and shard plans' arithmetic, and of the C leg of the oracle (oracle/bb_baseline.c) — SURVEY §5
// "race detection / sanitizers".  Built by tests/test_sanitizers.py with g++ -fsanitize=address,undefined -fno-sanitize-recover;
// any out-of-bounds access, overflow or misaligned access aborts the process.  Prints a checksum that the test compares with
// the library's own output so that the sanitized build is known to compute the same values.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "nxsig_internal.h"

namespace nxsig {
static thread_local std::string g_err;
int set_error(int code, const std::string& msg) { g_err = msg; return code; }
const char* last_error_cstr() { return g_err.c_str(); }
}  // namespace nxsig

extern "C" int64_t bb_stft_f32(const float* x, int64_t L, const float* w, int32_t N, int32_t hop, int32_t K, double eps, float* z, int32_t threads);
extern "C" int64_t bb_stft_f32_repeat(const float* x, int64_t L, const float* w, int32_t N, int32_t hop, int32_t K, double eps, float* z, int32_t reps, int32_t threads);

static uint64_t h64 = 1469598103934665603ull;
static void mix(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h64 ^= b[i]; h64 *= 1099511628211ull; } }

int main() {
  using namespace nxsig;
  std::vector<int> sizes;
  for (int n = 1; n <= 130; ++n) sizes.push_back(n);
  for (int n : {255, 256, 257, 400, 1000, 1024, 2048, 4097}) sizes.push_back(n);
  for (int n : sizes) {
    std::vector<float> w((size_t)n);
    for (int kind = 0; kind <= 6; ++kind)
      for (int per = 0; per <= 1; ++per) {
        if (window_f32(kind, n, per != 0, kind == 6 ? 8.6 : 0.0, 1e-7, w.data()) == 0) mix(w.data(), w.size() * 4);
      }
    for (int sc = 0; sc <= 2; ++sc) { float f = scaling_factor(w.data(), n, sc, 48000.0); mix(&f, 4); }
  }
  (void)window_f32(99, 8, true, 0.0, 1e-7, nullptr);  // unknown kind: error, no write
  for (int taps : {1, 2, 3, 4, 5, 31, 32, 101, 257, 1025}) {
    std::vector<float> h((size_t)taps);
    const double c1[] = {0.3}, c2[] = {0.2, 0.5}, c3[] = {0.1, 0.3, 0.6}, bad[] = {1.5}, unsorted[] = {0.5, 0.2};
    for (int kind : {0, 1, 3, 4, 5, 6})
      for (int pz = 0; pz <= 1; ++pz)
        for (int sc = 0; sc <= 1; ++sc) {
          if (firwin_f32(taps, c1, 1, kind, 5.0, pz, sc, 2.0, h.data()) == 0) mix(h.data(), h.size() * 4);
          if (firwin_f32(taps, c2, 2, kind, 5.0, pz, sc, 2.0, h.data()) == 0) mix(h.data(), h.size() * 4);
          if (firwin_f32(taps, c3, 3, kind, 5.0, pz, sc, 2.0, h.data()) == 0) mix(h.data(), h.size() * 4);
          if (firwin_f32(taps, unsorted, 2, kind, 5.0, pz, sc, 2.0, h.data()) == 0) mix(h.data(), h.size() * 4);
          (void)firwin_f32(taps, bad, 1, kind, 5.0, pz, sc, 2.0, h.data());
          (void)firwin_f32(taps, c1, 0, kind, 5.0, pz, sc, 2.0, h.data());
        }
  }
  for (int K : {2, 10, 16, 400, 512, 1024, 2048})
    for (int bins : {1, 5, 80, 128}) {
      std::vector<float> f((size_t)K * bins);
      mel_filters_f32(K, bins, 16000.0, 3016.0, 200.0 / 3.0, f.data());
      mix(f.data(), f.size() * 4);
      std::vector<float> fr((size_t)K);
      fft_frequencies_f32(16000.0, K, false, fr.data()); mix(fr.data(), fr.size() * 4);
      fft_frequencies_f32(16000.0, K, true, fr.data()); mix(fr.data(), fr.size() * 4);
    }
  for (int64_t M : {0, 1, 2, 184, 11247}) { std::vector<float> t((size_t)M + 1); stft_times_f32(1024, 48000.0, M, t.data()); mix(t.data(), (size_t)M * 4); }
  { std::vector<float> t(1000), o(1000); for (int i = 0; i < 1000; ++i) t[i] = (float)(i - 500) * 0.01f; sinc_f32(t.data(), 1000, o.data()); mix(o.data(), 4000); }
  // C leg of the oracle: ragged lengths, fft_length below / above the frame length, odd lengths (naive DFT branch)
  for (int cfg = 0; cfg < 5; ++cfg) {
    const int N = (int[]){64, 100, 48, 7, 256}[cfg], K = (int[]){64, 128, 36, 15, 256}[cfg], hop = (int[]){16, 25, 12, 3, 64}[cfg];
    const int64_t L = 1000 + cfg * 37;
    std::vector<float> x((size_t)L), w((size_t)N);
    for (int64_t i = 0; i < L; ++i) x[(size_t)i] = (float)((i * 2654435761u) % 1000) * 0.002f - 1.0f;
    window_f32(5, N, true, 0.0, 1e-7, w.data());
    const int64_t M = (L - N) / hop + 1;
    std::vector<float> z((size_t)M * K * 2);
    if (bb_stft_f32(x.data(), L, w.data(), N, hop, K, 1e-10, z.data(), 1) != M) return 2;
    mix(z.data(), z.size() * 4);
    if (bb_stft_f32_repeat(x.data(), L, w.data(), N, hop, K, 1e-10, z.data(), 3, 2) != 3 * M) return 3;
    mix(z.data(), z.size() * 4);
  }
  std::printf("sanitized host numerics ok, checksum %016llx\n", (unsigned long long)h64);
  return 0;
}
