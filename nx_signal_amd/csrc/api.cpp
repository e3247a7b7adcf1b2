// C-ABI entry points declared in include/nxsig.h.  Validates arguments the way the reference's
// deftransforms do (same failure cases -> NXSIG_ERR_INVALID_ARG, which the host mirrors turn into
// ArgumentError), resolves framing geometry, stages host buffers when asked to, and enqueues the HIP
// kernels on the context's stream.  No C++ exception may leave this file: every entry point is wrapped.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <thread>
#include <vector>
#ifdef __linux__
#include <sys/mman.h>
#endif

#include "nxsig_internal.h"
#include <strings.h>
#include <cstdio>

namespace nxsig {

static thread_local std::string g_last_error;

int set_error(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
const char* last_error_cstr() { return g_last_error.c_str(); }

static thread_local std::string g_dispatch;
void dispatch_reset() { g_dispatch.clear(); }
void dispatch_note(const char* family) {
  const std::string f(family);
  size_t at = 0;
  while ((at = g_dispatch.find(f, at)) != std::string::npos) {   // already noted (as a whole '+'-separated item)?
    const bool l = at == 0 || g_dispatch[at - 1] == '+', r = at + f.size() == g_dispatch.size() || g_dispatch[at + f.size()] == '+';
    if (l && r) return;
    at += f.size();
  }
  if (!g_dispatch.empty()) g_dispatch += '+';
  g_dispatch += f;
}
const char* dispatch_cstr() { return g_dispatch.c_str(); }
// one per compute entry point of the C ABI: resets the calling thread's record, and leaves a copy in the context on the way out (the
// BEAM's dirty schedulers move between threads from call to call: the NIF reads the context's copy)
struct DispatchScope {
  Ctx* c;
  explicit DispatchScope(Ctx* ctx) : c(ctx) { dispatch_reset(); }
  ~DispatchScope() { c->last_dispatch = g_dispatch; }
};

// In-process cache key (never persisted): FNV-1a over the tail bytes, and for the bulk four independent multiply-xor lanes over
// 8-byte words — a byte-at-a-time FNV costs one dependent multiply per byte, 0.7 ms for the 512 KB mel filterbank that
// stft_to_mel / mel_spectrogram look up on every call.
uint64_t fnv1a(uint64_t seed, const void* data, size_t bytes) {
  const uint64_t P = 1099511628211ull;
  uint64_t h = 1469598103934665603ull ^ seed;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  size_t i = 0;
  if (bytes >= 64) {
    uint64_t l0 = h ^ 0x9E3779B97F4A7C15ull, l1 = h ^ 0xC2B2AE3D27D4EB4Full, l2 = h ^ 0x165667B19E3779F9ull, l3 = h ^ 0x27D4EB2F165667C5ull;
    for (; i + 32 <= bytes; i += 32) {
      uint64_t w[4];
      std::memcpy(w, p + i, 32);
      l0 = (l0 ^ w[0]) * P; l0 ^= l0 >> 29;
      l1 = (l1 ^ w[1]) * P; l1 ^= l1 >> 29;
      l2 = (l2 ^ w[2]) * P; l2 ^= l2 >> 29;
      l3 = (l3 ^ w[3]) * P; l3 ^= l3 >> 29;
    }
    h = (((((l0 * P) ^ l1) * P) ^ l2) * P ^ l3) * P;
    h ^= h >> 32;
  }
  for (; i < bytes; ++i) {
    h ^= p[i];
    h *= P;
  }
  return h;
}

int make_framing(int64_t L, int32_t N, int32_t hop, int32_t pad_mode, int64_t pad_lo, int64_t pad_hi, Framing* out) {
  if (N < 1) return set_error(NXSIG_ERR_INVALID_ARG, "window_length must be >= 1");
  if (hop < 1)  // lib/nx_signal.ex:282-284
    return set_error(NXSIG_ERR_INVALID_ARG, "expected an integer >= 1 or a list of integers, got: " + std::to_string(hop));
  if (L < 1) return set_error(NXSIG_ERR_INVALID_ARG, "signal length must be >= 1");
  Framing f;
  f.L = L; f.N = N; f.hop = hop; f.reflect = 0; f.lo = 0; f.hi = 0;
  switch (pad_mode) {
    case NXSIG_PAD_VALID: break;
    case NXSIG_PAD_REFLECT: f.reflect = 1; f.lo = N / 2; f.hi = N / 2; break;                 // :262
    case NXSIG_PAD_SAME: { const int64_t tot = N - 1; f.lo = tot / 2; f.hi = tot - tot / 2; } break;  // :308-312
    case NXSIG_PAD_EXPLICIT: f.lo = pad_lo; f.hi = pad_hi; break;
    default:  // :325-329
      return set_error(NXSIG_ERR_INVALID_ARG,
                       "invalid padding mode specified, padding must be one of :valid, :same, or a padding configuration");
  }
  const int64_t Lp = L + f.lo + f.hi;
  if (Lp < N)
    return set_error(NXSIG_ERR_INVALID_ARG, "window of length " + std::to_string(N) +
                                                " does not fit the (padded) signal of length " + std::to_string(Lp));
  f.M = (Lp - N) / hop + 1;  // pooled shape of Nx.window_max, :289-298
  *out = f;
  return NXSIG_OK;
}

int ctx_twiddles(Ctx* c, int K, const float2** out) {
  auto it = c->twiddles.find(K);
  if (it == c->twiddles.end()) {
    std::vector<float2> tw(K);
    for (int j = 0; j < K; ++j) {
      const double ang = -2.0 * 3.14159265358979323846 * (double)j / (double)K;
      tw[j] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    DeviceTable t;
    t.bytes = (size_t)K * sizeof(float2);
    NXSIG_HIP_TRY(hipMalloc(&t.ptr, t.bytes));
    NXSIG_HIP_TRY(hipMemcpyAsync(t.ptr, tw.data(), t.bytes, hipMemcpyHostToDevice, c->stream));
    NXSIG_HIP_TRY(hipStreamSynchronize(c->stream));  // tw goes out of scope
    it = c->twiddles.emplace(K, t).first;
  }
  *out = reinterpret_cast<const float2*>(it->second.ptr);
  return NXSIG_OK;
}

int ctx_table(Ctx* c, uint64_t tag, const void* host, size_t bytes, const void** out) {
  // content-addressed: key = fnv1a(tag, content) ^ size.  A hit on a table of up to 1 MiB is VERIFIED against the kept host copy
  // (size + content): on a 64-bit collision the next key is probed instead of silently handing out another table (a wrong
  // window / filter).  Larger tables keep no host copy and are matched on hash + size only.
  uint64_t key = fnv1a(tag, host, bytes) ^ (uint64_t)bytes;
  for (int probe = 0; probe < 8; ++probe, key = key * 0x9E3779B97F4A7C15ull + 1) {
    auto it = c->tables.find(key);
    if (it == c->tables.end()) {
      DeviceTable t;
      t.bytes = bytes;
      NXSIG_HIP_TRY(hipMalloc(&t.ptr, bytes ? bytes : 4));
      NXSIG_HIP_TRY(hipMemcpyAsync(t.ptr, host, bytes, hipMemcpyHostToDevice, c->stream));
      NXSIG_HIP_TRY(hipStreamSynchronize(c->stream));
      t.has_host = bytes <= ((size_t)1 << 20);
      if (t.has_host) t.host.assign(static_cast<const unsigned char*>(host), static_cast<const unsigned char*>(host) + bytes);
      it = c->tables.emplace(key, std::move(t)).first;
      *out = it->second.ptr;
      return NXSIG_OK;
    }
    const DeviceTable& t = it->second;
    if (t.bytes == bytes && (!t.has_host || bytes == 0 || std::memcmp(t.host.data(), host, bytes) == 0)) {
      *out = t.ptr;
      return NXSIG_OK;
    }
  }
  return set_error(NXSIG_ERR_HIP, "table cache: repeated hash collisions");
}

static thread_local const Ctx* t_mel_defer = nullptr;
MelDeferScope::MelDeferScope(const Ctx* c) { t_mel_defer = c; }
MelDeferScope::~MelDeferScope() { t_mel_defer = nullptr; }
bool mel_deferred(const Ctx* c) { return t_mel_defer == c; }

int ctx_scratch(Ctx* c, int slot, size_t bytes, void** out) {
  if (c->scratch_bytes[slot] < bytes) {
    if (c->scratch[slot]) {
      NXSIG_HIP_TRY(hipStreamSynchronize(c->stream));
      NXSIG_HIP_TRY(hipFree(c->scratch[slot]));
      c->scratch[slot] = nullptr;
      c->scratch_bytes[slot] = 0;
    }
    NXSIG_HIP_TRY(hipMalloc(&c->scratch[slot], bytes));
    c->scratch_bytes[slot] = bytes;
  }
  *out = c->scratch[slot];
  return NXSIG_OK;
}

int ctx_window(Ctx* c, const float* w, int N, int K, const float** dev, const float** devK) {
  if (c->memo_dev && c->memo_K == K && (int)c->memo_win.size() == N &&
      std::memcmp(c->memo_win.data(), w, (size_t)N * sizeof(float)) == 0) {
    *dev = c->memo_dev; *devK = c->memo_devK;
    return NXSIG_OK;
  }
  const void *d = nullptr, *dk = nullptr;
  int rc = ctx_table(c, 0x57494Eull, w, (size_t)N * sizeof(float), &d);
  if (rc) return rc;
  std::vector<float> padded((size_t)K, 0.0f);
  for (int i = 0; i < K && i < N; ++i) padded[i] = w[i];
  rc = ctx_table(c, 0x57494E4Bull, padded.data(), padded.size() * sizeof(float), &dk);
  if (rc) return rc;
  c->memo_win.assign(w, w + N);
  c->memo_K = K;
  c->memo_dev = reinterpret_cast<const float*>(d);
  c->memo_devK = reinterpret_cast<const float*>(dk);
  *dev = c->memo_dev; *devK = c->memo_devK;
  return NXSIG_OK;
}

// declared here, implemented in the .hip files
int launch_stft_generic(Ctx* c, const StftLaunch& a);
int launch_istft_generic(Ctx* c, const IstftLaunch& a);
int launch_fir_generic(Ctx* c, const FirLaunch& a);
int launch_stft_wave(Ctx* c, const StftLaunch& a, bool* handled);
int launch_istft_wave(Ctx* c, const IstftLaunch& a, const float* window_host, bool* handled);
int launch_fir_wave(Ctx* c, const FirLaunch& a, bool* handled);

// More rows than one launch takes (the bounds-checked and generic kernels put the row on gridDim.y, <= 65 535) run as slabs of 65 504
// rows — a multiple of 32, so that every slab's rows keep the first slab's alignment (FirWaveArgs::row_mod).  Rows are independent in
// stft / istft / fir; the sinks whose result depends on the whole tensor (log-mel, dBFS) keep the 65 535-row limit.
static constexpr int32_t kSlabRows = 65504;

int launch_stft(Ctx* c, const StftLaunch& a) {
  if (a.batch > kSlabRows) {
    for (int32_t r0 = 0; r0 < a.batch; r0 += kSlabRows) {
      StftLaunch b = a;
      b.batch = a.batch - r0 < kSlabRows ? a.batch - r0 : kSlabRows;
      b.x = a.x + (size_t)r0 * a.batch_stride;
      b.z = a.z + (size_t)r0 * a.fr.M * a.K;
      int rc = launch_stft(c, b);
      if (rc) return rc;
    }
    return NXSIG_OK;
  }
  bool handled = false;
  int rc = launch_stft_wave(c, a, &handled);
  if (rc || handled) return rc;
  return launch_stft_generic(c, a);
}
int launch_istft_packed_wave(Ctx* c, const IstftLaunch& a, const float* window_host, bool* handled);
int launch_istft(Ctx* c, const IstftLaunch& a, const float* window_host) {
  if (a.batch > kSlabRows && !a.onesided) {
    const int64_t out_len = a.M * a.hop + (a.N - a.hop);
    for (int32_t r0 = 0; r0 < a.batch; r0 += kSlabRows) {
      IstftLaunch b = a;
      b.batch = a.batch - r0 < kSlabRows ? a.batch - r0 : kSlabRows;
      b.z = a.z + (size_t)r0 * a.M * a.K;
      b.y = a.y + (size_t)r0 * out_len;
      int rc2 = launch_istft(c, b, window_host);
      if (rc2) return rc2;
    }
    return NXSIG_OK;
  }
  bool handled = false;
  int rc;
  if (a.onesided) {  // packed half spectrum in, real signal out (nxsig_istft_packed_f32)
    if ((rc = launch_istft_packed_wave(c, a, window_host, &handled))) return rc;
    if (handled) {
      return launch_istft_fix(c, a, window_host);
    }
    // no fused kernel for this geometry: the Hermitian rows are written out, the complex path runs, its real part is kept
    const int64_t out_len = a.M * a.hop + (a.N - a.hop);
    void *zf = nullptr, *yf = nullptr;
    if ((rc = ctx_scratch(c, 24, (size_t)a.batch * a.M * a.K * sizeof(float2), &zf))) return rc;
    if ((rc = ctx_scratch(c, 25, (size_t)a.batch * out_len * sizeof(float2), &yf))) return rc;
    if ((rc = launch_full_from_packed(c, a.z, (int64_t)a.batch * a.M, a.K, reinterpret_cast<float2*>(zf)))) return rc;
    IstftLaunch b = a;
    b.onesided = false; b.z = reinterpret_cast<const float2*>(zf); b.y = reinterpret_cast<float2*>(yf);
    b.nf_list = nullptr; b.nf_frames_per_unit = 0;
    if ((rc = launch_istft(c, b, window_host))) return rc;
    return launch_real_from_c64(c, b.y, (int64_t)a.batch * out_len, reinterpret_cast<float*>(a.y));
  }
  rc = launch_istft_wave(c, a, window_host, &handled);
  if (rc) return rc;
  if (!handled && a.filt) {
    // no fused kernel for this geometry: the filter product is materialised once (the two-step chain), then the plain path runs
    void* zf = nullptr;
    if ((rc = ctx_scratch(c, 20, (size_t)a.batch * a.M * a.K * sizeof(float2), &zf))) return rc;
    if ((rc = launch_spectrum_mul(c, a.z, (int64_t)a.batch * a.M, a.K, a.filt, reinterpret_cast<float2*>(zf)))) return rc;
    IstftLaunch b = a;
    b.z = reinterpret_cast<const float2*>(zf);
    b.filt = nullptr;
    return launch_istft(c, b, window_host);
  }
  if (!handled && (rc = launch_istft_generic(c, a))) return rc;
  // ill-conditioned edge samples recomputed in double; kernels that invert several frames per transform reported their non-finite
  // units: those samples again, frame by frame (one launch for both)
  return launch_istft_fix(c, a, window_host);
}
// filters longer than the overlap-save kernels take (> 4096 taps: impulse responses of seconds): one transform of
// next_pow2(L + taps - 1) points per row, the way the reference's fftconvolve does it (lib/nx_signal/convolution.ex:252-329),
// through the device-side fft_nd fold (four-step rows up to 2^26 points); the requested slice is copied out of the full result
static int launch_fir_long(Ctx* c, const FirLaunch& a) {
  dispatch_note("fir.long");
  const int64_t full = a.L + a.taps - 1;
  if (full > ((int64_t)1 << 26))
    return set_error(NXSIG_ERR_UNSUPPORTED, "fir: more than 4096 taps with length + taps - 1 > 2^26 is not supported");
  const void* hd = nullptr;
  int rc = ctx_table(c, 0xF17A95ull ^ ((uint64_t)a.taps << 24), a.h_host, (size_t)a.taps * sizeof(float), &hd);
  if (rc) return rc;
  void* tmp = nullptr;
  if ((rc = ctx_scratch(c, 21, (size_t)full * sizeof(float), &tmp))) return rc;
  const int64_t s1 = a.L, s2 = a.taps;
  for (int32_t row = 0; row < a.batch; ++row) {
    if ((rc = launch_fftconvolve_nd(c, a.x + (size_t)row * a.batch_stride, true, &s1, hd, true, &s2, 1, NXSIG_CONV_FULL, tmp, nullptr))) return rc;
    NXSIG_HIP_TRY(hipMemcpyAsync(a.y + (size_t)row * a.out_len, static_cast<const float*>(tmp) + a.out_start, (size_t)a.out_len * sizeof(float),
                                 hipMemcpyDeviceToDevice, c->stream));
  }
  return NXSIG_OK;
}

int launch_fir_partition_sum(Ctx* c, int n, const float* const* src, const int64_t* i0, const int64_t* len, bool accumulate, bool clean, float scale,
                             float* y, int32_t batch, int64_t out_len);   // kernels_generic.hip

// 1 026 ... 32 768 taps (round 5): the filter is cut into P partitions of <= 1 025 taps (stride 256 ... 1 024), each one a FIR call of the tuned overlap-save
// kernels into a scratch tensor (full convolution of x with h[p S .. p S + taps_p), delayed by p S samples), and one pass sums them into
// y.  P x (8 B per sample at the 1 025-tap kernel's rate) + one (4 P + 4) B pass instead of the 8192-point workgroup kernel (0.05 of
// the roofline) or one 2^k-point transform per row (> 4 096 taps: 0.009).  Non-finite samples poison their row like everywhere else
// (FirLaunch::row_flags, shared by the partitions).  Scratch: at most ~4 GB per round of partitions; later rounds add to y.
// The reference's result is ONE Nx.ifft output: samples with |y| <= 1e-10 are exact zeros (convolution.ex:282).  The overlap-save
// kernels apply that clean-up to what they store — here partial sums, where it does not belong.  So the partitions run with h x 2^20
// (exact: the kernels' threshold then sits at 1e-16 of the true scale, far below what decides a sample of y), the summing pass
// multiplies by 2^-20 and cleans the finished sums.
static int launch_fir_partitioned(Ctx* c, const FirLaunch& a_in) {
  dispatch_note("fir.partitioned");
  FirLaunch a = a_in;
  int rc = fir_row_flags(c, a.batch, &a.row_flags);
  if (rc) return rc;
  constexpr float kUp = 1048576.0f, kDown = 1.0f / 1048576.0f;
  std::vector<float> hs((size_t)a.taps);
  for (int i = 0; i < a.taps; ++i) hs[i] = a.h_host[i] * kUp;
  // partition stride S: a multiple of 256 (the delays p S keep the rows' 16-byte alignment: the stream kernels decline unaligned
  // spans; and taps_p - 1 is what the real-block kernel pads to a multiple of 256 anyway); the last partition takes up to S + 1 taps
  const int P = (a.taps - 1 + 1023) / 1024;
  int S = ((a.taps - 1 + P - 1) / P + 255) / 256 * 256;
  if (S > 1024) S = 1024;
  const int64_t row_bytes = (int64_t)a.batch * (a.out_len + 4) * 4;
  int per_round = (int)(((int64_t)4 << 30) / (row_bytes > 0 ? row_bytes : 1));
  if (per_round < 1) per_round = 1;
  if (per_round > 8) per_round = 8;
  void* tmp = nullptr;
  const int n_parts = (a.taps - 1 + S - 1) / S;
  const int first_round = n_parts < per_round ? n_parts : per_round;
  if ((rc = ctx_scratch(c, 21, (size_t)first_round * (size_t)row_bytes, &tmp))) return rc;
  bool any = false;
  for (int p0 = 0; p0 < n_parts; p0 += per_round) {
    const float* src[8];
    int64_t i0s[8], lens[8];
    int n = 0;
    size_t off = 0;
    for (int p = p0; p < n_parts && p < p0 + per_round; ++p) {
      const int64_t delay = (int64_t)p * S;
      const int32_t taps_p = p == n_parts - 1 ? a.taps - p * S : S;
      const int64_t full_p = a.L + taps_p - 1;                       // length of x * h_p
      int64_t i0 = delay - a.out_start;                              // first output of y this partition reaches
      if (i0 < 0) i0 = 0;
      int64_t i1 = full_p - 1 + delay - a.out_start;                 // last one
      if (i1 > a.out_len - 1) i1 = a.out_len - 1;
      if (i1 < i0) continue;
      FirLaunch q = a;
      q.h_host = hs.data() + (size_t)p * S; q.taps = taps_p;
      q.out_start = a.out_start + i0 - delay;
      q.out_len = (i1 - i0 + 1 + 3) & ~(int64_t)3;   // whole 16-byte groups per row (the stream kernels decline unaligned rows); what lies
                                                      // beyond the partition's full convolution comes out as zeros
      q.y = reinterpret_cast<float*>(static_cast<char*>(tmp) + off);
      off += (size_t)a.batch * (size_t)q.out_len * 4;
      bool handled = false;
      if ((rc = launch_fir_wave(c, q, &handled))) return rc;
      if (!handled && (rc = launch_fir_generic(c, q))) return rc;
      src[n] = q.y; i0s[n] = i0; lens[n] = q.out_len; ++n;
    }
    const bool last = p0 + per_round >= n_parts;
    if (n == 0 && any && !last) continue;
    if ((rc = launch_fir_partition_sum(c, n, src, i0s, lens, any, last, kDown, a.y, a.batch, a.out_len))) return rc;
    any = true;
  }
  return launch_fir_poison(c, a);
}

int launch_fir(Ctx* c, const FirLaunch& a_in) {
  if (a_in.out_len <= 0 || a_in.batch == 0) return NXSIG_OK;
  if (a_in.batch > kSlabRows) {
    for (int32_t r0 = 0; r0 < a_in.batch; r0 += kSlabRows) {
      FirLaunch b = a_in;
      b.batch = a_in.batch - r0 < kSlabRows ? a_in.batch - r0 : kSlabRows;
      b.x = a_in.x + (size_t)r0 * a_in.batch_stride;
      b.y = a_in.y + (size_t)r0 * a_in.out_len;
      int rc = launch_fir(c, b);
      if (rc) return rc;
    }
    return NXSIG_OK;
  }
  if (a_in.taps > 32769) return launch_fir_long(c, a_in);   // one transform per row, like the reference: non-finite rows come out NaN
  if (a_in.taps > 1025 && !tune(c, kT_DISABLE_WAVE, 0) && tune(c, kT_FIR_DLINE, 1)) {
    // round 6: one forward transform per input block, the partitions applied as a frequency-domain delay line, one inverse per output
    // block (kernels_wave_firlong.hip); NXSIG_FIR_DLINE=0 keeps round 5's partition-by-partition form
    FirLaunch a = a_in;
    int rcd = fir_row_flags(c, a.batch, &a.row_flags);
    if (rcd) return rcd;
    bool hd = false;
    if ((rcd = launch_fir_dline(c, a, &hd))) return rcd;
    if (hd) return launch_fir_poison(c, a);
  }
  if (a_in.taps > 32768) return launch_fir_long(c, a_in);
  if (a_in.taps > 1025 && !tune(c, kT_DISABLE_WAVE, 0)) return launch_fir_partitioned(c, a_in);
  if (a_in.taps > 4096) return launch_fir_long(c, a_in);
  FirLaunch a = a_in;
  int rc = fir_row_flags(c, a.batch, &a.row_flags);
  if (rc) return rc;
  bool handled = false;
  rc = launch_fir_wave(c, a, &handled);
  if (rc) return rc;
  if (!handled && (rc = launch_fir_generic(c, a))) return rc;
  return launch_fir_poison(c, a);   // rows that held an Inf / NaN sample: NaN from end to end (FirLaunch::row_flags)
}

struct DeviceGuard {
  explicit DeviceGuard(Ctx* c) : lock(c->mu) {
    ok = (hipSetDevice(c->device) == hipSuccess);
    // bound the content-addressed table cache.  Done at API entry only, never while a call holds table pointers, and never
    // while the stream is being captured (a captured graph bakes table pointers in).  A graph captured EARLIER stays valid as
    // long as its context sees fewer than 1024 distinct tables afterwards; beyond that re-capture it (streams).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (ok && c->tables.size() > 1024 && hipStreamIsCapturing(c->stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
      (void)hipStreamSynchronize(c->stream);
      for (auto& kv : c->tables) (void)hipFree(kv.second.ptr);
      c->tables.clear();
      c->wave_tables.clear();
      c->f64_tables.clear();
      c->memo_dev = c->memo_devK = nullptr;
      c->memo.clear();
    }
  }
  std::lock_guard<std::mutex> lock;
  bool ok;
};

// staging of host signal buffers through scratch slots (convenience path, PCIe-bound)
struct Staged {
  Ctx* c;
  explicit Staged(Ctx* ctx) : c(ctx) {}
  // ---- round 6: pinned bounce slots.  A pageable hipMemcpy runs at 13-32 GB/s on this platform (the runtime stages it through its own
  // small pinned buffers, single-threaded); a DMA between HBM and PINNED host memory runs at the link's ~56 GB/s.  So transfers of
  // kPinMin bytes or more go in chunks of kPinChunk through two pinned slots per direction: while the DMA engine moves chunk k + 1,
  // kCopyThreads host threads copy chunk k between the slot and the caller's buffer (which also takes the first-touch page faults of a
  // freshly allocated result, in parallel).  NXSIG_HOST_PIPE=1 selects it (n >= 2: n copy threads).  Results are the same bytes either way.
  static constexpr size_t kPinChunk = (size_t)32 << 20, kPinMin = (size_t)8 << 20;
  static constexpr unsigned kCopyThreads = 8;
  int ensure_pins() {
    if (c->pin_bytes) return NXSIG_OK;
    for (int i = 0; i < 4; ++i) {
      NXSIG_HIP_TRY(hipHostMalloc(&c->pin[i], kPinChunk, hipHostMallocDefault));
      NXSIG_HIP_TRY(hipEventCreateWithFlags(&c->xfer_ev[i], hipEventDisableTiming));
    }
    NXSIG_HIP_TRY(hipEventCreateWithFlags(&c->xfer_ready, hipEventDisableTiming));
    NXSIG_HIP_TRY(hipStreamCreateWithFlags(&c->xfer_stream, hipStreamNonBlocking));
    c->pin_bytes = kPinChunk;
    return NXSIG_OK;
  }
  void parallel_memcpy(char* dst, const char* src, size_t bytes) const {
    unsigned hw = std::thread::hardware_concurrency();
    const int knob = tune(c, kT_HOST_PIPE, 0);   // >= 2: that many copy threads (sweeps)
    const unsigned T = knob >= 2 ? (unsigned)knob : (hw >= 16 ? kCopyThreads : (hw >= 4 ? 2 : 1));
    const size_t per = ((bytes / T) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) {
      const size_t o = (size_t)t * per;
      if (o < bytes) th.emplace_back([=] { std::memcpy(dst + o, src + o, bytes - o < per ? bytes - o : per); });
    }
    std::memcpy(dst, src, bytes < per ? bytes : per);
    for (auto& t : th) t.join();
  }
  // (default OFF: measured on 8 x config 2 — 92 MB up, 737 MB down, link floor 14.5 ms — the direct pageable copies with pre-faulting
  // below take 15.5 ms into a resident result buffer and 17.6 ms into a fresh one, this pipeline 16.3-16.5 / 16.9-17.5 ms: the runtime
  // pins resident user pages on the fly and DMAs straight into them, which the extra host copy cannot beat; profiles/r06/host_path.txt)
  bool piped(size_t bytes) const { return bytes >= kPinMin && tune(c, kT_HOST_PIPE, 0) != 0; }
  int in(int slot, const void* host, size_t bytes, const void** dev) {
    void* d = nullptr;
    int rc = ctx_scratch(c, slot, bytes ? bytes : 4, &d);
    if (rc) return rc;
    *dev = d;
    if (!piped(bytes)) {
      NXSIG_HIP_TRY(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, c->stream));
      return NXSIG_OK;
    }
    if ((rc = ensure_pins())) return rc;
    // the scratch slot may still be read by work queued on the compute stream: the transfer stream starts behind it
    NXSIG_HIP_TRY(hipEventRecord(c->xfer_ready, c->stream));
    NXSIG_HIP_TRY(hipStreamWaitEvent(c->xfer_stream, c->xfer_ready, 0));
    const char* h = static_cast<const char*>(host);
    char* dd = static_cast<char*>(d);
    int k = 0;
    for (size_t off = 0; off < bytes; off += kPinChunk, ++k) {
      const size_t len = bytes - off < kPinChunk ? bytes - off : kPinChunk;
      const int sl = 2 + (k & 1);
      if (k >= 2) NXSIG_HIP_TRY(hipEventSynchronize(c->xfer_ev[sl]));   // the DMA that last read this slot is done
      parallel_memcpy(static_cast<char*>(c->pin[sl]), h + off, len);
      NXSIG_HIP_TRY(hipMemcpyAsync(dd + off, c->pin[sl], len, hipMemcpyHostToDevice, c->xfer_stream));
      NXSIG_HIP_TRY(hipEventRecord(c->xfer_ev[sl], c->xfer_stream));
    }
    // the compute stream continues once the last chunk has landed
    NXSIG_HIP_TRY(hipEventRecord(c->xfer_ready, c->xfer_stream));
    NXSIG_HIP_TRY(hipStreamWaitEvent(c->stream, c->xfer_ready, 0));
    return NXSIG_OK;
  }
  int out_alloc(int slot, size_t bytes, void** dev) { return ctx_scratch(c, slot, bytes ? bytes : 4, dev); }
  // A pageable device-to-host copy runs at PCIe speed (56 GB/s measured) into RESIDENT pages but at 20-25 GB/s into a
  // freshly allocated result buffer (np.empty, enif_make_new_binary): first-touch page faults, taken one at a time
  // inside the copy.  So the result is copied in chunks, and while chunk k is on the wire a few threads pre-fault the
  // pages of chunk k+1 (MADV_POPULATE_WRITE, or a read-write touch that keeps the contents where madvise refuses).
  static void prefault_parallel(char* p, size_t bytes) {
    const uintptr_t page = 4096;
    uintptr_t a = (reinterpret_cast<uintptr_t>(p) + page - 1) & ~(page - 1);
    const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(page - 1);
    if (e <= a) return;
    unsigned hw = std::thread::hardware_concurrency();
    unsigned T = hw >= 8 ? 4 : 1;  // more threads contend on the process' mmap lock: 8 / 16 / 32 measured slower
    const size_t per = (((e - a) / T) + page - 1) & ~(size_t)(page - 1);
    auto work = [](uintptr_t s0, uintptr_t s1) {
      const uintptr_t page = 4096;
      if (s1 <= s0) return;
#ifdef __linux__
      if (madvise(reinterpret_cast<void*>(s0), s1 - s0, 23 /* MADV_POPULATE_WRITE */) == 0) return;
#endif
      for (uintptr_t q = s0; q < s1; q += page) { volatile char* b = reinterpret_cast<volatile char*>(q); *b = *b; }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) {
      const uintptr_t s0 = a + (uintptr_t)t * per, s1 = s0 + per < e ? s0 + per : e;
      if (s0 < e) th.emplace_back(work, s0, s1);
    }
    work(a, a + per < e ? a + per : e);
    for (auto& t : th) t.join();
  }
  int out_copy(void* host, const void* dev, size_t bytes) {
    if (piped(bytes)) return out_copy_piped(host, dev, bytes);
    const size_t CH = (size_t)32 << 20;
    if (bytes < ((size_t)32 << 20) || tune(c, kT_NO_PREFAULT, 0)) {
      NXSIG_HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
      NXSIG_HIP_TRY(hipStreamSynchronize(c->stream));
      return NXSIG_OK;
    }
    char* h = static_cast<char*>(host);
    const char* d = static_cast<const char*>(dev);
    prefault_parallel(h, bytes < CH ? bytes : CH);
    for (size_t off = 0; off < bytes; off += CH) {
      const size_t len = bytes - off < CH ? bytes - off : CH;
      std::thread pf;
      if (off + CH < bytes) {
        const size_t nlen = bytes - (off + CH) < CH ? bytes - (off + CH) : CH;
        pf = std::thread(prefault_parallel, h + off + CH, nlen);
      }
      hipError_t e1 = hipMemcpyAsync(h + off, d + off, len, hipMemcpyDeviceToHost, c->stream);
      hipError_t e2 = e1 == hipSuccess ? hipStreamSynchronize(c->stream) : e1;
      if (pf.joinable()) pf.join();
      NXSIG_HIP_TRY(e2);
    }
    return NXSIG_OK;
  }
  // device -> pinned slot (DMA, transfer stream) -> caller's buffer (host threads), two slots: DMA of chunk k + 1 beside the copy of chunk k
  int out_copy_piped(void* host, const void* dev, size_t bytes) {
    int rc = ensure_pins();
    if (rc) return rc;
    NXSIG_HIP_TRY(hipEventRecord(c->xfer_ready, c->stream));             // the kernels that produce the result
    NXSIG_HIP_TRY(hipStreamWaitEvent(c->xfer_stream, c->xfer_ready, 0));
    char* h = static_cast<char*>(host);
    const char* d = static_cast<const char*>(dev);
    const size_t n = (bytes + kPinChunk - 1) / kPinChunk;
    auto len_of = [&](size_t k) { const size_t off = k * kPinChunk; return bytes - off < kPinChunk ? bytes - off : kPinChunk; };
    NXSIG_HIP_TRY(hipMemcpyAsync(c->pin[0], d, len_of(0), hipMemcpyDeviceToHost, c->xfer_stream));
    NXSIG_HIP_TRY(hipEventRecord(c->xfer_ev[0], c->xfer_stream));
    for (size_t k = 0; k < n; ++k) {
      const int sl = (int)(k & 1);
      if (k + 1 < n) {   // slot (k + 1) & 1 was emptied by the copy of chunk k - 1 (sequential below)
        NXSIG_HIP_TRY(hipMemcpyAsync(c->pin[sl ^ 1], d + (k + 1) * kPinChunk, len_of(k + 1), hipMemcpyDeviceToHost, c->xfer_stream));
        NXSIG_HIP_TRY(hipEventRecord(c->xfer_ev[sl ^ 1], c->xfer_stream));
      }
      NXSIG_HIP_TRY(hipEventSynchronize(c->xfer_ev[sl]));
      parallel_memcpy(h + k * kPinChunk, static_cast<const char*>(c->pin[sl]), len_of(k));
    }
    // later work on the compute stream may overwrite the scratch slot the result came from: it is behind the last DMA already
    // (every DMA was waited for above); nothing else to order
    return NXSIG_OK;
  }
};

}  // namespace nxsig

using namespace nxsig;

#define NXSIG_API_BEGIN try {
#define NXSIG_API_END                                                                   \
  }                                                                                     \
  catch (const std::bad_alloc&) { return set_error(NXSIG_ERR_OOM, "host out of memory"); } \
  catch (const std::exception& e) { return set_error(NXSIG_ERR_INVALID_ARG, std::string("internal error: ") + e.what()); } \
  catch (...) { return set_error(NXSIG_ERR_INVALID_ARG, "internal error"); }

#define NXSIG_CHECK_CTX(ctx)                                                            \
  if (!(ctx)) return set_error(NXSIG_ERR_INVALID_ARG, "null context");                  \
  Ctx* c = reinterpret_cast<Ctx*>(ctx);                                                 \
  DeviceGuard guard(c);                                                                 \
  if (!guard.ok) return set_error(NXSIG_ERR_HIP, "hipSetDevice failed");

namespace nxsig {
static const char* const kTuneNames[kTuneCount] = {
#define NXSIG_X(n) "NXSIG_" #n,
    NXSIG_TUNABLES(NXSIG_X)
#undef NXSIG_X
};
const char* tuning_name(int key) { return key >= 0 && key < kTuneCount ? kTuneNames[key] : ""; }
int tuning_index(const char* name) {
  if (!name) return -1;
  for (int k = 0; k < kTuneCount; ++k)
    if (!std::strcmp(name, kTuneNames[k]) || !std::strcmp(name, kTuneNames[k] + 6)) return k;
  return -1;
}
// Admissible values of a switch.  The geometry knobs end up as divisors or loop counts in the launchers (ISTFT_RUNS_PER_CU = 0 was an
// integer division by zero in launch_istft_wave_R), so they are range-checked HERE, once, for both ways in — the environment and
// nxsig_ctx_set_tuning — instead of at every use site; the on / off switches take any non-negative value.
static bool tuning_in_range(int k, long v, long* lo, long* hi) {
  long a = 0, b = 1L << 30;
  switch (k) {
    case kT_ISTFT_RUNS_PER_CU: a = 1; b = 4096; break;         // resident waves per CU the run length is derived from
    case kT_ISTFT_MIN_RUN: a = 1; b = 1 << 20; break;
    case kT_WAVE_UNITS_PER_WAVE: a = 1; b = 4096; break;
    case kT_FIR_UNITS_PER_WAVE: a = 1; b = 4096; break;
    case kT_MEL_LDS_KB: a = 1; b = 160; break;               // LDS per CU on gfx950
    case kT_FFT_TILE_ELEMS: a = 1024; b = 16384; break;
    case kT_FFT_TILE_NT: a = 64; b = 1024; break;
    case kT_FFT_TILED_MIN: a = 2; break;
    case kT_STORE_POLICY: a = 0; b = 2; break;
    case kT_FIR_R2K: a = 0; b = 2; break;
    default: break;
  }
  if (lo) *lo = a;
  if (hi) *hi = b;
  return v >= a && v <= b;
}
// the library's ONE look at the process environment for its switches: at context creation, never in a launch.  A value that is not
// a number ("true", "on", "yes" / "false", "off", "no" aside, which mean 1 / 0) or lies outside the switch's range is IGNORED with
// one line on stderr — never silently read as 0.
void tuning_from_env(Tuning* t) {
  for (int k = 0; k < kTuneCount; ++k) {
    const char* v = std::getenv(kTuneNames[k]);
    if (!v || !*v) continue;
    char* end = nullptr;
    long val = std::strtol(v, &end, 10);
    while (end && (*end == ' ' || *end == '\t')) ++end;
    if (end == v || (end && *end)) {
      if (!strcasecmp(v, "true") || !strcasecmp(v, "on") || !strcasecmp(v, "yes")) val = 1;
      else if (!strcasecmp(v, "false") || !strcasecmp(v, "off") || !strcasecmp(v, "no")) val = 0;
      else { std::fprintf(stderr, "nxsig: %s=%s is not a number: ignored\n", kTuneNames[k], v); continue; }
    }
    long lo, hi;
    if (!tuning_in_range(k, val, &lo, &hi)) {
      std::fprintf(stderr, "nxsig: %s=%ld is outside [%ld, %ld]: ignored\n", kTuneNames[k], val, lo, hi);
      continue;
    }
    t->v[k] = (int32_t)val; t->set[k] = true;
  }
}
bool tuning_value_ok(int key, long value, long* lo, long* hi) { return tuning_in_range(key, value, lo, hi); }
}  // namespace nxsig

static int check_mem(int32_t mem) {
  if (mem != NXSIG_HOST && mem != NXSIG_DEVICE) return set_error(NXSIG_ERR_INVALID_ARG, "mem must be NXSIG_HOST or NXSIG_DEVICE");
  return NXSIG_OK;
}

extern "C" {

int nxsig_abi_version(void) { return NXSIG_ABI_VERSION; }

const char* nxsig_last_error(void) { return last_error_cstr(); }
const char* nxsig_last_dispatch(void) { return dispatch_cstr(); }

int nxsig_device_count(int* count) {
  NXSIG_API_BEGIN
  if (!count) return set_error(NXSIG_ERR_INVALID_ARG, "count is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return set_error(NXSIG_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  *count = n;
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_ctx_set_tuning(nxsig_ctx* ctx, const char* name, int32_t value) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  const int k = tuning_index(name);
  if (k < 0) return set_error(NXSIG_ERR_INVALID_ARG, std::string("no such switch: ") + (name ? name : "(null)"));
  long lo, hi;
  if (!tuning_value_ok(k, value, &lo, &hi))
    return set_error(NXSIG_ERR_INVALID_ARG, std::string(tuning_name(k)) + " must lie in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]");
  c->tuning.v[k] = value;   // (NXSIG_CHECK_CTX holds the context's mutex)
  c->tuning.set[k] = true;
  if (k == kT_POOL_MAX_MB) c->pool_cap = 0;  // decided again at the next nxsig_free
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_ctx_get_tuning(nxsig_ctx* ctx, const char* name, int32_t* value, int32_t* is_set) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  const int k = tuning_index(name);
  if (k < 0) return set_error(NXSIG_ERR_INVALID_ARG, std::string("no such switch: ") + (name ? name : "(null)"));
  if (value) *value = c->tuning.v[k];
  if (is_set) *is_set = c->tuning.set[k] ? 1 : 0;
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_ctx_clear_tuning(nxsig_ctx* ctx, const char* name) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  if (name == nullptr) { c->tuning = Tuning(); return NXSIG_OK; }
  const int k = tuning_index(name);
  if (k < 0) return set_error(NXSIG_ERR_INVALID_ARG, std::string("no such switch: ") + name);
  c->tuning.set[k] = false;
  c->tuning.v[k] = 0;
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_ctx_create(int device, nxsig_ctx** out) {
  NXSIG_API_BEGIN
  if (!out) return set_error(NXSIG_ERR_INVALID_ARG, "out is null");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return set_error(NXSIG_ERR_NO_DEVICE, "no ROCm-capable device: the nxsig hot path has no CPU fallback");
  if (device < 0 || device >= n) return set_error(NXSIG_ERR_INVALID_ARG, "device index out of range");
  NXSIG_HIP_TRY(hipSetDevice(device));
  Ctx* c = new Ctx();
  c->device = device;
  hipDeviceProp_t prop;
  NXSIG_HIP_TRY(hipGetDeviceProperties(&prop, device));
  c->num_cus = prop.multiProcessorCount;
  tuning_from_env(&c->tuning);
  c->dev_name = std::string(prop.name) + " " + prop.gcnArchName + " " + std::to_string(prop.multiProcessorCount) + " CUs";
  NXSIG_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  NXSIG_HIP_TRY(hipEventCreate(&c->ev_start));
  NXSIG_HIP_TRY(hipEventCreate(&c->ev_stop));
  *out = reinterpret_cast<nxsig_ctx*>(c);
  return NXSIG_OK;
  NXSIG_API_END
}

void nxsig_ctx_destroy(nxsig_ctx* ctx) {
  if (!ctx) return;
  Ctx* c = reinterpret_cast<Ctx*>(ctx);
  try {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->twiddles) (void)hipFree(kv.second.ptr);
    for (auto& kv : c->tables) (void)hipFree(kv.second.ptr);
    for (auto& s : c->scratch) if (s) (void)hipFree(s);
    for (auto& kv : c->pool_free) (void)hipFree(kv.second);
    for (auto& kv : c->pool_live) (void)hipFree(kv.first);   // blocks the caller never returned die with their context
    (void)hipEventDestroy(c->ev_start);
    (void)hipEventDestroy(c->ev_stop);
    for (auto e : c->lap_events) (void)hipEventDestroy(e);
    for (auto& pp : c->pin) if (pp) (void)hipHostFree(pp);
    for (auto& e : c->xfer_ev) if (e) (void)hipEventDestroy(e);
    if (c->xfer_ready) (void)hipEventDestroy(c->xfer_ready);
    if (c->xfer_stream) (void)hipStreamDestroy(c->xfer_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
  } catch (...) {
  }
  delete c;
}

int nxsig_ctx_last_dispatch(nxsig_ctx* ctx, char* buf, size_t buflen) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  if (!buf || buflen == 0) return set_error(NXSIG_ERR_INVALID_ARG, "last_dispatch: buf is null");
  std::snprintf(buf, buflen, "%s", c->last_dispatch.c_str());
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_device_name(nxsig_ctx* ctx, char* buf, size_t buflen) {
  NXSIG_API_BEGIN
  if (!ctx || !buf || buflen == 0) return set_error(NXSIG_ERR_INVALID_ARG, "bad arguments");
  Ctx* c = reinterpret_cast<Ctx*>(ctx);
  std::strncpy(buf, c->dev_name.c_str(), buflen - 1);
  buf[buflen - 1] = 0;
  return NXSIG_OK;
  NXSIG_API_END
}

// ---- caching allocator (see Ctx::pool_free).  Blocks are rounded up to 2 MiB (256 B below 1 MiB) so that results of nearly equal
// sizes share blocks; a cached block serves a request when it is at most 12.5 % + 2 MiB larger.
static size_t pool_round(size_t bytes) {
  if (bytes < 4) bytes = 4;
  const size_t g = bytes < ((size_t)1 << 20) ? 256 : ((size_t)2 << 20);
  return (bytes + g - 1) / g * g;
}
static void pool_trim(Ctx* c, size_t keep) {   // releases cached blocks, largest first, until at most `keep` bytes stay cached
  if (c->pool_cached <= keep) return;
  (void)hipStreamSynchronize(c->stream);
  while (c->pool_cached > keep && !c->pool_free.empty()) {
    auto it = std::prev(c->pool_free.end());
    (void)hipFree(it->second);
    c->pool_cached -= it->first;
    c->pool_free.erase(it);
  }
}

int nxsig_alloc(nxsig_ctx* ctx, size_t bytes, void** dptr) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  if (!dptr) return set_error(NXSIG_ERR_INVALID_ARG, "dptr is null");
  const size_t need = pool_round(bytes);
  auto it = c->pool_free.lower_bound(need);
  if (it != c->pool_free.end() && it->first <= need + need / 8 + ((size_t)2 << 20)) {
    *dptr = it->second;
    c->pool_live[it->second] = it->first;
    c->pool_cached -= it->first;
    c->pool_free.erase(it);
    return NXSIG_OK;
  }
  hipError_t e = hipMalloc(dptr, need);
  if (e != hipSuccess && !c->pool_free.empty()) {  // out of memory with blocks parked in the cache: give them back and retry
    (void)hipGetLastError();
    pool_trim(c, 0);
    e = hipMalloc(dptr, need);
  }
  NXSIG_HIP_TRY(e);
  c->pool_live[*dptr] = need;
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_free(nxsig_ctx* ctx, void* dptr) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  if (!dptr) return NXSIG_OK;
  auto it = c->pool_live.find(dptr);
  if (it == c->pool_live.end()) {  // not one of ours (or already released): plain free, after the stream has drained
    NXSIG_HIP_TRY(hipStreamSynchronize(c->stream));
    NXSIG_HIP_TRY(hipFree(dptr));
    return NXSIG_OK;
  }
  if (c->pool_cap == 0) {  // NXSIG_POOL_MAX_MB (0 disables caching); default: a quarter of the device memory
    size_t free_b = 0, total_b = 0;
    c->pool_cap = 1;
    if (c->tuning.set[kT_POOL_MAX_MB]) c->pool_cap = (size_t)(c->tuning.v[kT_POOL_MAX_MB] < 0 ? 0 : c->tuning.v[kT_POOL_MAX_MB]) << 20 | 1;
    else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) c->pool_cap = total_b / 4 | 1;
  }
  const size_t sz = it->second;
  c->pool_live.erase(it);
  if (sz > c->pool_cap) {  // never cacheable
    NXSIG_HIP_TRY(hipStreamSynchronize(c->stream));
    NXSIG_HIP_TRY(hipFree(dptr));
    return NXSIG_OK;
  }
  c->pool_free.emplace(sz, dptr);
  c->pool_cached += sz;
  pool_trim(c, c->pool_cap);
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_upload(nxsig_ctx* ctx, void* dst_device, const void* src_host, size_t bytes) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  NXSIG_HIP_TRY(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, c->stream));
  NXSIG_HIP_TRY(hipStreamSynchronize(c->stream));
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_download(nxsig_ctx* ctx, void* dst_host, const void* src_device, size_t bytes) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  if (!dst_host || !src_device) return set_error(NXSIG_ERR_INVALID_ARG, "download: null pointer");
  Staged st(c);  // large results: chunked copy with the next chunk's pages pre-faulted (freshly allocated destinations)
  return st.out_copy(dst_host, src_device, bytes);
  NXSIG_API_END
}

int nxsig_sync(nxsig_ctx* ctx) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  NXSIG_HIP_TRY(hipStreamSynchronize(c->stream));
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_mem_info(nxsig_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  size_t f = 0, t = 0;
  NXSIG_HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_set_stream(nxsig_ctx* ctx, void* hip_stream) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  NXSIG_HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->own_stream) NXSIG_HIP_TRY(hipStreamDestroy(c->stream));
  if (hip_stream) {
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
  } else {
    NXSIG_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  return NXSIG_OK;
  NXSIG_API_END
}

void* nxsig_get_stream(nxsig_ctx* ctx) { return ctx ? reinterpret_cast<Ctx*>(ctx)->stream : nullptr; }

int nxsig_timer_start(nxsig_ctx* ctx) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  NXSIG_HIP_TRY(hipEventRecord(c->ev_start, c->stream));
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_timer_stop(nxsig_ctx* ctx, float* elapsed_ms) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  if (!elapsed_ms) return set_error(NXSIG_ERR_INVALID_ARG, "elapsed_ms is null");
  NXSIG_HIP_TRY(hipEventRecord(c->ev_stop, c->stream));
  NXSIG_HIP_TRY(hipEventSynchronize(c->ev_stop));
  NXSIG_HIP_TRY(hipEventElapsedTime(elapsed_ms, c->ev_start, c->ev_stop));
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_timer_lap(nxsig_ctx* ctx) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  if (c->laps_used >= 4096) return set_error(NXSIG_ERR_INVALID_ARG, "timer_lap: at most 4096 laps per series");
  if (c->laps_used == c->lap_events.size()) {
    hipEvent_t e;
    NXSIG_HIP_TRY(hipEventCreate(&e));
    c->lap_events.push_back(e);
  }
  NXSIG_HIP_TRY(hipEventRecord(c->lap_events[c->laps_used], c->stream));
  ++c->laps_used;
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_timer_laps(nxsig_ctx* ctx, float* intervals_ms, int32_t capacity, int32_t* count) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  if (!count || (capacity > 0 && !intervals_ms)) return set_error(NXSIG_ERR_INVALID_ARG, "timer_laps: null output");
  const size_t n = c->laps_used;
  c->laps_used = 0;
  *count = 0;
  if (n == 0) return NXSIG_OK;
  NXSIG_HIP_TRY(hipEventSynchronize(c->lap_events[n - 1]));
  for (size_t i = 0; i + 1 < n && (int32_t)i < capacity; ++i) {
    NXSIG_HIP_TRY(hipEventElapsedTime(&intervals_ms[i], c->lap_events[i], c->lap_events[i + 1]));
    *count = (int32_t)i + 1;
  }
  return NXSIG_OK;
  NXSIG_API_END
}

/* ---------------------------------------------------------------- shape helpers */
int32_t nxsig_next_pow2(int32_t n) {
  int32_t p = 1;
  while (p < n && p < (1 << 30)) p <<= 1;
  return p;
}

int64_t nxsig_num_frames(int64_t length, int32_t frame_length, int32_t hop, int32_t pad_mode, int64_t pad_lo,
                         int64_t pad_hi) {
  Framing f;
  int rc = make_framing(length, frame_length, hop, pad_mode, pad_lo, pad_hi, &f);
  return rc ? (int64_t)rc : f.M;
}

int64_t nxsig_ola_length(int64_t num_frames, int32_t frame_length, int32_t hop) {
  if (num_frames < 0 || frame_length < 1 || hop < 1 || hop > frame_length)
    return set_error(NXSIG_ERR_INVALID_ARG, "ola_length: need num_frames >= 0 and 1 <= hop <= frame_length");
  return num_frames * hop + (frame_length - hop);
}

int64_t nxsig_conv_length(int64_t n1, int64_t n2, int32_t mode) {
  if (n1 < 1 || n2 < 1) return set_error(NXSIG_ERR_INVALID_ARG, "conv_length: lengths must be >= 1");
  switch (mode) {
    case NXSIG_CONV_FULL: return n1 + n2 - 1;
    case NXSIG_CONV_SAME: return n1;
    case NXSIG_CONV_VALID: return (n1 >= n2 ? n1 - n2 : n2 - n1) + 1;
    default: return set_error(NXSIG_ERR_INVALID_ARG, "expected mode to be one of [:full, :same, :valid]");
  }
}

/* ---------------------------------------------------------------- host generators */
int nxsig_window_f32(int32_t kind, int32_t n, int32_t is_periodic, double beta, double eps, float* out) {
  NXSIG_API_BEGIN
  return window_f32(kind, n, is_periodic != 0, beta, eps, out);
  NXSIG_API_END
}

int nxsig_sinc_f32(const float* t, int64_t n, float* out) {
  NXSIG_API_BEGIN
  if (n < 0 || (n > 0 && (!t || !out))) return set_error(NXSIG_ERR_INVALID_ARG, "sinc: bad arguments");
  sinc_f32(t, n, out);
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_firwin_f32(int32_t num_taps, const double* cutoff, int32_t n_cutoff, int32_t window_kind, double kaiser_beta,
                     int32_t pass_zero, int32_t scale, double sampling_rate, float* out) {
  NXSIG_API_BEGIN
  return firwin_f32(num_taps, cutoff, n_cutoff, window_kind, kaiser_beta, pass_zero != 0, scale != 0, sampling_rate, out);
  NXSIG_API_END
}

int nxsig_fft_frequencies_f32(double sampling_rate, int32_t fft_length, int32_t endpoint, float* out) {
  NXSIG_API_BEGIN
  if (fft_length < 1 || !out) return set_error(NXSIG_ERR_INVALID_ARG, "fft_frequencies: fft_length must be >= 1");
  fft_frequencies_f32(sampling_rate, fft_length, endpoint != 0, out);
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_stft_times_f32(int32_t frame_length, double sampling_rate, int64_t num_frames, float* out) {
  NXSIG_API_BEGIN
  if (num_frames < 0 || (num_frames > 0 && !out)) return set_error(NXSIG_ERR_INVALID_ARG, "stft_times: bad arguments");
  stft_times_f32(frame_length, sampling_rate, num_frames, out);
  return NXSIG_OK;
  NXSIG_API_END
}

/* ---------------------------------------------------------------- hot path */
static int check_scaling(int32_t s) {
  if (s != NXSIG_SCALE_NONE && s != NXSIG_SCALE_SPECTRUM && s != NXSIG_SCALE_PSD)  // lib/nx_signal.ex:124-126, :622-624
    return set_error(NXSIG_ERR_INVALID_ARG, "invalid :scaling, expected one of :spectrum, :psd or nil");
  return NXSIG_OK;
}

int nxsig_stft_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride,
                   const float* window, const nxsig_stft_params* p, nxsig_c64* z, int64_t* num_frames_out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!x || !window || !p || !z) return set_error(NXSIG_ERR_INVALID_ARG, "stft: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft: batch must be >= 1");   // (more than 65 504 rows: slabs, see launch_stft)
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "stft: batch_stride < length");
  if (p->fft_length < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft: fft_length must be >= 1");
  rc = check_scaling(p->scaling);
  if (rc) return rc;
  Framing fr;
  rc = make_framing(length, p->frame_length, p->hop, p->pad_mode, p->pad_lo, p->pad_hi, &fr);
  if (rc) return rc;
  if (num_frames_out) *num_frames_out = fr.M;

  StftLaunch a;
  a.fr = fr; a.batch = batch; a.batch_stride = batch_stride; a.K = p->fft_length;
  a.has_scale = p->scaling != NXSIG_SCALE_NONE;
  a.inv_scale_div = a.has_scale ? scaling_factor(window, p->frame_length, p->scaling, p->sampling_rate) : 1.0f;
  rc = ctx_window(c, window, p->frame_length, p->fft_length, &a.window, &a.window_padK);
  if (rc) return rc;
  const size_t zbytes = (size_t)batch * fr.M * p->fft_length * sizeof(float2);
  if (mem == NXSIG_DEVICE) {
    a.x = x; a.z = reinterpret_cast<float2*>(z);
    return launch_stft(c, a);
  }
  Staged st(c);
  const void* xd = nullptr; void* zd = nullptr;
  const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(float);
  if ((rc = st.in(1, x, xbytes, &xd))) return rc;
  if ((rc = st.out_alloc(2, zbytes, &zd))) return rc;
  a.x = reinterpret_cast<const float*>(xd); a.z = reinterpret_cast<float2*>(zd);
  if ((rc = launch_stft(c, a))) return rc;
  return st.out_copy(z, zd, zbytes);
  NXSIG_API_END
}

int nxsig_stft_c64(nxsig_ctx* ctx, const nxsig_c64* x, int64_t length, int32_t batch, int64_t batch_stride,
                   const float* window, const nxsig_stft_params* p, nxsig_c64* z, int64_t* num_frames_out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!x || !window || !p || !z) return set_error(NXSIG_ERR_INVALID_ARG, "stft: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft: batch must be >= 1");   // (more than 65 504 rows: slabs, see launch_stft_c64)
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "stft: batch_stride < length");
  if (p->fft_length < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft: fft_length must be >= 1");
  rc = check_scaling(p->scaling);
  if (rc) return rc;
  Framing fr;
  rc = make_framing(length, p->frame_length, p->hop, p->pad_mode, p->pad_lo, p->pad_hi, &fr);
  if (rc) return rc;
  if (num_frames_out) *num_frames_out = fr.M;
  StftLaunch a;
  a.fr = fr; a.batch = batch; a.batch_stride = batch_stride; a.K = p->fft_length;
  a.has_scale = p->scaling != NXSIG_SCALE_NONE;
  a.inv_scale_div = a.has_scale ? scaling_factor(window, p->frame_length, p->scaling, p->sampling_rate) : 1.0f;
  rc = ctx_window(c, window, p->frame_length, p->fft_length, &a.window, &a.window_padK);
  if (rc) return rc;
  const size_t zbytes = (size_t)batch * fr.M * p->fft_length * sizeof(float2);
  if (mem == NXSIG_DEVICE) {
    a.x = reinterpret_cast<const float*>(x); a.z = reinterpret_cast<float2*>(z);
    return launch_stft_c64(c, a);
  }
  Staged st(c);
  const void* xd = nullptr; void* zd = nullptr;
  const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(float2);
  if ((rc = st.in(1, x, xbytes, &xd))) return rc;
  if ((rc = st.out_alloc(2, zbytes, &zd))) return rc;
  a.x = reinterpret_cast<const float*>(xd); a.z = reinterpret_cast<float2*>(zd);
  if ((rc = launch_stft_c64(c, a))) return rc;
  return st.out_copy(z, zd, zbytes);
  NXSIG_API_END
}

static int istft_common(nxsig_ctx* ctx, const nxsig_c64* z, int64_t num_frames, int32_t batch, const float* window,
                        const nxsig_stft_params* p, const nxsig_c64* h, nxsig_c64* y, int32_t mem, bool onesided = false) {
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!z || !window || !p || !y) return set_error(NXSIG_ERR_INVALID_ARG, "istft: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1 || (batch > 65535 && (onesided || h)))
    return set_error(NXSIG_ERR_INVALID_ARG, "istft: batch must be >= 1 (and <= 65535 for the packed and the filtered forms)");
  if (num_frames < 1) return set_error(NXSIG_ERR_INVALID_ARG, "istft: num_frames must be >= 1");
  rc = check_scaling(p->scaling);
  if (rc) return rc;
  const int N = p->frame_length, hop = p->hop, K = p->fft_length;
  if (N < 1 || hop < 1) return set_error(NXSIG_ERR_INVALID_ARG, "istft: frame_length and hop must be >= 1");
  if (hop > N)  // overlap_length < 0 cannot be expressed; overlap >= N -> hop <= 0 handled above (lib/nx_signal.ex:692-695)
    return set_error(NXSIG_ERR_INVALID_ARG, "overlap_length must be a number less than the window size");
  if (K != N)
    return set_error(NXSIG_ERR_INVALID_ARG,
                     "istft: fft_length must equal the window length (the reference broadcasts {M,K} x {N}, lib/nx_signal.ex:628)");
  if (onesided && (K & 1)) return set_error(NXSIG_ERR_INVALID_ARG, "istft_packed: fft_length must be even");
  IstftLaunch a;
  a.M = num_frames; a.batch = batch; a.N = N; a.hop = hop; a.K = K; a.onesided = onesided;
  a.has_scale = p->scaling != NXSIG_SCALE_NONE;
  a.scale_mul = a.has_scale ? scaling_factor(window, N, p->scaling, p->sampling_rate) : 1.0f;
  const void* wdev = nullptr;
  rc = ctx_table(c, 0x57494Eull, window, (size_t)N * sizeof(float), &wdev);
  if (rc) return rc;
  a.window = reinterpret_cast<const float*>(wdev);
  if (h) {  // same table key as spectrum_mul: the two-step and the fused form share the filter's device copy
    const void* hd = nullptr;
    if ((rc = ctx_table(c, 0x5BEC0ull ^ (uint64_t)K, h, (size_t)K * sizeof(float2), &hd))) return rc;
    a.filt = reinterpret_cast<const float2*>(hd);
  }
  const int64_t out_len = num_frames * hop + (N - hop);
  const size_t zbytes = (size_t)batch * num_frames * (onesided ? K / 2 : K) * sizeof(float2);
  const size_t ybytes = (size_t)batch * out_len * (onesided ? sizeof(float) : sizeof(float2));
  if (mem == NXSIG_DEVICE) {
    a.z = reinterpret_cast<const float2*>(z); a.y = reinterpret_cast<float2*>(y);
    return launch_istft(c, a, window);
  }
  Staged st(c);
  const void* zd = nullptr; void* yd = nullptr;
  if ((rc = st.in(1, z, zbytes, &zd))) return rc;
  if ((rc = st.out_alloc(2, ybytes, &yd))) return rc;
  a.z = reinterpret_cast<const float2*>(zd); a.y = reinterpret_cast<float2*>(yd);
  if ((rc = launch_istft(c, a, window))) return rc;
  return st.out_copy(y, yd, ybytes);
}

int nxsig_istft_c64(nxsig_ctx* ctx, const nxsig_c64* z, int64_t num_frames, int32_t batch, const float* window,
                    const nxsig_stft_params* p, nxsig_c64* y, int32_t mem) {
  NXSIG_API_BEGIN
  return istft_common(ctx, z, num_frames, batch, window, p, nullptr, y, mem);
  NXSIG_API_END
}

int nxsig_istft_packed_f32(nxsig_ctx* ctx, const nxsig_c64* z, int64_t num_frames, int32_t batch, const float* window,
                           const nxsig_stft_params* p, float* y, int32_t mem) {
  NXSIG_API_BEGIN
  return istft_common(ctx, z, num_frames, batch, window, p, nullptr, reinterpret_cast<nxsig_c64*>(y), mem, true);
  NXSIG_API_END
}

int nxsig_istft_filtered_c64(nxsig_ctx* ctx, const nxsig_c64* z, int64_t num_frames, int32_t batch, const float* window,
                             const nxsig_stft_params* p, const nxsig_c64* h, nxsig_c64* y, int32_t mem) {
  NXSIG_API_BEGIN
  if (!h) return set_error(NXSIG_ERR_INVALID_ARG, "istft_filtered: null filter spectrum");
  return istft_common(ctx, z, num_frames, batch, window, p, h, y, mem);
  NXSIG_API_END
}

int nxsig_as_windowed_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride,
                          int32_t window_length, int32_t stride, int32_t pad_mode, int64_t pad_lo, int64_t pad_hi,
                          float* out, int64_t* num_frames_out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!x || !out) return set_error(NXSIG_ERR_INVALID_ARG, "as_windowed: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1 || batch > 65535) return set_error(NXSIG_ERR_INVALID_ARG, "as_windowed: batch must be in [1, 65535]");
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "as_windowed: batch_stride < length");
  Framing fr;
  rc = make_framing(length, window_length, stride, pad_mode, pad_lo, pad_hi, &fr);
  if (rc) return rc;
  if (num_frames_out) *num_frames_out = fr.M;
  const size_t obytes = (size_t)batch * fr.M * fr.N * sizeof(float);
  if (mem == NXSIG_DEVICE) return launch_as_windowed(c, x, batch_stride, batch, fr, out);
  Staged st(c);
  const void* xd = nullptr; void* od = nullptr;
  const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(float);
  if ((rc = st.in(1, x, xbytes, &xd))) return rc;
  if ((rc = st.out_alloc(2, obytes, &od))) return rc;
  if ((rc = launch_as_windowed(c, reinterpret_cast<const float*>(xd), batch_stride, batch, fr, reinterpret_cast<float*>(od)))) return rc;
  return st.out_copy(out, od, obytes);
  NXSIG_API_END
}

int nxsig_overlap_and_add(nxsig_ctx* ctx, const float* frames, int64_t num_frames, int32_t batch, int32_t frame_length,
                          int32_t overlap_length, int32_t components, float* out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!frames || !out) return set_error(NXSIG_ERR_INVALID_ARG, "overlap_and_add: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (components != 1 && components != 2) return set_error(NXSIG_ERR_INVALID_ARG, "overlap_and_add: components must be 1 (f32) or 2 (c64)");
  if (batch < 1 || batch > 65535 || num_frames < 1 || frame_length < 1)
    return set_error(NXSIG_ERR_INVALID_ARG, "overlap_and_add: batch, num_frames and frame_length must be >= 1");
  if (overlap_length >= frame_length)  // lib/nx_signal.ex:692-695 (message prints the window size twice, quirk B10)
    return set_error(NXSIG_ERR_INVALID_ARG, "overlap_length must be a number less than the window size " +
                                                std::to_string(frame_length) + ", got: " + std::to_string(frame_length));
  if (overlap_length < 0) return set_error(NXSIG_ERR_INVALID_ARG, "overlap_and_add: overlap_length must be >= 0");
  const int hop = frame_length - overlap_length;
  const int64_t out_len = num_frames * hop + overlap_length;
  const size_t ibytes = (size_t)batch * num_frames * frame_length * components * sizeof(float);
  const size_t obytes = (size_t)batch * out_len * components * sizeof(float);
  if (mem == NXSIG_DEVICE) return launch_overlap_and_add(c, frames, num_frames, batch, frame_length, hop, components, out);
  Staged st(c);
  const void* fd = nullptr; void* od = nullptr;
  if ((rc = st.in(1, frames, ibytes, &fd))) return rc;
  if ((rc = st.out_alloc(2, obytes, &od))) return rc;
  if ((rc = launch_overlap_and_add(c, reinterpret_cast<const float*>(fd), num_frames, batch, frame_length, hop, components,
                                   reinterpret_cast<float*>(od)))) return rc;
  return st.out_copy(out, od, obytes);
  NXSIG_API_END
}

int nxsig_fft(nxsig_ctx* ctx, const void* in, int32_t in_is_real, int64_t rows, int32_t n_in, int32_t fft_length,
              int32_t inverse, nxsig_c64* out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!in || !out) return set_error(NXSIG_ERR_INVALID_ARG, "fft: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (rows < 1 || n_in < 1 || fft_length < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fft: rows, n_in and fft_length must be >= 1");
  const size_t ibytes = (size_t)rows * n_in * (in_is_real ? sizeof(float) : sizeof(float2));
  const size_t obytes = (size_t)rows * fft_length * sizeof(float2);
  if (mem == NXSIG_DEVICE) return launch_fft(c, in, in_is_real != 0, rows, n_in, fft_length, inverse != 0, reinterpret_cast<float2*>(out));
  Staged st(c);
  const void* id = nullptr; void* od = nullptr;
  if ((rc = st.in(1, in, ibytes, &id))) return rc;
  if ((rc = st.out_alloc(2, obytes, &od))) return rc;
  if ((rc = launch_fft(c, id, in_is_real != 0, rows, n_in, fft_length, inverse != 0, reinterpret_cast<float2*>(od)))) return rc;
  return st.out_copy(out, od, obytes);
  NXSIG_API_END
}

static int fir_common(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* h,
                      int32_t num_taps, int64_t start, int64_t out_len, float* y, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!x || !h || !y) return set_error(NXSIG_ERR_INVALID_ARG, "fir: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1 || length < 1 || num_taps < 1)   // (more than 65 504 rows: slabs, see launch_fir)
    return set_error(NXSIG_ERR_INVALID_ARG, "fir: batch, length and num_taps must be >= 1");
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "fir: batch_stride < length");
  if (start < 0 || out_len < 1 || start + out_len > length + num_taps - 1)
    return set_error(NXSIG_ERR_INVALID_ARG, "fir: requested slice lies outside the full convolution");
  FirLaunch a;
  a.L = length; a.batch = batch; a.batch_stride = batch_stride; a.h_host = h; a.taps = num_taps;
  a.out_start = start; a.out_len = out_len;
  const size_t ybytes = (size_t)batch * out_len * sizeof(float);
  if (mem == NXSIG_DEVICE) {
    a.x = x; a.y = y;
    return launch_fir(c, a);
  }
  Staged st(c);
  const void* xd = nullptr; void* yd = nullptr;
  const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(float);
  if ((rc = st.in(1, x, xbytes, &xd))) return rc;
  if ((rc = st.out_alloc(2, ybytes, &yd))) return rc;
  a.x = reinterpret_cast<const float*>(xd); a.y = reinterpret_cast<float*>(yd);
  if ((rc = launch_fir(c, a))) return rc;
  return st.out_copy(y, yd, ybytes);
  NXSIG_API_END
}

int nxsig_fir_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* h,
                  int32_t num_taps, int32_t mode, float* y, int32_t mem) {
  if (length < 1 || num_taps < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fir: batch, length and num_taps must be >= 1");
  const int64_t full = length + num_taps - 1;
  int64_t out_len, start;
  switch (mode) {  // lib/nx_signal/convolution.ex:300-329: centered(out, shape) starts at div(full - new, 2)
    case NXSIG_CONV_FULL: out_len = full; start = 0; break;
    case NXSIG_CONV_SAME: out_len = length; start = (full - out_len) / 2; break;
    case NXSIG_CONV_VALID:
      out_len = (length >= num_taps ? length - num_taps : num_taps - length) + 1;
      start = (full - out_len) / 2;
      break;
    default:  // convolution.ex:41-44
      return set_error(NXSIG_ERR_INVALID_ARG, "expected mode to be one of [:full, :same, :valid]");
  }
  return fir_common(ctx, x, length, batch, batch_stride, h, num_taps, start, out_len, y, mem);
}

int nxsig_fir_slice_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* h,
                        int32_t num_taps, int64_t out_start, int64_t out_len, float* y, int32_t mem) {
  return fir_common(ctx, x, length, batch, batch_stride, h, num_taps, out_start, out_len, y, mem);
}

int nxsig_fft_nd(nxsig_ctx* ctx, const void* in, int32_t in_is_real, const int64_t* shape, int32_t rank, const int32_t* axes,
                 const int64_t* lengths, int32_t n_axes, int32_t inverse, nxsig_c64* out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!in || !out || !shape || (n_axes > 0 && (!axes || !lengths))) return set_error(NXSIG_ERR_INVALID_ARG, "fft_nd: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (rank < 1 || rank > 8 || n_axes < 0 || n_axes > 16) return set_error(NXSIG_ERR_INVALID_ARG, "fft_nd: rank must be in [1, 8]");
  std::vector<int64_t> osh(shape, shape + rank);
  int64_t n_in = 1, n_out = 1;
  for (int d = 0; d < rank; ++d) { if (shape[d] < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fft_nd: empty dimension"); n_in *= shape[d]; }
  for (int i = 0; i < n_axes; ++i) {
    const int ax = axes[i] < 0 ? axes[i] + rank : axes[i];
    if (ax < 0 || ax >= rank) return set_error(NXSIG_ERR_INVALID_ARG, "fft_nd: axis out of bounds");
    if (lengths[i] < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fft_nd: lengths must be positive");
    osh[ax] = lengths[i];
  }
  for (auto v : osh) n_out *= v;
  if (mem == NXSIG_DEVICE) return launch_fft_nd(c, in, in_is_real != 0, shape, rank, axes, lengths, n_axes, inverse != 0, reinterpret_cast<float2*>(out));
  void *di = nullptr, *dout = nullptr;
  const size_t ibytes = (size_t)n_in * (in_is_real ? sizeof(float) : sizeof(float2)), obytes = (size_t)n_out * sizeof(float2);
  if ((rc = ctx_scratch(c, 17, ibytes, &di))) return rc;
  if ((rc = ctx_scratch(c, 18, obytes, &dout))) return rc;
  NXSIG_HIP_TRY(hipMemcpyAsync(di, in, ibytes, hipMemcpyHostToDevice, c->stream));
  if ((rc = launch_fft_nd(c, di, in_is_real != 0, shape, rank, axes, lengths, n_axes, inverse != 0, reinterpret_cast<float2*>(dout)))) return rc;
  Staged st(c);
  return st.out_copy(out, dout, obytes);
  NXSIG_API_END
}

int nxsig_fftconvolve_nd(nxsig_ctx* ctx, const void* a, int32_t a_is_real, const int64_t* a_shape, const void* b, int32_t b_is_real,
                         const int64_t* b_shape, int32_t rank, int32_t mode, void* out, int64_t* out_shape, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!a || !b || !out || !a_shape || !b_shape) return set_error(NXSIG_ERR_INVALID_ARG, "fftconvolve: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (rank < 1 || rank > 8) return set_error(NXSIG_ERR_INVALID_ARG, "fftconvolve: rank must be in [1, 8]");
  if (mem == NXSIG_DEVICE) return launch_fftconvolve_nd(c, a, a_is_real != 0, a_shape, b, b_is_real != 0, b_shape, rank, mode, out, out_shape);
  int64_t na = 1, nb = 1, no = 1, osh[8];
  for (int d = 0; d < rank; ++d) {
    if (a_shape[d] < 1 || b_shape[d] < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fftconvolve: empty dimension");
    na *= a_shape[d]; nb *= b_shape[d]; no *= a_shape[d] + b_shape[d] - 1;  // upper bound of every mode's result
  }
  const bool real_out = a_is_real && b_is_real;
  void *da = nullptr, *db = nullptr, *dout = nullptr;
  const size_t abytes = (size_t)na * (a_is_real ? 4 : 8), bbytes = (size_t)nb * (b_is_real ? 4 : 8);
  if ((rc = ctx_scratch(c, 17, abytes, &da))) return rc;
  if ((rc = ctx_scratch(c, 18, bbytes, &db))) return rc;
  if ((rc = ctx_scratch(c, 19, (size_t)no * (real_out ? 4 : 8), &dout))) return rc;
  NXSIG_HIP_TRY(hipMemcpyAsync(da, a, abytes, hipMemcpyHostToDevice, c->stream));
  NXSIG_HIP_TRY(hipMemcpyAsync(db, b, bbytes, hipMemcpyHostToDevice, c->stream));
  if ((rc = launch_fftconvolve_nd(c, da, a_is_real != 0, a_shape, db, b_is_real != 0, b_shape, rank, mode, dout, osh))) return rc;
  int64_t nres = 1;
  for (int d = 0; d < rank; ++d) { nres *= osh[d]; if (out_shape) out_shape[d] = osh[d]; }
  Staged st(c);
  return st.out_copy(out, dout, (size_t)nres * (real_out ? 4 : 8));
  NXSIG_API_END
}

int nxsig_convolve_direct(nxsig_ctx* ctx, const void* a, int32_t a_is_real, const int64_t* a_shape, const void* b, int32_t b_is_real,
                          const int64_t* b_shape, int32_t rank, int32_t mode, void* out, int64_t* out_shape, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!a || !b || !out || !a_shape || !b_shape) return set_error(NXSIG_ERR_INVALID_ARG, "convolve: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (rank < 1 || rank > 8) return set_error(NXSIG_ERR_INVALID_ARG, "convolve: rank must be in [1, 8]");
  if (mem == NXSIG_DEVICE) return launch_convolve_direct(c, a, a_is_real != 0, a_shape, b, b_is_real != 0, b_shape, rank, mode, out, out_shape);
  int64_t na = 1, nb = 1, no = 1, osh[8];
  for (int d = 0; d < rank; ++d) {
    if (a_shape[d] < 1 || b_shape[d] < 1) return set_error(NXSIG_ERR_INVALID_ARG, "convolve: empty dimension");
    na *= a_shape[d]; nb *= b_shape[d]; no *= a_shape[d] + b_shape[d] - 1;  // upper bound of every mode's result
  }
  const bool real_out = a_is_real && b_is_real;
  void *da = nullptr, *db = nullptr, *dout = nullptr;
  const size_t abytes = (size_t)na * (a_is_real ? 4 : 8), bbytes = (size_t)nb * (b_is_real ? 4 : 8);
  if ((rc = ctx_scratch(c, 17, abytes, &da))) return rc;
  if ((rc = ctx_scratch(c, 18, bbytes, &db))) return rc;
  if ((rc = ctx_scratch(c, 19, (size_t)no * (real_out ? 4 : 8), &dout))) return rc;
  NXSIG_HIP_TRY(hipMemcpyAsync(da, a, abytes, hipMemcpyHostToDevice, c->stream));
  NXSIG_HIP_TRY(hipMemcpyAsync(db, b, bbytes, hipMemcpyHostToDevice, c->stream));
  if ((rc = launch_convolve_direct(c, da, a_is_real != 0, a_shape, db, b_is_real != 0, b_shape, rank, mode, dout, osh))) return rc;
  int64_t nres = 1;
  for (int d = 0; d < rank; ++d) { nres *= osh[d]; if (out_shape) out_shape[d] = osh[d]; }
  Staged st(c);
  return st.out_copy(out, dout, (size_t)nres * (real_out ? 4 : 8));
  NXSIG_API_END
}

int nxsig_fftconvolve_c64(nxsig_ctx* ctx, const nxsig_c64* a, int64_t n1, const nxsig_c64* b, int64_t n2, int32_t mode,
                          nxsig_c64* out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!a || !b || !out) return set_error(NXSIG_ERR_INVALID_ARG, "fftconvolve: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (n1 < 1 || n2 < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fftconvolve: lengths must be >= 1");
  const int64_t full = n1 + n2 - 1;
  int64_t out_len, start;
  switch (mode) {  // lib/nx_signal/convolution.ex:300-329
    case NXSIG_CONV_FULL: out_len = full; start = 0; break;
    case NXSIG_CONV_SAME: out_len = n1; start = (full - out_len) / 2; break;
    case NXSIG_CONV_VALID: out_len = (n1 >= n2 ? n1 - n2 : n2 - n1) + 1; start = (full - out_len) / 2; break;
    default: return set_error(NXSIG_ERR_INVALID_ARG, "expected mode to be one of [:full, :same, :valid]");
  }
  if (mem == NXSIG_DEVICE)
    return launch_fftconvolve_c64(c, reinterpret_cast<const float2*>(a), n1, reinterpret_cast<const float2*>(b), n2, start, out_len,
                                  reinterpret_cast<float2*>(out));
  Staged st(c);
  const void *ad = nullptr, *bd = nullptr;
  void* od = nullptr;
  if ((rc = st.in(1, a, (size_t)n1 * sizeof(float2), &ad))) return rc;
  if ((rc = st.in(2, b, (size_t)n2 * sizeof(float2), &bd))) return rc;
  if ((rc = ctx_scratch(c, 4, (size_t)(out_len > 8192 ? out_len : 8192) * sizeof(float2), &od))) return rc;
  if ((rc = launch_fftconvolve_c64(c, reinterpret_cast<const float2*>(ad), n1, reinterpret_cast<const float2*>(bd), n2, start, out_len,
                                   reinterpret_cast<float2*>(od)))) return rc;
  return st.out_copy(out, od, (size_t)out_len * sizeof(float2));
  NXSIG_API_END
}

int nxsig_mel_filters_f32(int32_t fft_length, int32_t mel_bins, double sampling_rate, double max_mel,
                          double mel_frequency_spacing, float* out) {
  NXSIG_API_BEGIN
  if (fft_length < 2 || mel_bins < 1 || !out || !(mel_frequency_spacing > 0.0))
    return set_error(NXSIG_ERR_INVALID_ARG, "mel_filters: fft_length >= 2, mel_bins >= 1 and a positive spacing are required");
  mel_filters_f32(fft_length, mel_bins, sampling_rate, max_mel, mel_frequency_spacing, out);
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_stft_to_mel(nxsig_ctx* ctx, const nxsig_c64* z, int64_t rows, int32_t fft_length, int32_t mel_bins,
                      const float* filters, float* out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!z || !filters || !out) return set_error(NXSIG_ERR_INVALID_ARG, "stft_to_mel: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (rows < 1 || fft_length < 2 || mel_bins < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft_to_mel: rows, fft_length, mel_bins must be positive");
  if (fft_length / 2 > 8192) return set_error(NXSIG_ERR_UNSUPPORTED, "stft_to_mel: fft_length > 16384 is not supported");
  const size_t zbytes = (size_t)rows * fft_length * sizeof(float2), obytes = (size_t)rows * mel_bins * sizeof(float);
  if (mem == NXSIG_DEVICE) return launch_stft_to_mel(c, reinterpret_cast<const float2*>(z), rows, fft_length, mel_bins, filters, out);
  Staged st(c);
  const void* zd = nullptr; void* od = nullptr;
  if ((rc = st.in(1, z, zbytes, &zd))) return rc;
  if ((rc = st.out_alloc(2, obytes, &od))) return rc;
  if ((rc = launch_stft_to_mel(c, reinterpret_cast<const float2*>(zd), rows, fft_length, mel_bins, filters, reinterpret_cast<float*>(od)))) return rc;
  return st.out_copy(out, od, obytes);
  NXSIG_API_END
}

int nxsig_spectrum_mul_c64(nxsig_ctx* ctx, const nxsig_c64* z, int64_t rows, int32_t fft_length, const nxsig_c64* h,
                           nxsig_c64* out, int32_t mem) {
  NXSIG_API_BEGIN
  if (!z || !h || !out) return set_error(NXSIG_ERR_INVALID_ARG, "spectrum_mul: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (rows < 0 || fft_length < 1) return set_error(NXSIG_ERR_INVALID_ARG, "spectrum_mul: rows >= 0 and fft_length >= 1 required");
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (rows == 0) return NXSIG_OK;
  const void* hd = nullptr;
  if ((rc = ctx_table(c, 0x5BEC0ull ^ (uint64_t)fft_length, h, (size_t)fft_length * sizeof(float2), &hd))) return rc;
  const size_t bytes = (size_t)rows * fft_length * sizeof(float2);
  if (mem == NXSIG_DEVICE)
    return launch_spectrum_mul(c, reinterpret_cast<const float2*>(z), rows, fft_length, reinterpret_cast<const float2*>(hd),
                               reinterpret_cast<float2*>(out));
  Staged st(c);
  const void* zd = nullptr;
  if ((rc = st.in(1, z, bytes, &zd))) return rc;
  float2* zw = reinterpret_cast<float2*>(const_cast<void*>(zd));
  if ((rc = launch_spectrum_mul(c, zw, rows, fft_length, reinterpret_cast<const float2*>(hd), zw))) return rc;
  return st.out_copy(out, zw, bytes);
  NXSIG_API_END
}

int nxsig_stft_mel_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride,
                       const float* window, const nxsig_stft_params* p, int32_t mel_bins, const float* filters, float* out,
                       int64_t* num_frames_out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!x || !window || !p || !filters || !out) return set_error(NXSIG_ERR_INVALID_ARG, "stft_mel: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1 || batch > 65535) return set_error(NXSIG_ERR_INVALID_ARG, "stft_mel: batch must be in [1, 65535]");
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "stft_mel: batch_stride < length");
  if (p->fft_length < 2 || mel_bins < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft_mel: fft_length >= 2 and mel_bins >= 1 required");
  rc = check_scaling(p->scaling);
  if (rc) return rc;
  Framing fr;
  rc = make_framing(length, p->frame_length, p->hop, p->pad_mode, p->pad_lo, p->pad_hi, &fr);
  if (rc) return rc;
  if (num_frames_out) *num_frames_out = fr.M;
  StftLaunch a;
  a.fr = fr; a.batch = batch; a.batch_stride = batch_stride; a.K = p->fft_length;
  a.has_scale = p->scaling != NXSIG_SCALE_NONE;
  a.inv_scale_div = a.has_scale ? scaling_factor(window, p->frame_length, p->scaling, p->sampling_rate) : 1.0f;
  rc = ctx_window(c, window, p->frame_length, p->fft_length, &a.window, &a.window_padK);
  if (rc) return rc;
  a.z = nullptr;
  const size_t obytes = (size_t)batch * fr.M * mel_bins * sizeof(float);
  Staged st(c);
  float* od = out;
  if (mem == NXSIG_DEVICE) {
    a.x = x;
  } else {
    const void* xd = nullptr; void* o2 = nullptr;
    const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(float);
    if ((rc = st.in(1, x, xbytes, &xd))) return rc;
    if ((rc = st.out_alloc(2, obytes, &o2))) return rc;
    a.x = reinterpret_cast<const float*>(xd); od = reinterpret_cast<float*>(o2);
  }
  bool handled = false;
  rc = launch_stft_mel_wave(c, a, mel_bins, filters, od, &handled);
  if (rc) return rc;
  if (!handled) {  // two-step path: spectrum in a scratch buffer, then the band-sum kernel
    void* zs = nullptr;
    if ((rc = ctx_scratch(c, 4, (size_t)batch * fr.M * p->fft_length * sizeof(float2), &zs))) return rc;
    a.z = reinterpret_cast<float2*>(zs);
    if ((rc = launch_stft(c, a))) return rc;
    if ((rc = launch_stft_to_mel(c, a.z, (int64_t)batch * fr.M, p->fft_length, mel_bins, filters, od))) return rc;
  }
  if (mem == NXSIG_DEVICE) return NXSIG_OK;
  return st.out_copy(out, od, obytes);
  NXSIG_API_END
}

}  // extern "C"
static int stft_onesided_impl(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* window,
                              const nxsig_stft_params* p, nxsig_c64* out, int64_t* num_frames_out, int32_t mem, bool packed);
extern "C" {
int nxsig_stft_onesided_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* window,
                            const nxsig_stft_params* p, nxsig_c64* out, int64_t* num_frames_out, int32_t mem) {
  return stft_onesided_impl(ctx, x, length, batch, batch_stride, window, p, out, num_frames_out, mem, false);
}
int nxsig_stft_packed_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* window,
                          const nxsig_stft_params* p, nxsig_c64* out, int64_t* num_frames_out, int32_t mem) {
  return stft_onesided_impl(ctx, x, length, batch, batch_stride, window, p, out, num_frames_out, mem, true);
}
}  // extern "C"
static int stft_onesided_impl(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* window,
                              const nxsig_stft_params* p, nxsig_c64* out, int64_t* num_frames_out, int32_t mem, bool packed) {
  NXSIG_API_BEGIN
  if (!x || !window || !p || !out) return set_error(NXSIG_ERR_INVALID_ARG, "stft_onesided: null pointer argument");
  if (packed && (p->fft_length & 1)) return set_error(NXSIG_ERR_INVALID_ARG, "stft_packed: fft_length must be even");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1 || batch > 65535) return set_error(NXSIG_ERR_INVALID_ARG, "stft_onesided: batch must be in [1, 65535]");
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "stft_onesided: batch_stride < length");
  if (p->fft_length < 2) return set_error(NXSIG_ERR_INVALID_ARG, "stft_onesided: fft_length >= 2 required");
  rc = check_scaling(p->scaling);
  if (rc) return rc;
  Framing fr;
  rc = make_framing(length, p->frame_length, p->hop, p->pad_mode, p->pad_lo, p->pad_hi, &fr);
  if (rc) return rc;
  if (num_frames_out) *num_frames_out = fr.M;
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  StftLaunch a;
  a.fr = fr; a.batch = batch; a.batch_stride = batch_stride; a.K = p->fft_length;
  a.has_scale = p->scaling != NXSIG_SCALE_NONE;
  a.inv_scale_div = a.has_scale ? scaling_factor(window, p->frame_length, p->scaling, p->sampling_rate) : 1.0f;
  rc = ctx_window(c, window, p->frame_length, p->fft_length, &a.window, &a.window_padK);
  if (rc) return rc;
  a.z = nullptr;
  const int half = p->fft_length / 2;
  const size_t obytes = (size_t)batch * fr.M * half * sizeof(float2);
  Staged st(c);
  float2* od = reinterpret_cast<float2*>(out);
  if (mem == NXSIG_DEVICE) {
    a.x = x;
  } else {
    const void* xd = nullptr; void* o2 = nullptr;
    const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(float);
    if ((rc = st.in(1, x, xbytes, &xd))) return rc;
    if ((rc = st.out_alloc(2, obytes, &o2))) return rc;
    a.x = reinterpret_cast<const float*>(xd); od = reinterpret_cast<float2*>(o2);
  }
  bool handled = false;
  rc = launch_stft_mag_wave(c, a, packed ? 4 : 3 /* complex bins */, reinterpret_cast<float*>(od), &handled);
  if (rc) return rc;
  if (!handled) {  // two-step path: full spectrum in a scratch buffer, then the slice
    void* zs = nullptr;
    if ((rc = ctx_scratch(c, 4, (size_t)batch * fr.M * p->fft_length * sizeof(float2), &zs))) return rc;
    a.z = reinterpret_cast<float2*>(zs);
    if ((rc = launch_stft(c, a))) return rc;
    if ((rc = launch_half_from_spectrum(c, a.z, (int64_t)batch * fr.M, p->fft_length, od, packed))) return rc;
  }
  if (mem == NXSIG_DEVICE) return NXSIG_OK;
  return st.out_copy(out, od, obytes);
  NXSIG_API_END
}
extern "C" {

int nxsig_stft_magnitude_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride,
                             const float* window, const nxsig_stft_params* p, int32_t kind, float* out,
                             int64_t* num_frames_out, int32_t mem) {
  NXSIG_API_BEGIN
  if (!x || !window || !p || !out) return set_error(NXSIG_ERR_INVALID_ARG, "stft_magnitude: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (kind != NXSIG_MAG_ABS && kind != NXSIG_MAG_POWER && kind != NXSIG_MAG_DBFS)
    return set_error(NXSIG_ERR_INVALID_ARG, "stft_magnitude: kind must be NXSIG_MAG_ABS, NXSIG_MAG_POWER or NXSIG_MAG_DBFS");
  if (batch < 1 || batch > 65535) return set_error(NXSIG_ERR_INVALID_ARG, "stft_magnitude: batch must be in [1, 65535]");
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "stft_magnitude: batch_stride < length");
  if (p->fft_length < 2) return set_error(NXSIG_ERR_INVALID_ARG, "stft_magnitude: fft_length >= 2 required");
  rc = check_scaling(p->scaling);
  if (rc) return rc;
  Framing fr;
  rc = make_framing(length, p->frame_length, p->hop, p->pad_mode, p->pad_lo, p->pad_hi, &fr);
  if (rc) return rc;
  if (num_frames_out) *num_frames_out = fr.M;
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  StftLaunch a;
  a.fr = fr; a.batch = batch; a.batch_stride = batch_stride; a.K = p->fft_length;
  a.has_scale = p->scaling != NXSIG_SCALE_NONE;
  a.inv_scale_div = a.has_scale ? scaling_factor(window, p->frame_length, p->scaling, p->sampling_rate) : 1.0f;
  rc = ctx_window(c, window, p->frame_length, p->fft_length, &a.window, &a.window_padK);
  if (rc) return rc;
  a.z = nullptr;
  const int half = p->fft_length / 2;
  const size_t obytes = (size_t)batch * fr.M * half * sizeof(float);
  Staged st(c);
  float* od = out;
  if (mem == NXSIG_DEVICE) {
    a.x = x;
  } else {
    const void* xd = nullptr; void* o2 = nullptr;
    const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(float);
    if ((rc = st.in(1, x, xbytes, &xd))) return rc;
    if ((rc = st.out_alloc(2, obytes, &o2))) return rc;
    a.x = reinterpret_cast<const float*>(xd); od = reinterpret_cast<float*>(o2);
  }
  bool handled = false;
  rc = launch_stft_mag_wave(c, a, kind, od, &handled);
  if (rc) return rc;
  if (!handled) {  // two-step path: spectrum in a scratch buffer, then the magnitude kernel
    void* zs = nullptr;
    if ((rc = ctx_scratch(c, 4, (size_t)batch * fr.M * p->fft_length * sizeof(float2), &zs))) return rc;
    a.z = reinterpret_cast<float2*>(zs);
    if ((rc = launch_stft(c, a))) return rc;
    if ((rc = launch_mag_from_spectrum(c, a.z, (int64_t)batch * fr.M, p->fft_length, kind, od))) return rc;
  }
  if (mem == NXSIG_DEVICE) return NXSIG_OK;
  return st.out_copy(out, od, obytes);
  NXSIG_API_END
}

/* ---------------------------------------------------------------- f64 / c128 tier (kernels_f64.hip) */
int nxsig_window_f64(int32_t kind, int32_t n, int32_t is_periodic, double beta, double eps, double* out) {
  NXSIG_API_BEGIN
  return window_f64(kind, n, is_periodic != 0, beta, eps, out);
  NXSIG_API_END
}

int nxsig_sinc_f64(const double* t, int64_t n, double* out) {
  NXSIG_API_BEGIN
  if (n < 0 || (n > 0 && (!t || !out))) return set_error(NXSIG_ERR_INVALID_ARG, "sinc: bad arguments");
  sinc_f64(t, n, out);
  return NXSIG_OK;
  NXSIG_API_END
}

int nxsig_firwin_f64(int32_t num_taps, const double* cutoff, int32_t n_cutoff, int32_t window_kind, double kaiser_beta,
                     int32_t pass_zero, int32_t scale, double sampling_rate, double* out) {
  NXSIG_API_BEGIN
  return firwin_f64(num_taps, cutoff, n_cutoff, window_kind, kaiser_beta, pass_zero != 0, scale != 0, sampling_rate, out);
  NXSIG_API_END
}

int nxsig_fft_frequencies_f64(double sampling_rate, int32_t fft_length, int32_t endpoint, double* out) {
  NXSIG_API_BEGIN
  if (fft_length < 1 || !out) return set_error(NXSIG_ERR_INVALID_ARG, "fft_frequencies: fft_length must be >= 1");
  fft_frequencies_f64(sampling_rate, fft_length, endpoint != 0, out);
  return NXSIG_OK;
  NXSIG_API_END
}

// the caller's window as f64 on the device; `wide` receives the exactly widened host copy of an f32 window
static int window_dev_f64(Ctx* c, const void* window, int32_t window_is_f64, int N, std::vector<double>& wide, const double** dev) {
  const double* wh = reinterpret_cast<const double*>(window);
  if (!window_is_f64) {
    wide.resize(N);
    for (int i = 0; i < N; ++i) wide[i] = (double)reinterpret_cast<const float*>(window)[i];
    wh = wide.data();
  }
  const void* d = nullptr;
  int rc = ctx_table(c, 0x57494E3634ull, wh, (size_t)N * sizeof(double), &d);
  if (rc) return rc;
  *dev = reinterpret_cast<const double*>(d);
  return NXSIG_OK;
}
// the scalar of :scaling in the window's own type (Nx.sum(window) / Nx.sqrt(fs * Nx.sum(window ** 2))), widened
static double scaling_of(const void* window, int32_t window_is_f64, int N, int scaling, double fs) {
  if (window_is_f64) return scaling_factor_f64(reinterpret_cast<const double*>(window), N, scaling, fs);
  return (double)scaling_factor(reinterpret_cast<const float*>(window), N, scaling, fs);
}

int nxsig_stft_f64(nxsig_ctx* ctx, const double* x, int64_t length, int32_t batch, int64_t batch_stride, const void* window,
                   int32_t window_is_f64, const nxsig_stft_params* p, nxsig_c128* z, int64_t* num_frames_out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!x || !window || !p || !z) return set_error(NXSIG_ERR_INVALID_ARG, "stft: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1 || batch > 65535) return set_error(NXSIG_ERR_INVALID_ARG, "stft: batch must be in [1, 65535]");
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "stft: batch_stride < length");
  if (p->fft_length < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft: fft_length must be >= 1");
  if ((rc = check_scaling(p->scaling))) return rc;
  Framing fr;
  if ((rc = make_framing(length, p->frame_length, p->hop, p->pad_mode, p->pad_lo, p->pad_hi, &fr))) return rc;
  if (num_frames_out) *num_frames_out = fr.M;
  StftLaunchD a;
  a.fr = fr; a.batch = batch; a.batch_stride = batch_stride; a.K = p->fft_length;
  a.has_scale = p->scaling != NXSIG_SCALE_NONE;
  a.div = a.has_scale ? scaling_of(window, window_is_f64, p->frame_length, p->scaling, p->sampling_rate) : 1.0;
  std::vector<double> wide;
  if ((rc = window_dev_f64(c, window, window_is_f64, p->frame_length, wide, &a.window))) return rc;
  const size_t zbytes = (size_t)batch * fr.M * p->fft_length * sizeof(double2);
  if (mem == NXSIG_DEVICE) {
    a.x = x; a.z = reinterpret_cast<double2*>(z);
    return launch_stft_f64(c, a);
  }
  Staged st(c);
  const void* xd = nullptr; void* zd = nullptr;
  const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(double);
  if ((rc = st.in(1, x, xbytes, &xd))) return rc;
  if ((rc = st.out_alloc(2, zbytes, &zd))) return rc;
  a.x = reinterpret_cast<const double*>(xd); a.z = reinterpret_cast<double2*>(zd);
  if ((rc = launch_stft_f64(c, a))) return rc;
  return st.out_copy(z, zd, zbytes);
  NXSIG_API_END
}

int nxsig_stft_c128(nxsig_ctx* ctx, const nxsig_c128* x, int64_t length, int32_t batch, int64_t batch_stride, const void* window,
                    int32_t window_is_f64, const nxsig_stft_params* p, nxsig_c128* z, int64_t* num_frames_out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!x || !window || !p || !z) return set_error(NXSIG_ERR_INVALID_ARG, "stft: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1 || batch > 65535) return set_error(NXSIG_ERR_INVALID_ARG, "stft: batch must be in [1, 65535]");
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "stft: batch_stride < length");
  if (p->fft_length < 1) return set_error(NXSIG_ERR_INVALID_ARG, "stft: fft_length must be >= 1");
  if ((rc = check_scaling(p->scaling))) return rc;
  Framing fr;
  if ((rc = make_framing(length, p->frame_length, p->hop, p->pad_mode, p->pad_lo, p->pad_hi, &fr))) return rc;
  if (num_frames_out) *num_frames_out = fr.M;
  StftLaunchD a;
  a.fr = fr; a.batch = batch; a.batch_stride = batch_stride; a.K = p->fft_length; a.x_is_complex = 1;
  a.has_scale = p->scaling != NXSIG_SCALE_NONE;
  a.div = a.has_scale ? scaling_of(window, window_is_f64, p->frame_length, p->scaling, p->sampling_rate) : 1.0;
  std::vector<double> wide;
  if ((rc = window_dev_f64(c, window, window_is_f64, p->frame_length, wide, &a.window))) return rc;
  dispatch_note("stft.f64.c128");
  const size_t zbytes = (size_t)batch * fr.M * p->fft_length * sizeof(double2);
  if (mem == NXSIG_DEVICE) {
    a.x = reinterpret_cast<const double*>(x); a.z = reinterpret_cast<double2*>(z);
    return launch_stft_f64(c, a);
  }
  Staged st(c);
  const void* xd = nullptr; void* zd = nullptr;
  const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(double2);
  if ((rc = st.in(1, x, xbytes, &xd))) return rc;
  if ((rc = st.out_alloc(2, zbytes, &zd))) return rc;
  a.x = reinterpret_cast<const double*>(xd); a.z = reinterpret_cast<double2*>(zd);
  if ((rc = launch_stft_f64(c, a))) return rc;
  return st.out_copy(z, zd, zbytes);
  NXSIG_API_END
}

int nxsig_istft_c128(nxsig_ctx* ctx, const nxsig_c128* z, int64_t num_frames, int32_t batch, const void* window, int32_t window_is_f64,
                     const nxsig_stft_params* p, nxsig_c128* y, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!z || !window || !p || !y) return set_error(NXSIG_ERR_INVALID_ARG, "istft: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1 || batch > 65535) return set_error(NXSIG_ERR_INVALID_ARG, "istft: batch must be in [1, 65535]");
  if (num_frames < 1) return set_error(NXSIG_ERR_INVALID_ARG, "istft: num_frames must be >= 1");
  if ((rc = check_scaling(p->scaling))) return rc;
  const int N = p->frame_length, hop = p->hop, K = p->fft_length;
  if (N < 1 || hop < 1) return set_error(NXSIG_ERR_INVALID_ARG, "istft: frame_length and hop must be >= 1");
  if (hop > N) return set_error(NXSIG_ERR_INVALID_ARG, "overlap_length must be a number less than the window size");
  if (K != N)
    return set_error(NXSIG_ERR_INVALID_ARG,
                     "istft: fft_length must equal the window length (the reference broadcasts {M,K} x {N}, lib/nx_signal.ex:628)");
  IstftLaunchD a;
  a.M = num_frames; a.batch = batch; a.N = N; a.hop = hop; a.K = K; a.window_f32 = window_is_f64 ? 0 : 1;
  a.has_scale = p->scaling != NXSIG_SCALE_NONE;
  a.scale_mul = a.has_scale ? scaling_of(window, window_is_f64, N, p->scaling, p->sampling_rate) : 1.0;
  std::vector<double> wide;
  if ((rc = window_dev_f64(c, window, window_is_f64, N, wide, &a.window))) return rc;
  const int64_t out_len = num_frames * hop + (N - hop);
  const size_t zbytes = (size_t)batch * num_frames * K * sizeof(double2), ybytes = (size_t)batch * out_len * sizeof(double2);
  if (mem == NXSIG_DEVICE) {
    a.z = reinterpret_cast<const double2*>(z); a.y = reinterpret_cast<double2*>(y);
    return launch_istft_f64(c, a);
  }
  Staged st(c);
  const void* zd = nullptr; void* yd = nullptr;
  if ((rc = st.in(1, z, zbytes, &zd))) return rc;
  if ((rc = st.out_alloc(2, ybytes, &yd))) return rc;
  a.z = reinterpret_cast<const double2*>(zd); a.y = reinterpret_cast<double2*>(yd);
  if ((rc = launch_istft_f64(c, a))) return rc;
  return st.out_copy(y, yd, ybytes);
  NXSIG_API_END
}

int nxsig_fft_c128(nxsig_ctx* ctx, const void* in, int32_t in_is_real, int64_t rows, int32_t n_in, int32_t fft_length, int32_t inverse,
                   nxsig_c128* out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!in || !out) return set_error(NXSIG_ERR_INVALID_ARG, "fft: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (rows < 1 || n_in < 1 || fft_length < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fft: rows, n_in and fft_length must be >= 1");
  const size_t ibytes = (size_t)rows * n_in * (in_is_real ? sizeof(double) : sizeof(double2));
  const size_t obytes = (size_t)rows * fft_length * sizeof(double2);
  if (mem == NXSIG_DEVICE) return launch_fft_f64(c, in, in_is_real != 0, rows, n_in, fft_length, inverse != 0, reinterpret_cast<double2*>(out));
  Staged st(c);
  const void* id = nullptr; void* od = nullptr;
  if ((rc = st.in(1, in, ibytes, &id))) return rc;
  if ((rc = st.out_alloc(2, obytes, &od))) return rc;
  if ((rc = launch_fft_f64(c, id, in_is_real != 0, rows, n_in, fft_length, inverse != 0, reinterpret_cast<double2*>(od)))) return rc;
  return st.out_copy(out, od, obytes);
  NXSIG_API_END
}

int nxsig_as_windowed_f64(nxsig_ctx* ctx, const double* x, int64_t length, int32_t batch, int64_t batch_stride, int32_t window_length,
                          int32_t stride, int32_t pad_mode, int64_t pad_lo, int64_t pad_hi, double* out, int64_t* num_frames_out,
                          int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!x || !out) return set_error(NXSIG_ERR_INVALID_ARG, "as_windowed: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1 || batch > 65535) return set_error(NXSIG_ERR_INVALID_ARG, "as_windowed: batch must be in [1, 65535]");
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "as_windowed: batch_stride < length");
  Framing fr;
  if ((rc = make_framing(length, window_length, stride, pad_mode, pad_lo, pad_hi, &fr))) return rc;
  if (num_frames_out) *num_frames_out = fr.M;
  const size_t obytes = (size_t)batch * fr.M * fr.N * sizeof(double);
  if (mem == NXSIG_DEVICE) return launch_as_windowed_f64(c, x, batch_stride, batch, fr, out);
  Staged st(c);
  const void* xd = nullptr; void* od = nullptr;
  const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(double);
  if ((rc = st.in(1, x, xbytes, &xd))) return rc;
  if ((rc = st.out_alloc(2, obytes, &od))) return rc;
  if ((rc = launch_as_windowed_f64(c, reinterpret_cast<const double*>(xd), batch_stride, batch, fr, reinterpret_cast<double*>(od)))) return rc;
  return st.out_copy(out, od, obytes);
  NXSIG_API_END
}

int nxsig_overlap_and_add_f64(nxsig_ctx* ctx, const double* frames, int64_t num_frames, int32_t batch, int32_t frame_length,
                              int32_t overlap_length, int32_t components, double* out, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!frames || !out) return set_error(NXSIG_ERR_INVALID_ARG, "overlap_and_add: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (components != 1 && components != 2) return set_error(NXSIG_ERR_INVALID_ARG, "overlap_and_add: components must be 1 (f64) or 2 (c128)");
  if (batch < 1 || batch > 65535 || num_frames < 1 || frame_length < 1)
    return set_error(NXSIG_ERR_INVALID_ARG, "overlap_and_add: batch, num_frames and frame_length must be >= 1");
  if (overlap_length >= frame_length)
    return set_error(NXSIG_ERR_INVALID_ARG, "overlap_length must be a number less than the window size " +
                                                std::to_string(frame_length) + ", got: " + std::to_string(frame_length));
  if (overlap_length < 0) return set_error(NXSIG_ERR_INVALID_ARG, "overlap_and_add: overlap_length must be >= 0");
  const int hop = frame_length - overlap_length;
  const int64_t out_len = num_frames * hop + overlap_length;
  const size_t ibytes = (size_t)batch * num_frames * frame_length * components * sizeof(double);
  const size_t obytes = (size_t)batch * out_len * components * sizeof(double);
  if (mem == NXSIG_DEVICE) return launch_ola_f64(c, frames, num_frames, batch, frame_length, hop, components, nullptr, false, false, out);
  Staged st(c);
  const void* fd = nullptr; void* od = nullptr;
  if ((rc = st.in(1, frames, ibytes, &fd))) return rc;
  if ((rc = st.out_alloc(2, obytes, &od))) return rc;
  if ((rc = launch_ola_f64(c, reinterpret_cast<const double*>(fd), num_frames, batch, frame_length, hop, components, nullptr, false, false,
                           reinterpret_cast<double*>(od)))) return rc;
  return st.out_copy(out, od, obytes);
  NXSIG_API_END
}

static int fir_common_f64(nxsig_ctx* ctx, const double* x, int64_t length, int32_t batch, int64_t batch_stride, const double* h,
                          int32_t num_taps, int64_t start, int64_t out_len, double* y, int32_t mem) {
  NXSIG_API_BEGIN
  NXSIG_CHECK_CTX(ctx)
  DispatchScope dispatch_scope(c);
  if (!x || !h || !y) return set_error(NXSIG_ERR_INVALID_ARG, "fir: null pointer argument");
  int rc = check_mem(mem);
  if (rc) return rc;
  if (batch < 1 || batch > 65535 || length < 1 || num_taps < 1)
    return set_error(NXSIG_ERR_INVALID_ARG, "fir: batch, length and num_taps must be >= 1");
  if (batch_stride < length) return set_error(NXSIG_ERR_INVALID_ARG, "fir: batch_stride < length");
  if (start < 0 || out_len < 1 || start + out_len > length + num_taps - 1)
    return set_error(NXSIG_ERR_INVALID_ARG, "fir: requested slice lies outside the full convolution");
  FirLaunchD a;
  a.L = length; a.batch = batch; a.batch_stride = batch_stride; a.h_host = h; a.taps = num_taps;
  a.out_start = start; a.out_len = out_len;
  const size_t ybytes = (size_t)batch * out_len * sizeof(double);
  if (mem == NXSIG_DEVICE) {
    a.x = x; a.y = y;
    return launch_fir_f64(c, a);
  }
  Staged st(c);
  const void* xd = nullptr; void* yd = nullptr;
  const size_t xbytes = ((size_t)(batch - 1) * batch_stride + length) * sizeof(double);
  if ((rc = st.in(1, x, xbytes, &xd))) return rc;
  if ((rc = st.out_alloc(2, ybytes, &yd))) return rc;
  a.x = reinterpret_cast<const double*>(xd); a.y = reinterpret_cast<double*>(yd);
  if ((rc = launch_fir_f64(c, a))) return rc;
  return st.out_copy(y, yd, ybytes);
  NXSIG_API_END
}

int nxsig_fir_f64(nxsig_ctx* ctx, const double* x, int64_t length, int32_t batch, int64_t batch_stride, const double* h,
                  int32_t num_taps, int32_t mode, double* y, int32_t mem) {
  if (length < 1 || num_taps < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fir: batch, length and num_taps must be >= 1");
  const int64_t full = length + num_taps - 1;
  int64_t out_len, start;
  switch (mode) {
    case NXSIG_CONV_FULL: out_len = full; start = 0; break;
    case NXSIG_CONV_SAME: out_len = length; start = (full - out_len) / 2; break;
    case NXSIG_CONV_VALID:
      out_len = (length >= num_taps ? length - num_taps : num_taps - length) + 1;
      start = (full - out_len) / 2;
      break;
    default:
      return set_error(NXSIG_ERR_INVALID_ARG, "expected mode to be one of [:full, :same, :valid]");
  }
  return fir_common_f64(ctx, x, length, batch, batch_stride, h, num_taps, start, out_len, y, mem);
}

int nxsig_fir_slice_f64(nxsig_ctx* ctx, const double* x, int64_t length, int32_t batch, int64_t batch_stride, const double* h,
                        int32_t num_taps, int64_t out_start, int64_t out_len, double* y, int32_t mem) {
  return fir_common_f64(ctx, x, length, batch, batch_stride, h, num_taps, out_start, out_len, y, mem);
}

}  // extern "C"
