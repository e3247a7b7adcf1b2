// Internal declarations shared by the C-ABI layer (api.cpp), the host-side generators
// (host_numerics.cpp) and the HIP launchers (*.hip).  Nothing here is exported.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/nxsig.h"

namespace nxsig {

// ---- error plumbing: thread-local message, never throws across the C ABI ----
int set_error(int code, const std::string& msg);
const char* last_error_cstr();

// ---- dispatch record (nxsig_last_dispatch): the kernel FAMILIES the calling thread's last compute call launched, "a+b+c" in launch
// order (each family once).  Every launcher that commits to a kernel family notes it; the compute entry points of the C ABI reset
// the record.  Thread-local like the error message: dirty schedulers calling concurrently see their own call's record.
void dispatch_reset();
void dispatch_note(const char* family);
const char* dispatch_cstr();

#define NXSIG_HIP_TRY(expr)                                                                          \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return ::nxsig::set_error(_e == hipErrorOutOfMemory ? NXSIG_ERR_OOM : NXSIG_ERR_HIP,           \
                                std::string(#expr) + ": " + hipGetErrorString(_e));                  \
  } while (0)

// ---- Nx.fft / Nx.ifft clean-up (SURVEY App. A rule 7; call sites lib/nx_signal.ex:102, :609, convolution.ex:282): the
// BinaryBackend zeroes every component of a transform's result whose magnitude is <= eps (1e-10, the default of the :eps
// option) before rounding to c64.  Every forward / inverse kernel applies it to the finished transform, ahead of any scaling
// or window product the reference applies afterwards.  NaN compares false and passes through.
constexpr float kFftEps = 1.0e-10f;
__host__ __device__ __forceinline__ float fft_eps0(float x) { return __builtin_fabsf(x) <= kFftEps ? 0.0f : x; }
__host__ __device__ __forceinline__ float2 fft_eps0(float2 v) { return make_float2(fft_eps0(v.x), fft_eps0(v.y)); }

// ---- host numerics (host_numerics.cpp) ----
int window_f32(int kind, int n, bool periodic, double beta, double eps, float* out);
void sinc_f32(const float* t, int64_t n, float* out);
int firwin_f32(int num_taps, const double* cutoff, int n_cutoff, int window_kind, double beta, bool pass_zero,
               bool scale, double sampling_rate, float* out);
void fft_frequencies_f32(double fs, int K, bool endpoint, float* out);
void stft_times_f32(int N, double fs, int64_t M, float* out);
float scaling_factor(const float* w, int N, int scaling, double fs);
void mel_filters_f32(int K, int mel_bins, double fs, double max_mel, double f_sp, float* out);
// the same generators evaluated in f64 (`type: {:f, 64}` of the reference: every op rounds to double)
int window_f64(int kind, int n, bool periodic, double beta, double eps, double* out);
void sinc_f64(const double* t, int64_t n, double* out);
int firwin_f64(int num_taps, const double* cutoff, int n_cutoff, int window_kind, double beta, bool pass_zero,
               bool scale, double sampling_rate, double* out);
void fft_frequencies_f64(double fs, int K, bool endpoint, double* out);
double scaling_factor_f64(const double* w, int N, int scaling, double fs);

// ---- framing geometry (as_windowed, lib/nx_signal.ex:257-331) ----
struct Framing {
  int64_t L;       // signal length
  int32_t N;       // frame (window) length
  int32_t hop;     // stride
  int32_t reflect; // 1 = mirror padding, 0 = zero padding
  int64_t lo, hi;  // padding (may be negative for explicit crop)
  int64_t M;       // number of frames
};
int make_framing(int64_t L, int32_t N, int32_t hop, int32_t pad_mode, int64_t pad_lo, int64_t pad_hi, Framing* out);

// ---- dispatch / geometry switches ----
// Read from the environment ONCE per context, in nxsig_ctx_create (tuning_from_env: the library's one getenv loop); afterwards a
// launch only reads the context's Tuning struct.  nxsig_ctx_set_tuning changes a switch of one context at run time (the A/B sweep
// tools, tests); nothing in a launch path looks at the process environment.  Names are the NXSIG_<NAME> variables of
// INTEGRATION.md.  The knobs of concluded experiments are gone (round 4): their winning value is a constant at the use site.
#define NXSIG_TUNABLES(X)                                                                                                      \
  X(DISABLE_WAVE) X(DISABLE_WAVE_ROWS) X(DISABLE_BLUE_WAVE) X(DISABLE_R20) X(DISABLE_RAB) X(DISABLE_8K) X(DISABLE_4K) X(DISABLE_FUSED_FILTER) \
  X(ISTFT_DEEP) X(ISTFT_HALF_DEEP) X(ISTFT_RUNS_PER_CU) X(ISTFT_MIN_RUN) X(ISTFT_REGOLA)                                                    \
  X(STORE_POLICY) X(WAVE_NO_SPLIT) X(NO_AL8) X(NO_STAGE) X(WAVE_UNITS_PER_WAVE) X(STAGE_PAD) X(WAVE_SMALL_W) X(WAVE_SMALL_CHUNK) \
  X(FIR32) X(FIR_PAD_TAPS) X(FIR_PHASE) X(FIR_HREG) X(FIR_UNITS_PER_WAVE) X(FIR_R2K) X(FIR_DLINE)                                                    \
  X(MEL_TILE) X(MEL_LDS_KB) X(FFT_TILED) X(FFT_TILE_ELEMS) X(FFT_TILE_NT) X(FFT_COLUMNS) X(FFT_TILED_MIN) X(CONV_POW2)          \
  X(DIRECT_FAST) X(POOL_MAX_MB) X(NO_PREFAULT) X(HOST_PIPE)
enum TuneKey : int {
#define NXSIG_X(n) kT_##n,
  NXSIG_TUNABLES(NXSIG_X)
#undef NXSIG_X
  kTuneCount
};
struct Tuning {
  int32_t v[kTuneCount] = {};
  bool set[kTuneCount] = {};
};
void tuning_from_env(Tuning* t);       // api.cpp
bool tuning_value_ok(int key, long value, long* lo, long* hi);   // the admissible range of a switch (api.cpp)
int tuning_index(const char* name);    // "NXSIG_FOO" or "FOO" -> TuneKey, -1 when there is no such switch
const char* tuning_name(int key);      // "NXSIG_FOO"

// ---- device tables cached per context ----
struct DeviceTable {
  void* ptr = nullptr;
  size_t bytes = 0;
  std::vector<unsigned char> host;  // copy of the content (tables <= 1 MiB): a hash hit is verified against it
  bool has_host = false;            // (a zero-byte table has a host copy too: an empty one)
};

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  std::vector<hipEvent_t> lap_events;  // nxsig_timer_lap series: events are created on demand and reused
  size_t laps_used = 0;
  std::mutex mu;
  int num_cus = 0;
  std::string dev_name;
  // twiddle tables w_K^j = exp(-2 pi i j / K), j in [0,K), generated in double, stored as float2
  std::map<int, DeviceTable> twiddles;
  // content-addressed small tables (windows, filter spectra ...): key = fnv1a(tag, bytes)
  std::map<uint64_t, DeviceTable> tables;
  // f64 tier (kernels_f64.hip): device pointers of its twiddle / chirp tables by (kind << 32 | length); the storage belongs to `tables`
  std::map<int64_t, const void*> f64_tables;
  // scratch buffer reused by multi-stage paths (generic istft, host staging)
  // slots: 0 multi-stage temporaries, 1/2 host staging in/out, 3 wave-kernel dummy sink, 4 fused-path spectrum, 5 reduction cells,
  // 6-9 four-step / Bluestein rows, 10-12 fft_nd ping-pong, 13-15 n-D fftconvolve, 16 long-transform stft frames, 17-19 host staging of n-D calls
  // 20 istft filter fallback, 21 long FIR, 22 FIR row flags, 23 istft non-finite unit list, 24/25 packed istft fallback,
  // 27 frames of the f64 istft
  void* scratch[32] = {};
  size_t scratch_bytes[32] = {};
  // per-K tables of the tuned wave kernels (pass-B / pass-C twiddles), built once
  struct WaveTables { const void* twB = nullptr; const void* twC = nullptr; const void* twI = nullptr;
                      const void* twBi = nullptr; const void* twCi = nullptr;    // ...i = conjugated (inverse transform)
                      const void* twQ[3] = {nullptr, nullptr, nullptr}; };       // quad front-ends J = 2, 4, 8
  std::map<int, WaveTables> wave_tables;
  // memo of the last window seen (the common case: the same window tensor call after call)
  std::vector<float> memo_win;
  int memo_K = 0;
  const float* memo_dev = nullptr;   // f32[N]
  const float* memo_devK = nullptr;  // f32[K], zero-padded / truncated to the fft length
  // memo of host-side derivations keyed by the content they derive from (filter spectrum of a tap vector, OLA normaliser
  // rows and edge-fix sample list of a window): a few machine words each, dropped together with `tables`
  std::map<uint64_t, std::vector<uint64_t>> memo;
  // caching allocator behind nxsig_alloc / nxsig_free: freed blocks are kept (no hipFree, no device synchronisation) and handed
  // out again to requests of a similar size.  A 3.3 GB hipMalloc costs ~100 ms on this platform — two hundred times the
  // stft that fills it — so a caller that allocates its result per call (the Python mirror, the NIF's *_dev functions) would
  // otherwise spend all its time in the allocator.  Reuse is ordered by the context's stream.
  std::multimap<size_t, void*> pool_free;   // size -> block
  std::map<void*, size_t> pool_live;        // blocks handed out by nxsig_alloc
  size_t pool_cached = 0, pool_cap = 0;     // bytes sitting in pool_free; cap (0 = not yet decided)
  // host-tensor calls (mem = NXSIG_HOST): two pinned bounce slots per direction + a transfer stream.  The result travels in chunks:
  // DMA of chunk k + 1 into one slot while host threads copy chunk k out of the other into the caller's (pageable) buffer
  // (Staged::out_copy, api.cpp); created on the first host-tensor call of at least kPinMin bytes, freed with the context
  void* pin[4] = {nullptr, nullptr, nullptr, nullptr};   // [0, 1] device -> host, [2, 3] host -> device
  size_t pin_bytes = 0;
  hipStream_t xfer_stream = nullptr;
  hipEvent_t xfer_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t xfer_ready = nullptr;
  std::string last_dispatch;                // kernel families of the last compute call on this context (nxsig_ctx_last_dispatch)
  Tuning tuning;                            // dispatch / geometry switches (environment at creation, nxsig_ctx_set_tuning later)
};
// value of a switch for this context, or the launcher's default when nobody set it
static inline int tune(const Ctx* c, TuneKey k, int dflt) { return c->tuning.set[k] ? (int)c->tuning.v[k] : dflt; }

// Units per wave of a short-lived-workgroup launch (round 6): the measured optimum `dflt` of a launch that runs for many rounds of
// workgroups — but a launch with fewer workgroups than three per CU is ONE round, what sets its time is the longest wave, and it should
// spread over the chip instead (a 10 s utterance at 16 kHz in 400 / 512 framing: 250 units = 16 workgroups at 4 units per wave, 24.4 us;
// at one unit per wave 63 workgroups, 15.4 us; config 1's 2048-point twin 23.5 -> 7.2 us: profiles/r06/small_launch_units_per_wave.txt)
static inline int fill_units_per_wave(const Ctx* c, int64_t total_units, int W, int dflt) {
  const int64_t slots = (int64_t)c->num_cus * 3 * W;
  const int64_t fill = (total_units + slots - 1) / slots;
  return fill < dflt ? (fill < 1 ? 1 : (int)fill) : dflt;
}

// Shortest run of the persistent inverse kernels (every run recomputes its halo, so long runs pay less of it — but a launch with fewer
// units than two per wave slot is ONE partial round, and what sets its time is the length of a run: 1 s of audio, 184 frames, ran as 23
// runs of 8 + 3 frames on 23 of 2 048 wave slots).  Such launches take runs of 2: N = 1024 one second 29.8 -> 19.9 us, N = 2048 ten
// seconds 61 -> 40 us, N = 256 36 -> 19 us (profiles/r06/small_launch_istft_min_run.txt).  NXSIG_ISTFT_MIN_RUN overrides.
static inline int istft_min_run(const Ctx* c, int64_t total_units, int64_t slots, int dflt) {
  if (c->tuning.set[kT_ISTFT_MIN_RUN]) return (int)c->tuning.v[kT_ISTFT_MIN_RUN];
  return total_units <= 2 * slots ? 2 : dflt;
}

// Set (for the calling THREAD and one context) by the sharded log-mel of group.cpp while it runs pass 1 on a member: the clamp
// pass of stft_to_mel / the fused mel sink is NOT launched — the running maximum of the member's shard must first be
// all-reduced with the other members'; the group launches the pass afterwards.  Thread-local on purpose: another thread that
// calls nxsig_stft_mel_f32 / stft_to_mel on the same context at the same time keeps its own clamp pass, and the scope guard
// clears the mark on every way out.
struct MelDeferScope {
  explicit MelDeferScope(const Ctx* c);
  ~MelDeferScope();
  MelDeferScope(const MelDeferScope&) = delete;
  MelDeferScope& operator=(const MelDeferScope&) = delete;
};
bool mel_deferred(const Ctx* c);

int ctx_twiddles(Ctx* c, int K, const float2** out);
int ctx_table(Ctx* c, uint64_t tag, const void* host, size_t bytes, const void** out);
int ctx_scratch(Ctx* c, int slot, size_t bytes, void** out);
// device copies of a host window: raw f32[N] and the K-point table (Nx.fft(length: K) pads / truncates rows)
int ctx_window(Ctx* c, const float* window_host, int N, int K, const float** dev, const float** devK);
uint64_t fnv1a(uint64_t seed, const void* data, size_t bytes);

// ---- kernel launchers (implemented in the .hip files; all enqueue on c->stream) ----
struct StftLaunch {
  const float* x;        // device, [batch][L] rows batch_stride apart
  int64_t batch_stride;
  int32_t batch;
  Framing fr;
  int32_t K;             // fft length
  const float* window;   // device f32[N]
  const float* window_padK = nullptr;  // device f32[K] (zero-padded / truncated), used by the wave kernel
  float inv_scale_div;   // spectrum is DIVIDED by this (1.0 = none); division kept exact like the reference
  int32_t has_scale;
  float2* z;             // device c64[batch][M][K]
};
int launch_stft(Ctx* c, const StftLaunch& a);
int launch_stft_c64(Ctx* c, const StftLaunch& a);   // complex samples: a.x holds c64[batch][L] (kernels_generic.hip)

struct IstftLaunch {
  const float2* z;       // device c64[batch][M][K]
  int64_t M;
  int32_t batch;
  int32_t N, hop, K;
  const float* window;   // device f32[N]
  float scale_mul;       // frames are MULTIPLIED by this (istft :614/:617)
  int32_t has_scale;
  float2* y;             // device c64[batch][M*hop + N-hop]
  const float2* filt = nullptr;  // optional device c64[K]: every frame's spectrum is multiplied by it first (z * H of the
                                 // STFT-domain filtering chain, guides/filtering.livemd:141), rounded to c64 like Nx.multiply
  // set by a wave launcher whose kernel inverts several frames with ONE transform: the device list of units that hold a
  // non-finite bin and the frames per unit; launch_istft then recomputes those units' samples frame by frame (k_istft_edge_fix's second role)
  mutable int* nf_list = nullptr;
  mutable int nf_frames_per_unit = 0;
  // packed one-sided form (nxsig_istft_packed_f32): z is c64[batch][M][K / 2] — bins 0 .. K/2 - 1 with Re X[K/2] in the imaginary
  // part of bin 0 — and y is REAL f32[batch][M*hop + N-hop]
  bool onesided = false;
};
int launch_half_from_spectrum(Ctx* c, const float2* z, int64_t rows, int32_t K, float2* out, bool packed);
int launch_full_from_packed(Ctx* c, const float2* zp, int64_t rows, int32_t K, float2* out);
int launch_real_from_c64(Ctx* c, const float2* in, int64_t n, float* out);
int istft_nf_list(Ctx* c, int64_t capacity, int** list);
int launch_istft_fix(Ctx* c, const IstftLaunch& s, const float* window_host);   // k_istft_edge_fix: edge samples + reported non-finite units
int launch_istft(Ctx* c, const IstftLaunch& a, const float* window_host);

int launch_as_windowed(Ctx* c, const float* x, int64_t batch_stride, int32_t batch, const Framing& fr, float* out);
int launch_overlap_and_add(Ctx* c, const float* frames, int64_t M, int32_t batch, int32_t N, int32_t hop, int32_t comps,
                           float* out);
// clean: apply the Nx.fft / Nx.ifft eps clean-up to the result (false for transforms that are sub-steps of a longer one)
int launch_fft(Ctx* c, const void* in, bool in_is_real, int64_t rows, int32_t n_in, int32_t K, bool inverse, float2* out, bool clean = true);

struct FirLaunch {
  const float* x;
  int64_t L, batch_stride;
  int32_t batch;
  const float* h_host;
  int32_t taps;
  int64_t out_start, out_len;  // requested slice of the full convolution
  float* y;                    // device f32[batch][out_len]
  // device int[batch], zero on entry: a kernel that loads a non-finite sample of row r sets row_flags[r]; launch_fir's last
  // pass (k_fir_poison) then makes the whole row NaN and clears the flag.  The reference filters by ONE transform of the whole
  // row (Convolution.fftconvolve, lib/nx_signal/convolution.ex:276-284), so an Inf / NaN sample anywhere leaves no finite
  // output in that row; block-wise overlap-save would otherwise confine it to the blocks (and block pairs) that hold it.
  int* row_flags = nullptr;
};
int launch_fir(Ctx* c, const FirLaunch& a);
int launch_fir_dline(Ctx* c, const FirLaunch& a, bool* handled);   // kernels_wave_firlong.hip: 1 026 ... 32 769 taps
int fir_row_flags(Ctx* c, int32_t batch, int** out);                                                     // kernels_generic.hip
int launch_fir_poison(Ctx* c, const FirLaunch& a);
int launch_fir_flags_from_output(Ctx* c, const float* y, int32_t rows, int64_t out_len, int* flags);     // sample-sharded FIR, group.cpp
int launch_mag_from_spectrum(Ctx* c, const float2* z, int64_t rows, int32_t K, int kind, float* out);
int launch_stft_mag_wave(Ctx* c, const StftLaunch& s, int kind, float* out, bool* handled);
int launch_spectrum_mul(Ctx* c, const float2* z, int64_t rows, int32_t K, const float2* h_dev, float2* out);
int launch_stft_to_mel(Ctx* c, const float2* z, int64_t rows, int32_t K, int32_t mel_bins, const float* filters_host,
                       float* out);
int launch_stft_mel_wave(Ctx* c, const StftLaunch& s, int mel_bins, const float* filters_host, float* out, bool* handled);
// the two cells {running maximum in ordered-int encoding, non-finite flag} of the log-mel paths and their clamp pass
int launch_mel_init(Ctx* c, int** gmax);
int launch_mel_finish(Ctx* c, float* out, int64_t n, int* gmax);
// kernels_nd.hip: rows of any length (four-step / Bluestein beyond the LDS-resident kernels), device-side fft_nd, n-D fftconvolve
int launch_fft_big(Ctx* c, const void* in, bool in_is_real, int64_t rows, int64_t n_in, int64_t K, bool inverse, float2* out, bool clean = true);
int64_t fft_tiled_min(const Ctx* c);   // power-of-two row lengths from here up run the two-pass tiled four-step of kernels_nd.hip
int launch_rows_post(Ctx* c, float2* a, int64_t rows, int64_t K, const float* window, float scale, bool has_scale, float div, bool has_div);
int launch_fft_nd(Ctx* c, const void* in, bool in_is_real, const int64_t* shape, int rank, const int32_t* axes, const int64_t* lengths,
                  int n_axes, bool inverse, float2* out);
int launch_convolve_direct(Ctx* c, const void* a, bool a_is_real, const int64_t* s1, const void* b, bool b_is_real, const int64_t* s2,
                           int rank, int mode, void* out, int64_t* out_shape);
int launch_fftconvolve_nd(Ctx* c, const void* a, bool a_is_real, const int64_t* s1, const void* b, bool b_is_real, const int64_t* s2,
                          int rank, int mode, void* out, int64_t* out_shape);
int launch_stft_big(Ctx* c, const StftLaunch& s);

// ---- f64 / c128 tier (kernels_f64.hip) ----
struct StftLaunchD {
  const double* x;       // device f64[batch][L], rows batch_stride apart; x_is_complex: c128[batch][L], batch_stride in complex elements
  int32_t x_is_complex = 0;
  int64_t batch_stride;
  int32_t batch;
  Framing fr;
  int32_t K;
  const double* window;  // device f64[N]
  double div;            // the spectrum is DIVIDED by this
  int32_t has_scale;
  double2* z;            // device c128[batch][M][K]
};
struct IstftLaunchD {
  const double2* z;      // device c128[batch][M][K]
  int64_t M;
  int32_t batch, N, hop, K;
  const double* window;  // device f64[N]
  double scale_mul;
  int32_t has_scale;
  int32_t window_f32;    // the caller's window was f32: the |w|^2 normaliser is formed with f32 roundings like the reference's
  double2* y;            // device c128[batch][M*hop + N-hop]
};
struct FirLaunchD {
  const double* x;
  int64_t L, batch_stride;
  int32_t batch;
  const double* h_host;
  int32_t taps;
  int64_t out_start, out_len;
  double* y;
};
int launch_stft_f64(Ctx* c, const StftLaunchD& s);
int launch_istft_f64(Ctx* c, const IstftLaunchD& s);
int launch_fft_f64(Ctx* c, const void* in, bool in_is_real, int64_t rows, int32_t n_in, int32_t K, bool inverse, double2* out,
                   const double* post_window = nullptr, double post_scale = 1.0, bool has_post_scale = false);
int launch_ola_f64(Ctx* c, const double* frames, int64_t M, int32_t batch, int32_t N, int32_t hop, int32_t comps, const double* window,
                   bool norm, bool window_f32, double* out);
int launch_as_windowed_f64(Ctx* c, const double* x, int64_t batch_stride, int32_t batch, const Framing& f, double* out);
int launch_fir_f64(Ctx* c, const FirLaunchD& s);
int launch_fftconvolve_c64(Ctx* c, const float2* a, int64_t n1, const float2* b, int64_t n2, int64_t start, int64_t len,
                           float2* out);

}  // namespace nxsig
