/*
 * nxsig.h — C ABI of the MI355X-native STFT / iSTFT / FIR hot path behind the NxSignal API.
 *
 * This is the drop-in boundary: everything a host language needs
 * (the Elixir dirty-NIF shim in nif/nxsig_nif.c, the Python mirror in nx_signal_amd/, the C++
 * tests) goes through these `extern "C"` entry points — plain pointers and sizes, no torch /
 * C++ types.  The reference has no native code; each entry point replaces the Nx primitives
 * the cited reference lines compose on Nx.BinaryBackend (file:line are relative to the
 * reference root, elixir-nx/nx_signal v0.3.0).
 *
 * Conventions
 *   - every function returns NXSIG_OK (0) or a negative nxsig_status; a human readable message
 *     for the calling thread is available from nxsig_last_error().  Nothing aborts, throws or
 *     longjmps across this boundary (dirty-NIF safe).
 *   - buffers are caller-allocated and never retained after the call returns.
 *   - `mem` says where the SIGNAL buffers (x, z, y ...) live: NXSIG_HOST (the library stages
 *     them through HBM — PCIe-bound, convenience path) or NXSIG_DEVICE (HBM pointers from
 *     nxsig_alloc or any other HIP allocator — the hot path; the call is asynchronous on the
 *     context's stream and returns once the kernels are enqueued).
 *   - small parameter tables (window, FIR taps) are ALWAYS host pointers: they are user-built
 *     tensors of a few KB, hashed and cached in HBM per context.
 *   - c64 = interleaved little-endian (f32 re, f32 im), row-major — Nx's native layout.
 *   - a context is bound to one GPU and one HIP stream; calls on one context are serialised
 *     by an internal mutex, different contexts may be used concurrently from any OS thread.
 */
#ifndef NXSIG_H
#define NXSIG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NXSIG_ABI_VERSION 1

typedef struct nxsig_ctx nxsig_ctx;

typedef struct nxsig_c64 {
  float re, im;
} nxsig_c64;

typedef struct nxsig_c128 {
  double re, im;
} nxsig_c128;

typedef enum nxsig_status {
  NXSIG_OK = 0,
  NXSIG_ERR_INVALID_ARG = -1, /* maps to ArgumentError on the Elixir side */
  NXSIG_ERR_UNSUPPORTED = -2, /* valid in the reference, not built here yet (message says what) */
  NXSIG_ERR_HIP = -3,         /* a HIP runtime call failed */
  NXSIG_ERR_NO_DEVICE = -4,
  NXSIG_ERR_OOM = -5
} nxsig_status;

typedef enum nxsig_mem { NXSIG_HOST = 0, NXSIG_DEVICE = 1 } nxsig_mem;

/* as_windowed padding modes — lib/nx_signal.ex:303-331, :343-352 */
typedef enum nxsig_pad {
  NXSIG_PAD_VALID = 0,    /* :valid (default of stft, :76) */
  NXSIG_PAD_REFLECT = 1,  /* :reflect — div(N,2) each side, mirror without edge repeat (:262, :348-349) */
  NXSIG_PAD_SAME = 2,     /* :same — k-1 total, floor left / ceil right (:308-312) */
  NXSIG_PAD_EXPLICIT = 3  /* [{lo, hi}] zero padding (negative = crop, as Nx.pad) (:314-323) */
} nxsig_pad;

/* stft/istft :scaling — lib/nx_signal.ex:113-127, :611-625 */
typedef enum nxsig_scaling { NXSIG_SCALE_NONE = 0, NXSIG_SCALE_SPECTRUM = 1, NXSIG_SCALE_PSD = 2 } nxsig_scaling;

/* NxSignal.Windows — lib/nx_signal/windows.ex */
typedef enum nxsig_window_kind {
  NXSIG_WIN_RECTANGULAR = 0, /* :33  */
  NXSIG_WIN_BARTLETT = 1,    /* :57  */
  NXSIG_WIN_TRIANGULAR = 2,  /* :98  */
  NXSIG_WIN_BLACKMAN = 3,    /* :160 */
  NXSIG_WIN_HAMMING = 4,     /* :225 */
  NXSIG_WIN_HANN = 5,        /* :278 */
  NXSIG_WIN_KAISER = 6       /* :341 */
} nxsig_window_kind;

/* Convolution modes — lib/nx_signal/convolution.ex:300-329 */
typedef enum nxsig_conv_mode { NXSIG_CONV_FULL = 0, NXSIG_CONV_SAME = 1, NXSIG_CONV_VALID = 2 } nxsig_conv_mode;

/* options of NxSignal.stft/3 (lib/nx_signal.ex:71-85) and istft/3 (:583) after default resolution */
typedef struct nxsig_stft_params {
  int32_t frame_length;  /* N = size of the window tensor (:69) */
  int32_t hop;           /* N - overlap_length (:99, :702) */
  int32_t fft_length;    /* K; :power_of_two already resolved by the caller (nxsig_next_pow2) */
  int32_t pad_mode;      /* nxsig_pad (stft only) */
  int64_t pad_lo, pad_hi;/* NXSIG_PAD_EXPLICIT only */
  int32_t scaling;       /* nxsig_scaling */
  int32_t reserved;
  double sampling_rate;  /* used by :psd scaling only */
} nxsig_stft_params;

/* ---------------------------------------------------------------- context / errors ---- */
int nxsig_abi_version(void);
int nxsig_device_count(int* count);
int nxsig_ctx_create(int device, nxsig_ctx** out);
void nxsig_ctx_destroy(nxsig_ctx* ctx);
/* Dispatch / geometry switches of ONE context ("Switches").  nxsig_ctx_create reads the NXSIG_<NAME> variables of
 * the process environment once; after that no call looks at the environment — a launch only reads the context's switches, and
 * these three functions change them at run time (A/B sweeps, tests).  `name` is "NXSIG_FOO" or "FOO"; unknown names are
 * NXSIG_ERR_INVALID_ARG.  clear(name = NULL) returns every switch to the launchers' defaults. */
int nxsig_ctx_set_tuning(nxsig_ctx* ctx, const char* name, int32_t value);
int nxsig_ctx_get_tuning(nxsig_ctx* ctx, const char* name, int32_t* value, int32_t* is_set);
int nxsig_ctx_clear_tuning(nxsig_ctx* ctx, const char* name);
const char* nxsig_last_error(void); /* thread-local, valid until the next call on this thread */
/* Which kernel families did the calling thread's LAST compute call (stft / istft / fir / fft / convolve / mel ...) launch?  A
 * '+'-separated list in launch order, each family once, e.g. "stft.pair", "stft.pair.1r+stft.pair.edge", "istft.wave+istft.edge_chunks",
 * "fir.pair+fir.poison", "stft.generic".  The names are the kernel families of DESIGN.md section 3; a shape that silently falls from a tuned
 * kernel to the generic ones shows here (tests/test_gpu_dispatch_table.py pins the family of every documented shape).  Thread-local,
 * valid until the next compute call on this thread; "" before the first one.  Diagnostic only: nothing in the library reads it. */
const char* nxsig_last_dispatch(void);
/* the same record of the last compute call made on THIS CONTEXT by any thread, copied into buf (truncated to buflen - 1 characters): what a
 * host whose calls hop between threads (the BEAM's dirty schedulers) reads */
int nxsig_ctx_last_dispatch(nxsig_ctx* ctx, char* buf, size_t buflen);
/* human readable device line ("AMD Instinct MI355X gfx950 256 CUs") into buf */
int nxsig_device_name(nxsig_ctx* ctx, char* buf, size_t buflen);

/* ---------------------------------------------------------------- memory / stream ----- */
/* Device memory of the context's GPU through a caching allocator: nxsig_free parks the block (no hipFree, no device
 * synchronisation) and a later nxsig_alloc of a similar size gets it back — a fresh 3.3 GB hipMalloc costs ~100 ms here, two
 * hundred times the stft that fills it.  Reuse is ordered by the context's stream; callers that touch a buffer on another
 * stream synchronise before freeing it.  The cache holds at most NXSIG_POOL_MAX_MB (default: a quarter of the device memory,
 * 0 = no caching) and is emptied when an allocation fails.  Blocks belong to their context and die with it.  nxsig_free also
 * accepts pointers from other HIP allocators (plain hipFree after the stream has drained). */
int nxsig_alloc(nxsig_ctx* ctx, size_t bytes, void** dptr);
int nxsig_free(nxsig_ctx* ctx, void* dptr);
int nxsig_upload(nxsig_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);   /* synchronous */
int nxsig_download(nxsig_ctx* ctx, void* dst_host, const void* src_device, size_t bytes); /* synchronous */
int nxsig_sync(nxsig_ctx* ctx);                       /* wait for everything enqueued on the ctx stream */
/* free / total memory of the context's GPU right now (hipMemGetInfo): bench.py's preflight reads the high-water mark off it */
int nxsig_mem_info(nxsig_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
int nxsig_set_stream(nxsig_ctx* ctx, void* hip_stream); /* adopt a caller-owned hipStream_t (NULL = own stream) */
void* nxsig_get_stream(nxsig_ctx* ctx);
/* HIP-event stopwatch on the ctx stream (used by bench.py: events live on the stream the kernels run on) */
int nxsig_timer_start(nxsig_ctx* ctx);
int nxsig_timer_stop(nxsig_ctx* ctx, float* elapsed_ms); /* records + synchronises the stop event */

/* ---------------------------------------------------------------- shape helpers (pure) - */
int32_t nxsig_next_pow2(int32_t n); /* :power_of_two of Nx.fft */
/* number of frames of as_windowed — lib/nx_signal.ex:289-298; negative status on error */
int64_t nxsig_num_frames(int64_t length, int32_t frame_length, int32_t hop, int32_t pad_mode, int64_t pad_lo,
                         int64_t pad_hi);
/* output length of overlap_and_add / istft: M*hop + (N-hop) — lib/nx_signal.ex:703 */
int64_t nxsig_ola_length(int64_t num_frames, int32_t frame_length, int32_t hop);
/* output length of convolve for a mode — lib/nx_signal/convolution.ex:300-347 */
int64_t nxsig_conv_length(int64_t n1, int64_t n2, int32_t mode);

/* ------------------------------------------ host-side generators (BinaryBackend rounding) */
/* NxSignal.Windows.* (n, is_periodic:, type: f32) — lib/nx_signal/windows.ex; beta/eps: kaiser only */
int nxsig_window_f32(int32_t kind, int32_t n, int32_t is_periodic, double beta, double eps, float* out);
/* NxSignal.Waveforms.sinc — lib/nx_signal/waveforms.ex:451-457 */
int nxsig_sinc_f32(const float* t, int64_t n, float* out);
/* NxSignal.Filters.firwin/3 — lib/nx_signal/filters.ex:147-279 (cutoff in sampling_rate units) */
int nxsig_firwin_f32(int32_t num_taps, const double* cutoff, int32_t n_cutoff, int32_t window_kind, double kaiser_beta,
                     int32_t pass_zero, int32_t scale, double sampling_rate, float* out);
/* NxSignal.fft_frequencies/2 — lib/nx_signal.ex:154-166 */
int nxsig_fft_frequencies_f32(double sampling_rate, int32_t fft_length, int32_t endpoint, float* out);
/* the `times` output of stft — lib/nx_signal.ex:108-111 */
int nxsig_stft_times_f32(int32_t frame_length, double sampling_rate, int64_t num_frames, float* out);

/* ---------------------------------------------------------------- the hot path -------- */
/*
 * NxSignal.stft/3 — lib/nx_signal.ex:68-130 (as_windowed :94-100, Nx.multiply :101, Nx.fft :102,
 * scaling :113-127) fused in one launch.
 *   x      f32[batch][length], rows `batch_stride` elements apart (batch = vectorized channels, B16)
 *   window f32[N] HOST
 *   z      c64[batch][M][K]
 *   M      = nxsig_num_frames(length, N, hop, pad...) written to *num_frames_out (may be NULL)
 */
int nxsig_stft_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride,
                   const float* window, const nxsig_stft_params* params, nxsig_c64* z, int64_t* num_frames_out,
                   int32_t mem);

/*
 * NxSignal.stft/3 of COMPLEX samples (c64 IQ data) — lib/nx_signal.ex:94-102 frames, multiplies (:101, c64 x f32 componentwise) and
 * transforms (:102, one Nx.fft row per frame: no pair packing) whatever tensor it is given.  Same parameters and outputs as
 * nxsig_stft_f32 with x c64[batch][length] (interleaved re, im), rows `batch_stride` COMPLEX elements apart.
 */
int nxsig_stft_c64(nxsig_ctx* ctx, const nxsig_c64* x, int64_t length, int32_t batch, int64_t batch_stride,
                   const float* window, const nxsig_stft_params* params, nxsig_c64* z, int64_t* num_frames_out,
                   int32_t mem);

/*
 * NxSignal.istft/3 — lib/nx_signal.ex:582-638 (Nx.ifft :609, inverse scaling :611-625, x window and
 * overlap_and_add :627-628, OLA(|w|^2) normaliser with the 1e-10 guard :630-637).
 *   z c64[batch][M][K]  ->  y c64[batch][M*hop + N-hop]   (complex output)
 *   requires fft_length == frame_length (the reference's broadcast {M,K} x {N} requires it too).
 */
int nxsig_istft_c64(nxsig_ctx* ctx, const nxsig_c64* z, int64_t num_frames, int32_t batch, const float* window,
                    const nxsig_stft_params* params, nxsig_c64* y, int32_t mem);

/* NxSignal.as_windowed/2 — lib/nx_signal.ex:249-364: f32[batch][length] -> f32[batch][M][N] */
int nxsig_as_windowed_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride,
                          int32_t window_length, int32_t stride, int32_t pad_mode, int64_t pad_lo, int64_t pad_hi,
                          float* out, int64_t* num_frames_out, int32_t mem);

/* NxSignal.overlap_and_add/2 — lib/nx_signal.ex:684-736: [batch][M][N] -> [batch][M*hop + N-hop].
 * `components` = 1 for f32 tensors, 2 for c64 (interleaved); deterministic (fixed frame order). */
int nxsig_overlap_and_add(nxsig_ctx* ctx, const float* frames, int64_t num_frames, int32_t batch, int32_t frame_length,
                          int32_t overlap_length, int32_t components, float* out, int32_t mem);

/* Nx.fft / Nx.ifft(length: K) over the last axis as NxSignal.Transforms.fft_nd / ifft_nd reach them
 * (lib/nx_signal/transforms.ex:5-21): rows of n_in elements are zero-padded / truncated to K.
 * in: c64[rows][n_in] (in_is_real = 0) or f32[rows][n_in] (in_is_real = 1); out: c64[rows][K]. */
int nxsig_fft(nxsig_ctx* ctx, const void* in, int32_t in_is_real, int64_t rows, int32_t n_in, int32_t fft_length,
              int32_t inverse, nxsig_c64* out, int32_t mem);

/*
 * NxSignal.Transforms.fft_nd / ifft_nd — lib/nx_signal/transforms.ex:5-21: Enum.zip_reduce(axes, lengths, t, &Nx.fft(&3, axis: &1,
 * length: &2)) (or Nx.ifft), entirely in HBM: an axis other than the last is brought to the back by a tiled transpose kernel,
 * transformed by the row kernels and moved back.  in: f32 (in_is_real) or c64 tensor of `rank` <= 8 dims, row-major; out: c64 tensor
 * whose dims at `axes` are `lengths` (zero-pad / truncate like Nx.fft(length:)).  Row lengths: any power of two up to 2^26
 * (four-step beyond 8192), any other length up to 2^22 (Bluestein).  `out` must not alias `in`.
 */
int nxsig_fft_nd(nxsig_ctx* ctx, const void* in, int32_t in_is_real, const int64_t* shape, int32_t rank, const int32_t* axes,
                 const int64_t* lengths, int32_t n_axes, int32_t inverse, nxsig_c64* out, int32_t mem);

/*
 * n-D Convolution.fftconvolve/3 — lib/nx_signal/convolution.ex:252-347: fft_nd of both operands over the axes where neither
 * dimension is 1 (lengths s1 + s2 - 1), broadcast product, ifft_nd, Nx.real for real operands, `centered` slice per mode
 * (:valid needs one operand at least as large as the other in every dimension, :335-347).  a, b: tensors of equal rank <= 8
 * (f32 when *_is_real, else c64).  out: f32 when both are real, else c64; its shape is written to out_shape[rank] (may be
 * NULL): full s1 + s2 - 1, same s1, valid |s1 - s2| + 1.
 */
int nxsig_fftconvolve_nd(nxsig_ctx* ctx, const void* a, int32_t a_is_real, const int64_t* a_shape, const void* b, int32_t b_is_real,
                         const int64_t* b_shape, int32_t rank, int32_t mode, void* out, int64_t* out_shape, int32_t mem);

/*
 * Convolution.convolve/3 with `method: :direct` (the reference's DEFAULT method) — lib/nx_signal/convolution.ex:38-58, :95-218:
 * Nx.conv of in1, zero-padded per mode (:full k-1 on both sides; :same (k-1) - div(k-1, 2) left, div(k-1, 2) right; :valid none,
 * the larger operand becomes the volume, :120-135), with in2 reversed along every axis.  Time-domain sums, products accumulated
 * in double in the BinaryBackend's window order and rounded once: integer-valued inputs come out exact (the reference's tests
 * compare with ==).  O(output x kernel) work — meant for short kernels; long filters belong to nxsig_fir_f32 / the FFT method.
 * Same argument convention as nxsig_fftconvolve_nd; out shape: full s1 + s2 - 1, same s1, valid |s1 - s2| + 1.
 */
int nxsig_convolve_direct(nxsig_ctx* ctx, const void* a, int32_t a_is_real, const int64_t* a_shape, const void* b, int32_t b_is_real,
                          const int64_t* b_shape, int32_t rank, int32_t mode, void* out, int64_t* out_shape, int32_t mem);

/*
 * FIR filtering: y = Convolution.convolve(x, h, method: :fft, mode:) for real 1-D x (per batch row) and
 * real taps h — lib/nx_signal/convolution.ex:252-329 as used by guides/filtering.livemd:126-128 —
 * computed by overlap-save block FFT convolution (the `Filters.fir` of BASELINE config 5; the reference
 * does one length-(L+K-1) FFT, results agree to fp32 rounding).  Filters of any length: up to 1025 taps the tuned wave kernels,
 * up to 4096 taps the generic overlap-save kernel, beyond that (impulse responses of seconds) one transform of
 * next_pow2(length + num_taps - 1) <= 2^26 points per row like the reference.
 *   x f32[batch][length], h f32[num_taps] HOST, y f32[batch][nxsig_conv_length(length, num_taps, mode)]
 */
int nxsig_fir_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* h,
                  int32_t num_taps, int32_t mode, float* y, int32_t mem);

/* Any slice [out_start, out_start + out_len) of the FULL convolution of every row with h (the modes of nxsig_fir_f32 are
 * three such slices; sharded filtering asks for the slice a rank owns): y f32[batch][out_len].
 * Non-finite samples: the reference filters a row by ONE transform (convolution.ex:276-284), so a row that holds an Inf / NaN has no
 * finite output, and nxsig_fir_f32 returns such a row as NaN from end to end.  A SLICE call only looks at the blocks its outputs
 * need: it returns NaN for the whole slice when one of THOSE samples is not finite, and finite values when the non-finite sample
 * lies elsewhere in the row — so the slices of a row assembled BY THE CALLER can be finite where the unsliced call is NaN
 * (nxsig_fir_sharded_f32 on the samples axis closes that itself: its members exchange one flag per row).  Callers that need the reference's whole-row behaviour for sliced rows test the row
 * themselves; finite rows are unaffected. */
int nxsig_fir_slice_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* h,
                        int32_t num_taps, int64_t out_start, int64_t out_len, float* y, int32_t mem);

/*
 * NxSignal.mel_filters/4 — lib/nx_signal.ex:397-445 (Slaney-style filterbank, host-side with BinaryBackend rounding):
 * out f32[mel_bins][fft_length].  Defaults of the reference: max_mel 3016, mel_frequency_spacing 200/3.
 */
int nxsig_mel_filters_f32(int32_t fft_length, int32_t mel_bins, double sampling_rate, double max_mel,
                          double mel_frequency_spacing, float* out);

/*
 * NxSignal.stft_to_mel/3 — lib/nx_signal.ex:486-513: |z|^2 of the first fft_length/2 bins x filterbank
 * -> log10(max(., 1e-10)) -> max(., global max - 8) -> (. + 4) / 4.   z c64[rows][fft_length] (rows = every frame of
 * every batch element: the reference's reduce_max runs over the whole tensor) -> out f32[rows][mel_bins].
 * `filters` f32[mel_bins][fft_length] is a HOST table (nxsig_mel_filters_f32 or user supplied).
 */
int nxsig_stft_to_mel(nxsig_ctx* ctx, const nxsig_c64* z, int64_t rows, int32_t fft_length, int32_t mel_bins,
                      const float* filters, float* out, int32_t mem);

/*
 * Fused NxSignal.stft/3 -> NxSignal.stft_to_mel/3: the log-mel spectrogram of x without ever writing the
 * complex spectrum to HBM (mel_bins * 4 B/frame leave the chip instead of fft_length * 8).  Same result as
 * nxsig_stft_f32 followed by nxsig_stft_to_mel, to fp32 rounding.  x f32[batch][length] -> out f32[batch][M][mel_bins];
 * the global maximum runs over the whole output.  Shapes the fused kernel does not cover run the two-step path.
 */
int nxsig_stft_mel_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride,
                       const float* window, const nxsig_stft_params* params, int32_t mel_bins, const float* filters,
                       float* out, int64_t* num_frames_out, int32_t mem);

/*
 * One-sided complex spectrum (opt-in; not in the reference API): the bins 0 .. fft_length/2 - 1 of NxSignal.stft/3 —
 * the slice the reference's own downstream code keeps for real signals (stft_to_mel: lib/nx_signal.ex:493-496; the spectrogram
 * guide: guides/spectrogram.livemd:80-82) — written straight from the transform: 4 KB per frame instead of 8 at fft_length 1024,
 * identical bits to the first half of nxsig_stft_f32's rows.  out c64[batch][M][fft_length / 2].
 */
int nxsig_stft_onesided_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* window,
                            const nxsig_stft_params* params, nxsig_c64* out, int64_t* num_frames_out, int32_t mem);

/*
 * The PACKED one-sided pair (opt-in; not in the reference API): the reference's STFT-domain filtering chain
 * stft -> z * H -> istft (guides/filtering.livemd:137-159) moves the full two-sided spectrum, 8 KB + 2 KB of HBM traffic per
 * frame at fft_length 1024, although for a real signal half of z mirrors the other half and the imaginary part of the istft
 * result is round-off (callers take Nx.real / Nx.as_type, lib/nx_signal.ex:550).  The packed pair keeps what is independent:
 *   nxsig_stft_packed_f32    out c64[batch][M][fft_length / 2]: bins 0 .. fft_length/2 - 1 exactly as nxsig_stft_onesided_f32
 *                            writes them, except that the imaginary part of bin 0 (zero for a real frame) carries
 *                            Re X[fft_length / 2], the Nyquist bin (real for a real frame) — nothing of a real frame's
 *                            spectrum is lost.  fft_length even.  4 KB per frame.
 *   nxsig_istft_packed_f32   the inverse of exactly that layout (a pointwise product z * H of two packed tensors must treat
 *                            bin 0 as the two reals it is): y REAL f32[batch][M * hop + frame_length - hop], equal to the real part
 *                            of nxsig_istft_c64 on the full Hermitian spectrum to fp32 round-off.  1 KB per frame.
 * frame_length == fft_length as for istft; fused kernel for 1024 / hop 128 ... 1024, other shapes go through the full layout.
 */
int nxsig_stft_packed_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride, const float* window,
                          const nxsig_stft_params* params, nxsig_c64* out, int64_t* num_frames_out, int32_t mem);
int nxsig_istft_packed_f32(nxsig_ctx* ctx, const nxsig_c64* z, int64_t num_frames, int32_t batch, const float* window,
                           const nxsig_stft_params* params, float* y, int32_t mem);

/*
 * Magnitude spectrogram fused with the STFT (opt-in, not in the reference API): what
 * guides/spectrogram.livemd:76-92 computes from NxSignal.stft/3 — Nx.abs(s) of the bins below fft_length / 2, optionally
 * as dBFS 20 * log(|s| / reduce_max|s|) / log(10) — without writing the complex spectrum to HBM: fft_length * 2 bytes per
 * frame leave the chip instead of fft_length * 8.   x f32[batch][length] -> out f32[batch][M][fft_length / 2].
 * kind: NXSIG_MAG_ABS |s|, NXSIG_MAG_POWER |s|^2, NXSIG_MAG_DBFS (the maximum runs over the whole output).
 */
#define NXSIG_MAG_ABS 0
#define NXSIG_MAG_POWER 1
#define NXSIG_MAG_DBFS 2
int nxsig_stft_magnitude_f32(nxsig_ctx* ctx, const float* x, int64_t length, int32_t batch, int64_t batch_stride,
                             const float* window, const nxsig_stft_params* params, int32_t kind, float* out,
                             int64_t* num_frames_out, int32_t mem);

/*
 * STFT-domain filtering kept on the device: the `Nx.multiply(z, hfft)` step of the reference's documented
 * workflow stft -> z * H -> istft (guides/filtering.livemd:137-159).  out[r][k] = z[r][k] * h[k], each component
 * computed in double and rounded once like Nx.BinaryBackend's complex multiply.  z, out c64[rows][fft_length] (out may
 * alias z); `h` c64[fft_length] is a HOST table (the DFT of the filter, e.g. from nxsig_fft).
 */
int nxsig_spectrum_mul_c64(nxsig_ctx* ctx, const nxsig_c64* z, int64_t rows, int32_t fft_length, const nxsig_c64* h,
                           nxsig_c64* out, int32_t mem);

/*
 * The last two steps of that workflow in one call: NxSignal.istft(Nx.multiply(z, hfft), window, opts)
 * (guides/filtering.livemd:141 + :150-157).  Same result, bit for bit, as nxsig_spectrum_mul_c64 followed by nxsig_istft_c64
 * (the product is formed in double and rounded to c64 before the inverse transform sees it), but for N = 1024 the filter
 * values ride in the registers of the inverse-STFT kernel and the filtered spectrogram is never written to or re-read from
 * HBM (26 KB of traffic per frame -> 10 at hop 256).  Other sizes run the two steps behind this entry.  z is not modified.
 *   z c64[batch][M][K], h c64[K] HOST  ->  y c64[batch][M*hop + N-hop]
 */
int nxsig_istft_filtered_c64(nxsig_ctx* ctx, const nxsig_c64* z, int64_t num_frames, int32_t batch, const float* window,
                             const nxsig_stft_params* params, const nxsig_c64* h, nxsig_c64* y, int32_t mem);

/*
 * 1-D complex case of Convolution.fftconvolve/3 — lib/nx_signal/convolution.ex:252-329 (tests: "FFT complex",
 * test/nx_signal/convolutions_test.exs:473-487): out = ifft(fft(a, P) * fft(b, P)) sliced per mode, with
 * P = next power of two >= n1 + n2 - 1 (same linear convolution as the reference's length n1 + n2 - 1).
 *   a c64[n1], b c64[n2], out c64[nxsig_conv_length(n1, n2, mode)]; n1 + n2 - 1 <= 2^26 (four-step transforms beyond 8192 points).
 */
int nxsig_fftconvolve_c64(nxsig_ctx* ctx, const nxsig_c64* a, int64_t n1, const nxsig_c64* b, int64_t n2, int32_t mode,
                          nxsig_c64* out, int32_t mem);

/* ---------------------------------------------------------------- f64 / c128 tier --------
 * The reference computes in the type of its operands: f64 samples or an f64 window promote the product of lib/nx_signal.ex:101
 * to f64 and Nx.fft returns c128 (:102); a c128 spectrum is inverted in c128 (:609); Windows.* / firwin / fft_frequencies take
 * `type: {:f, 64}` (lib/nx_signal/windows.ex:58, :99 ..., filters.ex:154, nx_signal.ex:155).  These entry points are that
 * tier: the argument conventions of their f32 namesakes with double / nxsig_c128 payloads, double arithmetic on the device
 * (workgroup-per-frame kernels, kernels_f64.hip — the tuned wave kernels stay the f32 path).  Limits: fft_length a power of two
 * <= 8192, any other length <= 4096 (Bluestein) or <= 65536 with at most 8192 samples per row (table DFT); FIR up to 4097 taps.
 * NXSIG_ERR_UNSUPPORTED beyond.  `window` is f64[N] when window_is_f64, else f32[N]: the reference forms the :scaling
 * scalar and the |w|^2 normaliser in the window's OWN type before promoting (Nx.sum(window) :116, :614; :630-633).
 */
/* NxSignal.Windows.*(n, type: {:f, 64}) — lib/nx_signal/windows.ex: every op of the f32 generator rounded to double instead */
int nxsig_window_f64(int32_t kind, int32_t n, int32_t is_periodic, double beta, double eps, double* out);
/* NxSignal.Waveforms.sinc on an f64 tensor — lib/nx_signal/waveforms.ex:451-457 (its pi() stays the f32 constant) */
int nxsig_sinc_f64(const double* t, int64_t n, double* out);
/* NxSignal.Filters.firwin(..., type: {:f, 64}) — lib/nx_signal/filters.ex:147-279 */
int nxsig_firwin_f64(int32_t num_taps, const double* cutoff, int32_t n_cutoff, int32_t window_kind, double kaiser_beta,
                     int32_t pass_zero, int32_t scale, double sampling_rate, double* out);
/* NxSignal.fft_frequencies(fs, type: {:f, 64}) — lib/nx_signal.ex:154-166 */
int nxsig_fft_frequencies_f64(double sampling_rate, int32_t fft_length, int32_t endpoint, double* out);
/* NxSignal.stft/3 on f64 samples — lib/nx_signal.ex:68-130: x f64[batch][length] -> z c128[batch][M][K] */
int nxsig_stft_f64(nxsig_ctx* ctx, const double* x, int64_t length, int32_t batch, int64_t batch_stride, const void* window,
                   int32_t window_is_f64, const nxsig_stft_params* params, nxsig_c128* z, int64_t* num_frames_out, int32_t mem);
/* NxSignal.stft/3 on COMPLEX f64 samples (c128 IQ data) — lib/nx_signal.ex:94-102 on whatever tensor it is given: frames, c128 x window
 * componentwise (:101), one Nx.fft row per frame (:102).  x c128[batch][length], rows `batch_stride` COMPLEX elements apart; the other
 * arguments as nxsig_stft_f64 */
int nxsig_stft_c128(nxsig_ctx* ctx, const nxsig_c128* x, int64_t length, int32_t batch, int64_t batch_stride, const void* window,
                    int32_t window_is_f64, const nxsig_stft_params* params, nxsig_c128* z, int64_t* num_frames_out, int32_t mem);
/* NxSignal.istft/3 on a c128 spectrum — lib/nx_signal.ex:582-638: z c128[batch][M][K] -> y c128[batch][M*hop + N-hop] */
int nxsig_istft_c128(nxsig_ctx* ctx, const nxsig_c128* z, int64_t num_frames, int32_t batch, const void* window, int32_t window_is_f64,
                     const nxsig_stft_params* params, nxsig_c128* y, int32_t mem);
/* Nx.fft / Nx.ifft(length: K) over rows in c128 — lib/nx_signal/transforms.ex:5-21: in f64 (in_is_real) or c128 */
int nxsig_fft_c128(nxsig_ctx* ctx, const void* in, int32_t in_is_real, int64_t rows, int32_t n_in, int32_t fft_length, int32_t inverse,
                   nxsig_c128* out, int32_t mem);
/* NxSignal.as_windowed/2 of an f64 tensor — lib/nx_signal.ex:249-364 */
int nxsig_as_windowed_f64(nxsig_ctx* ctx, const double* x, int64_t length, int32_t batch, int64_t batch_stride, int32_t window_length,
                          int32_t stride, int32_t pad_mode, int64_t pad_lo, int64_t pad_hi, double* out, int64_t* num_frames_out,
                          int32_t mem);
/* NxSignal.overlap_and_add/2 of an f64 (components = 1) or c128 (2) tensor — lib/nx_signal.ex:684-736 */
int nxsig_overlap_and_add_f64(nxsig_ctx* ctx, const double* frames, int64_t num_frames, int32_t batch, int32_t frame_length,
                              int32_t overlap_length, int32_t components, double* out, int32_t mem);
/* Convolution.convolve(x, h, method: :fft, mode:) of f64 rows with f64 taps — lib/nx_signal/convolution.ex:252-329 */
int nxsig_fir_f64(nxsig_ctx* ctx, const double* x, int64_t length, int32_t batch, int64_t batch_stride, const double* h,
                  int32_t num_taps, int32_t mode, double* y, int32_t mem);
int nxsig_fir_slice_f64(nxsig_ctx* ctx, const double* x, int64_t length, int32_t batch, int64_t batch_stride, const double* h,
                        int32_t num_taps, int64_t out_start, int64_t out_len, double* y, int32_t mem);

/* ============================================================== multi-GPU groups ====
 * The path shards with NO data-path exchange: channels (the reference's vectorized axes,
 * lib/nx_signal.ex:358-363) are independent and frames are independent given their samples, so rank r owns a
 * contiguous block of channels, or a contiguous frame range of one long stream plus an (N - hop)-sample input halo it
 * reads redundantly.  The only collective is the OPTIONAL final assembly: an RCCL all-gather over xGMI of the output
 * shards.  A group is either
 *   - LOCAL: one process drives every GPU of the node (the Elixir / dirty-NIF host): one nxsig_ctx + stream per device,
 *     communicators from ncclCommInitAll, collectives fused with ncclGroupStart / ncclGroupEnd; or
 *   - RANKED: one process per GPU (bench.py under a process launcher): ncclCommInitRank, the 128-byte ncclUniqueId
 *     travelling from rank 0 through a file on the node (nxsig_rendezvous_*).
 * librccl is dlopen()ed when the first group is created; a process that never creates a group never loads it.
 * Members that share one device (testing / replicas on a single GPU) get no communicator: their assembly runs as
 * device-to-device copies (LOCAL groups only).
 */
typedef struct nxsig_group nxsig_group;

typedef enum nxsig_shard_axis {
  NXSIG_SHARD_CHANNELS = 0, /* batch rows split into contiguous blocks (BASELINE configs 4 / 5) */
  NXSIG_SHARD_FRAMES = 1    /* frame ranges of each row, input halo read redundantly (one long stream) */
} nxsig_shard_axis;

/* contiguous near-equal split of range(total): the first total % parts members get one extra item (pure) */
int nxsig_shard_range(int64_t total, int32_t parts, int32_t index, int64_t* begin, int64_t* end);
/* frame range [m0, m1) of member `index` of a :valid-framed stream of `num_frames` frames and the sample span
 * [s0, s1) it reads: s0 = m0 * hop, s1 = (m1 - 1) * hop + frame_length (pure) */
int nxsig_shard_frames(int64_t num_frames, int32_t frame_length, int32_t hop, int32_t parts, int32_t index, int64_t* m0,
                       int64_t* m1, int64_t* s0, int64_t* s1);
/* output range [n0, n1) of member `index` of a FIR whose full-convolution slice starts at `out_start` (mode offset) and
 * the input span [s0, s1), clamped to [0, length), it reads: num_taps - 1 samples of halo (pure) */
int nxsig_shard_fir(int64_t length, int32_t num_taps, int32_t mode, int32_t parts, int32_t index, int64_t* n0, int64_t* n1,
                    int64_t* s0, int64_t* s1);

/* iSTFT over frame ranges: member `index` keeps output samples [n0, n1) = [m0 hop, m1 hop) of its frame share (the last member
 * also the tail up to M hop + N - hop) and needs input frames [f0, f1) = [m0 - (ceil(N / hop) - 1), m1) widened to multiples of
 * 8 frames (the kernels' frame groups): the halo FRAMES are recomputed instead of exchanging partial overlap-add sums (pure) */
int nxsig_shard_istft(int64_t num_frames, int32_t frame_length, int32_t hop, int32_t parts, int32_t index, int64_t* f0, int64_t* f1,
                      int64_t* n0, int64_t* n1);

/* file rendezvous on one node (pure host code): publish writes `bytes` bytes atomically (tmp file + rename), fetch polls
 * until the file exists, is complete and is younger than `max_age_s` seconds (stale files of earlier runs are ignored) */
int nxsig_rendezvous_publish(const char* path, const void* data, size_t bytes);
int nxsig_rendezvous_fetch(const char* path, void* data, size_t bytes, int32_t timeout_ms, int32_t max_age_s);

/* LOCAL group over `n` devices of this process (device_ids NULL = 0 .. n-1) */
int nxsig_group_create_local(int32_t n, const int32_t* device_ids, nxsig_group** out);
/* RANKED group: this process is rank `rank` of `world` and drives `device`.  Rank 0 creates the ncclUniqueId and
 * publishes it at `rendezvous_path`, the other ranks fetch it there (timeout_ms); world == 1 needs no path. */
int nxsig_group_create_rank(int32_t world, int32_t rank, int32_t device, const char* rendezvous_path, int32_t timeout_ms,
                            nxsig_group** out);
void nxsig_group_destroy(nxsig_group* g);
int32_t nxsig_group_world(const nxsig_group* g);        /* ranks in the group */
int32_t nxsig_group_local_count(const nxsig_group* g);  /* members driven by this process */
int32_t nxsig_group_rank(const nxsig_group* g, int32_t local_index);   /* global rank of a local member */
nxsig_ctx* nxsig_group_ctx(nxsig_group* g, int32_t local_index);       /* its context (owned by the group) */
int32_t nxsig_group_has_rccl(const nxsig_group* g);     /* 1 when the members hold RCCL communicators */
/* The RCCL library the groups of this process dlopen()ed (group.cpp looks for NXSIG_RCCL_LIB, librccl.so.1, /opt/rocm/lib/librccl.so.1,
 * librccl.so in that order — a process that imported torch finds torch's bundled copy first): *version = ncclGetVersion's code
 * (e.g. 22606 = 2.26.6; 0: no library / no such symbol), path_buf = the shared object's path.  Loads the library if nobody has yet. */
int nxsig_rccl_info(int32_t* version, char* path_buf, size_t buflen);
/* waits for every local stream, then (RCCL) all-reduces one word across the group and waits again */
int nxsig_group_barrier(nxsig_group* g);
/* element-wise reduction of `n` (<= 64) host doubles over the PROCESSES of the group; op 0 = max, 1 = sum */
int nxsig_group_allreduce_f64(nxsig_group* g, double* values, int32_t n, int32_t op);
/*
 * Final assembly of sharded device buffers: member r contributes `counts[r]` bytes (counts has nxsig_group_world
 * entries and is the same on every member).  For each local member i: send[i] = its shard (device), recv[i] = a device
 * buffer of sum(counts) bytes receiving the shards in rank order; send[i] may point into recv[i] at its own offset (in
 * place).  RCCL: one ncclBroadcast per rank inside ncclGroupStart / End (ncclAllGather when all counts are equal), on the
 * members' streams; asynchronous.  Members sharing a device: device-to-device copies.
 */
int nxsig_group_allgather(nxsig_group* g, const void* const* send, const int64_t* counts, void* const* recv);

/*
 * NxSignal.stft/3 sharded over the group (window_padding must be :valid — padding belongs to the stream ends).
 *   mem == NXSIG_HOST  : x[0] is the whole host tensor f32[batch][length], z[0] the whole host result c64[batch][M][K];
 *                        every local member uploads its part, computes it, and the result is assembled either by per-shard
 *                        downloads (gather == 0) or by the RCCL all-gather followed by one download (gather == 1).
 *                        RANKED groups (one process per GPU): every process passes the whole tensor and a full-size result;
 *                        gather == 0 writes only the process' own part of it, gather == 1 the whole result in every process.
 *   mem == NXSIG_DEVICE: x[i] is local member i's INPUT SHARD on its own device — rows [c0, c1) of the tensor
 *                        (channels axis: f32[c1 - c0][length], rows batch_stride apart) or the sample span [s0, s1) of
 *                        every row (frames axis: f32[batch][s1 - s0]).  batch_stride == 0 means DENSE per-member shards
 *                        (row stride = the member's own row length) and is what frame shards should pass: their spans
 *                        differ from member to member whenever the frame count does not divide evenly, so one stride
 *                        cannot describe them; a non-zero batch_stride applies to every member and must be at least
 *                        the member's row length (NXSIG_ERR_INVALID_ARG otherwise).  gather == 0: z[i] receives
 *                        the member's output shard (c64[c1 - c0][M][K] / c64[batch][m1 - m0][K]) and stays on its device.
 *                        gather == 1: z[i] is a full c64[batch][M][K] buffer on every member's device; shards are
 *                        computed in place and all-gathered (frames axis: batch must be 1).  Asynchronous.
 * `length`, `batch` always describe the WHOLE tensor; ranges come from nxsig_shard_range / nxsig_shard_frames.
 */
int nxsig_stft_sharded_f32(nxsig_group* g, const float* const* x, int64_t length, int32_t batch, int64_t batch_stride,
                           const float* window, const nxsig_stft_params* params, int32_t axis, int32_t gather,
                           nxsig_c64* const* z, int32_t mem);
/* NxSignal.istft/3 (nxsig_istft_c64) sharded over the group; same conventions.  Channels axis: rows of z c64[batch][M][K] split.
 * Frames axis: output sample ranges with halo frames (nxsig_shard_istft); DEVICE shards are dense c64[batch][f1 - f0][K] and
 * c64[batch][n1 - n0].  The result equals the unsharded call bit for bit: every frame that touches a kept sample is present on
 * the member that keeps it, and is added in the same order. */
int nxsig_istft_sharded_c64(nxsig_group* g, const nxsig_c64* const* z, int64_t num_frames, int32_t batch, const float* window,
                            const nxsig_stft_params* params, int32_t axis, int32_t gather, nxsig_c64* const* y, int32_t mem);
/* FIR filtering (nxsig_fir_f32) sharded over the group; same conventions.  Channels axis: rows split.  Frames axis here
 * means OUTPUT SAMPLE ranges of every row with a (num_taps - 1)-sample input halo (nxsig_shard_fir); it has ONE exchange step:
 * a row that holds an Inf / NaN anywhere has no finite output in the reference (one transform per row, convolution.ex:276-284),
 * so every member reads off its slice which rows its own samples poisoned, the flags (int32[batch]) are all-reduced with a maximum
 * (ncclAllReduce; members of one process that share a device: through the host) and every member turns the flagged rows NaN —
 * the sharded result equals the unsharded one for such rows too.  A ranked group without RCCL cannot do that: unsupported. */
int nxsig_fir_sharded_f32(nxsig_group* g, const float* const* x, int64_t length, int32_t batch, int64_t batch_stride,
                          const float* h, int32_t num_taps, int32_t mode, int32_t axis, int32_t gather, float* const* y,
                          int32_t mem);

/*
 * NxSignal.stft/3 |> NxSignal.stft_to_mel/3 (lib/nx_signal.ex:68-130, :486-513) sharded over the group — the one sharded entry
 * point with an EXCHANGE step: the clamp `max(log_spec, reduce_max(log_spec) - 8)` (:511) needs the maximum over the whole
 * tensor.  Every member runs the fused stft -> mel kernel on its shard (device-resident, laid out as for nxsig_stft_sharded_f32
 * with mem == NXSIG_DEVICE: rows [c0, c1) for the channels axis, the sample span [s0, s1) of every row for the frames axis),
 * or — mem == NXSIG_HOST, LOCAL groups — on the part of the host tensor x[0] the call uploads to it;
 * the members' running maxima (one int32 pair: ordered-int maximum, non-finite flag) are all-reduced — ncclAllReduce / ncclMax
 * on the members' streams inside one ncclGroupStart / End; members of one process that share a device reduce through the
 * host — and every member then clamps its own shard.  mem == NXSIG_DEVICE: out[i] is f32[c1 - c0][M][mel_bins] /
 * f32[batch][m1 - m0][mel_bins] on member i's device and the results stay sharded (assemble with nxsig_group_allgather if
 * needed); mem == NXSIG_HOST: out[0] is the whole host result f32[batch][M][mel_bins].  window_padding must be :valid.
 * `filters`: host f32[mel_bins][fft_length] (nxsig_mel_filters_f32).  Asynchronous on the members' streams.
 */
int nxsig_stft_mel_sharded_f32(nxsig_group* g, const float* const* x, int64_t length, int32_t batch, int64_t batch_stride,
                               const float* window, const nxsig_stft_params* params, int32_t mel_bins, const float* filters,
                               int32_t axis, float* const* out, int64_t* num_frames_out, int32_t mem);

/* per-launch stopwatch on the ctx stream: lap() records an event (at most 4096 per series), laps() synchronises and
 * returns the n - 1 intervals in milliseconds and clears the series */
int nxsig_timer_lap(nxsig_ctx* ctx);
int nxsig_timer_laps(nxsig_ctx* ctx, float* intervals_ms, int32_t capacity, int32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* NXSIG_H */
