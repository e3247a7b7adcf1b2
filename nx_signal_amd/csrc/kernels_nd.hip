// Device-resident multi-axis and large transforms:
//
//   k_transpose          [outer][R][C] -> [outer][C][R] of c64 through a padded 64 x 64 LDS tile (512-byte runs on both
//                        sides), optionally reading f32 / zero-padding / truncating on the way in and multiplying by the
//                        four-step twiddle w_K^(r c) on the way out
//   launch_fft_big       rows of ANY length beyond the LDS-resident kernels of kernels_generic.hip:
//                          power of two 8192 <= K <= 2^20: two-pass tiled four-step (k_fft_tile, below): K1-point column transforms
//                          + twiddle, then K2-point row transforms with the transposing store — 32 B per element of traffic;
//                          larger powers of two: four-step K = K1 K2 through explicit transposes — transpose, K1-point row FFTs,
//                          twiddle + transpose, K2-point row FFTs, transpose (five streaming passes, no host round trip);
//                          other K > 4096: Bluestein chirp-z through a power-of-two transform of P >= 2K - 1 points
//   launch_fft_nd        NxSignal.Transforms.fft_nd / ifft_nd (lib/nx_signal/transforms.ex:5-21): Enum.zip_reduce over
//                        (axes, lengths) of Nx.fft(axis:, length:) — an axis other than the last is brought to the back
//                        by k_transpose, transformed by the row kernels and moved back, all in HBM
//   launch_fftconvolve_nd  n-D Convolution.fftconvolve/3 (lib/nx_signal/convolution.ex:252-347): fft_nd of both operands
//                        over the axes where neither is 1 (lengths s1 + s2 - 1), broadcast product, ifft_nd, real part for
//                        real operands, `centered` slice per mode
//   launch_stft_big      stft for fft_length beyond the fused kernels: framing x window into a scratch tensor, then
//                        launch_fft_big over its rows, then the :spectrum / :psd division
//
// Twiddles come from host tables generated in double (two-level: w_K^j = hi[j >> 13] * lo[j & 8191]).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nxsig_internal.h"

namespace nxsig {

static constexpr int kT = 256;
static constexpr int kTile = 64;

__device__ __forceinline__ float2 ndmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

struct TrArgs {
  const void* in;
  float2* out;
  int64_t outer;
  int32_t R, C;
  int32_t in_is_real;
  int64_t plane_stride;   // elements between consecutive [R][C] planes of the input
  int64_t plane_valid;    // plane elements with linear index r * C + c >= plane_valid read as zero (zero-pad / truncate)
  int32_t tw_mode;        // 0 none, 1 multiply by w_K^(r c), 2 by its conjugate
  int32_t clean;          // 1: this transposition delivers a finished transform: Nx.fft / Nx.ifft eps clean-up on the way out
  int64_t K;
  const float2* tw_lo;    // [8192]  w_K^t
  const float2* tw_hi;    // [K / 8192] w_K^(8192 s)
};

__global__ __launch_bounds__(kT) void k_transpose(TrArgs a) {
  __shared__ float2 tile[kTile][kTile + 1];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const int64_t o = blockIdx.y;
  const int tiles_c = (a.C + kTile - 1) / kTile;   // tiles are numbered along blockIdx.x (2^31 of them): no 65535-row limit
  const int r0 = (int)(blockIdx.x / tiles_c) * kTile, c0 = (int)(blockIdx.x % tiles_c) * kTile;
  const int64_t pbase = o * a.plane_stride;
#pragma unroll 4
  for (int i = ty; i < kTile; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float2 v = make_float2(0.f, 0.f);
    if (r < a.R && c < a.C) {
      const int64_t lin = (int64_t)r * a.C + c;
      if (lin < a.plane_valid) {
        if (a.in_is_real) v.x = reinterpret_cast<const float*>(a.in)[pbase + lin];
        else v = reinterpret_cast<const float2*>(a.in)[pbase + lin];
      }
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  float2* op = a.out + o * (int64_t)a.R * a.C;
#pragma unroll 4
  for (int i = ty; i < kTile; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (r < a.R && c < a.C) {
      float2 v = tile[tx][i];
      if (a.tw_mode) {
        const int64_t j = ((int64_t)r * c) % a.K;
        float2 w = ndmul(a.tw_hi[j >> 13], a.tw_lo[j & 8191]);
        if (a.tw_mode == 2) w.y = -w.y;
        v = ndmul(v, w);
      }
      if (a.clean) v = fft_eps0(v);
      op[(int64_t)c * a.R + r] = v;
    }
  }
}

static int launch_transpose(Ctx* c, const void* in, bool in_is_real, int64_t outer, int R, int C, int64_t plane_stride,
                            int64_t plane_valid, float2* out, int tw_mode = 0, int64_t K = 0, const float2* lo = nullptr,
                            const float2* hi = nullptr, bool clean = false) {
  if (outer == 0 || R == 0 || C == 0) return NXSIG_OK;
  TrArgs a;
  a.in = in; a.out = out; a.R = R; a.C = C; a.in_is_real = in_is_real ? 1 : 0;
  a.plane_stride = plane_stride; a.plane_valid = plane_valid; a.tw_mode = tw_mode; a.K = K; a.tw_lo = lo; a.tw_hi = hi;
  a.clean = clean ? 1 : 0;
  const int64_t tiles = (int64_t)((C + kTile - 1) / kTile) * ((R + kTile - 1) / kTile);
  if (tiles > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "transpose: plane too large for one launch");
  for (int64_t o0 = 0; o0 < outer; o0 += 65535) {  // gridDim.y limit
    const int64_t no = outer - o0 < 65535 ? outer - o0 : 65535;
    a.outer = no;
    a.in = in_is_real ? static_cast<const void*>(reinterpret_cast<const float*>(in) + o0 * plane_stride)
                      : static_cast<const void*>(reinterpret_cast<const float2*>(in) + o0 * plane_stride);
    a.out = out + o0 * (int64_t)R * C;
    dim3 grid((unsigned)tiles, (unsigned)no);
    dispatch_note("fft.transpose");
    hipLaunchKernelGGL(k_transpose, grid, dim3(kT), 0, c->stream, a);
  }
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// ---- element-wise helpers
__global__ __launch_bounds__(kT) void k_copy_pad(const void* __restrict__ in, int in_is_real, int64_t rows, int64_t n_in, int64_t K,
                                                 float2* __restrict__ out) {  // rows of n_in -> rows of K (zero-pad / truncate)
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= rows * K) return;
  const int64_t r = i / K, k = i - r * K;
  float2 v = make_float2(0.f, 0.f);
  if (k < n_in) {
    if (in_is_real) v.x = reinterpret_cast<const float*>(in)[r * n_in + k];
    else v = reinterpret_cast<const float2*>(in)[r * n_in + k];
  }
  out[i] = v;
}
// Bluestein: A[r][p] = conj^INV(x[r][p]) * chirp[p] for p < min(n_in, K), zero up to P
__global__ __launch_bounds__(kT) void k_blue_in(const void* __restrict__ in, int in_is_real, int64_t rows, int64_t n_in, int64_t K, int64_t P,
                                                const float2* __restrict__ chirp, int inv, float2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= rows * P) return;
  const int64_t r = i / P, p = i - r * P;
  float2 v = make_float2(0.f, 0.f);
  if (p < n_in && p < K) {
    if (in_is_real) v.x = reinterpret_cast<const float*>(in)[r * n_in + p];
    else v = reinterpret_cast<const float2*>(in)[r * n_in + p];
    if (inv) v.y = -v.y;
    v = ndmul(v, chirp[p]);
  }
  out[i] = v;
}
__global__ __launch_bounds__(kT) void k_mul_rowtable(float2* __restrict__ a, const float2* __restrict__ t, int64_t rows, int64_t P) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= rows * P) return;
  a[i] = ndmul(a[i], t[i % P]);
}
// out[r][k] = conj^INV(A[r][k] * chirp[k]) (/ K for the inverse), k < K
__global__ __launch_bounds__(kT) void k_blue_out(const float2* __restrict__ A, int64_t rows, int64_t K, int64_t P,
                                                 const float2* __restrict__ chirp, int inv, int clean, float2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= rows * K) return;
  const int64_t r = i / K, k = i - r * K;
  float2 v = ndmul(A[r * P + k], chirp[k]);
  if (inv) { v.y = -v.y; v.x = v.x / (float)K; v.y = v.y / (float)K; }
  out[i] = clean ? fft_eps0(v) : v;
}
// istft epilogue / stft scaling on finished rows: v = (v * scale) * window[k]   or   v = v / div
__global__ __launch_bounds__(kT) void k_rows_post(float2* __restrict__ a, int64_t total, int64_t K, const float* __restrict__ window,
                                                  float scale, int has_scale, float div, int has_div) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= total) return;
  float2 v = a[i];
  if (has_scale) { v.x *= scale; v.y *= scale; }
  if (window) { const float w = window[i % K]; v.x *= w; v.y *= w; }
  if (has_div) { v.x = v.x / div; v.y = v.y / div; }
  a[i] = v;
}

static unsigned blocks_for(int64_t n) { return (unsigned)((n + kT - 1) / kT); }

// ---- host tables
static int twolevel_tables(Ctx* c, int64_t K, const float2** lo, const float2** hi) {
  const uint64_t key = 0x7B16000000000000ull ^ (uint64_t)K;
  auto hit = c->memo.find(key);
  if (hit != c->memo.end()) { *lo = reinterpret_cast<const float2*>(hit->second[0]); *hi = reinterpret_cast<const float2*>(hit->second[1]); return NXSIG_OK; }
  const double two_pi = 6.283185307179586476925286766559;
  const int64_t nhi = (K + 8191) / 8192;
  std::vector<float2> l(8192), h((size_t)nhi);
  for (int t = 0; t < 8192; ++t) { const double ang = -two_pi * (double)(t % K) / (double)K; l[t] = make_float2((float)std::cos(ang), (float)std::sin(ang)); }
  for (int64_t s = 0; s < nhi; ++s) { const double ang = -two_pi * (double)((s * 8192) % K) / (double)K; h[(size_t)s] = make_float2((float)std::cos(ang), (float)std::sin(ang)); }
  const void *dl = nullptr, *dh = nullptr;
  int rc = ctx_table(c, 0x7B161ull ^ ((uint64_t)K << 8), l.data(), l.size() * sizeof(float2), &dl);
  if (rc) return rc;
  if ((rc = ctx_table(c, 0x7B162ull ^ ((uint64_t)K << 8), h.data(), h.size() * sizeof(float2), &dh))) return rc;
  c->memo[key] = {reinterpret_cast<uint64_t>(dl), reinterpret_cast<uint64_t>(dh)};
  *lo = reinterpret_cast<const float2*>(dl); *hi = reinterpret_cast<const float2*>(dh);
  return NXSIG_OK;
}

// iterative radix-2 in double with one twiddle table (the per-butterfly cos/sin of the small-table helper would take seconds here)
static void host_fft_big(std::vector<double>& re, std::vector<double>& im) {
  const size_t n = re.size();
  std::vector<double> wr(n / 2 ? n / 2 : 1), wi(n / 2 ? n / 2 : 1);
  for (size_t k = 0; k < n / 2; ++k) { const double ang = -6.283185307179586476925286766559 * (double)k / (double)n; wr[k] = std::cos(ang); wi[k] = std::sin(ang); }
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t step = n / len;
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < len / 2; ++k) {
        const double cr = wr[k * step], ci = wi[k * step];
        const size_t u = i + k, v = u + len / 2;
        const double tr = re[v] * cr - im[v] * ci, ti = re[v] * ci + im[v] * cr;
        re[v] = re[u] - tr; im[v] = im[u] - ti;
        re[u] += tr; im[u] += ti;
      }
  }
}

static bool nd_is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

int launch_fft_big(Ctx* c, const void* in, bool in_is_real, int64_t rows, int64_t n_in, int64_t K, bool inverse, float2* out, bool clean);

// power-of-two K = K1 K2 > 8192 (both <= 8192): X[k1 + K1 k2] = sum_n2 [w_K^(n2 k1) sum_n1 x[K2 n1 + n2] w_K1^(n1 k1)] w_K2^(n2 k2)
static int fft_fourstep(Ctx* c, const void* in, bool in_is_real, int64_t rows, int64_t n_in, int64_t K, bool inverse, float2* out, bool clean) {
  int lg = 0;
  while (((int64_t)1 << lg) < K) ++lg;
  const int K1 = 1 << ((lg + 1) / 2), K2 = (int)(K / K1);
  if (K1 > 8192) return set_error(NXSIG_ERR_UNSUPPORTED, "fft: power-of-two lengths beyond 2^26 are not supported");
  const float2 *lo = nullptr, *hi = nullptr;
  int rc = twolevel_tables(c, K, &lo, &hi);
  if (rc) return rc;
  void *s1 = nullptr, *s2 = nullptr;
  const size_t bytes = (size_t)rows * K * sizeof(float2);
  if ((rc = ctx_scratch(c, 6, bytes, &s1))) return rc;
  if ((rc = ctx_scratch(c, 7, bytes, &s2))) return rc;
  float2* A = reinterpret_cast<float2*>(s1);
  float2* B = reinterpret_cast<float2*>(s2);
  // (a) [K1][K2] -> [K2][K1]   (zero-pad / truncate / real -> complex on the way in)
  if ((rc = launch_transpose(c, in, in_is_real, rows, K1, K2, n_in, n_in < K ? n_in : K, A))) return rc;
  // (b) K1-point transforms over n1
  if ((rc = launch_fft(c, A, false, rows * K2, K1, K1, inverse, B, false))) return rc;
  // (c) twiddle w_K^(n2 k1) and back to [K1][K2]
  if ((rc = launch_transpose(c, B, false, rows, K2, K1, (int64_t)K, (int64_t)K, A, inverse ? 2 : 1, K, lo, hi))) return rc;
  // (d) K2-point transforms over n2
  if ((rc = launch_fft(c, A, false, rows * K1, K2, K2, inverse, B, false))) return rc;
  // (e) Z[k1][k2] -> natural order X[k1 + K1 k2]  (+ the eps clean-up of the finished transform)
  return launch_transpose(c, B, false, rows, K1, K2, (int64_t)K, (int64_t)K, out, 0, 0, nullptr, nullptr, clean);
}

// ---- two-pass four-step (round 2): the transposes of fft_fourstep folded into the transforms.
// k_fft_tile takes a tile of T sequences of n <= 2048 points into LDS (element (position p, sequence t) at p * T + t), runs an
// in-place decimation-in-frequency FFT on all of them at once (one radix-2 stage when log2 n is odd, then radix-16 stages — two
// radix-4 levels fused in registers — and a closing radix-4 stage when four points are left; a thread owns butterfly j of
// sequence t with t fastest, so every stage reads and writes LDS in runs of T consecutive elements),
// and leaves through a position table (the output of frequency k sits at the digit-reversed position).  Global accesses:
//   pass A  sequences = columns n2 of the [K1][K2] view: every position is a run of T consecutive elements (T * 8 bytes), the
//           result goes back the same way with the twiddle w_K^(n2 k1) applied (real input / zero padding / truncation on load)
//   pass B  sequences = rows k1 of Y[k1][n2]: whole rows on load; X[k1 + K1 k2] on store — again runs of T consecutive k1
// Two streaming passes (32 B per element) instead of five (80 B): 0.70-0.85 -> 1.2-1.6 TB/s algorithmic on rows of 2^13 ... 2^20
// points (512 MB of c64 rows; each pass moves ~3 TB/s against ~5 for a plain copy: the LDS transform and the loads of a tile do
// not overlap inside a workgroup, and 124 VGPRs keep it to four waves per SIMD; a persistent variant that prefetched the next
// tile into registers needed 200+ VGPRs and was slower).  Above 2^20 the strided side would shrink to 32-byte runs: the
// five-pass form stays.
struct FtArgs {
  const void* in;
  float2* out;
  int32_t n, lg, lgT;
  int32_t in_real;
  int32_t load_along;       // 1: the input is contiguous along the sequence (in_pos_stride == 1), 0: across sequences
  int64_t in_seq_stride, in_pos_stride, in_row_stride, n_valid;
  int64_t out_seq_stride, out_pos_stride, out_row_stride;
  int64_t nseq;             // sequences per batch row (a multiple of T)
  int32_t inverse;
  float scale;
  int32_t clean;            // 1: this pass delivers the finished transform: Nx.fft / Nx.ifft eps clean-up after the scale
  int32_t tw_mode;          // 1: multiply (sequence s, frequency k) by w_K^(s k) (conjugated for the inverse)
  const float2 *tw_lo, *tw_hi, *tw_n;   // two-level w_K tables; w_n^j, j < n
};

template <int NT>
__global__ __launch_bounds__(NT) void k_fft_tile(FtArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ft_smem[];
  float2* s = reinterpret_cast<float2*>(ft_smem);   // element (position p, sequence t) at p * TP + t.  (An odd stride TP = T + 1
                                                    // against the bank conflicts of pass B's row-wise fill measured no faster and
                                                    // costs a tile per CU at n = 1024: the kernel is bound by issue + HBM, not LDS.)
  const int tid = threadIdx.x;
  const int n = a.n, lgT = a.lgT, T = 1 << lgT, TP = T;
  float2* s_tw = s + (size_t)n * TP;                // w_n^j, j < n (conjugated for the inverse), behind the tile
  __shared__ unsigned short s_pos[2048];
  const int64_t tiles_per_row = a.nseq >> lgT;
  const int64_t row = blockIdx.x / tiles_per_row;
  const int64_t seq0 = (blockIdx.x - row * tiles_per_row) << lgT;
  for (int k = tid; k < n; k += NT) {   // where frequency k ends up: its digits, lowest first, select the sub-blocks of each stage
    int pos = 0, span = n, kk = k;
    if (a.lg & 1) { span >>= 1; pos += (kk & 1) * span; kk >>= 1; }
    while (span >= 4) { span >>= 2; pos += (kk & 3) * span; kk >>= 2; }
    s_pos[k] = (unsigned short)pos;
    float2 w = a.tw_n[k];
    if (a.inverse) w.y = -w.y;
    s_tw[k] = w;
  }
  const int total = n << lgT;
  const int64_t ibase = row * a.in_row_stride;
  if (a.load_along) {
#pragma unroll 8
    for (int id = tid; id < total; id += NT) {
      const int t = id >> a.lg, p = id & (n - 1);
      const int64_t lin = (seq0 + t) * a.in_seq_stride + p;
      float2 v = make_float2(0.f, 0.f);
      if (lin < a.n_valid) {
        if (a.in_real) v.x = reinterpret_cast<const float*>(a.in)[ibase + lin];
        else v = reinterpret_cast<const float2*>(a.in)[ibase + lin];
      }
      s[p * TP + t] = v;
    }
  } else {
#pragma unroll 8
    for (int id = tid; id < total; id += NT) {
      const int p = id >> lgT, t = id & (T - 1);
      const int64_t lin = (seq0 + t) + (int64_t)p * a.in_pos_stride;
      float2 v = make_float2(0.f, 0.f);
      if (lin < a.n_valid) {
        if (a.in_real) v.x = reinterpret_cast<const float*>(a.in)[ibase + lin];
        else v = reinterpret_cast<const float2*>(a.in)[ibase + lin];
      }
      s[p * TP + t] = v;
    }
  }
  __syncthreads();
  const float sg = a.inverse ? 1.0f : -1.0f;   // the DFT4 rotation: -i forward, +i inverse
  int span = n;
  if (a.lg & 1) {
    const int h = n >> 1;
    for (int id = tid; id < (h << lgT); id += NT) {
      const int i = id >> lgT, t = id & (T - 1);
      const int i0 = i * TP + t, i1 = i0 + h * TP;
      const float2 x0 = s[i0], x1 = s[i1];
      s[i0] = make_float2(x0.x + x1.x, x0.y + x1.y);
      s[i1] = ndmul(make_float2(x0.x - x1.x, x0.y - x1.y), s_tw[i]);
    }
    span = h;
    __syncthreads();
  }
  while (span >= 16) {   // two radix-4 levels fused in registers: 16 elements per thread, one LDS round trip and one barrier
    const int q16 = span >> 4, lgq = 31 - __clz(q16), st = n / span;
    for (int id = tid; id < (total >> 4); id += NT) {
      const int j = id >> lgT, t = id & (T - 1);
      const int b = j >> lgq, o = j & (q16 - 1);
      const int i0 = ((b * span) + o) * TP + t, dq = q16 * TP;
      float2 e[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) e[m] = s[i0 + m * dq];
#pragma unroll
      for (int m0 = 0; m0 < 4; ++m0) {   // level 1: span, elements m0, m0 + 4, m0 + 8, m0 + 12; offset o + m0 q16
        const float2 x0 = e[m0], x1 = e[m0 + 4], x2 = e[m0 + 8], x3 = e[m0 + 12];
        const float2 sa = make_float2(x0.x + x2.x, x0.y + x2.y), sb = make_float2(x1.x + x3.x, x1.y + x3.y);
        const float2 da = make_float2(x0.x - x2.x, x0.y - x2.y), db = make_float2(x1.x - x3.x, x1.y - x3.y);
        const float2 idb = make_float2(-sg * db.y, sg * db.x);
        const int o1 = (o + m0 * q16) * st;
        e[m0] = make_float2(sa.x + sb.x, sa.y + sb.y);
        e[m0 + 4] = ndmul(make_float2(da.x + idb.x, da.y + idb.y), s_tw[o1]);
        e[m0 + 8] = ndmul(make_float2(sa.x - sb.x, sa.y - sb.y), s_tw[2 * o1]);
        e[m0 + 12] = ndmul(make_float2(da.x - idb.x, da.y - idb.y), s_tw[3 * o1]);
      }
      const int o2 = o * st * 4;
      const bool tw2 = q16 > 1;
      const float2 v1 = tw2 ? s_tw[o2] : make_float2(1.f, 0.f), v2 = tw2 ? s_tw[2 * o2] : make_float2(1.f, 0.f),
                   v3 = tw2 ? s_tw[3 * o2] : make_float2(1.f, 0.f);
#pragma unroll
      for (int r = 0; r < 4; ++r) {      // level 2: span / 4 inside quarter r: elements 4 r + {0, 1, 2, 3}; offset o
        const float2 x0 = e[4 * r], x1 = e[4 * r + 1], x2 = e[4 * r + 2], x3 = e[4 * r + 3];
        const float2 sa = make_float2(x0.x + x2.x, x0.y + x2.y), sb = make_float2(x1.x + x3.x, x1.y + x3.y);
        const float2 da = make_float2(x0.x - x2.x, x0.y - x2.y), db = make_float2(x1.x - x3.x, x1.y - x3.y);
        const float2 idb = make_float2(-sg * db.y, sg * db.x);
        float2 y0 = make_float2(sa.x + sb.x, sa.y + sb.y), y1 = make_float2(da.x + idb.x, da.y + idb.y);
        float2 y2 = make_float2(sa.x - sb.x, sa.y - sb.y), y3 = make_float2(da.x - idb.x, da.y - idb.y);
        if (tw2) { y1 = ndmul(y1, v1); y2 = ndmul(y2, v2); y3 = ndmul(y3, v3); }
        s[i0 + (4 * r) * dq] = y0; s[i0 + (4 * r + 1) * dq] = y1; s[i0 + (4 * r + 2) * dq] = y2; s[i0 + (4 * r + 3) * dq] = y3;
      }
    }
    span >>= 4;
    __syncthreads();
  }
  while (span >= 4) {
    const int q = span >> 2, lgq = 31 - __clz(q), st = n / span;
    for (int id = tid; id < (total >> 2); id += NT) {
      const int j = id >> lgT, t = id & (T - 1);
      const int b = j >> lgq, o = j & (q - 1);
      const int i0 = ((b * span) + o) * TP + t, dq = q * TP;
      const float2 x0 = s[i0], x1 = s[i0 + dq], x2 = s[i0 + 2 * dq], x3 = s[i0 + 3 * dq];
      const float2 sa = make_float2(x0.x + x2.x, x0.y + x2.y), sb = make_float2(x1.x + x3.x, x1.y + x3.y);
      const float2 da = make_float2(x0.x - x2.x, x0.y - x2.y), db = make_float2(x1.x - x3.x, x1.y - x3.y);
      const float2 idb = make_float2(-sg * db.y, sg * db.x);
      float2 y0 = make_float2(sa.x + sb.x, sa.y + sb.y);
      float2 y2 = make_float2(sa.x - sb.x, sa.y - sb.y);
      float2 y1 = make_float2(da.x + idb.x, da.y + idb.y);
      float2 y3 = make_float2(da.x - idb.x, da.y - idb.y);
      if (q > 1) { y1 = ndmul(y1, s_tw[o * st]); y2 = ndmul(y2, s_tw[2 * o * st]); y3 = ndmul(y3, s_tw[3 * o * st]); }
      s[i0] = y0; s[i0 + dq] = y1; s[i0 + 2 * dq] = y2; s[i0 + 3 * dq] = y3;
    }
    span = q;
    __syncthreads();
  }
  float2* ob = a.out + row * a.out_row_stride;
  // a thread's elements share t (NT is a multiple of T) and their k advance by NT / T: the four-step twiddle w_K^(s k) of pass A
  // runs as a recurrence w <- w * w_K^(s NT / T), re-seeded from the two-level table every eight elements
  const int t = tid & (T - 1);
  const int64_t sq = seq0 + t;
  const int dk = NT >> lgT;
  float2 wstep = make_float2(1.f, 0.f), w = make_float2(1.f, 0.f);
  if (a.tw_mode) {
    const int64_t js = sq * dk;
    wstep = ndmul(a.tw_hi[js >> 13], a.tw_lo[js & 8191]);
    if (a.inverse) wstep.y = -wstep.y;
  }
  int cnt = 0;
  for (int id = tid; id < total; id += NT, ++cnt) {
    const int k = id >> lgT;
    float2 v = s[(int)s_pos[k] * TP + t];
    if (a.tw_mode) {
      if ((cnt & 7) == 0) {
        const int64_t j = sq * k;   // < K1 K2 = K
        w = ndmul(a.tw_hi[j >> 13], a.tw_lo[j & 8191]);
        if (a.inverse) w.y = -w.y;
      } else {
        w = ndmul(w, wstep);
      }
      v = ndmul(v, w);
    }
    v.x *= a.scale; v.y *= a.scale;
    if (a.clean) v = fft_eps0(v);
    ob[sq * a.out_seq_stride + (int64_t)k * a.out_pos_stride] = v;
  }
}

static int fft_fourstep_tiled(Ctx* c, const void* in, bool in_is_real, int64_t rows, int64_t n_in, int64_t K, bool inverse, float2* out, bool clean) {
  int lg = 0;
  while (((int64_t)1 << lg) < K) ++lg;
  const int lg1 = (lg + 1) / 2, lg2 = lg - lg1;
  const int K1 = 1 << lg1, K2 = 1 << lg2;
  const int elems = [&] { const int e = tune(c, kT_FFT_TILE_ELEMS, 8192); return e >= 1024 && e <= 16384 ? e : 8192; }();
  const int ntk = tune(c, kT_FFT_TILE_NT, 512);
  const float2 *lo = nullptr, *hi = nullptr, *tw1 = nullptr, *tw2 = nullptr;
  int rc = twolevel_tables(c, K, &lo, &hi);
  if (rc) return rc;
  if ((rc = ctx_twiddles(c, K1, &tw1))) return rc;
  if ((rc = ctx_twiddles(c, K2, &tw2))) return rc;
  void* s1 = nullptr;
  if ((rc = ctx_scratch(c, 6, (size_t)rows * K * sizeof(float2), &s1))) return rc;
  float2* Y = reinterpret_cast<float2*>(s1);
  auto tile_lg = [&](int n, int64_t nseq) {
    int l = 0;
    while ((n << (l + 1)) <= elems && ((int64_t)1 << (l + 1)) <= nseq && l + 1 <= 6) ++l;
    return l;
  };
  constexpr int NT = 512;
  FtArgs a;
  // pass A: columns n2, K1 points each
  a.in = in; a.out = Y; a.n = K1; a.lg = lg1; a.lgT = tile_lg(K1, K2); a.in_real = in_is_real ? 1 : 0; a.load_along = 0;
  a.in_seq_stride = 1; a.in_pos_stride = K2; a.in_row_stride = n_in; a.n_valid = n_in < K ? n_in : K;
  a.out_seq_stride = 1; a.out_pos_stride = K2; a.out_row_stride = K; a.nseq = K2; a.inverse = inverse ? 1 : 0; a.scale = 1.0f; a.clean = 0;
  a.tw_mode = 1; a.tw_lo = lo; a.tw_hi = hi; a.tw_n = tw1;
  auto go = [&](int64_t blocks, size_t lds) -> int {
    if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "fft: too many tiles for one launch");
    if (ntk == 256) {
      if (lds > 64 * 1024) NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fft_tile<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      dispatch_note("fft.tiled");
      hipLaunchKernelGGL(k_fft_tile<256>, dim3((unsigned)blocks), dim3(256), lds, c->stream, a);
    } else {
      if (lds > 64 * 1024) NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fft_tile<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      dispatch_note("fft.tiled");
      hipLaunchKernelGGL(k_fft_tile<NT>, dim3((unsigned)blocks), dim3(NT), lds, c->stream, a);
    }
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  if ((rc = go(rows * (K2 >> a.lgT), ((size_t)K1 * ((1 << a.lgT) + 1)) * sizeof(float2)))) return rc;
  // pass B: rows k1, K2 points each; X[k1 + K1 k2]
  a.in = Y; a.out = out; a.n = K2; a.lg = lg2; a.lgT = tile_lg(K2, K1); a.in_real = 0; a.load_along = 1;
  a.in_seq_stride = K2; a.in_pos_stride = 1; a.in_row_stride = K; a.n_valid = K;
  a.out_seq_stride = 1; a.out_pos_stride = K1; a.out_row_stride = K; a.nseq = K1; a.scale = inverse ? 1.0f / (float)K : 1.0f; a.clean = clean ? 1 : 0;
  a.tw_mode = 0; a.tw_n = tw2;
  if ((rc = go(rows * (K1 >> a.lgT), ((size_t)K2 * ((1 << a.lgT) + 1)) * sizeof(float2)))) return rc;
  return NXSIG_OK;
}

// A power-of-two transform of 16 ... 1024 points along an axis that is NOT the fastest one, [outer][na][inner] -> [outer][K][inner],
// in ONE pass of k_fft_tile (sequences = the `inner` columns of a plane, T of them per workgroup, positions `inner` elements apart)
// instead of transpose + row transforms + transpose.  Needs inner to be a multiple of 8 (64-byte runs at least).
static int fft_columns_tiled(Ctx* c, const void* src, bool src_real, int64_t outer, int64_t na, int64_t inner, int64_t K, bool inverse,
                             float2* dst, bool* handled) {
  *handled = false;
  const bool on = tune(c, kT_FFT_COLUMNS, 1) != 0;
  if (!on || !nd_is_pow2(K) || K < 16 || K > 1024 || (inner & 7) != 0 || outer < 1) return NXSIG_OK;
  int lg = 0;
  while (((int64_t)1 << lg) < K) ++lg;
  int lgT = 3;
  while (lgT < 6 && (inner & (((int64_t)1 << (lgT + 1)) - 1)) == 0 && (K << (lgT + 1)) <= 8192) ++lgT;
  const int64_t blocks = outer * (inner >> lgT);
  if (blocks > 0x7fffffffLL) return NXSIG_OK;
  const float2* tw = nullptr;
  int rc = ctx_twiddles(c, (int)K, &tw);
  if (rc) return rc;
  FtArgs a;
  a.in = src; a.out = dst; a.n = (int)K; a.lg = lg; a.lgT = lgT; a.in_real = src_real ? 1 : 0; a.load_along = 0;
  a.in_seq_stride = 1; a.in_pos_stride = inner; a.in_row_stride = na * inner; a.n_valid = (na < K ? na : K) * inner;
  a.out_seq_stride = 1; a.out_pos_stride = inner; a.out_row_stride = K * inner; a.nseq = inner;
  a.inverse = inverse ? 1 : 0; a.scale = inverse ? 1.0f / (float)K : 1.0f; a.clean = 1;
  a.tw_mode = 0; a.tw_lo = nullptr; a.tw_hi = nullptr; a.tw_n = tw;
  const size_t lds = ((size_t)K * ((1 << lgT) + 1)) * sizeof(float2);
  if (lds > 64 * 1024) NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fft_tile<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dispatch_note("fft.tiled.columns");
  hipLaunchKernelGGL(k_fft_tile<512>, dim3((unsigned)blocks), dim3(512), lds, c->stream, a);
  NXSIG_HIP_TRY(hipGetLastError());
  *handled = true;
  return NXSIG_OK;
}

// any other K: chirp-z through a power-of-two convolution of P >= 2K - 1 points
static int fft_bluestein_big(Ctx* c, const void* in, bool in_is_real, int64_t rows, int64_t n_in, int64_t K, bool inverse, float2* out, bool clean) {
  if (K > ((int64_t)1 << 22)) return set_error(NXSIG_ERR_UNSUPPORTED, "fft: non-power-of-two lengths beyond 2^22 are not supported");
  int64_t P = 1;
  while (P < 2 * K - 1) P <<= 1;
  const uint64_t key = 0xB16B000000000000ull ^ (uint64_t)K;
  const float2 *chirp = nullptr, *Bf = nullptr;
  auto hit = c->memo.find(key);
  if (hit != c->memo.end()) { chirp = reinterpret_cast<const float2*>(hit->second[0]); Bf = reinterpret_cast<const float2*>(hit->second[1]); }
  else {
    std::vector<float2> ch((size_t)K), bf((size_t)P);
    std::vector<double> bre((size_t)P, 0.0), bim((size_t)P, 0.0);
    for (int64_t n = 0; n < K; ++n) {
      const int64_t q = (n * n) % (2 * K);  // exact phase index
      const double ang = -3.14159265358979323846 * (double)q / (double)K;
      const double cr = std::cos(ang), ci = std::sin(ang);
      ch[(size_t)n] = make_float2((float)cr, (float)ci);
      bre[(size_t)n] = cr; bim[(size_t)n] = -ci;
      if (n) { bre[(size_t)(P - n)] = cr; bim[(size_t)(P - n)] = -ci; }
    }
    host_fft_big(bre, bim);
    for (int64_t i = 0; i < P; ++i) bf[(size_t)i] = make_float2((float)bre[(size_t)i], (float)bim[(size_t)i]);  // unscaled: the inverse rows divide by P
    const void *dc = nullptr, *db = nullptr;
    int rc = ctx_table(c, 0xB16B1ull ^ ((uint64_t)K << 8), ch.data(), ch.size() * sizeof(float2), &dc);
    if (rc) return rc;
    if ((rc = ctx_table(c, 0xB16B2ull ^ ((uint64_t)K << 8), bf.data(), bf.size() * sizeof(float2), &db))) return rc;
    c->memo[key] = {reinterpret_cast<uint64_t>(dc), reinterpret_cast<uint64_t>(db)};
    chirp = reinterpret_cast<const float2*>(dc); Bf = reinterpret_cast<const float2*>(db);
  }
  void *s1 = nullptr, *s2 = nullptr;
  const size_t bytes = (size_t)rows * P * sizeof(float2);
  int rc;
  if ((rc = ctx_scratch(c, 8, bytes, &s1))) return rc;
  if ((rc = ctx_scratch(c, 9, bytes, &s2))) return rc;
  float2* A = reinterpret_cast<float2*>(s1);
  float2* B = reinterpret_cast<float2*>(s2);
  dispatch_note("fft.big.blue");
  hipLaunchKernelGGL(k_blue_in, dim3(blocks_for(rows * P)), dim3(kT), 0, c->stream, in, in_is_real ? 1 : 0, rows, n_in, K, P, chirp, inverse ? 1 : 0, A);
  NXSIG_HIP_TRY(hipGetLastError());
  if ((rc = launch_fft_big(c, A, false, rows, P, P, false, B, false))) return rc;
  hipLaunchKernelGGL(k_mul_rowtable, dim3(blocks_for(rows * P)), dim3(kT), 0, c->stream, B, Bf, rows, P);
  NXSIG_HIP_TRY(hipGetLastError());
  if ((rc = launch_fft_big(c, B, false, rows, P, P, true, A, false))) return rc;  // the inverse rows include the 1 / P
  hipLaunchKernelGGL(k_blue_out, dim3(blocks_for(rows * K)), dim3(kT), 0, c->stream, A, rows, K, P, chirp, inverse ? 1 : 0, clean ? 1 : 0, out);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

int64_t fft_tiled_min(const Ctx* c) {
  const int64_t m = tune(c, kT_FFT_TILED_MIN, 8192);
  return m < 8192 ? 8192 : m;
}

// rows of any length.  Lengths the LDS-resident kernels cover go straight to launch_fft.
int launch_fft_big(Ctx* c, const void* in, bool in_is_real, int64_t rows, int64_t n_in, int64_t K, bool inverse, float2* out, bool clean) {
  if (rows == 0) return NXSIG_OK;
  if (K < 1 || n_in < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fft: lengths must be >= 1");
  const bool small = (nd_is_pow2(K) && K <= 8192 && K < fft_tiled_min(c)) || (!nd_is_pow2(K) && K <= 4096);
  if (small) {
    if (n_in > 0x7fffffff) return set_error(NXSIG_ERR_UNSUPPORTED, "fft: rows longer than 2^31");
    return launch_fft(c, in, in_is_real, rows, (int32_t)n_in, (int32_t)K, inverse, out, clean);
  }
  if (nd_is_pow2(K)) {
    const bool tiled = tune(c, kT_FFT_TILED, 1) != 0;
    if (tiled && K <= ((int64_t)1 << 20)) return fft_fourstep_tiled(c, in, in_is_real, rows, n_in, K, inverse, out, clean);
    return fft_fourstep(c, in, in_is_real, rows, n_in, K, inverse, out, clean);
  }
  return fft_bluestein_big(c, in, in_is_real, rows, n_in, K, inverse, out, clean);
}

int launch_rows_post(Ctx* c, float2* a, int64_t rows, int64_t K, const float* window, float scale, bool has_scale, float div, bool has_div) {
  if (rows == 0 || (!window && !has_scale && !has_div)) return NXSIG_OK;
  hipLaunchKernelGGL(k_rows_post, dim3(blocks_for(rows * K)), dim3(kT), 0, c->stream, a, rows * K, K, window, scale, has_scale ? 1 : 0, div, has_div ? 1 : 0);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// ================================================================================================ fft_nd
// in: device f32 / c64 tensor of `rank` dims (row-major); out: c64 tensor whose dims at `axes` are `lengths`
int launch_fft_nd(Ctx* c, const void* in, bool in_is_real, const int64_t* shape, int rank, const int32_t* axes, const int64_t* lengths,
                  int n_axes, bool inverse, float2* out) {
  if (rank < 1 || rank > 8) return set_error(NXSIG_ERR_INVALID_ARG, "fft_nd: rank must be in [1, 8]");
  std::vector<int64_t> cur(shape, shape + rank);
  int64_t total = 1, max_total = 1;
  for (int d = 0; d < rank; ++d) { if (cur[d] < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fft_nd: empty dimension"); total *= cur[d]; }
  // the largest intermediate decides the scratch size
  {
    std::vector<int64_t> s2 = cur;
    max_total = total;
    for (int i = 0; i < n_axes; ++i) {
      int ax = axes[i] < 0 ? axes[i] + rank : axes[i];
      if (ax < 0 || ax >= rank) return set_error(NXSIG_ERR_INVALID_ARG, "fft_nd: axis out of bounds");
      if (lengths[i] < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fft_nd: lengths must be positive");
      s2[ax] = lengths[i];
      int64_t t = 1;
      for (auto v : s2) t *= v;
      if (t > max_total) max_total = t;
    }
  }
  if (n_axes == 0) {  // Enum.zip_reduce over nothing: the tensor itself (as c64)
    hipLaunchKernelGGL(k_copy_pad, dim3(blocks_for(total)), dim3(kT), 0, c->stream, in, in_is_real ? 1 : 0, (int64_t)1, total, total, out);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  }
  void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
  int rc;
  const size_t bytes = (size_t)max_total * sizeof(float2);
  if ((rc = ctx_scratch(c, 10, bytes, &p0))) return rc;
  if ((rc = ctx_scratch(c, 11, bytes, &p1))) return rc;
  if ((rc = ctx_scratch(c, 12, bytes, &p2))) return rc;
  float2* bufs[3] = {reinterpret_cast<float2*>(p0), reinterpret_cast<float2*>(p1), reinterpret_cast<float2*>(p2)};
  const void* src = in;
  bool src_real = in_is_real;
  int next = 0;
  auto take = [&]() { float2* b = bufs[next]; next = (next + 1) % 3; return b; };
  for (int i = 0; i < n_axes; ++i) {
    const int ax = axes[i] < 0 ? axes[i] + rank : axes[i];
    const int64_t K = lengths[i], na = cur[ax];
    int64_t outer = 1, inner = 1;
    for (int d = 0; d < ax; ++d) outer *= cur[d];
    for (int d = ax + 1; d < rank; ++d) inner *= cur[d];
    const bool last = i == n_axes - 1;
    if (inner == 1) {  // the axis is already the fastest one
      float2* dst = last ? out : take();
      if ((rc = launch_fft_big(c, src, src_real, outer, na, K, inverse, dst))) return rc;
      src = dst;
    } else {
      if (na > 0x7fffffff || inner > 0x7fffffff || K > 0x7fffffff) return set_error(NXSIG_ERR_UNSUPPORTED, "fft_nd: dimension beyond 2^31");
      {
        float2* dst = last ? out : take();
        bool handled = false;
        if (reinterpret_cast<const void*>(dst) != src) {
          if ((rc = fft_columns_tiled(c, src, src_real, outer, na, inner, K, inverse, dst, &handled))) return rc;
        }
        if (handled) { src = dst; src_real = false; cur[ax] = K; continue; }
        if (!last) next = (next + 2) % 3;   // give the buffer back: take() order is fixed
      }
      float2* t1 = take();  // [outer][na][inner] -> [outer][inner][na]
      if ((rc = launch_transpose(c, src, src_real, outer, (int)na, (int)inner, na * inner, na * inner, t1))) return rc;
      float2* t2 = take();
      if ((rc = launch_fft_big(c, t1, false, outer * inner, na, K, inverse, t2))) return rc;
      float2* dst = last ? out : take();  // [outer][inner][K] -> [outer][K][inner]
      if ((rc = launch_transpose(c, t2, false, outer, (int)inner, (int)K, inner * K, inner * K, dst))) return rc;
      src = dst;
    }
    src_real = false;
    cur[ax] = K;
  }
  return NXSIG_OK;
}

// ================================================================================================ n-D fftconvolve
struct BcastArgs {
  int32_t rank;
  int64_t oshape[8], astride[8], bstride[8];  // stride 0 where the operand's dimension is 1 (broadcast)
  int64_t total;
};
__global__ __launch_bounds__(kT) void k_bcast_mul(const float2* __restrict__ a, const float2* __restrict__ b, BcastArgs g, float2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= g.total) return;
  int64_t rem = i, ia = 0, ib = 0;
  for (int d = g.rank - 1; d >= 0; --d) {
    const int64_t q = rem / g.oshape[d], x = rem - q * g.oshape[d];
    ia += x * g.astride[d]; ib += x * g.bstride[d];
    rem = q;
  }
  out[i] = ndmul(a[ia], b[ib]);
}
struct SliceArgs {
  int32_t rank, real_out;
  int64_t oshape[8], istride[8], start[8];
  int64_t total;
};
__global__ __launch_bounds__(kT) void k_slice_out(const float2* __restrict__ in, SliceArgs g, void* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= g.total) return;
  int64_t rem = i, ii = 0;
  for (int d = g.rank - 1; d >= 0; --d) {
    const int64_t q = rem / g.oshape[d], x = rem - q * g.oshape[d];
    ii += (x + g.start[d]) * g.istride[d];
    rem = q;
  }
  const float2 v = in[ii];
  if (g.real_out) reinterpret_cast<float*>(out)[i] = v.x;  // Nx.real of the ifft for real operands (convolution.ex:286-291)
  else reinterpret_cast<float2*>(out)[i] = v;
}

// a, b: device tensors of equal rank (f32 when *_is_real, else c64); out: f32 when both are real, else c64, shape per mode
int launch_fftconvolve_nd(Ctx* c, const void* a, bool a_is_real, const int64_t* s1, const void* b, bool b_is_real, const int64_t* s2,
                          int rank, int mode, void* out, int64_t* out_shape) {
  if (rank < 1 || rank > 8) return set_error(NXSIG_ERR_INVALID_ARG, "fftconvolve: rank must be in [1, 8]");
  int64_t full[8], padded[8], res[8], start[8];
  const bool pow2_lengths = tune(c, kT_CONV_POW2, 1) != 0;
  std::vector<int32_t> axes;
  std::vector<int64_t> lens;
  for (int d = 0; d < rank; ++d) {
    if (s1[d] < 1 || s2[d] < 1) return set_error(NXSIG_ERR_INVALID_ARG, "fftconvolve: empty dimension");
    full[d] = s1[d] + s2[d] - 1;
    padded[d] = full[d];
    if (s1[d] != 1 && s2[d] != 1) {  // convolution.ex:266-276
      // The reference transforms at exactly s1 + s2 - 1 points.  Any length >= that gives the same linear convolution (the
      // circular wrap never reaches the first s1 + s2 - 1 samples), so a non-power-of-two length runs at the next power of two:
      // the tuned row kernels instead of Bluestein (which would pad to >= 2 (s1 + s2 - 1) - 1 internally anyway).
      if (pow2_lengths) while (padded[d] & (padded[d] - 1)) padded[d] = (padded[d] | (padded[d] - 1)) + 1;
      axes.push_back(d); lens.push_back(padded[d]);
    }
  }
  switch (mode) {  // apply_mode / centered, convolution.ex:300-347
    case NXSIG_CONV_FULL: for (int d = 0; d < rank; ++d) { res[d] = full[d]; start[d] = 0; } break;
    case NXSIG_CONV_SAME: for (int d = 0; d < rank; ++d) { res[d] = s1[d]; start[d] = (full[d] - s1[d]) / 2; } break;
    case NXSIG_CONV_VALID: {
      bool ok1 = true, ok2 = true;
      for (int d = 0; d < rank; ++d) { ok1 = ok1 && s1[d] >= s2[d]; ok2 = ok2 && s2[d] >= s1[d]; }
      if (!ok1 && !ok2)
        return set_error(NXSIG_ERR_INVALID_ARG, "For 'valid' mode, one must be at least as large as the other in every dimension.");
      for (int d = 0; d < rank; ++d) { res[d] = (ok1 ? s1[d] - s2[d] : s2[d] - s1[d]) + 1; start[d] = (full[d] - res[d]) / 2; }
    } break;
    default: return set_error(NXSIG_ERR_INVALID_ARG, "expected mode to be one of [:full, :same, :valid]");
  }
  // spectra: dims at the transformed axes become full[d]; the other dims keep each operand's own size (one of them is 1)
  int64_t sh1[8], sh2[8], osh[8], n1 = 1, n2 = 1, no = 1;
  for (int d = 0; d < rank; ++d) {
    const bool tr = s1[d] != 1 && s2[d] != 1;
    sh1[d] = tr ? padded[d] : s1[d]; sh2[d] = tr ? padded[d] : s2[d];
    osh[d] = padded[d];  // = max of the two for a broadcast axis
    n1 *= sh1[d]; n2 *= sh2[d]; no *= osh[d];
  }
  void *pa = nullptr, *pb = nullptr, *pc = nullptr;
  int rc;
  if ((rc = ctx_scratch(c, 13, (size_t)(n1 > no ? n1 : no) * sizeof(float2), &pa))) return rc;  // A, later the inverse transform's result
  if ((rc = ctx_scratch(c, 14, (size_t)n2 * sizeof(float2), &pb))) return rc;
  if ((rc = ctx_scratch(c, 15, (size_t)no * sizeof(float2), &pc))) return rc;
  float2 *A = reinterpret_cast<float2*>(pa), *B = reinterpret_cast<float2*>(pb), *C = reinterpret_cast<float2*>(pc);
  if ((rc = launch_fft_nd(c, a, a_is_real, s1, rank, axes.data(), lens.data(), (int)axes.size(), false, A))) return rc;
  if ((rc = launch_fft_nd(c, b, b_is_real, s2, rank, axes.data(), lens.data(), (int)axes.size(), false, B))) return rc;
  BcastArgs g;
  g.rank = rank; g.total = no;
  int64_t st1 = 1, st2 = 1;
  for (int d = rank - 1; d >= 0; --d) {
    g.oshape[d] = osh[d];
    g.astride[d] = sh1[d] == 1 && osh[d] != 1 ? 0 : st1;
    g.bstride[d] = sh2[d] == 1 && osh[d] != 1 ? 0 : st2;
    st1 *= sh1[d]; st2 *= sh2[d];
  }
  dispatch_note("fftconvolve_nd");
  hipLaunchKernelGGL(k_bcast_mul, dim3(blocks_for(no)), dim3(kT), 0, c->stream, A, B, g, C);
  NXSIG_HIP_TRY(hipGetLastError());
  // ifft_nd over the same axes (lengths = current sizes) into A's slot: A is dead once the product is queued (stream order)
  float2* D = A;
  if ((rc = launch_fft_nd(c, C, false, osh, rank, axes.data(), lens.data(), (int)axes.size(), true, D))) return rc;
  SliceArgs sl;
  sl.rank = rank; sl.real_out = (a_is_real && b_is_real) ? 1 : 0; sl.total = 1;
  int64_t st = 1;
  for (int d = rank - 1; d >= 0; --d) { sl.oshape[d] = res[d]; sl.start[d] = start[d]; sl.istride[d] = st; st *= osh[d]; sl.total *= res[d]; }
  hipLaunchKernelGGL(k_slice_out, dim3(blocks_for(sl.total)), dim3(kT), 0, c->stream, D, sl, out);
  NXSIG_HIP_TRY(hipGetLastError());
  if (out_shape) for (int d = 0; d < rank; ++d) out_shape[d] = res[d];
  return NXSIG_OK;
}

// ================================================================================================ direct convolution
// Convolution.convolve(method: :direct) — lib/nx_signal/convolution.ex:95-218: Nx.conv of in1 (zero-padded per mode) with the
// kernel reversed along every axis.  One thread per output element; the products are accumulated in double in the order the
// BinaryBackend walks the kernel window (row-major, ascending in1 index) and rounded once, so integer-valued inputs come out
// exact like in the reference's tests and random inputs match a sequential double restatement bit for bit.  O(out x kernel):
// the time-domain form, for short kernels; long filters belong to the FFT method (k_fir_wave / fftconvolve).
struct DirectArgs {
  int32_t rank, a_real, b_real;
  int64_t oshape[8], ashape[8], kshape[8], astride[8], kstride[8], shift[8];  // in1 index = out index + window index + shift
  int64_t total, ktotal;
};
__global__ __launch_bounds__(kT) void k_conv_direct(const void* __restrict__ a, const void* __restrict__ k, DirectArgs g, void* __restrict__ out) {
  const int64_t o = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (o >= g.total) return;
  int64_t base[8], j[8];
  int64_t rem = o;
  for (int d = g.rank - 1; d >= 0; --d) {
    const int64_t q = rem / g.oshape[d];
    base[d] = rem - q * g.oshape[d] + g.shift[d];
    rem = q;
    j[d] = 0;
  }
  const float* af = reinterpret_cast<const float*>(a);
  const float2* ac = reinterpret_cast<const float2*>(a);
  const float* kf = reinterpret_cast<const float*>(k);
  const float2* kc = reinterpret_cast<const float2*>(k);
  double acc_re = 0.0, acc_im = 0.0;
  for (int64_t t = 0; t < g.ktotal; ++t) {
    bool inside = true;
    int64_t ia = 0, ik = 0;
    for (int d = 0; d < g.rank; ++d) {
      const int64_t i = base[d] + j[d];
      inside = inside && i >= 0 && i < g.ashape[d];
      ia += i * g.astride[d];
      ik += (g.kshape[d] - 1 - j[d]) * g.kstride[d];   // the window holds the REVERSED kernel
    }
    if (inside) {
      double ar, ai = 0.0, br, bi = 0.0;
      if (g.a_real) ar = (double)af[ia]; else { const float2 v = ac[ia]; ar = (double)v.x; ai = (double)v.y; }
      if (g.b_real) br = (double)kf[ik]; else { const float2 v = kc[ik]; br = (double)v.x; bi = (double)v.y; }
      // f32 x f32 products are exact in double: each component of the complex product is rounded once, like Complex.multiply
      acc_re += ar * br - ai * bi;
      acc_im += ar * bi + ai * br;
    }
    for (int d = g.rank - 1; d >= 0; --d) {  // row-major odometer over the window
      if (++j[d] < g.kshape[d]) break;
      j[d] = 0;
    }
  }
  if (g.a_real && g.b_real) reinterpret_cast<float*>(out)[o] = (float)acc_re;
  else reinterpret_cast<float2*>(out)[o] = make_float2((float)acc_re, (float)acc_im);
}

// Real x real operands whose kernel spans the last two axes at most (leading axes of extent 1 in the kernel are a batch): every
// thread owns R = 8 consecutive outputs of a row and slides a register window over the input — one new input value and one
// kernel value per tap feed eight double-precision FMAs — instead of decoding an n-D odometer per product.  Same sums: each
// output accumulates its products in double in the window's row-major order (f32 x f32 is exact in double, so the fused
// multiply-add rounds exactly like multiply-then-add), inputs outside the operand contribute a signed zero like the padding.
struct DirectFast {
  const float* a; const float* k; float* out;
  int64_t nb, H, W, OH, OW, XB;   // batch planes; input rows / columns; output rows / columns; 8-wide output blocks per row
  int32_t K1, K2;
  int64_t sh1, sh2;               // input index = output index + tap index + shift
};

template <bool CHECK>
__device__ __forceinline__ double df_ld(const float* __restrict__ row, int64_t ix, int64_t W) {
  if (CHECK) return (ix >= 0 && ix < W) ? (double)row[ix] : 0.0;
  return (double)row[ix];
}

template <bool CHECK>
__device__ __forceinline__ void df_row(const float* __restrict__ ar, const float* __restrict__ kr, int K2, int64_t ix0, int64_t W,
                                       double (&acc)[8]) {
  constexpr int R = 8;
  double w[2 * R - 1];
#pragma unroll
  for (int r = 0; r < R - 1; ++r) w[r] = df_ld<CHECK>(ar, ix0 + r, W);
  int j = 0;
  for (; j + R <= K2; j += R) {
#pragma unroll
    for (int u = 0; u < R; ++u) w[R - 1 + u] = df_ld<CHECK>(ar, ix0 + j + R - 1 + u, W);
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const double kv = (double)kr[K2 - 1 - (j + u)];   // the window holds the REVERSED kernel
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = fma(w[u + r], kv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < R - 1; ++r) w[r] = w[R + r];
  }
  for (; j < K2; ++j) {
    w[R - 1] = df_ld<CHECK>(ar, ix0 + j + R - 1, W);
    const double kv = (double)kr[K2 - 1 - j];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = fma(w[r], kv, acc[r]);
#pragma unroll
    for (int r = 0; r < R - 1; ++r) w[r] = w[r + 1];
  }
}

__global__ __launch_bounds__(kT) void k_conv_direct_rr(DirectFast g) {
  constexpr int R = 8;
  const int64_t o = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (o >= g.nb * g.OH * g.XB) return;
  const int64_t by = o / g.XB, xb = o - by * g.XB;
  const int64_t b = by / g.OH, y = by - b * g.OH;
  const int64_t x0 = xb * R, ix0 = x0 + g.sh2;
  double acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.0;
  const bool interior = ix0 >= 0 && ix0 + (R - 1) + (g.K2 - 1) < g.W;
  for (int j1 = 0; j1 < g.K1; ++j1) {
    const int64_t iy = y + j1 + g.sh1;
    if (iy < 0 || iy >= g.H) continue;
    const float* ar = g.a + (b * g.H + iy) * g.W;
    const float* kr = g.k + (int64_t)(g.K1 - 1 - j1) * g.K2;
    if (interior) df_row<false>(ar, kr, g.K2, ix0, g.W, acc);
    else df_row<true>(ar, kr, g.K2, ix0, g.W, acc);
  }
  float* orow = g.out + (b * g.OH + y) * g.OW + x0;
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (x0 + r < g.OW) orow[r] = (float)acc[r];
}

// The same register window for complex operands (either or both): four outputs per thread, the products formed exactly like the
// first kernel's — re = ar br - ai bi and im = ar bi + ai br, each rounded once (a real operand carries a literal 0.0 imaginary
// part THROUGH the arithmetic, as there), then added to the running sums.
template <bool AC, bool CHECK>
__device__ __forceinline__ void dfc_ld(const void* __restrict__ row, int64_t ix, int64_t W, double& re, double& im) {
  re = 0.0; im = 0.0;
  if (CHECK && !(ix >= 0 && ix < W)) return;
  if (AC) { const float2 v = reinterpret_cast<const float2*>(row)[ix]; re = (double)v.x; im = (double)v.y; }
  else re = (double)reinterpret_cast<const float*>(row)[ix];
}

template <bool AC, bool KC, bool CHECK>
__device__ __forceinline__ void dfc_row(const void* __restrict__ ar, const void* __restrict__ kr, int K2, int64_t ix0, int64_t W,
                                        double (&acc_re)[4], double (&acc_im)[4]) {
  constexpr int R = 4;
  double wr[R], wi[R];
#pragma unroll
  for (int r = 0; r < R - 1; ++r) dfc_ld<AC, CHECK>(ar, ix0 + r, W, wr[r], wi[r]);
  for (int j = 0; j < K2; ++j) {
    dfc_ld<AC, CHECK>(ar, ix0 + j + R - 1, W, wr[R - 1], wi[R - 1]);
    double br, bi = 0.0;
    if (KC) { const float2 v = reinterpret_cast<const float2*>(kr)[K2 - 1 - j]; br = (double)v.x; bi = (double)v.y; }
    else br = (double)reinterpret_cast<const float*>(kr)[K2 - 1 - j];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc_re[r] += wr[r] * br - wi[r] * bi;   // outside the operand the window holds the padding's zeros, as in Nx.conv
      acc_im[r] += wr[r] * bi + wi[r] * br;
    }
#pragma unroll
    for (int r = 0; r < R - 1; ++r) { wr[r] = wr[r + 1]; wi[r] = wi[r + 1]; }
  }
}

template <bool AC, bool KC>
__global__ __launch_bounds__(kT) void k_conv_direct_cx(DirectFast g) {
  constexpr int R = 4;
  const int64_t o = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (o >= g.nb * g.OH * g.XB) return;
  const int64_t by = o / g.XB, xb = o - by * g.XB;
  const int64_t b = by / g.OH, y = by - b * g.OH;
  const int64_t x0 = xb * R, ix0 = x0 + g.sh2;
  double acc_re[R], acc_im[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { acc_re[r] = 0.0; acc_im[r] = 0.0; }
  const bool interior = ix0 >= 0 && ix0 + (R - 1) + (g.K2 - 1) < g.W;
  const size_t ea = AC ? sizeof(float2) : sizeof(float), ek = KC ? sizeof(float2) : sizeof(float);
  for (int j1 = 0; j1 < g.K1; ++j1) {
    const int64_t iy = y + j1 + g.sh1;
    if (iy < 0 || iy >= g.H) continue;
    const void* ar = reinterpret_cast<const char*>(g.a) + (size_t)((b * g.H + iy) * g.W) * ea;
    const void* kr = reinterpret_cast<const char*>(g.k) + (size_t)((int64_t)(g.K1 - 1 - j1) * g.K2) * ek;
    if (interior) dfc_row<AC, KC, false>(ar, kr, g.K2, ix0, g.W, acc_re, acc_im);
    else dfc_row<AC, KC, true>(ar, kr, g.K2, ix0, g.W, acc_re, acc_im);
  }
  float2* orow = reinterpret_cast<float2*>(g.out) + (b * g.OH + y) * g.OW + x0;
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (x0 + r < g.OW) orow[r] = make_float2((float)acc_re[r], (float)acc_im[r]);
}

// a, b: device tensors of equal rank (f32 when *_is_real, else c64); out f32 when both are real, else c64
int launch_convolve_direct(Ctx* c, const void* a, bool a_is_real, const int64_t* s1, const void* b, bool b_is_real, const int64_t* s2,
                           int rank, int mode, void* out, int64_t* out_shape) {
  if (rank < 1 || rank > 8) return set_error(NXSIG_ERR_INVALID_ARG, "convolve: rank must be in [1, 8]");
  for (int d = 0; d < rank; ++d)
    if (s1[d] < 1 || s2[d] < 1) return set_error(NXSIG_ERR_INVALID_ARG, "convolve: empty dimension");
  const void *vol = a, *ker = b;
  bool vol_real = a_is_real, ker_real = b_is_real;
  const int64_t *sv = s1, *sk = s2;
  if (mode == NXSIG_CONV_VALID) {  // convolution.ex:120-135: the larger operand becomes the volume
    bool ok1 = true, ok2 = true;
    for (int d = 0; d < rank; ++d) { ok1 = ok1 && s1[d] >= s2[d]; ok2 = ok2 && s1[d] <= s2[d]; }
    if (!ok1 && !ok2)
      return set_error(NXSIG_ERR_INVALID_ARG, "For :valid mode, one must be at least as large as the other in every dimension");
    if (!ok1) { vol = b; ker = a; vol_real = b_is_real; ker_real = a_is_real; sv = s2; sk = s1; }
  } else if (mode != NXSIG_CONV_FULL && mode != NXSIG_CONV_SAME) {
    return set_error(NXSIG_ERR_INVALID_ARG, "expected mode to be one of [:full, :same, :valid]");
  }
  DirectArgs g;
  g.rank = rank; g.a_real = vol_real ? 1 : 0; g.b_real = ker_real ? 1 : 0; g.total = 1; g.ktotal = 1;
  int64_t sa = 1, sb = 1;
  for (int d = rank - 1; d >= 0; --d) {
    const int64_t k = sk[d];
    int64_t pad_left, res;
    switch (mode) {  // padding of Nx.conv, convolution.ex:157-190
      case NXSIG_CONV_FULL: pad_left = k - 1; res = sv[d] + k - 1; break;
      case NXSIG_CONV_SAME: pad_left = (k - 1) - (k - 1) / 2; res = sv[d]; break;
      default: pad_left = 0; res = sv[d] - k + 1; break;
    }
    g.oshape[d] = res; g.ashape[d] = sv[d]; g.kshape[d] = k; g.astride[d] = sa; g.kstride[d] = sb; g.shift[d] = -pad_left;
    sa *= sv[d]; sb *= k;
    g.total *= res; g.ktotal *= k;
    if (out_shape) out_shape[d] = res;
  }
  if (g.total > (int64_t)0x7fffffff * kT) return set_error(NXSIG_ERR_UNSUPPORTED, "convolve: result too large for one launch");
  {
    const bool fast_on = tune(c, kT_DIRECT_FAST, 1) != 0;
    bool lead_one = true;
    for (int d = 0; d + 2 < rank; ++d) lead_one = lead_one && sk[d] == 1;
    if (fast_on && lead_one && sk[rank - 1] <= 0x7fffffff && (rank < 2 || sk[rank - 2] <= 0x7fffffff)) {
      DirectFast f;
      f.a = reinterpret_cast<const float*>(vol); f.k = reinterpret_cast<const float*>(ker); f.out = reinterpret_cast<float*>(out);
      f.nb = 1;
      for (int d = 0; d + 2 < rank; ++d) f.nb *= sv[d];
      f.W = sv[rank - 1]; f.OW = g.oshape[rank - 1]; f.K2 = (int32_t)sk[rank - 1]; f.sh2 = g.shift[rank - 1];
      if (rank >= 2) { f.H = sv[rank - 2]; f.OH = g.oshape[rank - 2]; f.K1 = (int32_t)sk[rank - 2]; f.sh1 = g.shift[rank - 2]; }
      else { f.H = 1; f.OH = 1; f.K1 = 1; f.sh1 = 0; }
      const bool rr = vol_real && ker_real;
      f.XB = rr ? (f.OW + 7) / 8 : (f.OW + 3) / 4;
      const int64_t threads = f.nb * f.OH * f.XB;
      dispatch_note("convolve_direct.fast");
      if (rr) hipLaunchKernelGGL(k_conv_direct_rr, dim3(blocks_for(threads)), dim3(kT), 0, c->stream, f);
      else if (!vol_real && !ker_real) hipLaunchKernelGGL((k_conv_direct_cx<true, true>), dim3(blocks_for(threads)), dim3(kT), 0, c->stream, f);
      else if (!vol_real) hipLaunchKernelGGL((k_conv_direct_cx<true, false>), dim3(blocks_for(threads)), dim3(kT), 0, c->stream, f);
      else hipLaunchKernelGGL((k_conv_direct_cx<false, true>), dim3(blocks_for(threads)), dim3(kT), 0, c->stream, f);
      NXSIG_HIP_TRY(hipGetLastError());
      return NXSIG_OK;
    }
  }
  dispatch_note("convolve_direct.generic");
  hipLaunchKernelGGL(k_conv_direct, dim3(blocks_for(g.total)), dim3(kT), 0, c->stream, vol, ker, g, out);
  NXSIG_HIP_TRY(hipGetLastError());
  return NXSIG_OK;
}

// ================================================================================================ stft for long transforms
__global__ __launch_bounds__(kT) void k_frames_windowed(const float* __restrict__ x, int64_t batch_stride, int64_t L, int64_t lo, int32_t reflect,
                                                        int64_t M, int32_t N, int32_t hop, const float* __restrict__ w, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;  // (m, n) of one row
  if (i >= M * N) return;
  const int64_t m = i / N;
  const int n = (int)(i - m * N);
  int64_t pos = m * hop + n - lo;
  const float* xr = x + (size_t)blockIdx.y * batch_stride;
  float v;
  if (reflect) {
    if (L == 1) v = xr[0];
    else {
      const int64_t period = 2 * (L - 1);
      pos %= period;
      if (pos < 0) pos += period;
      if (pos >= L) pos = period - pos;
      v = xr[pos];
    }
  } else v = (pos >= 0 && pos < L) ? xr[pos] : 0.0f;
  out[(size_t)blockIdx.y * M * N + i] = v * w[n];  // exact f32 product like lib/nx_signal.ex:101
}

int launch_stft_big(Ctx* c, const StftLaunch& s) {
  if (s.fr.M == 0 || s.batch == 0) return NXSIG_OK;
  void* fr = nullptr;
  int rc = ctx_scratch(c, 16, (size_t)s.batch * s.fr.M * s.fr.N * sizeof(float), &fr);
  if (rc) return rc;
  const int64_t per_row = s.fr.M * s.fr.N;
  if ((per_row + kT - 1) / kT > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "stft: too many frames for one launch");
  dispatch_note("stft.big");
  hipLaunchKernelGGL(k_frames_windowed, dim3(blocks_for(per_row), (unsigned)s.batch), dim3(kT), 0, c->stream, s.x, s.batch_stride, s.fr.L, s.fr.lo,
                     s.fr.reflect, s.fr.M, s.fr.N, s.fr.hop, s.window, reinterpret_cast<float*>(fr));
  NXSIG_HIP_TRY(hipGetLastError());
  const int64_t rows = (int64_t)s.batch * s.fr.M;
  if ((rc = launch_fft_big(c, fr, true, rows, s.fr.N, s.K, false, s.z))) return rc;
  return launch_rows_post(c, s.z, rows, s.K, nullptr, 1.0f, false, s.inv_scale_div, s.has_scale != 0);
}

}  // namespace nxsig
