// Bound of a matrix-core formulation of the FIR (tools only, not linked into libnxsig.so).
//
// Direct-form FIR y[n] = sum_t h[t] x[n - t] as a block-Toeplitz product on v_mfma_f32_32x32x16_f16 with a TWO-term f16 split of
// both operands (x = x1 + x2, h = h1 + h2 after an exact power-of-two scale; products x1 h1 + x1 h2 + x2 h1 accumulate in f32; the
// dropped x2 h2 is 2^-22 relative) — f32-class error at 3 MFMA products instead of the 6 a three-term bf16 split needs.
//
// One wave = one TILE of 2048 consecutive outputs, viewed as 32 blocks (m) x 64 samples (r = 32 nt + r'):
//   D_nt[m][r'] = sum_j A_j[m][k] B_{j - 2 nt}[k][r'],   A_j[m][k] = x[n0 - TP + 64 m + 16 j + k]       (data, from LDS)
//                                                         B_q[k][r'] = h[r' - k + TP - 16 q]              (constant, in registers)
// with TP = taps - 1 rounded up to 16.  The Toeplitz structure makes the 2 x 18 tiles of the two output halves the SAME 18 register
// tiles (x 2 split terms = 144 VGPRs), the data operand is one ds_read_b128 per (j, term) = 40 LDS reads per 108 MFMAs, and the
// accumulator layout (lanes = 32 consecutive samples, registers = blocks) stores whole 128-byte lines.
//
// Variants: FULL (HBM in, HBM out), NOHBM (the same tile re-read from L2, one store per tile elided) to bound the matrix-core side.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fir_mfma tools/fir_mfma.hip && tools/fir_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int TILE = 2048;
constexpr int TP = 256;               // (taps - 1) rounded up to 16
constexpr int NJ = TP / 16 + 4;       // data chunks per tile
constexpr int NQ = TP / 16 + 2;       // Toeplitz register tiles
constexpr int NCOL = (TILE + TP) / 64;
constexpr int PITCH = 144;            // bytes per LDS column of 64 halves (128 + 16: conflict-free ds_read_b128, see DESIGN)
constexpr int NLD = (TILE + TP) / 256;  // 16-byte loads per lane per tile

struct Args {
  const float* x; float* y; const h8* btab;
  long L;                 // row length (x and y rows are L apart)
  long tiles_per_row, total_tiles, tiles_per_wave;
  long n_first;           // full-convolution index of the first tile's first output
  long out_shift;         // y index = n - out_shift
  float inv_hscale;
  int nohbm;
  unsigned long long* prof;   // MODE 2: per-phase cycle sums {wait+scale+split, matrix phase, stores, tiles} accumulated by lane 0 of every wave
};

template <int MODE, bool WIDE = true, int PRIO = 0>
__global__ __launch_bounds__(256, 2) void k_fir_mfma(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* l1 = smem + wave * (2 * NCOL * PITCH);
  unsigned char* l2 = l1 + NCOL * PITCH;
  h8 B1[NQ], B2[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { B1[q] = a.btab[(q * 2 + 0) * 64 + lane]; B2[q] = a.btab[(q * 2 + 1) * 64 + lane]; }

  const long w_global = (long)blockIdx.x * 4 + wave;
  long t0 = w_global * a.tiles_per_wave, t1 = t0 + a.tiles_per_wave;
  if (t1 > a.total_tiles) t1 = a.total_tiles;
  if (t0 >= t1) return;
  const int m = lane & 31, g = lane >> 5;
  const int abase = m * PITCH + g * 16;
  const int wbase = (lane >> 4) * PITCH + (lane & 15) * 8;

  v4f pf[NLD];
  auto issue = [&](long t) {
    const long row = t / a.tiles_per_row, ti = t - row * a.tiles_per_row;
    const long n0 = a.n_first + (MODE == 1 ? 0 : ti * TILE);
    const v4f* src = reinterpret_cast<const v4f*>(a.x + row * a.L + (n0 - TP)) + lane;
#pragma unroll
    for (int i = 0; i < NLD; ++i) pf[i] = src[64 * i];
  };
  issue(t0);
  unsigned long long c_conv = 0, c_mat = 0, c_st = 0;
  for (long t = t0; t < t1; ++t) {
    const unsigned long long ta = MODE == 2 ? __builtin_readcyclecounter() : 0;
    // ---- scale: exact power of two that puts the tile's largest |x| in [2^13, 2^14)
    unsigned mx = 0;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      mx = max(mx, __float_as_uint(pf[i].x) & 0x7fffffffu); mx = max(mx, __float_as_uint(pf[i].y) & 0x7fffffffu);
      mx = max(mx, __float_as_uint(pf[i].z) & 0x7fffffffu); mx = max(mx, __float_as_uint(pf[i].w) & 0x7fffffffu);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    int e = (int)(mx >> 23);
    e = e < 14 ? 14 : (e > 254 ? 254 : e);
    const float sc = __uint_as_float((unsigned)(267 - e) << 23);
    const float isc = __uint_as_float((unsigned)(e - 13) << 23);
    // ---- split into two f16 terms, park in the wave's LDS columns
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const v4f s = pf[i] * sc;
      const h4 hi = __builtin_convertvector(s, h4);
      const v4f r = s - __builtin_convertvector(hi, v4f);
      const h4 lo = __builtin_convertvector(r, h4);
      *reinterpret_cast<h4*>(l1 + wbase + 4 * i * PITCH) = hi;
      *reinterpret_cast<h4*>(l2 + wbase + 4 * i * PITCH) = lo;
    }
    if (t + 1 < t1) issue(t + 1);   // next tile's samples land during the matrix phase
    const unsigned long long tb = MODE == 2 ? __builtin_readcyclecounter() : 0;
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
    f16v acc0 = {0}, acc1 = {0};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int off = abase + (j >> 2) * PITCH + (j & 3) * 32;
      const h8 a1 = *reinterpret_cast<const h8*>(l1 + off);
      const h8 a2 = *reinterpret_cast<const h8*>(l2 + off);
      if (j < NQ) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, B1[j], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, B2[j < NQ ? j : 0], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, B1[j < NQ ? j : 0], acc0, 0, 0, 0);
      }
      if (j >= 2) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, B1[j - 2 >= 0 ? j - 2 : 0], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, B2[j - 2 >= 0 ? j - 2 : 0], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, B1[j - 2 >= 0 ? j - 2 : 0], acc1, 0, 0, 0);
      }
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);   // the load / split / store phases run ahead of the partner's matrix phase
    if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
    if (MODE == 2) { asm volatile("" :: "v"(acc0), "v"(acc1)); }
    const unsigned long long tc = MODE == 2 ? __builtin_readcyclecounter() : 0;
    const long row = t / a.tiles_per_row, ti = t - row * a.tiles_per_row;
    const long n0 = a.n_first + ti * TILE;
    float* yp = a.y + row * a.L + (n0 - a.out_shift) + m;
    const float os = a.inv_hscale;
    if (MODE != 1 || t + 1 == t1) {
      if (WIDE) {
        // the accumulators (lanes = 32 consecutive samples, registers = blocks) turn through the wave's idle LDS columns into
        // 16 bytes per lane: 8 store instructions of 1 KiB instead of 32 of 256 B
        float* tl = reinterpret_cast<float*>(l1);
        const float osc = os * isc;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int mm = (v & 3) + 8 * (v >> 2) + 4 * g;
          tl[64 * mm + m] = acc0[v] * osc;
          tl[64 * mm + 32 + m] = acc1[v] * osc;
        }
        v4f* yo = reinterpret_cast<v4f*>(a.y + row * a.L + (n0 - a.out_shift)) + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(reinterpret_cast<const v4f*>(tl)[64 * i + lane], yo + 64 * i);
      } else {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int mm = (v & 3) + 8 * (v >> 2) + 4 * g;
        __builtin_nontemporal_store(acc0[v] * os * isc, yp + 64 * mm);
        __builtin_nontemporal_store(acc1[v] * os * isc, yp + 64 * mm + 32);
      }
      }
    } else {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) s += acc0[v] + acc1[v];
      if (s == 123456.789f) yp[0] = s;   // keeps the matrix phase alive
    }
    if (MODE == 2) { const unsigned long long td = __builtin_readcyclecounter(); c_conv += tb - ta; c_mat += tc - tb; c_st += td - tc; }
  }
  if (MODE == 2 && lane == 0) { atomicAdd(a.prof + 0, c_conv); atomicAdd(a.prof + 1, c_mat); atomicAdd(a.prof + 2, c_st); atomicAdd(a.prof + 3, (unsigned long long)(t1 - t0)); }
}

static std::vector<double> firwin_lp(int taps, double fc, double fs) {   // Hamming-windowed sinc, DC gain 1 (tool only)
  std::vector<double> h(taps);
  const double c = fc / (fs / 2), al = 0.5 * (taps - 1);
  double s = 0;
  for (int i = 0; i < taps; ++i) {
    const double mm = i - al, w = 0.54 - 0.46 * std::cos(2 * M_PI * i / (taps - 1));
    const double v = (mm == 0 ? c : std::sin(M_PI * c * mm) / (M_PI * mm)) * w;
    h[i] = v; s += v;
  }
  for (auto& v : h) v /= s;
  return h;
}

int main(int argc, char** argv) {
  const long rows = 8, L = 28800000;
  const int taps = 257;
  long tpw = argc > 1 ? atol(argv[1]) : 0;
  std::vector<double> hd = firwin_lp(taps, 4000, 48000);
  std::vector<float> hf(taps);
  for (int i = 0; i < taps; ++i) hf[i] = (float)hd[i];
  float hmax = 0; for (float v : hf) hmax = std::fmax(hmax, std::fabs(v));
  int he; std::frexp(hmax, &he);            // hmax = f * 2^he, f in [0.5, 1)
  const float hscale = std::ldexp(1.0f, 14 - he);   // hmax * hscale in [2^13, 2^14)
  std::vector<_Float16> bt((size_t)NQ * 2 * 64 * 8);
  for (int q = 0; q < NQ; ++q)
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 8; ++e) {
        const int rp = l & 31, k = 8 * (l >> 5) + e, t = rp - k + TP - 16 * q;
        float v = (t >= 0 && t < taps) ? hf[t] * hscale : 0.f;
        const _Float16 h1 = (_Float16)v, h2 = (_Float16)(v - (float)h1);
        bt[((size_t)(q * 2 + 0) * 64 + l) * 8 + e] = h1;
        bt[((size_t)(q * 2 + 1) * 64 + l) * 8 + e] = h2;
      }
  float *x, *y; h8* btd;
  CK(hipMalloc(&x, rows * L * 4)); CK(hipMalloc(&y, rows * L * 4)); CK(hipMalloc(&btd, bt.size() * 2));
  CK(hipMemcpy(btd, bt.data(), bt.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> hx(L);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  for (long i = 0; i < L; i += 2) { const double u1 = rnd() + 1e-300, u2 = rnd(); const double r = std::sqrt(-2 * std::log(u1)); hx[i] = (float)(r * std::cos(2 * M_PI * u2)); if (i + 1 < L) hx[i + 1] = (float)(r * std::sin(2 * M_PI * u2)); }
  for (long r = 0; r < rows; ++r) CK(hipMemcpy(x + r * L, hx.data(), L * 4, hipMemcpyHostToDevice));   // same stream in every row (tool)
  CK(hipMemset(y, 0, rows * L * 4));
  Args a{};
  a.x = x; a.y = y; a.btab = btd; a.L = L;
  a.n_first = 2048; a.out_shift = 128;   // mode :same of 257 taps
  a.tiles_per_row = (L - a.n_first - TILE) / TILE;
  a.total_tiles = rows * a.tiles_per_row;
  a.inv_hscale = 1.0f / hscale;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = 4 * 2 * NCOL * PITCH;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fir_mfma<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fir_mfma<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fir_mfma<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fir_mfma<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fir_mfma<0, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fir_mfma<0, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fir_mfma<2, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipMalloc(&a.prof, 64)); CK(hipMemset(a.prof, 0, 64));
  const double bytes = (double)a.total_tiles * TILE * 8;
  auto run = [&](int mode, long tiles_per_wave, int reps) {
    a.tiles_per_wave = tiles_per_wave; a.nohbm = mode;
    const unsigned grid = (unsigned)((a.total_tiles + 4 * tiles_per_wave - 1) / (4 * tiles_per_wave));
    auto go = [&] {
      switch (mode) {
        case 0: hipLaunchKernelGGL((k_fir_mfma<0>), dim3(grid), dim3(256), lds, 0, a); break;
        case 1: hipLaunchKernelGGL((k_fir_mfma<1>), dim3(grid), dim3(256), lds, 0, a); break;
        case 2: hipLaunchKernelGGL((k_fir_mfma<2>), dim3(grid), dim3(256), lds, 0, a); break;
        case 3: hipLaunchKernelGGL((k_fir_mfma<0, false>), dim3(grid), dim3(256), lds, 0, a); break;
        case 4: hipLaunchKernelGGL((k_fir_mfma<0, true, 1>), dim3(grid), dim3(256), lds, 0, a); break;
        case 5: hipLaunchKernelGGL((k_fir_mfma<0, true, 2>), dim3(grid), dim3(256), lds, 0, a); break;
        case 6: hipLaunchKernelGGL((k_fir_mfma<2, true, 1>), dim3(grid), dim3(256), lds, 0, a); break;
      }
    };
    for (int i = 0; i < 10; ++i) go();
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%s  tiles/wave %4ld  grid %6u  %8.1f us  %7.1f GB/s algorithmic (8 B/sample)  %6.1f G samples/s  mfma %5.0f TF\n", (const char*[]){"FULL wide ", "NOHBM     ", "PROF wide ", "FULL dword", "FULL prio1", "FULL prio2", "PROF prio1"}[mode], tiles_per_wave, grid,
           ms * 1e3, bytes / ms / 1e6, bytes / 8 / ms / 1e6, (double)a.total_tiles * 108 * 32768.0 / ms / 1e9);
    return ms;
  };
  if (argc > 2 && !strcmp(argv[1], "loop")) {   // keep the kernel running for argv[2] seconds (rocm-smi beside it); argv[3] = mode
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const double secs = atof(argv[2]);
    double done = 0; while (done < secs) done += run(mode, 55, 200) * 200 / 1e3;
    return 0;
  }
  if (tpw > 0) { run(0, tpw, 30); }
  else {
    for (long t : {14L, 28L, 55L}) run(0, t, 30);
    for (int md : {3, 4, 5, 0, 3, 4, 5}) run(md, 55, 30);
    for (long t : {28L, 55L}) run(1, t, 30);
    for (int md : {2, 6}) {
      CK(hipMemset(a.prof, 0, 64));
      run(md, 55, 30);
      unsigned long long pr[4]; CK(hipMemcpy(pr, a.prof, 32, hipMemcpyDeviceToHost));
      printf("PROF cycles per tile (one wave's view, 2 waves per SIMD): load-wait + scale + split %.0f, matrix phase %.0f, stores %.0f  (tiles %llu)\n", (double)pr[0] / pr[3], (double)pr[1] / pr[3], (double)pr[2] / pr[3], pr[3]);
    }
  }
  // ---- correctness of the FULL variant against direct convolution in double
  CK(hipMemset(y, 0, rows * L * 4));
  run(0, 28, 1);
  std::vector<float> hy(1 << 20);
  const long off = 5 * TILE + 2048 - 128;
  CK(hipMemcpy(hy.data(), y + 3 * L + off, hy.size() * 4, hipMemcpyDeviceToHost));
  double emax = 0, rmax = 0;
  for (long i = 0; i < (long)hy.size(); i += 7) {
    const long n = off + i + 128;   // full-convolution index
    double acc = 0;
    for (int t = 0; t < taps; ++t) acc += (double)hf[t] * (double)hx[n - t];
    emax = std::fmax(emax, std::fabs(acc - hy[i])); rmax = std::fmax(rmax, std::fabs(acc));
  }
  printf("max|err| / max|ref| = %.3e  (ref max %.3f) over %zu samples of row 3\n", emax / rmax, rmax, hy.size() / 7);
  // the last tile of the last row too
  const long lastn = a.n_first + (a.tiles_per_row - 1) * TILE;
  CK(hipMemcpy(hy.data(), y + 7 * L + lastn - 128, TILE * 4, hipMemcpyDeviceToHost));
  emax = 0;
  for (long i = 0; i < TILE; ++i) { const long n = lastn + i; double acc = 0; for (int t = 0; t < taps; ++t) acc += (double)hf[t] * (double)hx[n - t]; emax = std::fmax(emax, std::fabs(acc - hy[i])); }
  printf("last tile of row 7: max|err| / max|ref| = %.3e\n", emax / rmax);
  return 0;
}
