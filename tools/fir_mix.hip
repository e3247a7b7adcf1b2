// No-math ceiling of the FIR kernel's traffic (tools only).  Geometry of k_fir_wave<1024, true, 14>: 257 taps, V = 768; a wave
// takes one block PAIR per iteration: reads the pair's input and writes 2 V = 1536 contiguous outputs; 14 waves per workgroup,
// `chunk` pairs per workgroup.  Variants:
//   L8  : the kernel's loads — two blocks of 1024 samples, 8 B per lane, stride 512 B (the 256-sample overlap inside the pair is
//         fetched twice)
//   L16 : the pair's 1792-sample span once, 16 B per lane (7 loads of 1 KiB)
//   S8 / S16 : 8- or 16-byte streaming stores (sc1 nt through a buffer descriptor)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long v = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}

template <int L16, int S16, int W>
__global__ __launch_bounds__(64 * W) void k_firmix(const float* __restrict__ x, float* __restrict__ y, long L, long pairs_per_row, long total, long chunk) {
  constexpr int V = 768, TM1 = 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long p0 = (long)blockIdx.x * chunk;
  long p1 = p0 + chunk; if (p1 > total) p1 = total;
  for (long p = p0 + wave; p < p1; p += W) {
    const long row = p / pairs_per_row, pi = p - row * pairs_per_row;
    const float* src = x + row * L + pi * 2 * V;         // span [src, src + V + 1024)
    float* dst = y + row * L + pi * 2 * V + TM1;
    v2f acc = {0.f, 0.f};
    v4f acc4 = {0.f, 0.f, 0.f, 0.f};
    if (L16) {
      const v4f* s4 = reinterpret_cast<const v4f*>(src) + lane;
#pragma unroll
      for (int j = 0; j < 7; ++j) acc4 += s4[64 * j];
      acc = v2f{acc4.x + acc4.z, acc4.y + acc4.w};
    } else {
      const v2f* s2 = reinterpret_cast<const v2f*>(src) + lane;
#pragma unroll
      for (int q = 0; q < 8; ++q) { acc += s2[64 * q]; acc += s2[V / 2 + 64 * q]; }
      acc4 = v4f{acc.x, acc.y, acc.x, acc.y};
    }
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(dst, 2 * V * 4);
    if (S16) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const v4f o = acc4 * (float)(j + 1);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, o), rs, lane * 16 + 1024 * j, 0, 18);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const v2f o = acc * (float)(j + 1);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, o), rs, lane * 8 + 512 * j, 0, 18);
      }
    }
  }
}

int main() {
  const long rows = 8, L = 28800000, V = 768;
  const long pairs_per_row = (L - 1024 - V) / (2 * V);
  const long total = rows * pairs_per_row;
  float *a, *b;
  CK(hipMalloc(&a, rows * L * 4)); CK(hipMalloc(&b, rows * L * 4));
  CK(hipMemset(a, 1, rows * L * 4)); CK(hipMemset(b, 0, rows * L * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch) { for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e0)); for (int i = 0; i < 30; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 30; };
  const double bytes = (double)total * 2 * V * 8;   // algorithmic: 4 B in + 4 B out per output sample
#define RUN(L16, S16, W, CH) { const long chunk = CH; const unsigned grid = (unsigned)((total + chunk - 1) / chunk); \
    float ms = time([&] { hipLaunchKernelGGL((k_firmix<L16, S16, W>), dim3(grid), dim3(64 * W), 0, 0, a, b, L, pairs_per_row, total, chunk); }); \
    printf("loads %2d B  stores %2d B  %2d waves/WG  chunk %4ld pairs  %7.1f GB/s algorithmic (8 B/sample)\n", L16 ? 16 : 8, S16 ? 16 : 8, W, chunk, bytes / ms / 1e6); }
  for (long ch : {14L, 28L, 56L, 112L}) {
    RUN(0, 0, 14, ch) RUN(1, 0, 14, ch) RUN(0, 1, 14, ch) RUN(1, 1, 14, ch)
  }
  RUN(0, 0, 4, 16) RUN(1, 1, 4, 16) RUN(0, 0, 8, 32) RUN(1, 1, 8, 32)
  return 0;
}
