// Which gfx950 store cache policy streams an STFT-shaped write fastest?  (tools only)
// Pattern = the headline kernel's: every wave writes 2 frames x 8 KiB (1 KiB per wave instruction), a workgroup of 4 waves
// covers a contiguous 128 KiB chunk (2 pairs per wave), ~22 000 short-lived workgroups; optionally 1 KiB read per frame.
// Policies: global_store_dwordx4 with every combination of the sc0 / sc1 / nt bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int POL>
__device__ __forceinline__ void st(v4f* p, v4f v) {
  if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
  if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int POL, int READ>
__global__ __launch_bounds__(256) void k_mix(const v4f* __restrict__ in, v4f* __restrict__ out, size_t pairs, int ppw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t p0 = (size_t)blockIdx.x * 4 * ppw;
  for (int it = 0; it < ppw; ++it) {
    const size_t pr = p0 + wave + 4 * it;
    if (pr >= pairs) return;
    v4f v = {1.f, 2.f, 3.f, (float)lane};
    if (READ) { v = in[pr * 128 + lane]; v += in[pr * 128 + 64 + lane]; }
    v4f* o = out + pr * 1024 + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) st<POL>(o + 64 * j, v + (float)j);
  }
}

int main() {
  const size_t pairs = 32 * 11247 / 2;  // the bench's launch: 3.3 GB
  v4f *a, *b;
  CK(hipMalloc(&a, pairs * 2048)); CK(hipMalloc(&b, pairs * 16384));
  CK(hipMemset(a, 1, pairs * 2048)); CK(hipMemset(b, 0, pairs * 16384));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch) { for (int i = 0; i < 40; ++i) launch(); CK(hipEventRecord(e0)); for (int i = 0; i < 40; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 40; };
  const char* names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  for (int ppw : {2, 8}) {
    const unsigned grid = (unsigned)((pairs + 4 * ppw - 1) / (4 * ppw));
#define RUN(P)                                                                                                             \
    {                                                                                                                      \
      float w = time([&] { hipLaunchKernelGGL((k_mix<P, 0>), dim3(grid), dim3(256), 0, 0, a, b, pairs, ppw); });            \
      float m = time([&] { hipLaunchKernelGGL((k_mix<P, 1>), dim3(grid), dim3(256), 0, 0, a, b, pairs, ppw); });            \
      printf("ppw %d  %-11s fill %7.1f GB/s   mix(1 KiB read + 8 KiB write per frame) %7.1f GB/s\n", ppw, names[P],       \
             pairs * 16384.0 / w / 1e6, pairs * 18432.0 / m / 1e6);                                                        \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
  }
  return 0;
}
