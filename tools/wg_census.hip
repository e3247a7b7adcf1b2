// Where do the workgroups of a ONE-ROUND launch land?  (round 6, config 2 as written)  G workgroups of T threads with LDS bytes each
// spin for ~5 us and record {XCC id, HW id, start, end}; the host prints workgroups per CU and how many started after the first one ended.
//   hipcc --offload-arch=gfx950 -O3 tools/wg_census.hip -o tools/wg_census && tools/wg_census 256 768 122880
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
extern __shared__ unsigned char smem[];
__global__ void k_census(unsigned long long* out, int spin) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  smem[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  unsigned long long t1 = t0;
  while (t1 - t0 < (unsigned long long)spin) t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = xcc & 0xf;
    out[4 * blockIdx.x + 1] = hw;
    out[4 * blockIdx.x + 2] = t0;
    out[4 * blockIdx.x + 3] = t1 + smem[5];
  }
}
int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 768, lds = argc > 3 ? atoi(argv[3]) : 122880;
  unsigned long long* d;
  hipMalloc(&d, (size_t)G * 32);
  hipFuncSetAttribute((const void*)k_census, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_census, dim3(G), dim3(T), lds, 0, d, 10000);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h((size_t)G * 4);
  hipMemcpy(h.data(), d, (size_t)G * 32, hipMemcpyDeviceToHost);
  std::map<unsigned long long, int> per_cu;
  std::map<int, int> per_xcc;
  unsigned long long first_end = ~0ull, t_min = ~0ull;
  for (int b = 0; b < G; ++b) {
    const unsigned hw = (unsigned)h[4 * b + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per_cu[(h[4 * b] << 16) | (se << 8) | (sh << 4) | cu]++;
    per_xcc[(int)h[4 * b]]++;
    if (h[4 * b + 3] < first_end) first_end = h[4 * b + 3];
    if (h[4 * b + 2] < t_min) t_min = h[4 * b + 2];
  }
  int late = 0, hist[8] = {};
  for (int b = 0; b < G; ++b) if (h[4 * b + 2] >= first_end) ++late;
  for (auto& kv : per_cu) hist[kv.second < 7 ? kv.second : 7]++;
  printf("G=%d T=%d lds=%d: distinct CUs %zu; CUs holding 1/2/3/4+ workgroups: %d %d %d %d; workgroups that started after the first one ended: %d\n",
         G, T, lds, per_cu.size(), hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7], late);
  printf("  per XCC:");
  for (auto& kv : per_xcc) printf(" %d:%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
