// PMC calibration (tools only): known byte counts in the access patterns of the STFT kernel, so FETCH_SIZE / WRITE_SIZE
// of rocprofv3 can be converted to bytes with measured factors (MI355X_MICROARCH.md: "calibrate on a known byte
// count in your own access pattern").  Buffers are 2 GiB (>> 256 MiB Infinity Cache).
//   k_read_dword : every lane reads 4 B, 256 B contiguous per wave instruction (the STFT frame loader), sums, writes 4 B/wave
//   k_read_x4    : 16 B per lane streaming read
//   k_write_x4nt : 16 B per lane non-temporal stores (the STFT spectrum stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read_dword(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
  if (acc == 12345.678f) out[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_read_x4(const v4f* __restrict__ in, float* __restrict__ out, size_t n) {
  v4f acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[threadIdx.x] = acc.x;
}
__global__ __launch_bounds__(256) void k_write_x4nt(v4f* __restrict__ out, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(v4f{v, v + 1, v + 2, v + 3}, &out[i]);
}
// 16 B per lane through a buffer descriptor with the "sc1 nt" cache policy (the STFT spectrum stores since round 2)
__global__ __launch_bounds__(256) void k_write_x4_sc1nt(v4f* __restrict__ out, size_t n, float v) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  const size_t per_block = 4096;  // elements: each block walks 64 KiB pieces, one descriptor per piece
  for (size_t p = (size_t)blockIdx.x * per_block; p < n; p += (size_t)gridDim.x * per_block) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + p, 0, (int)(per_block * 16), 0x00020000);
    for (int i = threadIdx.x; i < (int)per_block; i += 256)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v4f{v, v + 1, v + 2, v + 3}), r, i * 16, 0, 18);
  }
}
int main() {
  const size_t bytes = (size_t)2048 << 20;
  float *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes));
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_read_dword, dim3(8192), dim3(256), 0, 0, a, b, bytes / 4);
  hipLaunchKernelGGL(k_read_x4, dim3(8192), dim3(256), 0, 0, (const v4f*)a, b, bytes / 16);
  hipLaunchKernelGGL(k_write_x4nt, dim3(8192), dim3(256), 0, 0, (v4f*)b, bytes / 16, 1.0f);
  hipLaunchKernelGGL(k_write_x4_sc1nt, dim3(8192), dim3(256), 0, 0, (v4f*)b, bytes / 16, 2.0f);
  CK(hipDeviceSynchronize());
  printf("calibration kernels moved %zu bytes each\n", bytes);
  return 0;
}
