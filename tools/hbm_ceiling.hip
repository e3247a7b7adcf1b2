// Empirical HBM ceilings on MI355X for the STFT traffic mix (tools only, not part of the library).
//   copy   : float4 read + float4 write (the 6.29 TB/s figure of MI355X_MICROARCH.md)
//   fill   : 16 B/lane stores only
//   stftmix: per "frame" read 1 KiB (hop*4) and write 8 KiB (K*8), no FFT — the best any STFT kernel can do
// usage: hbm_ceiling [MiB_out]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_fill(float4* __restrict__ out, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = make_float4(v, v + 1, v + 2, v + 3);
}
// one wave per frame: lane reads float4 (1 KiB per frame... 64 lanes x 16 B), writes 8 x float4 (8 KiB)
template <int NT>
__global__ __launch_bounds__(256) void k_stftmix(const float4* __restrict__ in, float4* __restrict__ out, size_t frames) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * 256) >> 6;
  for (size_t f = wave; f < frames; f += nw) {
    float4 v = in[f * 64 + lane];
    float4* o = out + f * 512;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float4 r = make_float4(v.x + j, v.y - j, v.z * (j + 1), v.w);
      typedef float v4f __attribute__((ext_vector_type(4)));
      v4f rv = {r.x, r.y, r.z, r.w};
      if (NT) __builtin_nontemporal_store(rv, reinterpret_cast<v4f*>(&o[j * 64 + lane])); else o[j * 64 + lane] = r;
    }
  }
}


// read-only: float4 loads, xor-reduced, one conditional store per thread (never taken)
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  if (acc.x == 12345.678f) out[threadIdx.x] = acc;
}
// istft mix: one wave per frame reads 8 KiB (LB bytes per lane per load) and NT-stores 2 KiB
template <int LB>
__global__ __launch_bounds__(256) void k_istftmix(const float* __restrict__ in, float4* __restrict__ out, size_t frames, size_t chunk) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef float v2f __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  size_t f0 = wave * chunk, f1 = f0 + chunk; if (f1 > frames) f1 = frames;
  for (size_t f = f0; f < f1; ++f) {
    v4f acc = {0, 0, 0, 0};
    if (LB == 16) {
      const v4f* p = reinterpret_cast<const v4f*>(in + f * 2048) + lane;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += p[64 * j];
    } else {
      const v2f* p = reinterpret_cast<const v2f*>(in + f * 2048) + lane;
#pragma unroll
      for (int j = 0; j < 16; ++j) { v2f v = p[64 * j]; acc.x += v.x; acc.y += v.y; }
    }
    v4f* o = reinterpret_cast<v4f*>(out) + f * 128 + lane;
    __builtin_nontemporal_store(acc, o);
    __builtin_nontemporal_store(acc + 1.0f, o + 64);
  }
}


// NT fill: each wave writes CH consecutive KiB (CH x 64 lanes x 16 B), waves take consecutive chunks
template <int CH>
__global__ __launch_bounds__(256) void k_fill_nt(float4* __restrict__ out, size_t n16) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * 256) >> 6;
  const size_t chunks = n16 / (64 * CH);
  for (size_t c = wave; c < chunks; c += nw) {
    v4f* o = reinterpret_cast<v4f*>(out) + c * 64 * CH + lane;
#pragma unroll
    for (int j = 0; j < CH; ++j) { v4f v = {1.f + j, 2.f, 3.f, (float)lane}; __builtin_nontemporal_store(v, o + 64 * j); }
  }
}

int main(int argc, char** argv) {
  size_t mib = argc > 1 ? atol(argv[1]) : 2048;
  size_t bytes = mib << 20;
  float4 *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t n = bytes / 16;
  auto time = [&](auto launch, int reps) { for (int i = 0; i < 3; ++i) launch(); CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
  for (int grid : {2048, 4096, 8192, 16384}) {
    float ms = time([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, 20);
    printf("copy     grid %6d: %8.3f ms  %7.1f GB/s (r+w)\n", grid, ms, 2.0 * bytes / ms / 1e6);
  }
  for (int grid : {2048, 8192}) {
    float ms = time([&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, b, n, 1.0f); }, 20);
    printf("fill     grid %6d: %8.3f ms  %7.1f GB/s (w)\n", grid, ms, 1.0 * bytes / ms / 1e6);
  }
  size_t frames = bytes / 8192;
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    float ms = time([&] { hipLaunchKernelGGL(k_stftmix<0>, dim3(grid), dim3(256), 0, 0, a, b, frames); }, 20);
    printf("stftmix  grid %6d: %8.3f ms  %7.1f GB/s (9216 B/frame) %7.1f Mframes/s\n", grid, ms, frames * 9216.0 / ms / 1e6, frames / ms / 1e3);
    ms = time([&] { hipLaunchKernelGGL(k_stftmix<1>, dim3(grid), dim3(256), 0, 0, a, b, frames); }, 20);
    printf("stftmix nt    %6d: %8.3f ms  %7.1f GB/s\n", grid, ms, frames * 9216.0 / ms / 1e6);
  }
  for (int grid : {2048, 8192, 16384}) {
    float ms = time([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, n); }, 20);
    printf("read     grid %6d: %8.3f ms  %7.1f GB/s (r)\n", grid, ms, 1.0 * bytes / ms / 1e6);
  }
  {
    size_t fr = bytes / 8192;
    for (size_t chunk : {8, 16, 64, 256}) {
      unsigned grid = (unsigned)((fr + chunk * 4 - 1) / (chunk * 4));
      float ms = time([&] { hipLaunchKernelGGL(k_istftmix<16>, dim3(grid), dim3(256), 0, 0, (const float*)a, b, fr, chunk); }, 20);
      printf("istftmix 16B chunk %4zu: %8.3f ms  %7.1f GB/s (10240 B/frame)\n", chunk, ms, fr * 10240.0 / ms / 1e6);
      ms = time([&] { hipLaunchKernelGGL(k_istftmix<8>, dim3(grid), dim3(256), 0, 0, (const float*)a, b, fr, chunk); }, 20);
      printf("istftmix  8B chunk %4zu: %8.3f ms  %7.1f GB/s\n", chunk, ms, fr * 10240.0 / ms / 1e6);
    }
  }
  for (int grid : {2048, 8192, 32768}) {
    float ms = time([&] { hipLaunchKernelGGL(k_fill_nt<1>, dim3(grid), dim3(256), 0, 0, b, n); }, 20);
    printf("fill nt 1KiB/wave grid %6d: %8.3f ms  %7.1f GB/s (w)\n", grid, ms, 1.0 * bytes / ms / 1e6);
    ms = time([&] { hipLaunchKernelGGL(k_fill_nt<8>, dim3(grid), dim3(256), 0, 0, b, n); }, 20);
    printf("fill nt 8KiB/wave grid %6d: %8.3f ms  %7.1f GB/s (w)\n", grid, ms, 1.0 * bytes / ms / 1e6);
    ms = time([&] { hipLaunchKernelGGL(k_fill_nt<16>, dim3(grid), dim3(256), 0, 0, b, n); }, 20);
    printf("fill nt 16KiB/wave grid %6d: %8.3f ms  %7.1f GB/s (w)\n", grid, ms, 1.0 * bytes / ms / 1e6);
  }
  return 0;
}
