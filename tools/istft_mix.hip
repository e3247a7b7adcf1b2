// Does a read-dominated stream (8 KiB read + 2 KiB written per "frame", the iSTFT mix) go faster when every wave reads larger
// contiguous bursts?  (tools only)  Each wave walks a run of consecutive frames of its row like k_istft_wave; per iteration it
// loads BURST frames (8 KiB each, 16-byte loads, 1 KiB per wave instruction) before consuming them, then stores 2 KiB per frame.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int BURST>
__global__ __launch_bounds__(256) void k_mix(const v4f* __restrict__ in, v4f* __restrict__ out, size_t frames, size_t run) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  size_t f0 = wave * run, f1 = f0 + run;
  if (f1 > frames) f1 = frames;
  for (size_t f = f0; f < f1; f += BURST) {
    v4f r[BURST][8];
#pragma unroll
    for (int b = 0; b < BURST; ++b)
#pragma unroll
      for (int j = 0; j < 8; ++j) r[b][j] = in[(f + b < f1 ? f + b : f) * 512 + 64 * j + lane];
#pragma unroll
    for (int b = 0; b < BURST; ++b) {
      v4f acc = r[b][0];
#pragma unroll
      for (int j = 1; j < 8; ++j) acc += r[b][j];
      if (f + b < f1) {
        __builtin_nontemporal_store(acc, out + (f + b) * 128 + lane);
        __builtin_nontemporal_store(acc * 2.0f, out + (f + b) * 128 + 64 + lane);
      }
    }
  }
}

// the STFT kernels' geometry for the same mix: short-lived workgroups, each takes a contiguous chunk of `cpw` frames per wave
// (wave w reads frames w, w + 4, ... of the chunk); `halo` extra frames are read (not written) in front of every chunk
__global__ __launch_bounds__(256) void k_chunk(const v4f* __restrict__ in, v4f* __restrict__ out, size_t frames, int cpw, int halo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t c0 = (size_t)blockIdx.x * 4 * cpw;
  v4f carry = {0, 0, 0, 0};
  for (int h = wave; h < halo; h += 4) {  // halo frames: read only
    const size_t f = c0 >= (size_t)halo ? c0 - halo + h : h;
#pragma unroll
    for (int j = 0; j < 8; ++j) carry += in[f * 512 + 64 * j + lane];
  }
  for (int it = 0; it < cpw; ++it) {
    const size_t f = c0 + wave + 4 * it;
    if (f >= frames) return;
    v4f acc = carry;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += in[f * 512 + 64 * j + lane];
    __builtin_nontemporal_store(acc, out + f * 128 + lane);
    __builtin_nontemporal_store(acc * 2.0f, out + f * 128 + 64 + lane);
  }
}

// contiguous sub-runs: wave w of a workgroup takes frames c0 + off_w .. of its chunk one after the other (a register-resident
// overlap-add like k_istft_wave's), wave 0 reads `halo` frames in front of the chunk first and takes `halo` fewer frames of its own
// (every wave processes `cpw` frames): the geometry of a chunked iSTFT with a tail hand-off between the waves of a workgroup
__global__ __launch_bounds__(256) void k_chunk_runs(const v4f* __restrict__ in, v4f* __restrict__ out, size_t frames, int cpw, int halo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per_chunk = (size_t)4 * cpw - halo;
  const size_t c0 = (size_t)blockIdx.x * per_chunk;
  size_t f0 = c0 + (wave == 0 ? 0 : (size_t)wave * cpw - halo);
  const size_t n_own = wave == 0 ? cpw - halo : cpw;
  v4f carry = {0, 0, 0, 0};
  if (wave == 0)
    for (int h = 0; h < halo; ++h) {
      const size_t f = c0 >= (size_t)halo ? c0 - halo + h : h;
#pragma unroll
      for (int j = 0; j < 8; ++j) carry += in[f * 512 + 64 * j + lane];
    }
  for (size_t it = 0; it < n_own; ++it) {
    const size_t f = f0 + it;
    if (f >= frames) return;
    v4f acc = carry;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += in[f * 512 + 64 * j + lane];
    __builtin_nontemporal_store(acc, out + f * 128 + lane);
    __builtin_nontemporal_store(acc * 2.0f, out + f * 128 + 64 + lane);
  }
}

int main() {
  const size_t frames = 16 * 11247;  // config 3
  v4f *a, *b;
  CK(hipMalloc(&a, frames * 8192)); CK(hipMalloc(&b, frames * 2048));
  CK(hipMemset(a, 1, frames * 8192)); CK(hipMemset(b, 0, frames * 2048));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch) { for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e0)); for (int i = 0; i < 30; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 30; };
  for (int wpc : {8, 12, 16, 24, 32}) {
    const size_t waves = (size_t)256 * wpc, run = (frames + waves - 1) / waves;
    const unsigned grid = (unsigned)((waves + 3) / 4);
#define RUN(B) { float ms = time([&] { hipLaunchKernelGGL((k_mix<B>), dim3(grid), dim3(256), 0, 0, a, b, frames, run); }); \
      printf("waves/CU %2d  burst %d frame(s) = %2d KiB per wave  %7.1f GB/s (10240 B/frame)\n", wpc, B, 8 * B, frames * 10240.0 / ms / 1e6); }
    RUN(1) RUN(2) RUN(4)
  }
  for (int cpw : {2, 4, 8, 16})
    for (int halo : {0, 3}) {
      const unsigned grid = (unsigned)((frames + 4 * cpw - 1) / (4 * cpw));
      float ms = time([&] { hipLaunchKernelGGL(k_chunk, dim3(grid), dim3(256), 0, 0, a, b, frames, cpw, halo); });
      printf("chunk geometry: %2d frames per wave, halo %d frames per chunk  %7.1f GB/s (10240 B/frame, halo reads not counted)\n", cpw, halo, frames * 10240.0 / ms / 1e6);
    }
  for (int cpw : {4, 5, 6, 8, 10, 12, 16, 24})
    for (int halo : {0, 3}) {
      const size_t per_chunk = (size_t)4 * cpw - halo;
      const unsigned grid = (unsigned)((frames + per_chunk - 1) / per_chunk);
      float ms = time([&] { hipLaunchKernelGGL(k_chunk_runs, dim3(grid), dim3(256), 0, 0, a, b, frames, cpw, halo); });
      printf("chunked runs: %2d frames per wave (wave 0: %d halo + %d own)  %7.1f GB/s (10240 B/frame, halo reads not counted)\n", cpw, halo, cpw - halo,
             frames * 10240.0 / ms / 1e6);
    }
  return 0;
}
