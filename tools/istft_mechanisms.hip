// Round 6, VERDICT r05 item 1: the two mechanisms never built for the N = 1024 / hop = 256 iSTFT, each bounded by the no-math traffic
// model of its BEST CASE (the kernel's loads and stores in its geometry, a trivial reduction instead of the transform), timed
// interleaved with the model of the shipped kernel on the same buffers (config 3: 16 x 11 247 frames, 8 KiB read + 2 KiB written each).
//   (a) spectrum prefetch by LDS-DMA (global_load_lds_dwordx4 into a wave-private LDS slot) instead of 32 prefetch VGPRs per frame in
//       flight: SLOTS frames ahead per wave, W waves per CU.  The real kernel could hold 8 waves x 1 slot (148 KB with its exchange
//       buffers and tables); 12 x 1 and 8 x 2 do not fit 160 KB and are modelled anyway as upper bounds.
//   (b) short in-order chunks with a CARRY EXCHANGE instead of R - 1 = 3 recomputed halo frames: a chunk writes its three pending tail
//       segments (6 KiB) to scratch and its first three segments un-normalised; a fix-up pass reads both (12 KiB) and writes the three
//       finished segments (6 KiB) per chunk.  Chunk kernel + fix-up kernel are timed together.
//   hipcc --offload-arch=gfx950 -O3 tools/istft_mechanisms.hip -o tools/istft_mechanisms && tools/istft_mechanisms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// ---- the shipped kernel's stream: persistent runs, 8-byte non-temporal loads two frames ahead in registers, halo frames read
__global__ __launch_bounds__(256) void k_shipped(const v2f* __restrict__ z, v4f* __restrict__ y, long frames, long run_len, int halo) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long j0 = wave * run_len;
  long j1 = j0 + run_len; if (j1 > frames) j1 = frames;
  if (j0 >= j1) return;
  const long m0 = j0 >= halo ? j0 - halo : 0;
  v2f r0[16], r1[16];
  auto issue = [&](v2f (&r)[16], long m) {
    const v2f* p = z + (size_t)(m < frames ? m : frames - 1) * 1024 + lane;
#pragma unroll
    for (int s = 0; s < 16; ++s) r[s] = __builtin_nontemporal_load(p + 64 * s);
  };
  auto consume = [&](v2f (&r)[16], long m) {
    v2f a = r[0];
#pragma unroll
    for (int s = 1; s < 16; ++s) a += r[s];
    if (m >= j0 && m < j1) {
      __builtin_nontemporal_store(v4f{a.x, a.y, a.y, a.x}, y + (size_t)m * 128 + lane);
      __builtin_nontemporal_store(v4f{a.y, a.x, a.x, a.y}, y + (size_t)m * 128 + 64 + lane);
    }
  };
  issue(r0, m0); issue(r1, m0 + 1);
  for (long m = m0; m < j1; m += 2) {
    consume(r0, m); issue(r0, m + 2);
    consume(r1, m + 1); issue(r1, m + 3);
  }
}

// ---- (a) LDS-DMA prefetch: W waves per workgroup (one workgroup per CU), SLOTS frames ahead in wave-private LDS slots of 8 KiB
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {   // 64 lanes x 16 B -> 1 KiB at lds_dst (wave-uniform)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int W, int SLOTS>
__global__ __launch_bounds__(64 * W) void k_ldsdma(const v4f* __restrict__ z, v4f* __restrict__ y, long frames, long run_len, int halo) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wave = (long)blockIdx.x * W + wv;
  const long j0 = wave * run_len;
  long j1 = j0 + run_len; if (j1 > frames) j1 = frames;
  if (j0 >= j1) return;
  const long m0 = j0 >= halo ? j0 - halo : 0;
  const unsigned base = (unsigned)(size_t)(smem) + (unsigned)wv * SLOTS * 8192u;   // LDS byte address of this wave's ring
  const unsigned ubase = __builtin_amdgcn_readfirstlane(base);
  auto issue = [&](long m, int slot) {
    const v4f* p = z + (size_t)(m < frames ? m : frames - 1) * 512 + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s) glds16(p + 64 * s, ubase + (unsigned)slot * 8192u + 1024u * s);
  };
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) issue(m0 + s, s);
  int slot = 0;
  for (long m = m0; m < j1; ++m) {
    // the oldest frame has landed when at most 8 (SLOTS - 1) DMA pieces are outstanding
    if (SLOTS == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (SLOTS == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    const v4f* ls = reinterpret_cast<const v4f*>(smem + (size_t)wv * SLOTS * 8192 + (size_t)slot * 8192) + lane;
    v4f a = ls[0];
#pragma unroll
    for (int s = 1; s < 8; ++s) a += ls[64 * s];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot is read before the DMA refills it
    issue(m + SLOTS, slot);
    if (m >= j0) {
      __builtin_nontemporal_store(a, y + (size_t)m * 128 + lane);
      __builtin_nontemporal_store(v4f{a.y, a.x, a.w, a.z}, y + (size_t)m * 128 + 64 + lane);
    }
    slot = slot + 1 == SLOTS ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- (b) carry exchange: chunks of `run_len` frames without halo; 6 KiB of pending tail segments per chunk to scratch; fix-up pass
__global__ __launch_bounds__(256) void k_carry_chunks(const v2f* __restrict__ z, v4f* __restrict__ y, v4f* __restrict__ carry, long frames, long run_len) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long j0 = wave * run_len;
  long j1 = j0 + run_len; if (j1 > frames) j1 = frames;
  if (j0 >= j1) return;
  v2f r0[16], r1[16], hold = v2f{0.f, 0.f};
  auto issue = [&](v2f (&r)[16], long m) {
    const v2f* p = z + (size_t)(m < frames ? m : frames - 1) * 1024 + lane;
#pragma unroll
    for (int s = 0; s < 16; ++s) r[s] = __builtin_nontemporal_load(p + 64 * s);
  };
  auto consume = [&](v2f (&r)[16], long m) {
    v2f a = r[0];
#pragma unroll
    for (int s = 1; s < 16; ++s) a += r[s];
    hold += a;
    if (m < j1) {
      __builtin_nontemporal_store(v4f{a.x, a.y, a.y, a.x}, y + (size_t)m * 128 + lane);
      __builtin_nontemporal_store(v4f{a.y, a.x, a.x, a.y}, y + (size_t)m * 128 + 64 + lane);
    }
  };
  issue(r0, j0); issue(r1, j0 + 1);
  for (long m = j0; m < j1; m += 2) {
    consume(r0, m); issue(r0, m + 2);
    consume(r1, m + 1); issue(r1, m + 3);
  }
#pragma unroll
  for (int t = 0; t < 6; ++t) __builtin_nontemporal_store(v4f{hold.x, hold.y, hold.y + (float)t, hold.x}, carry + (size_t)wave * 384 + 64 * t + lane);
}
__global__ __launch_bounds__(256) void k_carry_fix(v4f* __restrict__ y, const v4f* __restrict__ carry, long frames, long run_len, long chunks) {
  const int lane = threadIdx.x & 63;
  const long c = (((long)blockIdx.x * 256 + threadIdx.x) >> 6) + 1;   // chunk c >= 1 takes the carry of chunk c - 1
  if (c >= chunks) return;
  const long j0 = c * run_len;
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    if (j0 + t / 2 >= frames) break;
    v4f* p = y + (size_t)j0 * 128 + 64 * t + lane;
    *p = *p + carry[(size_t)(c - 1) * 384 + 64 * t + lane];
  }
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const long frames = 16 * 11247;
  v4f *z, *y, *carry;
  CK(hipMalloc(&z, (size_t)frames * 8192)); CK(hipMalloc(&y, (size_t)frames * 2048)); CK(hipMalloc(&carry, (size_t)256 << 20));
  {  // random data (constant data reads 3 % faster: bus toggling)
    std::vector<unsigned> h((size_t)1 << 22);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 9) | 0x3f000000u; }
    for (size_t off = 0; off < (size_t)frames * 8192; off += h.size() * 4)
      CK(hipMemcpy(reinterpret_cast<char*>(z) + off, h.data(), std::min(h.size() * 4, (size_t)frames * 8192 - off), hipMemcpyHostToDevice));
  }
  CK(hipMemset(y, 0, (size_t)frames * 2048));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch) { for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 20; };
  const double alg = (double)frames * 10240.0;
  auto shipped = [&](int wpc) {
    const long waves = 256L * wpc, run = (frames + waves - 1) / waves;
    return time([&] { hipLaunchKernelGGL(k_shipped, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, reinterpret_cast<const v2f*>(z), y, frames, run, 3); });
  };
  for (int round = 0; round < 3; ++round) {
    printf("== round %d (GB/s on the algorithmic 10 240 B per frame; halo / carry / fix-up traffic not counted)\n", round);
    printf("shipped stream, 8 waves per CU, registers two frames ahead, 3 halo frames per run : %7.1f\n", alg / shipped(8) / 1e6);
#define LDSDMA(W, SLOTS) { const long waves = 256L * W, run = (frames + waves - 1) / waves; const size_t lds = (size_t)W * SLOTS * 8192; \
      CK(hipFuncSetAttribute((const void*)k_ldsdma<W, SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      float ms = time([&] { hipLaunchKernelGGL((k_ldsdma<W, SLOTS>), dim3(256), dim3(64 * W), lds, 0, z, y, frames, run, 3); }); \
      printf("(a) LDS-DMA prefetch, %2d waves per CU x %d slot(s) of 8 KiB = %3zu KiB of LDS%s : %7.1f\n", W, SLOTS, lds >> 10, \
             (lds + 14336 + (size_t)W * 9024 <= 163840) ? " (fits beside the kernel's tables and exchange buffers)" : " (upper bound: does NOT fit beside them)", alg / ms / 1e6); }
    LDSDMA(8, 1) LDSDMA(8, 2) LDSDMA(12, 1) LDSDMA(16, 1) LDSDMA(6, 3)
    printf("shipped stream again                                                              : %7.1f\n", alg / shipped(8) / 1e6);
    for (int wpc : {8, 16, 32, 64}) {
      const long waves = 256L * wpc, run = (frames + waves - 1) / waves, chunks = (frames + run - 1) / run;
      float ms = time([&] {
        hipLaunchKernelGGL(k_carry_chunks, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, 0, reinterpret_cast<const v2f*>(z), y, carry, frames, run);
        hipLaunchKernelGGL(k_carry_fix, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, 0, y, carry, frames, run, chunks);
      });
      float ms_h = shipped(wpc);
      printf("(b) carry exchange, %2d chunks per CU (%3ld frames each): chunks + fix-up %7.1f   | halo recompute in the same geometry %7.1f\n", wpc, run,
             alg / ms / 1e6, alg / ms_h / 1e6);
    }
  }
  return 0;
}
