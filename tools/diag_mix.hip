// No-math traffic models of the iSTFT and FIR kernels in their SHIPPED launch geometry, as a tiny shared library that bench.py
// loads next to libnxsig.so so that `roofline_istft.mix_ceiling` / `roofline_fir.mix_ceiling` are measured in the same process, on
// the same box and stream as the kernels they bound.  Measurement infrastructure: libnxsig.so never links or
// loads this file, and nothing here computes a result anybody uses.
//
//   nxdiag_istft_mix : k_istft_wave<1024, R = 4, DEEP>'s stream — 8 resident waves per CU, each walks one run of consecutive frames
//                      (R - 1 halo frames in front are read, not written), 8 KiB non-temporal 8-byte loads per frame issued two
//                      frames ahead, 2 KiB of non-temporal 16-byte stores per frame.
//   nxdiag_fir_mix   : k_fir_wave<1024, STREAM, 4, HREG>'s stream — 257 taps, V = 768: a wave takes one block pair per iteration
//                      (two 1024-sample blocks read with 8-byte loads, default policy; 2 V outputs leave through 8-byte `sc1 nt`
//                      buffer stores), 8 pairs per wave, 4 waves per workgroup, one pair prefetched.
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long v = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}

template <int LB, bool NT = true, int WB = 1>   // LB = bytes per lane per load: 8 (the shipped kernel's loads) or 16; NT: non-temporal loads (shipped) or default policy
                                                // WB: frames per burst of stores (1 = shipped; 4 / 8: the same bytes in fewer, longer bursts; 0: no stores at all)
__global__ __launch_bounds__(256) void k_istft_mix(const v2f* __restrict__ z, v4f* __restrict__ y, long frames, long run_len, int halo) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long j0 = wave * run_len;
  long j1 = j0 + run_len; if (j1 > frames) j1 = frames;
  if (j0 >= j1) return;
  const long m0 = j0 >= halo ? j0 - halo : 0;
  v2f r0[16], r1[16], hold = v2f{0.f, 0.f};
  auto issue = [&](v2f (&r)[16], long m) {
    if (LB == 8) {
      const v2f* p = z + (size_t)(m < frames ? m : frames - 1) * 1024 + lane;
#pragma unroll
      for (int s = 0; s < 16; ++s) r[s] = NT ? __builtin_nontemporal_load(p + 64 * s) : p[64 * s];
    } else {
      const v4f* p = reinterpret_cast<const v4f*>(z + (size_t)(m < frames ? m : frames - 1) * 1024) + lane;
#pragma unroll
      for (int s = 0; s < 8; ++s) { const v4f t = NT ? __builtin_nontemporal_load(p + 64 * s) : p[64 * s]; r[2 * s] = v2f{t.x, t.y}; r[2 * s + 1] = v2f{t.z, t.w}; }
    }
  };
  auto consume = [&](v2f (&r)[16], long m) {
    v2f a = r[0];
#pragma unroll
    for (int s = 1; s < 16; ++s) a += r[s];
    if (WB == 0) { if (a.x == 1.2345f && a.y == 5.4321f) __builtin_nontemporal_store(v4f{a.x, a.y, a.y, a.x}, y + (size_t)m * 128 + lane); return; }
    if (WB == 1) {
      if (m >= j0 && m < j1) {
        __builtin_nontemporal_store(v4f{a.x, a.y, a.y, a.x}, y + (size_t)m * 128 + lane);
        __builtin_nontemporal_store(v4f{a.y, a.x, a.x, a.y}, y + (size_t)m * 128 + 64 + lane);
      }
    } else {
      hold += a;
      if (m >= j0 && m < j1 && ((m - j0) % WB == WB - 1 || m == j1 - 1)) {
        const long mb = m - (m - j0) % WB;
#pragma unroll
        for (int t = 0; t < 2 * WB; ++t)
          if (mb + t / 2 <= m) __builtin_nontemporal_store(v4f{hold.x, hold.y, hold.y + (float)t, hold.x}, y + (size_t)mb * 128 + 64 * t + lane);
      }
    }
  };
  issue(r0, m0); issue(r1, m0 + 1);
  for (long m = m0; m < j1; m += 2) {
    consume(r0, m); issue(r0, m + 2);
    consume(r1, m + 1); issue(r1, m + 3);
  }
}


// the same persistent stream with the cache-policy bits of loads and stores as template parameters (buffer instructions: aux bit 0 = sc0,
// bit 1 = nt, bit 4 = sc1) and, optionally, the stores going through a per-wave LDS-free register burst of WB frames
template <int LAUX, int SAUX>
__global__ __launch_bounds__(256) void k_istft_mix_pol(const v2f* __restrict__ z, v4f* __restrict__ y, long frames, long run_len) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long j0 = wave * run_len;
  long j1 = j0 + run_len; if (j1 > frames) j1 = frames;
  if (j0 >= j1) return;
  v2i r0[16], r1[16];
  auto issue = [&](v2i (&r)[16], long m) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(z + (size_t)(m < frames ? m : frames - 1) * 1024, 8192);
#pragma unroll
    for (int s = 0; s < 16; ++s) r[s] = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8 + 512 * s, 0, LAUX);
  };
  auto consume = [&](v2i (&r)[16], long m) {
    v2i a = r[0];
#pragma unroll
    for (int s = 1; s < 16; ++s) a += r[s];
    if (m < j1) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(y + (size_t)m * 128, 2048);
      __builtin_amdgcn_raw_buffer_store_b128(v4i{a.x, a.y, a.y, a.x}, rs, lane * 16, 0, SAUX);
      __builtin_amdgcn_raw_buffer_store_b128(v4i{a.y, a.x, a.x, a.y}, rs, lane * 16 + 1024, 0, SAUX);
    }
  };
  issue(r0, j0); issue(r1, j0 + 1);
  for (long m = j0; m < j1; m += 2) {
    consume(r0, m); issue(r0, m + 2);
    consume(r1, m + 1); issue(r1, m + 3);
  }
}

template <int WIDE>   // 0: the shipped kernel's 8-byte accesses; 1: the pair's 1792-sample span once with 16-byte loads, 16-byte stores
__global__ __launch_bounds__(256) void k_fir_mix(const float* __restrict__ x, float* __restrict__ y, long L, long pairs_per_row, long total, long chunk) {
  constexpr int V = 768, TM1 = 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long p0 = (long)blockIdx.x * chunk;
  long p1 = p0 + chunk; if (p1 > total) p1 = total;
  v2f r[16];
  auto issue = [&](long p) {
    const long row = p / pairs_per_row, pi = p - row * pairs_per_row;
    if (WIDE) {
      const v4f* s4 = reinterpret_cast<const v4f*>(x + row * L + pi * 2 * V) + lane;
#pragma unroll
      for (int q = 0; q < 7; ++q) { const v4f t = s4[64 * q]; r[2 * q] = v2f{t.x, t.y}; r[2 * q + 1] = v2f{t.z, t.w}; }
      r[14] = r[0]; r[15] = r[1];
    } else {
      const v2f* s2 = reinterpret_cast<const v2f*>(x + row * L + pi * 2 * V) + lane;
#pragma unroll
      for (int q = 0; q < 8; ++q) { r[q] = s2[64 * q]; r[8 + q] = s2[V / 2 + 64 * q]; }
    }
  };
  if (p0 + wave < p1) issue(p0 + wave);
  for (long p = p0 + wave; p < p1; p += 4) {
    v2f acc = r[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) acc += r[q];
    issue(p + 4 < p1 ? p + 4 : p);
    const long row = p / pairs_per_row, pi = p - row * pairs_per_row;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(y + row * L + pi * 2 * V + TM1, 2 * V * 4);
    if (WIDE) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const v2f o = acc * (float)(j + 1);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v4f{o.x, o.y, o.y, o.x}), rs, lane * 16 + 1024 * j, 0, 18);
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

// The headline STFT kernel's stream (k_stft_wave<1024, pair>): short-lived 4-wave workgroups, `ppw` frame pairs per wave handed out in
// dispatch order; per pair 2 x 16 four-byte loads per lane (frames A and B, hop apart: 75 % of the bytes re-read from cache), next pair
// prefetched; 2 x 8 sixteen-byte `sc1 nt` buffer stores per lane (two 8 KiB spectrum rows); 12 KB of tables read per workgroup.
__global__ __launch_bounds__(256) void k_stft_mix(const float* __restrict__ x, v4f* __restrict__ zout, const float* __restrict__ tab, long L, long M,
                                                  long pairs_per_row, long total_pairs, int ppw, int hop) {
  __shared__ float s_tab[3072];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 3072; i += 256) s_tab[i] = tab[i];
  __syncthreads();
  const long p0 = (long)blockIdx.x * 4 * ppw;
  long p1 = p0 + 4L * ppw; if (p1 > total_pairs) p1 = total_pairs;
  float ra[16], rb[16];
  auto issue = [&](long p) {
    const long row = p / pairs_per_row, pi = p - row * pairs_per_row;
    const float* pa = x + row * L + (2 * pi) * hop + lane;
    const float* pb = pa + hop;
#pragma unroll
    for (int s = 0; s < 16; ++s) { ra[s] = pa[64 * s]; rb[s] = pb[64 * s]; }
  };
  if (p0 + wave < p1) issue(p0 + wave);
  for (long p = p0 + wave; p < p1; p += 4) {
    float a = s_tab[lane], b = s_tab[1024 + lane];
#pragma unroll
    for (int s = 0; s < 16; ++s) { a += ra[s]; b += rb[s]; }
    issue(p + 4 < p1 ? p + 4 : p);
    const long row = p / pairs_per_row, pi = p - row * pairs_per_row;
    v4f* zr = zout + ((row * M + 2 * pi) * 1024L) / 2;      // row of 1024 c64 = 512 v4f
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(zr, 8192), rB = make_rsrc(zr + 512, 8192);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v4f{a, b, a + (float)q, b}), rA, lane * 16 + 1024 * q, 0, 18);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v4f{b, a, b + (float)q, a}), rB, lane * 16 + 1024 * q, 0, 18);
    }
  }
}

// Config 4's kernel (k_stft_wave<1024, real-2x>: ONE 2048-sample frame per unit as even / odd samples): `upw` frames per wave handed out
// like the pairs above; per frame 16 eight-byte loads per lane (hop = 512: three quarters of them re-read from cache), next frame
// prefetched; 16 sixteen-byte `sc1 nt` buffer stores per lane (one 16 KiB spectrum row); the same 12 KB of tables per workgroup.
__global__ __launch_bounds__(256) void k_stft2048_mix(const float* __restrict__ x, v4f* __restrict__ zout, const float* __restrict__ tab, long L, long M,
                                                      long total, int upw, int hop) {
  __shared__ float s_tab[3072];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 3072; i += 256) s_tab[i] = tab[i];
  __syncthreads();
  const long p0 = (long)blockIdx.x * 4 * upw;
  long p1 = p0 + 4L * upw; if (p1 > total) p1 = total;
  v2f r[16];
  auto issue = [&](long p) {
    const long row = p / M, m = p - row * M;
    const v2f* pa = reinterpret_cast<const v2f*>(x + row * L + m * hop) + lane;
#pragma unroll
    for (int s = 0; s < 16; ++s) r[s] = pa[64 * s];
  };
  if (p0 + wave < p1) issue(p0 + wave);
  for (long p = p0 + wave; p < p1; p += 4) {
    v2f a = v2f{s_tab[lane], s_tab[1024 + lane]};
#pragma unroll
    for (int s = 0; s < 16; ++s) a += r[s];
    issue(p + 4 < p1 ? p + 4 : p);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(zout + p * 1024L, 16384);    // row of 2048 c64 = 1024 v4f
#pragma unroll
    for (int q = 0; q < 16; ++q)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v4f{a.x, a.y, a.x + (float)q, a.y}), rs, lane * 16 + 1024 * q, 0, 18);
  }
}

// ---- plain yardsticks: bandwidth figures that owe NOTHING to the geometry of a product kernel — grid-stride loops over
// 1 KiB wave rows (64 lanes x 16 bytes), default cache policy, no tables, no descriptors.  bench.py times them in its own process on its
// own buffers (`yardsticks` in the JSON line) so that "a 4 : 1 read / write stream tops out at 0.6x" is a measurement of the box and not
// an argument made with the iSTFT's own traffic model.
//   k_y_copy : float4 copy (the guide's 6.29 TB/s figure)      k_y_read : loads only        k_y_fill : stores only
//   k_y_mix<RD, WR> : per step a wave reads RD KiB and writes WR KiB (4 : 1 = the iSTFT's ratio, 1 : 8 = the STFT's, 1 : 1 = the FIR's)
__global__ __launch_bounds__(256) void k_y_copy(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_y_fill(v4f* __restrict__ out, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v4f{v, v + 1.f, v + 2.f, v + 3.f};
}
__global__ __launch_bounds__(256) void k_y_read(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n) {
  v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
  if (acc.x == 12345.678f) out[threadIdx.x] = acc;   // never taken: keeps the loads alive
}
// SPW = 0: grid-stride (long-lived workgroups); SPW > 0: short-lived workgroups in dispatch order, SPW consecutive steps per wave
template <int RD, int WR>
__global__ __launch_bounds__(256) void k_y_mix(const v4f* __restrict__ in, v4f* __restrict__ out, size_t steps, int spw) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256) >> 6;
  size_t c0 = wave, c1 = steps, inc = nw;
  if (spw > 0) { c0 = wave * (size_t)spw; c1 = c0 + spw < steps ? c0 + spw : steps; inc = 1; }
  for (size_t c = c0; c < c1; c += inc) {
    const v4f* p = in + c * (64 * RD) + lane;
    v4f acc = RD > 0 ? p[0] : v4f{1.f, 2.f, 3.f, (float)lane};
#pragma unroll
    for (int j = 1; j < RD; ++j) acc += p[64 * j];
    v4f* o = out + c * (64 * (WR > 0 ? WR : 1)) + lane;
    if (WR == 0) { if (acc.x == 12345.678f) o[0] = acc; }   // read-only: never taken, keeps the loads alive
#pragma unroll
    for (int j = 0; j < WR; ++j) o[64 * j] = acc + (float)j;
  }
}

extern "C" {
// the link itself: `bytes` of device memory into PINNED host memory with one hipMemcpy (and back), GB/s of the better of three runs each —
// the yardstick bench.py's `host_path` block puts beside the host-tensor call (nothing faster can cross PCIe)
int nxdiag_pcie_pinned(const void* dev, size_t bytes, double* d2h_GBps, double* h2d_GBps) {
  void* pin = nullptr;
  if (hipHostMalloc(&pin, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return 1; }
  hipEvent_t e0, e1;
  int bad = hipEventCreate(&e0) != hipSuccess;
  bad |= hipEventCreate(&e1) != hipSuccess;
  double best[2] = {0.0, 0.0};
  for (int dir = 0; dir < 2 && !bad; ++dir)
    for (int rep = 0; rep < 3; ++rep) {
      bad |= hipEventRecord(e0, 0) != hipSuccess;
      if (dir == 0) bad |= hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, 0) != hipSuccess;
      else bad |= hipMemcpyAsync(const_cast<void*>(dev), pin, bytes, hipMemcpyHostToDevice, 0) != hipSuccess;
      bad |= hipEventRecord(e1, 0) != hipSuccess;
      bad |= hipEventSynchronize(e1) != hipSuccess;
      float ms = 0.f;
      bad |= hipEventElapsedTime(&ms, e0, e1) != hipSuccess;
      const double g = ms > 0.f ? (double)bytes / (ms * 1e-3) / 1e9 : 0.0;
      if (g > best[dir]) best[dir] = g;
    }
  bad |= hipEventDestroy(e0) != hipSuccess;
  bad |= hipEventDestroy(e1) != hipSuccess;
  bad |= hipHostFree(pin) != hipSuccess;
  if (d2h_GBps) *d2h_GBps = best[0];
  if (h2d_GBps) *h2d_GBps = best[1];
  const int last = (int)hipGetLastError();   // ALWAYS read: a sticky error left behind here fails the next launch check of libnxsig.so
  return bad ? 1 : last;                     // (round 6: that made rank 0 skip an assembly its peers were already waiting in)
}
// x: f32[rows][L], z: c64[rows][M][2048] with M = (L - 2048) / hop + 1
int nxdiag_stft2048_mix(void* stream, const void* x, void* z, const void* tab, long rows, long L, int hop, int units_per_wave) {
  const long M = (L - 2048) / hop + 1, total = rows * M;
  const long per_wg = 4L * units_per_wave;
  hipLaunchKernelGGL(k_stft2048_mix, dim3((unsigned)((total + per_wg - 1) / per_wg)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (v4f*)z,
                     (const float*)tab, L, M, total, units_per_wave, hop);
  return (int)hipGetLastError();
}
// x: f32[rows][L], z: c64[rows][M][1024] with M = (L - 1024) / hop + 1 (even frame counts per row are walked; an odd last frame is skipped)
int nxdiag_stft_mix(void* stream, const void* x, void* z, const void* tab, long rows, long L, int hop, int pairs_per_wave) {
  const long M = (L - 1024) / hop + 1, ppr = M / 2, total = rows * ppr;
  const long per_wg = 4L * pairs_per_wave;
  hipLaunchKernelGGL(k_stft_mix, dim3((unsigned)((total + per_wg - 1) / per_wg)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (v4f*)z,
                     (const float*)tab, L, M, ppr, total, pairs_per_wave, hop);
  return (int)hipGetLastError();
}
// z: c64[frames][1024] (8 KiB per frame), y: c64[frames][256] (2 KiB per frame); returns 0 or a hipError_t
int nxdiag_istft_mix2(void* stream, const void* z, void* y, long frames, int waves_per_cu, int halo, int load_bytes) {
  int dev = 0; hipDeviceProp_t pr;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 1;
  const long waves = (long)pr.multiProcessorCount * waves_per_cu;
  const long run_len = (frames + waves - 1) / waves;
  const unsigned grid = (unsigned)(((frames + run_len - 1) / run_len + 3) / 4);
  // load_bytes: 8 / 16 = non-temporal loads, 108 / 116 = default cache policy
  if (load_bytes == 16) hipLaunchKernelGGL((k_istft_mix<16, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, run_len, halo);
  else if (load_bytes == 116) hipLaunchKernelGGL((k_istft_mix<16, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, run_len, halo);
  else if (load_bytes == 108) hipLaunchKernelGGL((k_istft_mix<8, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, run_len, halo);
  else hipLaunchKernelGGL((k_istft_mix<8, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, run_len, halo);
  return (int)hipGetLastError();
}
// the same stream handed out in SHORT runs to short-lived workgroups in dispatch order (the waves in flight then read one moving
// window of the spectrum instead of 2048 far-apart ones): run_len frames per wave, grid = frames / (4 run_len)
int nxdiag_istft_mix3(void* stream, const void* z, void* y, long frames, int run_len, int halo, int load_bytes) {
  const unsigned grid = (unsigned)(((frames + run_len - 1) / run_len + 3) / 4);
  if (load_bytes == 16) hipLaunchKernelGGL((k_istft_mix<16, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, (long)run_len, halo);
  else hipLaunchKernelGGL((k_istft_mix<8, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, (long)run_len, halo);
  return (int)hipGetLastError();
}
int nxdiag_istft_mix_pol(void* stream, const void* z, void* y, long frames, int waves_per_cu, int laux, int saux) {
  int dev = 0; hipDeviceProp_t pr;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 1;
  const long waves = (long)pr.multiProcessorCount * waves_per_cu;
  const long run_len = (frames + waves - 1) / waves;
  const unsigned grid = (unsigned)(((frames + run_len - 1) / run_len + 3) / 4);
#define POL(L_, S_) if (laux == L_ && saux == S_) { hipLaunchKernelGGL((k_istft_mix_pol<L_, S_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, run_len); return (int)hipGetLastError(); }
  POL(2, 0) POL(2, 2) POL(2, 16) POL(2, 18) POL(2, 1) POL(2, 17) POL(2, 19) POL(2, 3)
  POL(0, 2) POL(0, 18) POL(16, 18) POL(18, 18) POL(0, 0) POL(1, 18) POL(17, 18) POL(3, 18)
#undef POL
  return 2;
}
// the persistent geometry with the stores in bursts of `wb` frames (0: no stores: the read side alone)
int nxdiag_istft_mix4(void* stream, const void* z, void* y, long frames, int waves_per_cu, int halo, int wb) {
  int dev = 0; hipDeviceProp_t pr;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 1;
  const long waves = (long)pr.multiProcessorCount * waves_per_cu;
  const long run_len = (frames + waves - 1) / waves;
  const unsigned grid = (unsigned)(((frames + run_len - 1) / run_len + 3) / 4);
  if (wb == 0) hipLaunchKernelGGL((k_istft_mix<8, true, 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, run_len, halo);
  else if (wb == 4) hipLaunchKernelGGL((k_istft_mix<8, true, 4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, run_len, halo);
  else if (wb == 8) hipLaunchKernelGGL((k_istft_mix<8, true, 8>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, run_len, halo);
  else hipLaunchKernelGGL((k_istft_mix<8, true, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v2f*)z, (v4f*)y, frames, run_len, halo);
  return (int)hipGetLastError();
}
int nxdiag_istft_mix(void* stream, const void* z, void* y, long frames, int waves_per_cu, int halo) { return nxdiag_istft_mix2(stream, z, y, frames, waves_per_cu, halo, 8); }
// x, y: f32[rows][L]
int nxdiag_fir_mix2(void* stream, const void* x, void* y, long rows, long L, int pairs_per_wave, int wide) {
  const long V = 768;
  const long pairs_per_row = (L - 1024 - V) / (2 * V);
  const long total = rows * pairs_per_row, chunk = 4L * pairs_per_wave;
  const dim3 grid((unsigned)((total + chunk - 1) / chunk));
  if (wide) hipLaunchKernelGGL(k_fir_mix<1>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, L, pairs_per_row, total, chunk);
  else hipLaunchKernelGGL(k_fir_mix<0>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, L, pairs_per_row, total, chunk);
  return (int)hipGetLastError();
}
int nxdiag_fir_mix(void* stream, const void* x, void* y, long rows, long L, int pairs_per_wave) { return nxdiag_fir_mix2(stream, x, y, rows, L, pairs_per_wave, 0); }
// ---- plain yardsticks: sizes in BYTES of the written side (copy / fill / mix) or of the read side (read); `grid` workgroups of 256
int nxdiag_y_copy(void* stream, const void* in, void* out, size_t bytes, int grid) {
  hipLaunchKernelGGL(k_y_copy, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v4f*)in, (v4f*)out, bytes / 16);
  return (int)hipGetLastError();
}
int nxdiag_y_fill(void* stream, void* out, size_t bytes, int grid) {
  hipLaunchKernelGGL(k_y_fill, dim3(grid), dim3(256), 0, (hipStream_t)stream, (v4f*)out, bytes / 16, 1.0f);
  return (int)hipGetLastError();
}
int nxdiag_y_read(void* stream, const void* in, void* out, size_t bytes, int grid) {
  hipLaunchKernelGGL(k_y_read, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v4f*)in, (v4f*)out, bytes / 16);
  return (int)hipGetLastError();
}
// rd : wr in {4:1, 1:8, 1:1, 2:1}; `steps` wave steps, each reads rd KiB from `in` and writes wr KiB to `out`
// grid > 0: grid-stride over `grid` workgroups; grid < 0: short-lived workgroups, -grid consecutive steps per wave (4 waves per workgroup)
int nxdiag_y_mix(void* stream, const void* in, void* out, size_t steps, int rd, int wr, int grid) {
  int spw = 0;
  if (grid < 0) { spw = -grid; grid = (int)((steps + 4 * (size_t)spw - 1) / (4 * (size_t)spw)); }
#define YM(R_, W_) if (rd == R_ && wr == W_) { hipLaunchKernelGGL((k_y_mix<R_, W_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const v4f*)in, (v4f*)out, steps, spw); return (int)hipGetLastError(); }
  YM(4, 1) YM(1, 8) YM(1, 1) YM(2, 1) YM(8, 2) YM(2, 16) YM(4, 0) YM(0, 4) YM(4, 4)
#undef YM
  return 2;
}
// hipMemcpyDtoDAsync on the same stream (the runtime's own copy path: blit kernel or SDMA, whatever it picks)
int nxdiag_y_memcpy(void* stream, const void* in, void* out, size_t bytes) {
  return (int)hipMemcpyDtoDAsync((hipDeviceptr_t)out, (hipDeviceptr_t)in, bytes, (hipStream_t)stream);
}
}
