// Overlap-save FIR (Filters.fir of BASELINE config 5; Convolution.convolve(x, h, method: :fft) of
// lib/nx_signal/convolution.ex:252-329) on a 32 x 32 factorisation of the 1024-point transform: ONE LDS exchange per FFT.
//
// k_fir_wave (kernels_wave.hip) runs each 1024-point transform as 16 * 16 * 4 over 64 lanes: two exchanges per transform,
// four per block pair, and its SQ counters say the LDS pipe (stores above all) is what the two transforms of a pair queue on
// (profiles/r02/fir_sq_counters.txt; dropping one transform makes the kernel memory-bound).  Here a wave works on TWO block
// pairs at once, one per 32-lane half: a half holds its 1024 complex points as 32 per lane (point n on lane n % 32, register
// n / 32), and
//     X[k1 + 32 k2] = sum_n2 w_32^(n2 k2) . w_1024^(n2 k1) . sum_n1 x[32 n1 + n2] w_32^(n1 k1)
// is a 32-point DFT in registers (over n1), one twiddle, ONE 32 x 32 transposition through LDS (XOR-swizzled: conflict-free
// without padding, so two 4-wave workgroups fit a CU's 160 KiB), and a second 32-point DFT in registers (over n2).  Input
// and output use the same lane layout and natural bin order, so forward -> x H/K -> inverse needs nothing in between, and
// every global access is a 128-byte run per half wave.  Two consecutive real blocks ride as re / im like in k_fir_wave.
// The launcher hands this kernel the interior block pairs of every row two at a time; edge pairs (and an odd leftover) stay
// with k_fir_wave's bounds-checked path.  Needs only (taps - 1) % 32 == 0 (4-byte accesses: no alignment conditions).
#include "wave_stft.hpp"

namespace nxsig {

// cos / -sin of 2 pi m / 32
__device__ constexpr float kC32[32] = {
    1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f, 0.55557023301960229f,
    0.38268343236508984f, 0.19509032201612833f, 0.f, -0.19509032201612819f, -0.38268343236508973f, -0.55557023301960196f,
    -0.70710678118654746f, -0.83146961230254535f, -0.92387953251128674f, -0.98078528040323043f, -1.f, -0.98078528040323043f,
    -0.92387953251128685f, -0.83146961230254546f, -0.70710678118654768f, -0.55557023301960218f, -0.38268343236509034f,
    -0.19509032201612866f, 0.f, 0.1950903220161283f, 0.38268343236509f, 0.55557023301960184f, 0.70710678118654735f,
    0.83146961230254524f, 0.92387953251128652f, 0.98078528040323032f};
__device__ constexpr float kS32[32] = {
    0.f, -0.19509032201612825f, -0.38268343236508978f, -0.55557023301960218f, -0.70710678118654746f, -0.83146961230254524f,
    -0.92387953251128674f, -0.98078528040323043f, -1.f, -0.98078528040323043f, -0.92387953251128674f, -0.83146961230254546f,
    -0.70710678118654757f, -0.55557023301960218f, -0.38268343236508989f, -0.19509032201612861f, 0.f, 0.19509032201612836f,
    0.38268343236508967f, 0.55557023301960196f, 0.70710678118654746f, 0.83146961230254524f, 0.92387953251128652f,
    0.98078528040323032f, 1.f, 0.98078528040323043f, 0.92387953251128663f, 0.83146961230254546f, 0.70710678118654768f,
    0.55557023301960218f, 0.38268343236509039f, 0.19509032201612872f};

// a * b for a wave-uniform constant b held in an SGPR pair: the twelve distinct W32 constants of dft32 would otherwise be
// materialised in VGPRs and hoisted out of the main loop (24+ registers held for the whole kernel)
__device__ __forceinline__ v2f wcmul_sconst(v2f a, v2f b) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
  return r;
}

// a * W32^m (conjugated for INV), m a compile-time constant
template <bool INV, int M>
__device__ __forceinline__ v2f tw32c(v2f a) {
  if (M == 0) return a;
  if (M == 4) return rot45<INV>(a);
  if (M == 8) return rot90<INV>(a);
  if (M == 12) return rot135<INV>(a);
  return wcmul_sconst(a, v2f{kC32[M], INV ? -kS32[M] : kS32[M]});
}

// natural-order 32-point DFT on registers: t = 8 t1 + t0, r = r0 + 4 r1;
// A[t0][r0] = DFT4 over t1 of u[8 t1 + t0];  A *= W32^(t0 r0);  X[r0 + 4 r1] = DFT8 over t0 of A[.][r0]
template <bool INV, int T0>
__device__ __forceinline__ void dft32_twiddle_row(v2f* a) {   // a = A[T0][0..3]
  a[1] = tw32c<INV, T0 * 1>(a[1]);
  a[2] = tw32c<INV, T0 * 2>(a[2]);
  a[3] = tw32c<INV, T0 * 3>(a[3]);
}
template <bool INV>
__device__ __forceinline__ void dft32(v2f* u) {
  v2f A[8][4];
#pragma unroll
  for (int t0 = 0; t0 < 8; ++t0) {
    A[t0][0] = u[t0]; A[t0][1] = u[8 + t0]; A[t0][2] = u[16 + t0]; A[t0][3] = u[24 + t0];
    dft4<INV>(A[t0][0], A[t0][1], A[t0][2], A[t0][3]);
  }
  dft32_twiddle_row<INV, 1>(A[1]); dft32_twiddle_row<INV, 2>(A[2]); dft32_twiddle_row<INV, 3>(A[3]);
  dft32_twiddle_row<INV, 4>(A[4]); dft32_twiddle_row<INV, 5>(A[5]); dft32_twiddle_row<INV, 6>(A[6]);
  dft32_twiddle_row<INV, 7>(A[7]);
#pragma unroll
  for (int r0 = 0; r0 < 4; ++r0) {
    v2f b[8];
#pragma unroll
    for (int t0 = 0; t0 < 8; ++t0) b[t0] = A[t0][r0];
    dft8<INV>(b);
#pragma unroll
    for (int r1 = 0; r1 < 8; ++r1) u[r0 + 4 * r1] = b[r1];
  }
}

// 1024-point DFT of one 32-lane half: in / out d[r] = point (l + 32 r).  xh = the half's 1024-entry exchange region, s_tw =
// w_1024^(k1 n2) as [k1][n2].  INV: unscaled inverse (twiddles conjugated inside the multiplies).
typedef __attribute__((address_space(3))) v2f lds_v2f;
__device__ __forceinline__ lds_v2f* lds_at(uint32_t byte_addr) { return reinterpret_cast<lds_v2f*>(byte_addr); }

template <bool INV>
__device__ __forceinline__ void half_fft_32x32(v2f* d, const uint32_t xh, const v2f* s_tw, const int l) {
  // xh = LDS byte address of the half's region, 8 KiB aligned: (xh + 8 l) ^ (8 k) never carries, so every swizzled address is
  // ONE v_xor and the row offset rides in the instruction's immediate
  uint32_t wbase = xh + 8 * l;            // write: row k1, column l ^ k1
  uint32_t rbase = xh + 256 * l + 8 * l;  // read : row l,  column n2 ^ l
  // the 64 swizzled addresses are loop-invariant: left alone the compiler computes them once and keeps them in 64 registers
  // for the whole kernel (spills).  Opaque bases make each one a single v_xor next to its use.
  asm volatile("" : "+v"(wbase), "+v"(rbase));
  dft32<INV>(d);                                   // over n1: register index becomes k1
#pragma unroll
  for (int k1 = 0; k1 < 32; ++k1) {
    if (k1 > 0) d[k1] = INV ? wcmul_conj(d[k1], s_tw[k1 * 32 + l]) : wcmul(d[k1], s_tw[k1 * 32 + l]);
    lds_at(wbase ^ (8 * k1))[32 * k1] = d[k1];     // 32 distinct banks per half
    if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four table values in flight at a time, not all 31 (62 registers)
  }
  wave_lds_fence();
#pragma unroll
  for (int n2 = 0; n2 < 32; ++n2) d[n2] = *lds_at(rbase ^ (8 * n2));
  wave_lds_fence();
  dft32<INV>(d);                                   // over n2: register index becomes k2, the lane holds X[l + 32 k2]
}

struct Fir32Args {
  const float* x;
  int64_t batch_stride;
  int32_t V, tm1, r_lo;                // V = 1024 - (taps - 1); tm1 = taps - 1 = 32 r_lo
  int64_t first_block, pb_lo;          // pair pb covers blocks first_block + 2 pb, + 1
  int64_t dp_per_row, total_dp, chunk; // double pairs (two consecutive interior pairs) per row / in all / per workgroup
  int64_t out_start, out_len;
  const v2f* H;                        // c64[1024] natural order, pre-scaled by 1/1024
  const v2f* tw;                       // c64[32][32]: w_1024^(k1 n2)
  float* y;
  int* row_flags;                      // FirLaunch::row_flags
  int32_t row_mod = 0;                 // per-row grid phase, see FirWaveArgs::row_mod (kernels_wave.hip)
};
__device__ __forceinline__ int fir32_row_shift(const int32_t row_mod, const int64_t row) {
  return row_mod ? (int)((32 - (int)((row * row_mod) & 31)) & 31) : 0;
}

// RLO = (taps - 1) / 32 as a compile-time constant for the common filter lengths (the stores of registers below it vanish at
// compile time), or -1: decided at run time by pushing the unwanted stores' offsets out of the descriptor's range
template <int W, int RLO>
__global__ __launch_bounds__(64 * W, 2) void k_fir_wave32(Fir32Args a) {
  v2f* s_tw = reinterpret_cast<v2f*>(g_wave_smem);
  v2f* s_H = s_tw + 1024;
  v2f* s_x = s_H + 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l = lane & 31;
  for (int i = tid; i < 1024; i += 64 * W) { s_tw[i] = a.tw[i]; s_H[i] = a.H[i]; }
  __syncthreads();
  const uint32_t xh = (uint32_t)(uintptr_t)(lds_v2f*)(s_x + wave * 2048 + h * 1024);   // LDS byte address
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // uniform: the index arithmetic below stays on the scalar unit
  const int64_t p_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t p_end = p_begin + a.chunk;
  if (p_end > a.total_dp) p_end = a.total_dp;
  int64_t p = p_begin + wave_u;
  if (p >= p_end) return;   // whole wave; no barrier follows
  int64_t row = p / a.dp_per_row, q = p - row * a.dp_per_row;   // one division per wave; then advanced incrementally
  int64_t nrow = row, nq = q;
  auto advance = [&](int64_t& r, int64_t& qq) {
    qq += W;
    while (qq >= a.dp_per_row) { qq -= a.dp_per_row; ++r; }
  };
  advance(nrow, nq);
  // this half's pair of double pair (row, q): x[row][b1 V - tm1 + n] (block 1) and + V (block 2), b1 = first block of pair 2 q + h
  const int lane_off = h * 2 * a.V + l;
  auto src_of = [&](int64_t rw, int64_t qq) -> const float* {
    const int64_t b1 = a.first_block + 2 * (a.pb_lo + 2 * qq);
    return a.x + (size_t)rw * a.batch_stride + (b1 * (int64_t)a.V - a.tm1 + fir32_row_shift(a.row_mod, rw)) + lane_off;
  };
  v2f nd[32];   // the next double pair's samples land straight in (re, im) position
  auto issue_loads = [&](const float* s) {
#pragma unroll
    for (int r = 0; r < 32; ++r) { nd[r].x = s[32 * r]; nd[r].y = s[a.V + 32 * r]; }
  };
  issue_loads(src_of(row, q));
  for (; p < p_end; p += W) {
    v2f d[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) d[r] = nd[r];
    {  // a non-finite sample poisons its whole row, like the reference's single transform (FirLaunch::row_flags)
      v2f t = d[0];
#pragma unroll
      for (int r = 1; r < 32; ++r) t += d[r];
      if (wave_any_nonfinite(t.x, t.y) && lane == 0) atomicOr(a.row_flags + row, 1);
    }
    const bool more = p + W < p_end;
    issue_loads(src_of(more ? nrow : row, more ? nq : q));  // unconditional prefetch keeps the loop branch-free
    __builtin_amdgcn_sched_barrier(0);
    half_fft_32x32<false>(d, xh, s_tw, l);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      d[r] = wcmul(d[r], s_H[l + 32 * r]);   // Z H / K
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    half_fft_32x32<true>(d, xh, s_tw, l);                              // d[r] = (y1[n], y2[n]), n = l + 32 r
    __builtin_amdgcn_sched_barrier(0);
    // the double pair's valid outputs are one contiguous run of 4 V samples: streaming stores through one row descriptor
    const int64_t b1 = a.first_block + 2 * (a.pb_lo + 2 * q);
    const StreamRow ys(a.y + (size_t)row * a.out_len + (b1 * (int64_t)a.V - a.out_start + fir32_row_shift(a.row_mod, row)), (uint32_t)(4 * a.V) * 4);
    const int base = (h * 2 * a.V + l - a.tm1) * 4, base2 = base + a.V * 4;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (RLO >= 0) {
        if (r >= RLO) { ys.st4(fft_eps0(d[r].x), base + 128 * r); ys.st4(fft_eps0(d[r].y), base2 + 128 * r); }
      } else {
        const int poison = r >= a.r_lo ? 0 : 0x40000000;   // beyond num_records: the hardware drops the store
        ys.st4(fft_eps0(d[r].x), base + poison + 128 * r);  // fft_eps0: the Nx.ifft clean-up of fftconvolve (convolution.ex:282)
        ys.st4(fft_eps0(d[r].y), base2 + poison + 128 * r);
      }
    }
    row = nrow; q = nq;
    advance(nrow, nq);
  }
}

// stream part of launch_fir_wave_W (kernels_wave.hip): interior pairs [pb_lo, pb_lo + 2 dp_per_row) of every row
int launch_fir_wave32(Ctx* c, const float* x, int64_t batch_stride, int32_t batch, int32_t taps, int64_t first_block, int64_t pb_lo,
                      int64_t dp_per_row, int64_t out_start, int64_t out_len, const float2* H_dev, float* y, int* row_flags, int row_mod) {
  if (dp_per_row <= 0 || batch <= 0) return NXSIG_OK;
  constexpr int W = 4;
  Fir32Args a;
  a.x = x; a.batch_stride = batch_stride; a.V = 1024 - (taps - 1); a.tm1 = taps - 1; a.r_lo = (taps - 1) / 32; a.first_block = first_block; a.pb_lo = pb_lo;
  a.dp_per_row = dp_per_row; a.total_dp = dp_per_row * batch;
  a.out_start = out_start; a.out_len = out_len; a.H = reinterpret_cast<const v2f*>(H_dev); a.y = y; a.row_flags = row_flags; a.row_mod = row_mod;
  {
    const uint64_t key = 0xF1320000ull;
    auto hit = c->memo.find(key);
    if (hit != c->memo.end()) a.tw = reinterpret_cast<const v2f*>(hit->second[0]);
    else {
      std::vector<float2> tw(1024);
      for (int k1 = 0; k1 < 32; ++k1)
        for (int n2 = 0; n2 < 32; ++n2) {
          const double ang = -6.283185307179586476925286766559 * (double)(k1 * n2) / 1024.0;
          tw[(size_t)k1 * 32 + n2] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
      const void* d = nullptr;
      int rc = ctx_table(c, 0xF132ull, tw.data(), tw.size() * sizeof(float2), &d);
      if (rc) return rc;
      c->memo[key] = {reinterpret_cast<uint64_t>(d)};
      a.tw = reinterpret_cast<const v2f*>(d);
    }
  }
  const int units_per_wave = fill_units_per_wave(c, a.total_dp, W, 4);
  a.chunk = (int64_t)W * (units_per_wave < 1 ? 1 : units_per_wave);
  const int64_t blocks = (a.total_dp + a.chunk - 1) / a.chunk;
  if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "fir: signal too long for one launch");
  const size_t lds = (size_t)2 * 1024 * 8 + (size_t)W * 2048 * 8;   // 80 KiB: two workgroups per CU
  auto go = [&](auto kernel) -> int {
    NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note("fir.wave32");
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  switch (a.r_lo) {
    case 4: return go(k_fir_wave32<W, 4>);     // 129 taps
    case 8: return go(k_fir_wave32<W, 8>);     // 257 taps (BASELINE config 5)
    case 16: return go(k_fir_wave32<W, 16>);   // 513 taps
    default: return go(k_fir_wave32<W, -1>);
  }
}

}  // namespace nxsig
