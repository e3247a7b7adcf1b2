// Tuned gfx950 kernels built around ONE wave-private complex FFT core (wave_fft_core / wave_fft_core_T, 1024 or 2048
// points): 64 lanes x 16 (32) points, radix 16 x 16 x 4 (8), two LDS exchanges in an 8.8 (17.5) KB wave-private buffer,
// no workgroup barrier after the table preload.  Every kernel is a different way of feeding that core and draining it.
// This header holds what the three wave translation units share:
//
//   wave_fft_core / _core_T   forward, transposed and inverse-direction cores, packed-FP32 butterflies
//   stft_wave_body            pair   : two adjacent real frames as re / im                        fft_length 1024
//                             real-2x: one frame as even / odd samples                            fft_length 2048, 4096
//                             quad   : 2J frames, J complex sequences interleaved (J = 2, 4, 8)   fft_length 512 / 256 / 128
//                             sinks  : complex spectrum (k_stft_wave), log-mel (k_stft_mel_wave), magnitude (k_stft_mag_wave)
//   k_stft_blue_wave          non-power-of-two fft_length <= 1024: Bluestein chirp-z on the same cores, same sinks
//   launch_wave / launch_blue_wave   launch templates (interior / edge split, staged quad input, chunk geometry);
//                             the SINK template argument selects the kernel family a translation unit instantiates
//
//   kernels_wave.hip      iSTFT (k_istft_wave, _half, _quad, _dbl), overlap-save FIR (k_fir_wave), plain STFT launcher
//   kernels_wave_mel.hip  log-mel launcher          kernels_wave_mag.hip  magnitude launcher
//
// The core in pair mode, step by step:
//   global load (frame slice x window fused, lib/nx_signal.ex:94-101)          64 lanes x P = K/64 points
//   pass A  radix-16, registers                      -> LDS exchange 1 (padded e + e/16: conflict-free)
//   pass B  radix-16, twiddles w_256^(t k) from LDS  -> LDS exchange 2
//   pass C  radix-4, butterflies i = 2l+e+128u so every lane owns ADJACENT bins
//   Hermitian untangle of the two real spectra through partner lanes (ds_bpermute, no LDS storage):
//           XA[k] = (Z[k] + conj Z[K-k]) / 2 ,  XB[k] = -i (Z[k] - conj Z[K-k]) / 2
//   optional :spectrum / :psd division (lib/nx_signal.ex:113-127), 16-byte non-temporal stores of the full
//   two-sided c64 spectrum (:129), 1 KiB per wave instruction.
//
// The STFT path is HBM-bound (9 216 algorithmic B/frame at K=1024, 89 % stores), MFMA is deliberately unused.
// Index math and LDS bank behaviour are modelled lane by lane in tools/emulate_wave_fft.py.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nxsig_internal.h"

namespace nxsig {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
// wave-uniform: true when one of the two packed values is Inf / NaN on some lane
__device__ __forceinline__ bool wave_any_nonfinite(float sx, float sy) {
  const bool nf = ((__float_as_uint(sx) & 0x7f800000u) == 0x7f800000u) || ((__float_as_uint(sy) & 0x7f800000u) == 0x7f800000u);
  return __builtin_amdgcn_ballot_w64(nf) != 0;
}
__device__ __forceinline__ v2f fft_eps0(v2f v) { return v2f{fft_eps0(v.x), fft_eps0(v.y)}; }
__device__ __forceinline__ v4f fft_eps0(v4f v) { return v4f{fft_eps0(v.x), fft_eps0(v.y), fft_eps0(v.z), fft_eps0(v.w)}; }

// Nx.ifft's clean-up (|x| <= 1e-10 -> +0 on the transform's result x = zz / K, ahead of scale and window; lib/nx_signal.ex:609) in
// its COLD form (round 4): the threshold concerns digital silence only, so a lane first takes the minimum magnitude of the
// components it holds (one v_min3 per two values) and the compare-and-select per component runs only when some lane of the wave
// holds one at or below the threshold (wave-uniform branch).  |zz / K| <= eps  <=>  |zz| <= eps K exactly (K a power of two);
// NaN passes through both ways (minNum ignores it, the select keeps it).  2 VALU per component became 1/2.
// min(|a|, |b|, acc) in ONE instruction (the source-level fminf chain costs a canonicalising v_max per operand on top)
__device__ __forceinline__ float min3abs(float a, float b, float acc) {
  float r;
  asm("v_min3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
  return r;
}
template <int NQ>
__device__ __forceinline__ void ifft_eps_cold(v2f (*zz)[NQ], const float thr) {
  // The usual caller inverts the spectrum of a REAL signal: the imaginary parts of the result are round-off, and some of them are
  // exact zeros (conjugate bins cancel inside one butterfly) — a cold test over all components tripped on every frame of such input
  // (SQ_INSTS_VALU per frame unchanged, profiles/r04/README.md).  So: real parts by the cold test, imaginary parts eagerly.
  float amin = 3.0e38f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) amin = min3abs(zz[0][q].x, zz[1][q].x, amin);
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int q = 0; q < NQ; ++q) zz[e][q].y = __builtin_fabsf(zz[e][q].y) <= thr ? 0.0f : zz[e][q].y;
  if (__builtin_amdgcn_ballot_w64(amin <= thr) != 0) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int q = 0; q < NQ; ++q) zz[e][q].x = __builtin_fabsf(zz[e][q].x) <= thr ? 0.0f : zz[e][q].x;
  }
}

// Run length of the persistent inverse kernels (runs never cross rows; every run recomputes `halo` units of the previous one).  The
// plain rule — total / wave slots — leaves thousands of short rows with ONE run per row, i.e. ceil(rows / slots) rounds of whole rows
// (4 096 rows on 2 816 slots: two rounds, the second 45 % full).  Here: among 1 ... 8 equal runs per row and the plain rule, the length
// with the smallest  rounds x (run + halo).
inline int64_t istft_balanced_run_len(int64_t units_per_row, int64_t rows, int64_t slots, int64_t halo, int64_t min_run) {
  auto cdiv = [](int64_t a, int64_t b) { return (a + b - 1) / b; };
  auto cost = [&](int64_t len) {
    const int64_t rpr = cdiv(units_per_row, len);
    return cdiv(rpr * rows, slots) * ((len < units_per_row ? len : units_per_row) + halo);
  };
  int64_t best = cdiv(units_per_row * rows, slots);
  if (best < min_run) best = min_run;
  int64_t best_cost = cost(best);
  for (int r = 1; r <= 8; ++r) {
    int64_t len = cdiv(units_per_row, r);
    if (len < min_run) len = min_run;
    const int64_t c = cost(len);
    if (c < best_cost || (c == best_cost && len > best)) { best = len; best_cost = c; }
  }
  return best;
}

// front-ends of the C-point complex core
enum : int {
  kModePair = 0,    // two adjacent real frames of length C as re / im          (fft_length == C)
  kModeReal2x = 1,  // ONE real frame of length 2C as even / odd samples        (fft_length == 2C)
  kModeQuad = 2,    // 2J adjacent real frames of length C/J (J = 2, 4, 8): J complex sequences c_j = frame 2j + i frame 2j+1
                    // are interleaved, z[J n + j] = c_j[n], so Z[k0 + (C/J) m] = sum_j w_J^(jm) w_C^(j k0) C_j[k0]; the J values
                    // k0 + (C/J) m share a lane, so the C_j separate with a lane-local inverse radix-J butterfly  (fft_length == C/J)
};

// ---- packed-FP32 helpers.  gfx950 issues v_pk_{mul,add,fma}_f32 at full rate (two floats per lane per instruction),
// and VOP3P source modifiers (op_sel / op_sel_hi pick the low or high half per result half, neg_lo / neg_hi negate per
// half) make a complex multiply two instructions and a +-i rotation free inside the add that consumes it.  The compiler
// folds neither per-half negation nor the half swap (it emits 4-5 instructions per complex multiply), hence the asm.
// a * b
__device__ __forceinline__ v2f wcmul(v2f a, v2f b) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));             // (a.y b.y, a.y b.x)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));  // (a.x b.x - t.x, a.x b.y + t.y)
  return r;
}
// a * conj(b): lets an inverse-direction core read the FORWARD twiddle tables (no second copy in LDS)
__device__ __forceinline__ v2f wcmul_conj(v2f a, v2f b) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));             // (a.y b.y, a.y b.x)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));  // (a.x b.x + t.x, -a.x b.y + t.y)
  return r;
}
// x + (-i) y = (x.x + y.y, x.y - y.x)      and      x + (+i) y = (x.x - y.y, x.y + y.x)
__device__ __forceinline__ v2f add_mi(v2f x, v2f y) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ v2f add_pi(v2f x, v2f y) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// a * (-i) for the forward transform, a * (+i) for the inverse (INV)
template <bool INV>
__device__ __forceinline__ v2f rot90(v2f a) { return INV ? v2f{-a.y, a.x} : v2f{a.y, -a.x}; }
// a * conj^INV(c + i s) for a compile-time constant twiddle given as (cos, -sin) of the forward transform
template <bool INV>
__device__ __forceinline__ v2f cmulc(v2f a, float c, float ms) { return wcmul(a, v2f{c, INV ? -ms : ms}); }
// a * (1 -+ i)/sqrt2  and  a * (-1 -+ i)/sqrt2:  (1 - i) a = a + (-i) a = add_mi(a, a),  (1 + i) a = add_pi(a, a)
template <bool INV>
__device__ __forceinline__ v2f rot45(v2f a) {
  const float h = 0.70710678118654752f;
  return (INV ? add_pi(a, a) : add_mi(a, a)) * h;
}
template <bool INV>
__device__ __forceinline__ v2f rot135(v2f a) {
  const float h = -0.70710678118654752f;   // (-1 - i) = -(1 + i),  (-1 + i) = -(1 - i)
  return (INV ? add_mi(a, a) : add_pi(a, a)) * h;
}

// natural-order DFTs on registers (forward: e^{-2 pi i ..}; INV: e^{+2 pi i ..}, unscaled)
template <bool INV = false>
__device__ __forceinline__ void dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
  const v2f s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, e13 = a1 - a3;
  a0 = s02 + s13; a2 = s02 - s13;
  a1 = INV ? add_pi(d02, e13) : add_mi(d02, e13);
  a3 = INV ? add_mi(d02, e13) : add_pi(d02, e13);
}

template <bool INV = false>
__device__ __forceinline__ void dft8(v2f* u) {
  // t = 2 t1 + t0: A[t0][r0] = DFT4_{t1}(u[2 t1 + t0]); A[1][r0] *= W8^r0; v[r0] = A0 + A1, v[r0+4] = A0 - A1
  v2f a0 = u[0], a1 = u[2], a2 = u[4], a3 = u[6];
  v2f b0 = u[1], b1 = u[3], b2 = u[5], b3 = u[7];
  dft4<INV>(a0, a1, a2, a3);
  dft4<INV>(b0, b1, b2, b3);
  b1 = rot45<INV>(b1);
  b2 = rot90<INV>(b2);
  b3 = rot135<INV>(b3);
  u[0] = a0 + b0; u[4] = a0 - b0;
  u[1] = a1 + b1; u[5] = a1 - b1;
  u[2] = a2 + b2; u[6] = a2 - b2;
  u[3] = a3 + b3; u[7] = a3 - b3;
}

template <bool INV = false>
__device__ __forceinline__ void dft16(v2f* u) {
  // t = 4 t1 + t0, r = r0 + 4 r1:  A[t0][r0] = DFT4_{t1}(u[4 t1 + t0]); A *= W16^(t0 r0); v[r0 + 4 r1] = DFT4_{t0}(A[.][r0])
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;
  v2f A[4][4];
#pragma unroll
  for (int t0 = 0; t0 < 4; ++t0) {
    A[t0][0] = u[t0]; A[t0][1] = u[4 + t0]; A[t0][2] = u[8 + t0]; A[t0][3] = u[12 + t0];
    dft4<INV>(A[t0][0], A[t0][1], A[t0][2], A[t0][3]);
  }
  // W16^j = (cos, -sin)(2 pi j / 16), conjugated for INV
  A[1][1] = cmulc<INV>(A[1][1], c1, -s1);   // W^1
  A[1][2] = rot45<INV>(A[1][2]);            // W^2
  A[1][3] = cmulc<INV>(A[1][3], s1, -c1);   // W^3
  A[2][1] = rot45<INV>(A[2][1]);            // W^2
  A[2][2] = rot90<INV>(A[2][2]);            // W^4
  A[2][3] = rot135<INV>(A[2][3]);           // W^6
  A[3][1] = cmulc<INV>(A[3][1], s1, -c1);   // W^3
  A[3][2] = rot135<INV>(A[3][2]);           // W^6
  A[3][3] = cmulc<INV>(A[3][3], -c1, s1);   // W^9
#pragma unroll
  for (int r0 = 0; r0 < 4; ++r0) {
    dft4<INV>(A[0][r0], A[1][r0], A[2][r0], A[3][r0]);
    u[r0] = A[0][r0]; u[r0 + 4] = A[1][r0]; u[r0 + 8] = A[2][r0]; u[r0 + 12] = A[3][r0];
  }
}

// compiler-level ordering of this wave's LDS traffic (the hardware executes a wave's DS ops in order;
// no s_barrier, no s_waitcnt vmcnt: outstanding global stores keep flying)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The K-point complex forward FFT of one wave: in  d[s] = x[lane + 64 s]  (P = K/64 points per lane),
// out zz[par][q] = X[2 lane + par + 128 q].  xb = this wave's private LDS exchange buffer (XCH complex).
// INV = true computes the UNSCALED inverse DFT: pass the conjugated twiddle tables (twBi / twCi), or the forward tables with
// CT = true (the conjugation then happens inside the multiply, at the same instruction count).
template <int K, bool INV = false, bool CT = false>
__device__ __forceinline__ void wave_fft_core(const v2f* d, v2f (*zz)[K / 128], v2f* xb, const v2f* s_twB, const v2f* s_twC,
                                              const int lane) {
  constexpr int P = K / 64;
  constexpr int R3 = K / 256;
  constexpr int B12 = P / 16;
  {
    // ---- pass A: radix-16, p = 1; butterfly i = lane + 64 u takes points i + t K/16  (= d[u + B12 t])
#pragma unroll
    for (int u = 0; u < B12; ++u) {
      v2f b[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) b[t] = d[u + B12 * t];
      dft16<INV>(b);
      const int base = 17 * (lane + 64 * u);  // pad1(16 i + r) = 17 i + r
#pragma unroll
      for (int r = 0; r < 16; ++r) xb[base + r] = b[r];
    }
    wave_lds_fence();

    // ---- pass B: radix-16, p = 16; reads pad1(i + t K/16), twiddle w_256^(t k), k = lane & 15
    const int k16 = lane & 15;
    v2f e[B12][16];
#pragma unroll
    for (int u = 0; u < B12; ++u) {
      const int base = lane + (lane >> 4) + 68 * u;  // pad1(l + 64 u + 64 B12 t) = l + l/16 + 68 u + 68 B12 t
#pragma unroll
      for (int t = 0; t < 16; ++t) e[u][t] = xb[base + 68 * B12 * t];
    }
#pragma unroll
    for (int u = 0; u < B12; ++u) {
#pragma unroll
      for (int t = 1; t < 16; ++t) e[u][t] = CT ? wcmul_conj(e[u][t], s_twB[t * 16 + k16]) : wcmul(e[u][t], s_twB[t * 16 + k16]);
      dft16<INV>(e[u]);
    }
    wave_lds_fence();  // every exchange-1 read is issued before exchange 2 overwrites the buffer
#pragma unroll
    for (int u = 0; u < B12; ++u) {
      const int base = 16 * (lane + 64 * u) - 15 * k16;  // (i - k) 16 + k
#pragma unroll
      for (int r = 0; r < 16; ++r) xb[base + 16 * r] = e[u][r];
    }
    wave_lds_fence();

    // ---- pass C: radix-R3, p = 256; butterflies i = 2 lane + par + 128 u2 (adjacent pair per 16-byte LDS read)
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2) {
      v2f c0[R3], c1[R3];
      const int i0 = 2 * lane + 128 * u2;
#pragma unroll
      for (int t = 0; t < R3; ++t) {
        const v4f v = *reinterpret_cast<const v4f*>(&xb[i0 + 256 * t]);
        c0[t] = v2f{v.x, v.y};
        c1[t] = v2f{v.z, v.w};
        if (t > 0) {
          const v4f w = *reinterpret_cast<const v4f*>(&s_twC[t * 256 + i0]);
          c0[t] = CT ? wcmul_conj(c0[t], v2f{w.x, w.y}) : wcmul(c0[t], v2f{w.x, w.y});
          c1[t] = CT ? wcmul_conj(c1[t], v2f{w.z, w.w}) : wcmul(c1[t], v2f{w.z, w.w});
        }
      }
      if (R3 == 4) { dft4<INV>(c0[0], c0[1], c0[2], c0[3]); dft4<INV>(c1[0], c1[1], c1[2], c1[3]); }
      else { dft8<INV>(c0); dft8<INV>(c1); }
#pragma unroll
      for (int r = 0; r < R3; ++r) { zz[0][u2 + 2 * r] = c0[r]; zz[1][u2 + 2 * r] = c1[r]; }
    }
    wave_lds_fence();  // next iteration's pass-A writes come after these reads
  }
}

// The same K-point forward DFT computed by the TRANSPOSED passes in reverse order (the DFT matrix is symmetric:
// F = C B A = A^T B^T C^T): in  zz[par][q] = x[2 lane + par + 128 q]  (the layout wave_fft_core PRODUCES),
// out d[s] = X[lane + 64 s]  (the layout wave_fft_core CONSUMES).  Chaining core_T -> pointwise -> core gives
// FFT -> multiply -> inverse FFT with no transposition pass in between (overlap-save FIR).
template <int K>
__device__ __forceinline__ void wave_fft_core_T(const v2f (*zz)[K / 128], v2f* d, v2f* xb, const v2f* s_twB,
                                                const v2f* s_twC, const int lane) {
  constexpr int P = K / 64;
  constexpr int R3 = K / 256;
  constexpr int B12 = P / 16;
  // ---- C^T: butterflies i = 2 lane + par + 128 u2 over the points i + 256 r; output t gets w_K^(t i); 16-byte writes
#pragma unroll
  for (int u2 = 0; u2 < 2; ++u2) {
    v2f c0[R3], c1[R3];
    const int i0 = 2 * lane + 128 * u2;
#pragma unroll
    for (int r = 0; r < R3; ++r) { c0[r] = zz[0][u2 + 2 * r]; c1[r] = zz[1][u2 + 2 * r]; }
    if (R3 == 4) { dft4(c0[0], c0[1], c0[2], c0[3]); dft4(c1[0], c1[1], c1[2], c1[3]); }
    else { dft8(c0); dft8(c1); }
#pragma unroll
    for (int t = 0; t < R3; ++t) {
      if (t > 0) {
        const v4f w = *reinterpret_cast<const v4f*>(&s_twC[t * 256 + i0]);
        c0[t] = wcmul(c0[t], v2f{w.x, w.y});
        c1[t] = wcmul(c1[t], v2f{w.z, w.w});
      }
      *reinterpret_cast<v4f*>(&xb[i0 + 256 * t]) = v4f{c0[t].x, c0[t].y, c1[t].x, c1[t].y};
    }
  }
  wave_lds_fence();
  // ---- B^T: butterfly i = lane + 64 u reads (i - k) 16 + k + 16 r, output t gets w_256^(t k), goes to pad1(i + t K/16)
  const int k16 = lane & 15;
  v2f e[B12][16];
#pragma unroll
  for (int u = 0; u < B12; ++u) {
    const int base = 16 * (lane + 64 * u) - 15 * k16;
#pragma unroll
    for (int r = 0; r < 16; ++r) e[u][r] = xb[base + 16 * r];
  }
#pragma unroll
  for (int u = 0; u < B12; ++u) {
    dft16(e[u]);
#pragma unroll
    for (int t = 1; t < 16; ++t) e[u][t] = wcmul(e[u][t], s_twB[t * 16 + k16]);
  }
  wave_lds_fence();
#pragma unroll
  for (int u = 0; u < B12; ++u) {
    const int base = lane + (lane >> 4) + 68 * u;
#pragma unroll
    for (int t = 0; t < 16; ++t) xb[base + 68 * B12 * t] = e[u][t];
  }
  wave_lds_fence();
  // ---- A^T: butterfly i reads pad1(16 i + r) = 17 i + r, outputs X[i + t K/16]
#pragma unroll
  for (int u = 0; u < B12; ++u) {
    v2f b[16];
    const int base = 17 * (lane + 64 * u);
#pragma unroll
    for (int r = 0; r < 16; ++r) b[r] = xb[base + r];
    dft16(b);
#pragma unroll
    for (int t = 0; t < 16; ++t) d[u + B12 * t] = b[t];
  }
  wave_lds_fence();
}

// Streaming stores of one wave into a contiguous output row: a wave-uniform buffer descriptor (base forced into SGPRs
// with readfirstlane: it usually depends on the wave index) + per-lane byte offsets, cache policy "sc1 nt".  On the
// headline STFT launch the sc1 bit is worth +5 % over a plain non-temporal global store (see ST above).
struct StreamRow {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ StreamRow(const void* uniform_base, uint32_t bytes) {
    const uint64_t v = reinterpret_cast<uint64_t>(uniform_base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, bytes, 0x00020000);
  }
  static constexpr int kAux = 18;  // gfx940+ cache policy bits: 1 = sc0, 2 = nt, 16 = sc1
  __device__ __forceinline__ void st16(v4f v, int byte_off) const {
    typedef int v4i __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), r, byte_off, 0, kAux);
  }
  __device__ __forceinline__ void st4(float v, int byte_off) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, byte_off, 0, kAux);
  }
  template <int AUX = kAux>
  __device__ __forceinline__ void st8(v2f v, int byte_off) const {
    typedef int v2i __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, byte_off, 0, AUX);
  }
};

struct WaveArgs {
  const float* x;
  int64_t batch_stride, L, lo, M;
  int32_t N, hop, reflect, batch;
  int64_t pairs_per_row;      // work units per row: ceil(M / 2) frame pairs (pair mode) or M frames (real-2x mode)
  int64_t total_pairs;        // units in this launch: batch * units_per_row (k_stft_wave), batch * pairs_per_row (others)
  int64_t units_per_row;      // k_stft_wave: units of each row covered by this launch (interior or edge set)
  int64_t u_split, u_add0, u_add1;  // unit u of the launch is unit-in-row u + (u < u_split ? u_add0 : u_add1)
  int64_t chunk;              // units per workgroup (contiguous)
  const float* wtab;          // device f32[fft_length]: window zero-padded / truncated to the fft length
  const v2f* twB;             // device c64[16][16]: w_256^(t k)
  const v2f* twC;             // device c64[R3][256]: w_C^(t i)
  const v2f* twR;             // device c64[C]: w_2C^k (real-2x mode only)
  float div;
  int32_t has_scale;
  v2f* z;
  v2f* dummy;                 // device c64[K]: sink for the phantom second frame of an odd tail (keeps the loop branch-free)
};

__device__ __forceinline__ float fetch_any(const float* __restrict__ x, const WaveArgs& a, int64_t q) {
  int64_t pos = q - a.lo;
  if (a.reflect) {
    if (a.L == 1) return x[0];
    const int64_t period = 2 * (a.L - 1);
    // one mirror on either side covers every padding shorter than the signal (the usual case) without the 64-bit modulo, which costs
    // more than a hundred instructions per sample; anything farther out takes it
    if (pos < 0) pos = -pos;
    if (pos >= a.L) pos = period - pos;
    if (pos < 0 || pos >= a.L) {
      pos %= period;
      if (pos < 0) pos += period;
      if (pos >= a.L) pos = period - pos;
    }
    return x[pos];
  }
  return (pos >= 0 && pos < a.L) ? x[pos] : 0.0f;
}

extern __shared__ __attribute__((aligned(16))) unsigned char g_wave_smem[];
#ifdef NXSIG_TRACE   // diagnostic builds only (tools/trace_small.py): per-wave time stamps of the pair-mode kernel's phases
static __device__ unsigned long long* g_wave_trace = nullptr;
#endif

typedef __attribute__((address_space(1))) v4f gv4f;
typedef __attribute__((address_space(1))) v2f gv2f;
typedef __attribute__((address_space(3))) float lds_f32;  // explicit global address space: global_store, not flat_store

// GENERAL = false: :valid framing with every existing frame fully inside the signal (the streaming case);
// GENERAL = true : any padding mode / ragged tail, per-sample bounds and mirror math.  SCALE: :spectrum / :psd.
// W = waves per workgroup (tables in LDS are shared by the W waves; waves never synchronise with each other).
// ---- fused STFT -> log-mel: NxSignal.stft (lib/nx_signal.ex:68-130) followed by stft_to_mel (:486-513) in
// ONE kernel, so that mel_bins * 4 bytes per frame leave the chip instead of the 8 * fft_length byte spectrum.  The MEL
// variants of k_stft_wave untangle only the bins below fft_length / 2, put |X|^2 of the unit's frames into the wave's
// (then idle) exchange buffer, and every lane sums its mel bands over the sparse (triangular) filter rows held in LDS
// as CSR; log10; per-wave running max -> one atomicMax.  A second tiny pass (k_mel_pass2) applies max(., gmax - 8) and
// (. + 4) / 4 once the global maximum is known.  Every front-end (pair / real-2x / quad) and the interior / edge split
// are shared with the plain STFT.
enum { kSinkSpectrum = 0, kSinkMel = 1, kSinkMag = 2 };

struct MelWaveArgs {
  WaveArgs w;                 // framing / tables of the STFT front half (z unused)
  int32_t mel_bins, nnz;
  const float* csr_w;         // [nnz] filter weights, band after band
  const int* csr_off;         // [mel_bins + 1]
  const int* csr_lo;          // [mel_bins] first bin of each band
  float ln10;
  float* out;                 // f32[batch][M][mel_bins] (log10 power, before the clamp pass) / f32[batch][M][fft_length/2] (MAG)
  int* gmax;
  int32_t mag_kind;           // MAG sink: 0 = |X|, 1 = |X|^2, 2 = |X| with a running maximum (dBFS pass follows),
                              // 3 = the complex bins themselves (one-sided spectrum: out is c64[...][fft_length / 2]),
                              // 4 = the same with Re X[fft_length / 2] packed into the imaginary part of bin 0 (pair mode only)
};

// ST (pair / real-2x front-ends, complex-spectrum sink, streaming kernel): cache policy of the spectrum stores.
// 1 (default) = buffer_store_dwordx4 ... sc1 nt through a wave-uniform buffer descriptor of the frame's row; 0 = global_store
// ... nt (__builtin_nontemporal_store), 2 = buffer_store ... nt.  Measured on the headline launch, interleaved in one process
// (tools/sweep_stft.py NXSIG_STORE_POLICY 0 1 2): 5.71 / 6.00 / 5.69 TB/s — the sc1 bit (system-scope write-through) is worth
// +5 %, the buffer addressing itself nothing.
template <int K, int MODE, bool GENERAL, bool SCALE, int W, int J, bool NPRED, int SINK, int STG = 0, int ST = 1>
__device__ __forceinline__ void stft_wave_body(const WaveArgs& a, const MelWaveArgs* mp) {
  // STG > 0 (quad streaming kernels): the unit's contiguous input span travels as STG 16-byte loads per lane and is
  // re-distributed through the wave's exchange buffer instead of 32 strided 4-byte loads per lane (see issue_loads)
  constexpr bool STAGED = STG > 0 && MODE == kModeQuad && !GENERAL;  // (pair mode: measured slower, 5.49 vs 5.85 TB/s)
  // STG == 5 (round 4): 4 loads per lane AND hop == KOUT / 4 known at compile time -> the parked span is PADDED against the bank
  // conflicts of the gather.  Lanes of one ds_read_b32 that share a sample index but belong to different frame pairs (lane % J) read
  // addresses 2 hop = KOUT / 2 floats apart — a multiple of 32 banks: J-way conflicts in every gather instruction (15 % of the
  // kernel's LDS cycles, profiles/r03/stft512_sq_counters_input_in_hbm.txt).  With PADF = 32 / J floats inserted after every block of
  // KOUT / 2 floats the J frame pairs land 32 / J banks apart and a 32-lane group covers every bank once.  The block index of a sample
  // changes INSIDE a frame (at s = P / 2 for frame A; at s = P / 4 and 3 P / 4 for frame B = A + hop), which is a compile-time offset
  // only when the hop is: hence the dedicated instantiation for the default 75 % overlap.
  constexpr bool HQP = STG == 5;
  constexpr int NST = HQP ? 4 : STG;       // 16-byte loads per lane
  constexpr int PADF = 32 / J;             // floats of padding per block (HQP)
  constexpr int BLK = (MODE == kModeQuad ? K / J : K) / 2;   // block = 2 hop = fft_length / 2 floats (HQP)
  constexpr bool HALFW = MODE == kModeReal2x;   // the untangle's 1/2 rides in the staged window (see stage_tables)
  constexpr bool MEL = SINK == kSinkMel;   // |X|^2 -> LDS -> sparse mel filterbank -> log10
  constexpr bool MAG = SINK == kSinkMag;   // |X| or |X|^2 of the bins below fft_length / 2 straight to HBM as f32
  constexpr int P = K / 64;     // complex points per lane
  constexpr int R3 = K / 256;   // last radix: 4 or 8
  constexpr int NQ = K / 128;   // bins per lane per parity
  constexpr int XCH = K + K / 16 + 16;  // padded exchange buffer, complex elements (keeps 16-B alignment)
  constexpr int KOUT = MODE == kModeReal2x ? 2 * K : (MODE == kModeQuad ? K / J : K);  // fft_length = bins per frame
  constexpr int TWQ = (J - 1) * (K / J);  // quad-mode separation twiddles conj(w_K^(j k0)), j = 1..J-1
  constexpr int kWavesPerBlock = W;
  constexpr int kWaveThreads = 64 * W;

  // ---- LDS carve: [window KOUT f32][twB 256 c64][twC R3*256 c64][twR K c64 (real-2x)][W x exchange]
  float* s_w = reinterpret_cast<float*>(g_wave_smem);
  v2f* s_twB = reinterpret_cast<v2f*>(s_w + KOUT);
  v2f* s_twC = s_twB + 256;
  v2f* s_twR = s_twC + R3 * 256;
  v2f* s_x = s_twR + (MODE == kModeReal2x ? K : (MODE == kModeQuad ? TWQ : 0));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef NXSIG_TRACE
  unsigned long long* tr = g_wave_trace ? g_wave_trace + ((size_t)blockIdx.x * W + wave) * 8 : nullptr;
  int tri = 0;
  auto stamp = [&]() { if (tr && lane == 0 && tri < 8) tr[tri] = wall_clock64(); ++tri; };
#else
  auto stamp = [] {};
#endif
  // the tables are staged (and the workgroup's only barrier passed) AFTER the first unit's sample loads have been issued, see
  // below: the two memory round trips of a workgroup's start-up overlap (a short launch is mostly start-up: config 2 as written)
  float* s_csr = reinterpret_cast<float*>(s_x + W * XCH);
  int* s_off = reinterpret_cast<int*>(s_csr + (MEL ? mp->nnz : 0));
  int* s_lo = s_off + (MEL ? mp->mel_bins + 1 : 0);
  auto stage_tables = [&]() {
    // the window travels HALVED (round 5): the 1/2 of the Hermitian untangle XA = (Z + conj Z') / 2 rides in the window — an exact power of
    // two, so (x w / 2) and every butterfly sum downstream are the same bits scaled by 1/2 (underflow aside: |x w| < 2^-125, far
    // below the 1e-10 clean-up threshold) — and the drains lose four packed multiplies per bin group (32 of ~660 VALU per frame pair)
    // Real-2x front-end only (A/B against the previous build, tools/ab_libs.py: config 4's shard +2.2 %; the pair kernel's schedule
    // LOST 1 % to the same change and keeps its multiplies)
    for (int i = tid; i < KOUT; i += kWaveThreads) s_w[i] = HALFW ? a.wtab[i] * 0.5f : a.wtab[i];
    for (int i = tid; i < 256; i += kWaveThreads) s_twB[i] = a.twB[i];
    for (int i = tid; i < R3 * 256; i += kWaveThreads) s_twC[i] = a.twC[i];
    if (MODE == kModeReal2x)
      for (int i = tid; i < K; i += kWaveThreads) s_twR[i] = a.twR[i];
    if (MODE == kModeQuad)
      for (int i = tid; i < TWQ; i += kWaveThreads) s_twR[i] = a.twR[i];  // [j-1][k0] = conj(w_K^(j k0))
    // MEL: [nnz] filter weights, [mel_bins + 1] offsets, [mel_bins] first bins after the exchange buffers
    if (MEL) {
      for (int i = tid; i < mp->nnz; i += kWaveThreads) s_csr[i] = mp->csr_w[i];
      for (int i = tid; i <= mp->mel_bins; i += kWaveThreads) s_off[i] = mp->csr_off[i];
      for (int i = tid; i < mp->mel_bins; i += kWaveThreads) s_lo[i] = mp->csr_lo[i];
    }
    __syncthreads();  // the only workgroup barrier: tables are read-only afterwards
  };
  v2f* xb = s_x + wave * XCH;
  // MEL: after the core the exchange buffer is idle: |X|^2 of frame f of the unit at mags[f * KOUT/2 + k], k < KOUT/2
  float* mags = reinterpret_cast<float*>(xb);
  constexpr int FPU = MODE == kModePair ? 2 : (MODE == kModeQuad ? 2 * J : 1);  // frames per unit
  constexpr int KH = KOUT / 2;
  float vmax = -3.0e38f;
  // log-mel: a non-finite |z|^2 poisons the WHOLE tensor in the reference (its dense Nx.dot forms inf x 0 with the zero weights of every
  // band, the NaN reaches reduce_max, lib/nx_signal.ex:505-511) and so it does in the two-step k_mel_tile; the sparse band sums here
  // never form that product, so the kernel raises the non-finite flag (gmax[1]) itself: when a unit's windowed samples are not all
  // finite, or a band sum is not
  bool melbad = false;
  auto mel_tail = [&](int64_t crow, int64_t mA) {
    wave_lds_fence();
    // ---- sparse filterbank + log10
    float* o0p = mp->out + ((size_t)crow * a.M + mA) * mp->mel_bins;
    // the partial pass (mel_bins mod 64 lanes) takes the NARROWEST bands 0 .. rot - 1, the full passes the rest: a wave instruction
    // costs its widest lane, and with 80 bands the partial pass otherwise holds the 16 widest ones
    const int rot = mp->mel_bins & 63;
    for (int b0 = lane; b0 < mp->mel_bins; b0 += 64) {
      const int b = b0 + rot < mp->mel_bins ? b0 + rot : b0 + rot - mp->mel_bins;
      const int o0 = s_off[b], o1 = s_off[b + 1], k0 = s_lo[b];
      float acc[FPU];
#pragma unroll
      for (int f = 0; f < FPU; ++f) acc[f] = 0.0f;
      for (int j = o0; j < o1; ++j) {
        const float wv = s_csr[j];
#pragma unroll
        for (int f = 0; f < FPU; ++f) acc[f] = fmaf(mags[f * KH + k0 + (j - o0)], wv, acc[f]);
      }
#pragma unroll
      for (int f = 0; f < FPU; ++f) {
        melbad |= !(acc[f] < INFINITY);
        const float av = acc[f] > 1.0e-10f ? acc[f] : 1.0e-10f;
        // hardware log2 (v_log_f32, ~1 ulp) * log10(2): |error| ~ 1e-7, far inside the 1e-4 the reference's tests use
        const float v = __log2f(av) * 0.30102999566398120f;
        if (mA + f < a.M) { o0p[(size_t)f * mp->mel_bins + b] = v; vmax = v > vmax ? v : vmax; }
      }
    }
    wave_lds_fence();  // the power spectrum is consumed before the next pass A overwrites the buffer
  };
  // MAG sink: two adjacent bins of one frame (p2 = |X[k]|^2, |X[k+1]|^2) -> f32 row of fft_length / 2 values
  auto mag_store = [&](float* rowp, v2f p2) {
    v2f v = p2;
    if (mp->mag_kind != 1) v = v2f{__builtin_sqrtf(p2.x), __builtin_sqrtf(p2.y)};
    const float mx = v.x > v.y ? v.x : v.y;
    vmax = mx > vmax ? mx : vmax;
    __builtin_nontemporal_store(v, (gv2f*)rowp);
  };

  // workgroup b takes chunk b: the dispatcher deals consecutive chunks round-robin over the 8 XCDs (an XCD-contiguous walk and
  // persistent variants measured slower: profiles/r02, profiles/r03/negative_results.md)
  const int64_t p_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t p_end = p_begin + a.chunk;
  if (p_end > a.total_pairs) p_end = a.total_pairs;

  // Streaming kernels are software-pipelined one pair deep.  Iteration i: issue the raw loads of pair i+1 ->
  // FFT passes of pair i -> multiply the (long since landed) samples of pair i+1 by the window -> untangle and
  // store pair i.  The loads are consumed BEFORE this pair's stores are issued, so the only VMEM ops ahead of
  // them in gfx9's in-order queue are the previous iteration's stores (a whole iteration old): no wait ever
  // drains fresh stores, and HBM latency hides under the butterflies.
  float ra[STAGED ? 1 : P], rb[STAGED ? 1 : P];
  v4f rs[STAGED ? NST : 1];
  constexpr int FPU_IN = MODE == kModeQuad ? 2 * J : 2;  // frames per unit
  // floats of one unit's input span, 16-byte multiple; the unpadded form also takes spans that do not START on a 16-byte boundary (rows of
  // odd length, host slices): the loads begin `mis` floats early and the reads out of the parked span skip them (round 5: such rows fell
  // back to strided 4-byte gathers, 0.49 against 0.61-0.68 of the roofline for N = 512 / 256 / 128)
  const int span4 = STAGED ? ((((FPU_IN - 1) * a.hop + KOUT + 3) & ~3) + (HQP ? 0 : 4)) : 0;
  int mis = 0;
  auto issue_loads = [&](int64_t row, int64_t pin) {
    if (STAGED) {
      // the unit's frames 2J pin .. 2J pin + 2J - 1 read x[unit start .. + span): one contiguous run
      const float* pu = a.x + (size_t)row * a.batch_stride + (pin * FPU_IN * (int64_t)a.hop - a.lo);
      if (!HQP) mis = (int)((reinterpret_cast<uintptr_t>(pu) >> 2) & 3);   // wave-uniform
      const v4f* p4 = reinterpret_cast<const v4f*>(pu - mis) + lane;
#pragma unroll
      for (int c = 0; c < (STAGED ? NST : 0); ++c)
        rs[c] = (256 * c + 4 * lane < span4) ? p4[64 * c] : v4f{0.f, 0.f, 0.f, 0.f};
    } else if (MODE == kModePair) {
      const int64_t mA = pin * 2;
      const float* pa = a.x + (size_t)row * a.batch_stride + (mA * a.hop - a.lo) + lane;
      const float* pb = pa + ((mA + 1 < a.M) ? a.hop : 0);  // phantom frame B of an odd tail: reload A, never stored
#pragma unroll
      for (int s = 0; s < P; ++s) { ra[s] = pa[64 * s]; rb[s] = pb[64 * s]; }
    } else if (MODE == kModeReal2x) {  // complex point n = (x[2n], x[2n+1])
      const float* pa = a.x + (size_t)row * a.batch_stride + (pin * a.hop - a.lo) + 2 * lane;
      if (STG == 2) {  // every frame starts on an 8-byte boundary (checked by the launcher): one 8-byte load per point
#pragma unroll
        for (int s = 0; s < P; ++s) { const v2f t = *reinterpret_cast<const v2f*>(pa + 128 * s); ra[s] = t.x; rb[s] = t.y; }
      } else {
#pragma unroll
        for (int s = 0; s < P; ++s) { ra[s] = pa[128 * s]; rb[s] = pa[128 * s + 1]; }
      }
    } else {  // quad: lane carries sequence j = lane % J = frames (m0 + 2j, m0 + 2j + 1); core point lane + 64 s = J n + j
      const int64_t fa = pin * (2 * J) + 2 * (lane % J), fb = fa + 1, last = a.M - 1;
      const float* base = a.x + (size_t)row * a.batch_stride - a.lo + (lane / J);
      const float* pa = base + (fa < last ? fa : last) * a.hop;  // phantom frames of a ragged tail: reload, never stored
      const float* pb = base + (fb < last ? fb : last) * a.hop;
#pragma unroll
      for (int s = 0; s < P; ++s) ra[s] = pa[(64 / J) * s];
#pragma unroll
      for (int s = 0; s < P; ++s) rb[s] = pb[(64 / J) * s];
    }
  };
  auto window_mul = [&](v2f* d) {
    if (STAGED) {
      // runs at the end of the iteration: the exchange buffer is idle until the next pass A.  Park the span, then
      // every lane picks its 2 P samples: frames 2 (lane % J), +1 of the unit, sample lane / J + (64 / J) s.
      float* xsf = reinterpret_cast<float*>(xb);
#pragma unroll
      for (int c = 0; c < (STAGED ? NST : 0); ++c) {
        const int idx = 256 * c + 4 * lane;                       // (a 16-byte piece never straddles a block: BLK % 4 == 0)
        if (idx < span4) *reinterpret_cast<v4f*>(&xsf[HQP ? idx + (idx / BLK) * PADF : idx]) = rs[c];
      }
      wave_lds_fence();
      constexpr int JJ = MODE == kModeQuad ? J : 1;  // pair mode: frames 0, 1 of the unit, sample lane + 64 s
      if constexpr (HQP) {
        // frame A of pair j = lane % J starts block j; its sample n0 + (64 / J) s lies in block j + (s >= P / 2); frame B = A + hop
        // = A + BLK / 2 lies in block j + (s >= P / 4) + (s >= 3 P / 4)
        const float* fa = xsf + (lane % JJ) * (BLK + PADF) + (lane / JJ);
#pragma unroll
        for (int s = 0; s < P; ++s) {
          const float w = s_w[(lane / JJ) + (64 / JJ) * s];
          const int oa = (64 / JJ) * s + (s >= P / 2 ? PADF : 0);
          const int ob = BLK / 2 + (64 / JJ) * s + ((s >= P / 4 ? 1 : 0) + (s >= 3 * P / 4 ? 1 : 0)) * PADF;
          d[s] = v2f{fa[oa] * w, fa[ob] * w};
        }
        wave_lds_fence();
        return;
      }
      const float* fa = xsf + mis + (2 * (lane % JJ)) * a.hop + (lane / JJ);
      const float* fb = fa + a.hop;
#pragma unroll
      for (int s = 0; s < P; ++s) {
        const int nn = (lane / JJ) + (64 / JJ) * s;
        const float w = s_w[nn];
        d[s] = v2f{fa[(64 / JJ) * s] * w, fb[(64 / JJ) * s] * w};
        if (NPRED && nn >= a.N) d[s] = v2f{0.f, 0.f};
      }
      wave_lds_fence();
      return;
    }
#pragma unroll
    for (int s = 0; s < P; ++s) {
      // NPRED (frame_length < fft_length): samples past the frame are loaded (the unit is interior) but must not
      // reach the transform even as 0 * x, which would turn an Inf / NaN outside the frame into NaN
      if (MODE == kModePair) {
        const float w = s_w[lane + 64 * s];
        d[s] = v2f{ra[s] * w, rb[s] * w};
        if (NPRED && lane + 64 * s >= a.N) d[s] = v2f{0.f, 0.f};
      } else if (MODE == kModeReal2x) {
        const v2f w = *reinterpret_cast<const v2f*>(&s_w[2 * (lane + 64 * s)]);
        d[s] = v2f{ra[s] * w.x, rb[s] * w.y};
        if (NPRED && 2 * (lane + 64 * s) >= a.N) d[s].x = 0.f;
        if (NPRED && 2 * (lane + 64 * s) + 1 >= a.N) d[s].y = 0.f;
      } else {
        const float w = s_w[(lane / J) + (64 / J) * s];
        d[s] = v2f{ra[s] * w, rb[s] * w};
        if (NPRED && (lane / J) + (64 / J) * s >= a.N) d[s] = v2f{0.f, 0.f};
      }
    }
  };
  // (row, pair-in-row) of this wave's current and next pair, advanced incrementally (no division in the loop)
  // A launch covers units_per_row of each row's pairs_per_row units: the interior ones (every sample of every frame
  // inside the signal: the streaming kernels) or the few edge ones (GENERAL).  unit index u -> unit-in-row:
  auto pinof = [&](int64_t u) { return u + (u < a.u_split ? a.u_add0 : a.u_add1); };
  int64_t row = (p_begin + wave) / a.units_per_row;
  int64_t uin = (p_begin + wave) - row * a.units_per_row;
  int64_t nrow = row, nuin = uin;
  auto advance = [&](int64_t& r, int64_t& q) {
    q += kWavesPerBlock;
    while (q >= a.units_per_row) { q -= a.units_per_row; ++r; }
  };
  advance(nrow, nuin);
  v2f d[P];  // windowed samples of the current pair: re = frame A, im = frame B (exact f32 products, :101)
  const bool have_first = !GENERAL && p_begin + wave < p_end;
  stamp();
  stage_tables();
  stamp();
  if (have_first) issue_loads(row, pinof(uin));
  if (have_first) window_mul(d);
  stamp();

  // ---- Nx.fft's clean-up (SURVEY App. A rule 7; call site lib/nx_signal.ex:102): every component of the finished spectrum
  // with |x| <= eps = 1e-10 becomes +0, BEFORE the :spectrum / :psd division (:113-127).  NaN compares false and stays.
  // Round 4, spectrum sink of the streaming kernels: the clean-up is SPECULATIVE.  The threshold concerns digital silence only, so
  // the first drain of a unit stores the untangled spectrum as it is while every lane tracks the minimum magnitude of what it stores
  // (one v_min3 per two components instead of a compare + select per component); when some lane of the wave met a component at or
  // below the threshold, the unit is drained AGAIN with the clean-up (cold; the wave's own later stores to the same addresses land
  // after the first ones).  NaN never trips the test (minNum) and passes through both drains unchanged.  (A run-time switch between
  // this and the eager form put a third copy of the drain into the kernel and cost the headline 5.5 % — measured against the previous
  // build side by side in one process, tools/ab_libs.py — so there is none: the form is a compile-time property of the sink.)
  constexpr bool SPEC_CLEAN = SINK == kSinkSpectrum && !GENERAL;
  // ---- non-finite samples.  The reference transforms every frame on its own (one Nx.fft row per frame, lib/nx_signal.ex:94-102),
  // so an Inf / NaN sample reaches only the frames that contain it.  Frames that share one complex transform here (2 in pair mode,
  // 2J in quad mode) would share it: a unit whose windowed samples are not all finite therefore leaves the paired route and runs
  // each of its frames ALONE through the core (real part of sequence 0, everything else zero).  The test is one packed add per
  // point on values the loop holds anyway: the sum of the unit's windowed samples is finite iff they all are (an overflowing sum
  // only sends a finite unit down the solo route, which computes the same spectra).  Wave-uniform branch, cold path.
  constexpr bool CAN_SOLO = FPU > 1 && !MEL;  // log-mel: any non-finite |z|^2 poisons the whole tensor in the reference (reduce_max)
  auto unit_nonfinite = [&](const v2f* dd) -> bool {
    v2f t = dd[0];
#pragma unroll
    for (int s = 1; s < P; ++s) t += dd[s];
    const bool nf = ((__float_as_uint(t.x) & 0x7f800000u) == 0x7f800000u) || ((__float_as_uint(t.y) & 0x7f800000u) == 0x7f800000u);
    return __builtin_amdgcn_ballot_w64(nf) != 0;
  };
  // GENERAL kernels: bounds-checked, padded / mirrored samples of the unit's frames (solo < 0), or frame mA + solo alone
  auto load_general = [&](v2f* dd, int64_t lrow, int64_t mA, int solo) {
    const float* xr = a.x + (size_t)lrow * a.batch_stride;
#pragma unroll
    for (int s = 0; s < P; ++s) {
      const int n = lane + 64 * s;
      if (MODE == kModePair) {
        const int64_t fa = solo < 0 ? mA : mA + solo;
        const bool useB = solo < 0 && mA + 1 < a.M;
        const float w = s_w[n];
        const float va = (n < a.N) ? fetch_any(xr, a, fa * a.hop + n) : 0.0f;
        const float vb = (useB && n < a.N) ? fetch_any(xr, a, (fa + 1) * a.hop + n) : 0.0f;
        dd[s] = v2f{va * w, vb * w};
      } else if (MODE == kModeReal2x) {
        const int64_t qA = mA * a.hop;
        const float va = (2 * n < a.N) ? fetch_any(xr, a, qA + 2 * n) : 0.0f;
        const float vb = (2 * n + 1 < a.N) ? fetch_any(xr, a, qA + 2 * n + 1) : 0.0f;
        dd[s] = v2f{va * s_w[2 * n], vb * s_w[2 * n + 1]};
      } else {
        const int nn = (lane / J) + (64 / J) * s;               // sample index inside the K/J-sample frames
        const int jj = lane % J;                                 // lane % J selects the frame pair
        const int64_t fa = solo < 0 ? mA + 2 * jj : mA + solo, fb = fa + 1;
        const bool okA = solo < 0 ? fa < a.M : jj == 0, okB = solo < 0 && fb < a.M;
        const float w = s_w[nn];
        const float va = (okA && nn < a.N) ? fetch_any(xr, a, fa * a.hop + nn) : 0.0f;
        const float vb = (okB && nn < a.N) ? fetch_any(xr, a, fb * a.hop + nn) : 0.0f;
        dd[s] = v2f{va * w, vb * w};
      }
    }
  };
  // streaming kernels, solo route: frame m alone, straight from memory (the unit is interior: every sample exists)
  auto load_solo = [&](v2f* dd, int64_t lrow, int64_t m) {
    const float* xf = a.x + (size_t)lrow * a.batch_stride + (m * a.hop - a.lo);
#pragma unroll
    for (int s = 0; s < P; ++s) {
      const int nn = MODE == kModeQuad ? (lane / J) + (64 / J) * s : lane + 64 * s;
      const bool use = (MODE != kModeQuad || (lane % J) == 0) && (!NPRED || nn < a.N);
      const float v = xf[nn] * s_w[nn];
      dd[s] = v2f{use ? v : 0.0f, 0.0f};
    }
  };

  // ---- Hermitian untangle through partner lanes + eps clean-up + scaling + store.  SOLO: only slot 0 of the unit (real part of
  // sequence 0) is meaningful; it is frame mS of the row and leaves through plain non-temporal stores.
  auto drain = [&](auto solo_c, auto spec_c, v2f (*zz)[NQ], const int64_t crow, const int64_t mA, const bool haveB, const int64_t mS) -> bool {
    constexpr bool SOLO = decltype(solo_c)::value;
    constexpr bool SPEC = decltype(spec_c)::value;   // store uncleaned, report whether anything needed the clean-up
    float amin = 3.0e38f;
    // skip_y: component .y of this value is a STRUCTURAL zero — the imaginary part of a real frame's DC / Nyquist bin, formed as
    // x - x = +0 exactly (the bin is its own Hermitian partner) — and must not trip the test: every unit holds those.
    auto eps_clean = [&](v4f v, const bool skip_y = false) -> v4f {
      if constexpr (SPEC) {
        amin = min3abs(v.x, skip_y ? 3.0e38f : v.y, amin);
        amin = min3abs(v.z, v.w, amin);
        return v;
      } else {
        return fft_eps0(v);
      }
    };
    const int src0 = ((64 - lane) & 63) << 2, src1 = (63 - lane) << 2;
    if constexpr (MODE == kModeQuad) {
      constexpr int HQ = NQ / J;  // bins per lane per parity of the short spectra
      constexpr int JN = SOLO ? 1 : J;
      // C_j[k0] = conj(w_K^(j k0)) / J * sum_m Z[k0 + (K/J) m] conj(w_J^(jm)),  k0 = 2 lane + par + 128 q, q < HQ
      v2f cs[JN][2][HQ];
#pragma unroll
      for (int q = 0; q < HQ; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          v2f u[J];
#pragma unroll
          for (int m = 0; m < J; ++m) u[m] = zz[e][q + HQ * m];
          if constexpr (J == 2) { const v2f s0 = u[0] + u[1], s1 = u[0] - u[1]; u[0] = s0; u[1] = s1; }
          else if constexpr (J == 4) dft4<true>(u[0], u[1], u[2], u[3]);
          else dft8<true>(u);
#pragma unroll
          for (int j = 0; j < JN; ++j) {
            v2f v = u[j] * (1.0f / (float)J);
            if (j > 0) v = wcmul(v, s_twR[(j - 1) * (K / J) + 2 * lane + e + 128 * q]);
            cs[j][e][q] = v;
          }
        }
      const int64_t m0 = SOLO ? mS : mA;
      v2f* z0 = a.z + ((size_t)crow * a.M + m0) * KOUT + 2 * lane;
      v2f* dm = a.dummy + 2 * lane;
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        v2f* zfa = (m0 + 2 * j < a.M) ? z0 + (size_t)(2 * j) * KOUT : dm;      // frame m0 + 2j      (real part of c_j)
        v2f* zfb = (m0 + 2 * j + 1 < a.M) ? z0 + (size_t)(2 * j + 1) * KOUT : dm;  // frame m0 + 2j + 1 (imaginary part)
        const StreamRow ra(zfa - 2 * lane, KOUT * 8), rb2(zfb - 2 * lane, KOUT * 8);  // wave-uniform row descriptors (SALU only)
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
          const v2f own0 = cs[j][0][(HQ - q) % HQ];
          v2f p0, p1;
          p0.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src0, __float_as_int(cs[j][0][HQ - 1 - q].x)));
          p0.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src0, __float_as_int(cs[j][0][HQ - 1 - q].y)));
          p1.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src1, __float_as_int(cs[j][1][HQ - 1 - q].x)));
          p1.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src1, __float_as_int(cs[j][1][HQ - 1 - q].y)));
          if (lane == 0) p0 = own0;
          const v2f z0v = cs[j][0][q], z1v = cs[j][1][q];
          // bins 0 and KOUT / 2 of the short spectra: lane 0 at q = 0 and q = HQ / 2 (J = 8: KOUT / 2 = 64 sits on lane 32, q = 0)
          const bool self = HQ >= 2 ? (lane == 0 && (q == 0 || q == HQ / 2)) : ((lane & 31) == 0);
          v4f xa = eps_clean(v4f{z0v.x + p0.x, z0v.y - p0.y, z1v.x + p1.x, z1v.y - p1.y} * 0.5f, self);
          v4f xbv = eps_clean(v4f{z0v.y + p0.y, p0.x - z0v.x, z1v.y + p1.y, p1.x - z1v.x} * 0.5f, self);
          if (SCALE) { xa = xa / a.div; xbv = xbv / a.div; }
          const bool stA = SOLO || m0 + 2 * j < a.M, stB = !SOLO && m0 + 2 * j + 1 < a.M;
          if (MEL || MAG) {
            if (HQ >= 2 ? (q < HQ / 2) : (lane < 32)) {  // bins k0 = 2 lane + par + 128 q below fft_length / 2
              const v2f pa2 = v2f{xa.x * xa.x + xa.y * xa.y, xa.z * xa.z + xa.w * xa.w};
              const v2f pb2 = v2f{xbv.x * xbv.x + xbv.y * xbv.y, xbv.z * xbv.z + xbv.w * xbv.w};
              if (MEL) {
                *reinterpret_cast<v2f*>(&mags[(2 * j) * KH + 2 * lane + 128 * q]) = pa2;
                *reinterpret_cast<v2f*>(&mags[(2 * j + 1) * KH + 2 * lane + 128 * q]) = pb2;
              } else {
                if (mp->mag_kind == 3) {  // one-sided complex spectrum: two adjacent bins per 16-byte store
                  v2f* c0 = reinterpret_cast<v2f*>(mp->out) + ((size_t)crow * a.M + m0 + 2 * j) * KH + 2 * lane + 128 * q;
                  if (stA) __builtin_nontemporal_store(xa, (gv4f*)c0);
                  if (stB) __builtin_nontemporal_store(xbv, (gv4f*)(c0 + KH));
                } else {
                  float* r0 = mp->out + ((size_t)crow * a.M + m0 + 2 * j) * KH + 2 * lane + 128 * q;
                  if (stA) mag_store(r0, pa2);
                  if (stB) mag_store(r0 + KH, pb2);
                }
              }
            }
          } else if (SOLO) {
            __builtin_nontemporal_store(xa, (gv4f*)(zfa + 128 * q));
          } else if (ST > 0 && !GENERAL) {  // streaming kernel: "sc1 nt" stores through the two frames' row descriptors
            ra.st16(xa, lane * 16 + 1024 * q);
            rb2.st16(xbv, lane * 16 + 1024 * q);
          } else {
            __builtin_nontemporal_store(xa, (gv4f*)(zfa + 128 * q));
            __builtin_nontemporal_store(xbv, (gv4f*)(zfb + 128 * q));
          }
        }
      }
    } else {
      const int64_t m0 = SOLO ? mS : mA;
      v2f* zA = a.z + ((size_t)crow * a.M + m0) * KOUT + 2 * lane;
      v2f* zB = (GENERAL || haveB) ? zA + K : a.dummy + 2 * lane;  // pair: frame B; real-2x: bins K..2K-1
      constexpr bool BUFST = ST > 0 && (MODE == kModePair || MODE == kModeReal2x) && !GENERAL && SINK == kSinkSpectrum && !SOLO;
      __amdgpu_buffer_rsrc_t rsA, rsB;
      if (BUFST) {  // wave-uniform row descriptors (the row base depends on the wave index: make it an SGPR pair explicitly)
        auto uni = [](const v2f* p) -> void* {
          const uint64_t v = reinterpret_cast<uint64_t>(p);
          const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
          return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
        };
        rsA = __builtin_amdgcn_make_buffer_rsrc(uni(zA - 2 * lane), 0, K * 8, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc(uni(zB - 2 * lane), 0, K * 8, 0x00020000);
      }
      const bool stB = !SOLO && (MODE == kModePair ? haveB : true);
      constexpr int QN = ((MEL || MAG) && MODE == kModePair) ? NQ / 2 : NQ;  // MEL / MAG: only the bins below fft_length / 2
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        // partner of bin k = 2l+par+128q is K-k = 2l'+par+128(NQ-1-q) on lane l' (lane 0 / par 0: own (NQ-q) % NQ)
        const v2f own0 = zz[0][(NQ - q) % NQ];
        v2f p0, p1;
        p0.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src0, __float_as_int(zz[0][NQ - 1 - q].x)));
        p0.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src0, __float_as_int(zz[0][NQ - 1 - q].y)));
        p1.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src1, __float_as_int(zz[1][NQ - 1 - q].x)));
        p1.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src1, __float_as_int(zz[1][NQ - 1 - q].y)));
        if (lane == 0) p0 = own0;
        const v2f z0 = zz[0][q], z1 = zz[1][q];
        // XA = ((a + c), (b - d)) / 2 ; XB = ((b + d), (c - a)) / 2   with Z = a + ib, Z[K-k] = c + id (Z arrives halved)
        v4f xa = v4f{z0.x + p0.x, z0.y - p0.y, z1.x + p1.x, z1.y - p1.y};
        v4f xbv = v4f{z0.y + p0.y, p0.x - z0.x, z1.y + p1.y, p1.x - z1.x};
        if (!HALFW) { xa = xa * 0.5f; xbv = xbv * 0.5f; }   // (HALFW: the 1/2 rides in the window, see stage_tables)
        if (MODE == kModeReal2x) {
          // xa = E[k], xbv = O[k] (spectra of the even / odd samples): X[k] = E + w_2K^k O, X[k+K] = E - w_2K^k O
          const v4f t = *reinterpret_cast<const v4f*>(&s_twR[2 * lane + 128 * q]);
          const v2f o0 = wcmul(v2f{xbv.x, xbv.y}, v2f{t.x, t.y}), o1 = wcmul(v2f{xbv.z, xbv.w}, v2f{t.z, t.w});
          const v4f to = v4f{o0.x, o0.y, o1.x, o1.y};
          xbv = xa - to;
          xa = xa + to;
        }
        // pair: bins 0 and K / 2 (lane 0, q = 0 and NQ / 2) are their own partners; real-2x: bins 0 and K of the 2K-point spectrum
        // (lane 0, q = 0: E and O are real there and w = 1)
        const bool self = lane == 0 && (q == 0 || (MODE == kModePair && q == NQ / 2));
        xa = eps_clean(xa, self);
        if (!SOLO) xbv = eps_clean(xbv, self);
        if (SCALE) { xa = xa / a.div; xbv = xbv / a.div; }  // true division like the reference (:116/:119)
        if (MEL) {
          const v2f pa2 = v2f{xa.x * xa.x + xa.y * xa.y, xa.z * xa.z + xa.w * xa.w};      // |X[k]|^2, |X[k+1]|^2
          *reinterpret_cast<v2f*>(&mags[2 * lane + 128 * q]) = pa2;
          if (MODE == kModePair) {
            const v2f pb2 = v2f{xbv.x * xbv.x + xbv.y * xbv.y, xbv.z * xbv.z + xbv.w * xbv.w};
            *reinterpret_cast<v2f*>(&mags[KH + 2 * lane + 128 * q]) = pb2;
          }
        } else if (MAG) {
          if (mp->mag_kind >= 3) {  // one-sided complex spectrum: two adjacent bins per 16-byte store
            if (MODE == kModePair && mp->mag_kind == 4 && q == 0 && lane == 0) {
              // packed form (kind 4, nxsig_stft_packed_f32): the Nyquist bin X[K/2] of a real frame is real and bin 0's imaginary
              // part is zero -> Re X[K/2] rides there.  Z[K/2] sits on lane 0 (zz[0][NQ/2]): its real part belongs to frame A,
              // its imaginary part to frame B (XA[K/2] = Re Z, XB[K/2] = Im Z); clean-up and scaling as for every bin
              float nyA = fft_eps0(zz[0][NQ / 2].x), nyB = fft_eps0(zz[0][NQ / 2].y);
              if (SCALE) { nyA = nyA / a.div; nyB = nyB / a.div; }
              xa.y = nyA; xbv.y = nyB;
            }
            v2f* c0 = reinterpret_cast<v2f*>(mp->out) + ((size_t)crow * a.M + m0) * KH + 2 * lane + 128 * q;
            __builtin_nontemporal_store(xa, (gv4f*)c0);
            if (MODE == kModePair && stB) __builtin_nontemporal_store(xbv, (gv4f*)(c0 + KH));
          } else {
            float* r0 = mp->out + ((size_t)crow * a.M + m0) * KH + 2 * lane + 128 * q;
            mag_store(r0, v2f{xa.x * xa.x + xa.y * xa.y, xa.z * xa.z + xa.w * xa.w});
            if (MODE == kModePair && stB) mag_store(r0 + KH, v2f{xbv.x * xbv.x + xbv.y * xbv.y, xbv.z * xbv.z + xbv.w * xbv.w});
          }
        } else if (BUFST) {
          typedef int v4i __attribute__((ext_vector_type(4)));
          constexpr int AUX = ST == 1 ? 18 : 2;  // gfx940+ cache policy bits: 1 = sc0, 2 = nt, 16 = sc1
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, xa), rsA, lane * 16 + 1024 * q, 0, AUX);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, xbv), rsB, lane * 16 + 1024 * q, 0, AUX);
        } else {
          __builtin_nontemporal_store(xa, (gv4f*)(zA + 128 * q));
          if (!SOLO && (!GENERAL || haveB)) __builtin_nontemporal_store(xbv, (gv4f*)(zB + 128 * q));
        }
      }
    }
    return SPEC && __builtin_amdgcn_ballot_w64(amin <= kFftEps) != 0;
  };

  for (int64_t pr = p_begin + wave; pr < p_end; pr += kWavesPerBlock) {
    const int64_t pin = pinof(uin);
    const int64_t mA = MODE == kModePair ? pin * 2 : (MODE == kModeQuad ? pin * (2 * J) : pin), mB = mA + 1;
    const bool haveB = MODE == kModeReal2x ? true : (mB < a.M);
    const int64_t crow = row;
    if (!GENERAL) {
      // unconditional prefetch (the last iteration harmlessly re-reads its own unit) keeps the loop branch-free
      const bool more = pr + kWavesPerBlock < p_end;
      issue_loads(more ? nrow : row, more ? pinof(nuin) : pin);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      load_general(d, row, mA, -1);
    }

    // next unit's raw samples -> windowed d[].  Pair / real-2x: right after the butterflies (the loads had the whole core
    // to land and their registers are free for the untangle).  Quad: at the very end of the iteration (LATE): with its
    // input in HBM rather than in the Infinity Cache the quad kernels were waiting here (4.0 instead of 6.0 TB/s).
    constexpr bool LATE = MODE == kModeQuad || STAGED;
    v2f zz[2][NQ];  // zz[par][q] = Z[2 lane + par + 128 q]
    if (MEL && unit_nonfinite(d)) melbad = true;
    if (CAN_SOLO && unit_nonfinite(d)) {
#pragma nounroll
      for (int f = 0; f < FPU && mA + f < a.M; ++f) {
        v2f ds[P];
        if (GENERAL) load_general(ds, crow, mA, f); else load_solo(ds, crow, mA + f);
        wave_fft_core<K>(ds, zz, xb, s_twB, s_twC, lane);
        drain(std::true_type{}, std::false_type{}, zz, crow, mA, haveB, mA + f);
      }
      if (!GENERAL) {
        __builtin_amdgcn_sched_barrier(0);
        window_mul(d);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      wave_fft_core<K>(d, zz, xb, s_twB, s_twC, lane);
      if (!GENERAL && !LATE) {
        __builtin_amdgcn_sched_barrier(0);
        window_mul(d);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (drain(std::false_type{}, std::integral_constant<bool, SPEC_CLEAN>{}, zz, crow, mA, haveB, mA))
        drain(std::false_type{}, std::false_type{}, zz, crow, mA, haveB, mA);   // cold: the unit again, with the clean-up
      if (MEL) mel_tail(crow, mA);
      if (!GENERAL && LATE) {
        __builtin_amdgcn_sched_barrier(0);
        window_mul(d);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    row = nrow; uin = nuin;
    advance(nrow, nuin);
    stamp();
  }
  if (MEL || (MAG && mp->mag_kind == 2)) {  // one atomic per wave: running maximum in ordered-int encoding
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(vmax, off); vmax = o > vmax ? o : vmax; }
    if (lane == 0 && p_begin + wave < p_end) {
      const int i = __float_as_int(vmax);
      atomicMax(mp->gmax, i >= 0 ? i : i ^ 0x7fffffff);
    }
    if (MEL && __builtin_amdgcn_ballot_w64(melbad) != 0 && lane == 0) atomicOr(mp->gmax + 1, 1);   // k_mel_pass2: everything becomes NaN
  }
}

template <int K, int MODE, bool GENERAL, bool SCALE, int W, int J = 2, bool NPRED = false, int STG = 0, int ST = 1>
__global__ __launch_bounds__(64 * W) void k_stft_wave(WaveArgs a) {
  stft_wave_body<K, MODE, GENERAL, SCALE, W, J, NPRED, kSinkSpectrum, STG, ST>(a, nullptr);
}
template <int K, int MODE, bool GENERAL, bool SCALE, int W, int J = 2, bool NPRED = false, int STG = 0>
__global__ __launch_bounds__(64 * W) void k_stft_mel_wave(MelWaveArgs m) {
  stft_wave_body<K, MODE, GENERAL, SCALE, W, J, NPRED, kSinkMel, STG>(m.w, &m);
}
template <int K, int MODE, bool GENERAL, bool SCALE, int W, int J = 2, bool NPRED = false, int STG = 0>
__global__ __launch_bounds__(64 * W) void k_stft_mag_wave(MelWaveArgs m) {
  stft_wave_body<K, MODE, GENERAL, SCALE, W, J, NPRED, kSinkMag, STG>(m.w, &m);
}

// dBFS of a magnitude spectrogram (guides/spectrogram.livemd:88-90): 20 * log(|s| / max|s|) / log(10), f32 steps
static unsigned mag_db_blocks(const Ctx* c, int64_t n) {
  const int64_t want = ((n >> 2) + 255) / 256, cap = (int64_t)c->num_cus * 32;
  return (unsigned)(want < 1 ? 1 : (want > cap ? cap : want));
}
__device__ __forceinline__ float mag_to_db(float v, float mx) {
  const float l = logf(v / mx);  // <= 1 ulp from the correctly rounded double log the reference takes
  return (20.0f * l) / 2.3025851f;
}
// in place, 16 bytes per lane (the buffer is a whole number of rows of fft_length / 2 floats; n4 = n / 4, the tail is scalar)
static __global__ __launch_bounds__(256) void k_mag_db_pass2(float* __restrict__ out, int64_t n, const int* __restrict__ gmax) {
  const int g = *gmax;
  const float mx = __int_as_float(g >= 0 ? g : g ^ 0x7fffffff);
  const int64_t n4 = n >> 2;
  v4f* o4 = reinterpret_cast<v4f*>(out);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    v4f v = o4[i];
    v = v4f{mag_to_db(v.x, mx), mag_to_db(v.y, mx), mag_to_db(v.z, mx), mag_to_db(v.w, mx)};
    __builtin_nontemporal_store(v, (gv4f*)(o4 + i));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const int64_t i = (n4 << 2) + threadIdx.x; out[i] = mag_to_db(out[i], mx); }
}

// ============================================================================================ Bluestein on the wave core
// Non-power-of-two fft_length Kb <= C/2 (e.g. 400-point frames of 25 ms speech at 16 kHz): the chirp-z identity
//   X[k] = c[k] * sum_n (u[n] c[n]) conj(c)[k - n],   c[n] = exp(-i pi n^2 / Kb)
// turns the Kb-point DFT into one circular convolution of length C, which is exactly the overlap-save FIR chain of this
// file: transposed core -> x Bf (spectrum of the conj-chirp kernel, / C) -> inverse core, with no transposition pass.
// The DFT is linear over C, so TWO real frames ride through it as u = (frame A) + i (frame B) and are separated
// afterwards with the Hermitian partner U[(Kb - k) mod Kb], fetched through the wave's own exchange buffer.
struct BlueWaveArgs {
  WaveArgs w;            // framing, window (f32[Kb], zero beyond N), forward tables, output
  int32_t Kb;            // fft_length
  const v2f* chirp;      // c64[Kb]
  const v2f* Bf;         // c64[C], pre-scaled by 1/C
  const v2f* twBi;
  const v2f* twCi;
  // sinks other than the complex spectrum (same fields as MelWaveArgs)
  int32_t mel_bins, nnz;
  const float* csr_w;
  const int* csr_off;
  const int* csr_lo;
  float* out;
  int* gmax;
  int32_t mag_kind;
};

// SINK: kSinkSpectrum (c64 rows of Kb bins), kSinkMel (log-mel of the bins below Kb / 2), kSinkMag (|X| / |X|^2 of them)
template <int C, bool SCALE, int W, int SINK = kSinkSpectrum>
__global__ __launch_bounds__(64 * W) void k_stft_blue_wave(BlueWaveArgs b) {
  const WaveArgs& a = b.w;
  constexpr bool MEL = SINK == kSinkMel, MAG = SINK == kSinkMag;
  constexpr int P = C / 64;
  constexpr int R3 = C / 256;
  constexpr int NQ = C / 128;
  constexpr int XCH = C + C / 16 + 16;
  v2f* s_twB = reinterpret_cast<v2f*>(g_wave_smem);
  v2f* s_twC = s_twB + 256;
  v2f* s_twBi = s_twC + R3 * 256;
  v2f* s_twCi = s_twBi + 256;
  v2f* s_Bf = s_twCi + R3 * 256;
  v2f* s_ch = s_Bf + C;                                   // [C/2]
  float* s_w = reinterpret_cast<float*>(s_ch + C / 2);    // [C/2]
  v2f* s_x = reinterpret_cast<v2f*>(s_w + C / 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Kb = b.Kb;
  for (int i = tid; i < 256; i += 64 * W) { s_twB[i] = a.twB[i]; s_twBi[i] = b.twBi[i]; }
  for (int i = tid; i < R3 * 256; i += 64 * W) { s_twC[i] = a.twC[i]; s_twCi[i] = b.twCi[i]; }
  for (int i = tid; i < C; i += 64 * W) s_Bf[i] = b.Bf[i];
  for (int i = tid; i < Kb; i += 64 * W) { s_ch[i] = b.chirp[i]; s_w[i] = a.wtab[i]; }
  float* s_csr = reinterpret_cast<float*>(s_x + W * XCH);
  int* s_off = reinterpret_cast<int*>(s_csr + (MEL ? b.nnz : 0));
  int* s_lo = s_off + (MEL ? b.mel_bins + 1 : 0);
  if (MEL) {
    for (int i = tid; i < b.nnz; i += 64 * W) s_csr[i] = b.csr_w[i];
    for (int i = tid; i <= b.mel_bins; i += 64 * W) s_off[i] = b.csr_off[i];
    for (int i = tid; i < b.mel_bins; i += 64 * W) s_lo[i] = b.csr_lo[i];
  }
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  const int half = Kb / 2;
  float* mags = reinterpret_cast<float*>(xb + C / 2);  // MEL: |XA|^2 at [k], |XB|^2 at [half + k]; U lives in xb[0 .. Kb)
  float vmax = -3.0e38f;
  bool melbad = false;   // log-mel: a non-finite |z|^2 poisons the whole tensor (see stft_wave_body)

  const int64_t p_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t p_end = p_begin + a.chunk;
  if (p_end > a.total_pairs) p_end = a.total_pairs;
  const int nuse = a.N < Kb ? a.N : Kb;
  for (int64_t pr = p_begin + wave; pr < p_end; pr += W) {
    const int64_t row = pr / a.pairs_per_row;
    const int64_t pin = pr - row * a.pairs_per_row;
    const int64_t mA = 2 * pin, mB = mA + 1;
    const bool haveB = mB < a.M;
    const float* xr = a.x + (size_t)row * a.batch_stride;
    v2f zz[2][NQ];
    // One pass = frames (m0, m0 + 1 if two) as re / im of one transform pair.  The paired route is (mA, haveB).  The reference transforms
    // every frame alone (lib/nx_signal.ex:94-102): a pair whose windowed samples are not all finite is redone as (mA, alone) and
    // (mB, alone) — the SAME code with the second frame absent: the imaginary slot is zero and only slot 0 is stored (round 5: the
    // earlier `sel` selector inside load and sink doubled the kernel's registers: 234 -> 121 at C = 1024, 556 B of scratch -> 0 at 2048)
    int64_t m0 = mA;
    bool two = haveB;
    int stage = 0;   // 0: paired, 1: frame A alone, 2: frame B alone
#pragma nounroll
    for (;;) {
      const int64_t qA = m0 * a.hop, qB = qA + a.hop;
      // every sample of the pass's frames inside the signal: plain loads; otherwise per-sample padding / mirror math
      const bool inside = qA - a.lo >= 0 && (two ? qB : qA) - a.lo + nuse <= a.L;
      v2f sum = v2f{0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int n = 2 * lane + e + 128 * q;
          v2f v = v2f{0.f, 0.f};
          if (128 * q < nuse && n < nuse) {
            float va, vb;
            if (inside) { va = xr[qA - a.lo + n]; vb = two ? xr[qB - a.lo + n] : 0.0f; }
            else { va = fetch_any(xr, a, qA + n); vb = two ? fetch_any(xr, a, qB + n) : 0.0f; }
            const float w = s_w[n];
            const v2f u = v2f{va * w, vb * w};
            sum += u;
            v = wcmul(u, s_ch[n]);   // windowed samples (exact f32 products, :101) times the chirp
          }
          zz[e][q] = v;
        }
      const bool nfl = ((__float_as_uint(sum.x) & 0x7f800000u) == 0x7f800000u) || ((__float_as_uint(sum.y) & 0x7f800000u) == 0x7f800000u);
      const bool unit_nf = __builtin_amdgcn_ballot_w64(nfl) != 0;
      if (stage == 0 && unit_nf) {
        if (MEL) melbad = true;                                   // log-mel: the whole tensor is poisoned instead (see stft_wave_body)
        else if (haveB) { stage = 1; two = false; continue; }     // leave the paired route
      }
      v2f d[P];
      wave_fft_core_T<C>(zz, d, xb, s_twB, s_twC, lane);
#pragma unroll
      for (int s = 0; s < P; ++s) d[s] = wcmul(d[s], s_Bf[lane + 64 * s]);
      v2f y[2][NQ];
      wave_fft_core<C, true>(d, y, xb, s_twBi, s_twCi, lane);
      // U[k] = y[k] c[k] = XA[k] + i XB[k], k < Kb: park it in the exchange buffer for the partner read
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int k = 2 * lane + e + 128 * q;
          if (128 * q < Kb && k < Kb) { y[e][q] = wcmul(y[e][q], s_ch[k]); xb[k] = y[e][q]; }
        }
      wave_lds_fence();
      v2f* zA = a.z + ((size_t)row * a.M + m0) * Kb;
      v2f* zB = zA + Kb;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int k = 2 * lane + e + 128 * q;
          if (128 * q < Kb && k < Kb) {
            const v2f u = y[e][q];
            const v2f p = xb[k == 0 ? 0 : Kb - k];
            v2f xa = fft_eps0(v2f{u.x + p.x, u.y - p.y} * 0.5f);
            v2f xv = fft_eps0(v2f{u.y + p.y, p.x - u.x} * 0.5f);
            if (SCALE) { xa = xa / a.div; xv = xv / a.div; }
            if (SINK == kSinkSpectrum) {
              __builtin_nontemporal_store(xa, (gv2f*)(zA + k));
              if (two) __builtin_nontemporal_store(xv, (gv2f*)(zB + k));
            } else if (k < half) {
              const float pa = xa.x * xa.x + xa.y * xa.y, pb = xv.x * xv.x + xv.y * xv.y;
              if (MEL) { mags[k] = pa; mags[half + k] = pb; }
              else {
                const float va = b.mag_kind == 1 ? pa : __builtin_sqrtf(pa), vb = b.mag_kind == 1 ? pb : __builtin_sqrtf(pb);
                float* o = b.out + ((size_t)row * a.M + m0) * half + k;
                o[0] = va; vmax = va > vmax ? va : vmax;
                if (two) { o[half] = vb; vmax = vb > vmax ? vb : vmax; }
              }
            }
          }
        }
      if (stage != 1) break;
      wave_lds_fence();   // the partner reads of this pass are done before the next pass parks its spectrum
      stage = 2; m0 = mB;
    }
    if (MEL) {
      wave_lds_fence();
      float* o0p = b.out + ((size_t)row * a.M + mA) * b.mel_bins;
      const int rot = b.mel_bins & 63;   // the partial pass takes the narrowest bands (see stft_wave_body)
      for (int mb0 = lane; mb0 < b.mel_bins; mb0 += 64) {
        const int mb = mb0 + rot < b.mel_bins ? mb0 + rot : mb0 + rot - b.mel_bins;
        const int o0 = s_off[mb], o1 = s_off[mb + 1], k0 = s_lo[mb];
        float accA = 0.0f, accB = 0.0f;
        for (int j = o0; j < o1; ++j) {
          const float wv = s_csr[j];
          accA = fmaf(mags[k0 + (j - o0)], wv, accA);
          accB = fmaf(mags[half + k0 + (j - o0)], wv, accB);
        }
        melbad |= !(accA < INFINITY) || (haveB && !(accB < INFINITY));
        accA = accA > 1.0e-10f ? accA : 1.0e-10f;
        accB = accB > 1.0e-10f ? accB : 1.0e-10f;
        const float vA = __log2f(accA) * 0.30102999566398120f, vB = __log2f(accB) * 0.30102999566398120f;
        o0p[mb] = vA;
        vmax = vA > vmax ? vA : vmax;
        if (haveB) { o0p[b.mel_bins + mb] = vB; vmax = vB > vmax ? vB : vmax; }
      }
    }
    wave_lds_fence();  // partner reads complete before the next unit's transposed pass writes the buffer
  }
  if (MEL || (MAG && b.mag_kind == 2)) {  // one atomic per wave: running maximum in ordered-int encoding
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(vmax, off); vmax = o > vmax ? o : vmax; }
    if (lane == 0 && p_begin + wave < p_end) {
      const int i = __float_as_int(vmax);
      atomicMax(b.gmax, i >= 0 ? i : i ^ 0x7fffffff);
    }
    if (MEL && __builtin_amdgcn_ballot_w64(melbad) != 0 && lane == 0) atomicOr(b.gmax + 1, 1);   // see stft_wave_body
  }
}

// ============================================================================================ host side (shared by the wave translation units)
static int ensure_wave_tables(Ctx* c, const int C) {
  const int R3 = C / 256;
  const double two_pi = 6.283185307179586476925286766559;
  Ctx::WaveTables& wt = c->wave_tables[C];
  if (wt.twB) return NXSIG_OK;
  std::vector<float2> twB(256), twC((size_t)R3 * 256), twR((size_t)C);
  for (int t = 0; t < 16; ++t)
    for (int k = 0; k < 16; ++k) {
      const double ang = -two_pi * (double)(t * k) / 256.0;
      twB[t * 16 + k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
  for (int t = 0; t < R3; ++t)
    for (int i = 0; i < 256; ++i) {
      const double ang = -two_pi * (double)(t * i) / (double)C;
      twC[(size_t)t * 256 + i] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
  for (int k = 0; k < C; ++k) {
    const double ang = -two_pi * (double)k / (double)(2 * C);
    twR[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
  }
  int rc = ctx_table(c, 0x7742ull, twB.data(), twB.size() * sizeof(float2), &wt.twB);
  if (rc) return rc;
  rc = ctx_table(c, 0x7743ull ^ (uint64_t)C, twC.data(), twC.size() * sizeof(float2), &wt.twC);
  if (rc) { wt.twB = nullptr; return rc; }
  rc = ctx_table(c, 0x7744ull ^ (uint64_t)C, twR.data(), twR.size() * sizeof(float2), &wt.twI);
  if (rc) { wt.twB = nullptr; return rc; }
  for (int ji = 0; ji < 3 && C == 1024; ++ji) {  // quad front-ends J = 2, 4, 8: [j-1][k0] = conj(w_C^(j k0)), k0 < C/J
    const int Jv = 2 << ji, KO = C / Jv;
    std::vector<float2> twQ((size_t)(Jv - 1) * KO);
    for (int j = 1; j < Jv; ++j)
      for (int k0 = 0; k0 < KO; ++k0) {
        const double ang = two_pi * (double)((int64_t)j * k0 % C) / (double)C;
        twQ[(size_t)(j - 1) * KO + k0] = make_float2((float)std::cos(ang), (float)std::sin(ang));
      }
    rc = ctx_table(c, 0x7747ull ^ ((uint64_t)C << 8) ^ (uint64_t)Jv, twQ.data(), twQ.size() * sizeof(float2), &wt.twQ[ji]);
    if (rc) { wt.twB = nullptr; return rc; }
  }
  for (auto& t : twB) t.y = -t.y;  // conjugated copies for the inverse transform
  for (auto& t : twC) t.y = -t.y;
  rc = ctx_table(c, 0x7745ull, twB.data(), twB.size() * sizeof(float2), &wt.twBi);
  if (rc) { wt.twB = nullptr; return rc; }
  rc = ctx_table(c, 0x7746ull ^ (uint64_t)C, twC.data(), twC.size() * sizeof(float2), &wt.twCi);
  if (rc) { wt.twB = nullptr; return rc; }
  return NXSIG_OK;
}

static int ensure_wave_tables_1024(Ctx* c) { return ensure_wave_tables(c, 1024); }


// C = complex core size (1024 here), MODE = front-end, W = waves per workgroup
int launch_mel_finish(Ctx* c, float* out, int64_t n, int* gmax);
int launch_mel_init(Ctx* c, int** gmax);

struct MelLaunch {
  int mel_bins;
  const float* filters_host;  // [mel_bins][fft_length]
  float* out;                 // device f32[batch][M][mel_bins]  (magnitude sink: f32[batch][M][fft_length / 2])
  bool* handled;              // set once the CSR fits LDS and the launch is committed
  int mag_kind = -1;          // >= 0: magnitude sink (0 |X|, 1 |X|^2, 2 dBFS) instead of the mel filterbank
};

// SINK selects the kernel family this translation unit instantiates (spectrum / log-mel / magnitude: one TU each)
template <int C, int MODE, int W, int J = 2, int SINK = kSinkSpectrum>
static int launch_wave(Ctx* c, const StftLaunch& s, const MelLaunch* mel = nullptr, const int64_t small_chunk = 0) {
  constexpr int R3 = C / 256;
  constexpr int XCH = C + C / 16 + 16;
  constexpr int KOUT = MODE == kModeReal2x ? 2 * C : (MODE == kModeQuad ? C / J : C);
  constexpr int TWQ = (J - 1) * (C / J);
  WaveArgs a;
  a.x = s.x; a.batch_stride = s.batch_stride; a.L = s.fr.L; a.lo = s.fr.lo; a.M = s.fr.M;
  a.N = s.fr.N; a.hop = s.fr.hop; a.reflect = s.fr.reflect; a.batch = s.batch;
  a.pairs_per_row = MODE == kModePair ? (s.fr.M + 1) / 2 : (MODE == kModeQuad ? (s.fr.M + 2 * J - 1) / (2 * J) : s.fr.M);
  a.total_pairs = a.pairs_per_row * s.batch;
  a.div = s.inv_scale_div; a.has_scale = s.has_scale; a.z = reinterpret_cast<v2f*>(s.z);

  static_assert(C == 1024 || C == 2048, "wave_fft_core covers 1024 (16*16*4) and 2048 (16*16*8, two butterflies per lane)");
  { int rc = ensure_wave_tables(c, C); if (rc) return rc; }
  Ctx::WaveTables& wt = c->wave_tables[C];
  a.twB = reinterpret_cast<const v2f*>(wt.twB);
  a.twC = reinterpret_cast<const v2f*>(wt.twC);
  a.twR = reinterpret_cast<const v2f*>(MODE == kModeQuad ? wt.twQ[J == 2 ? 0 : (J == 4 ? 1 : 2)] : wt.twI);
  a.wtab = s.window_padK;
  void* dummy = nullptr;
  { int rc2 = ctx_scratch(c, 3, (size_t)8192 * sizeof(float2), &dummy); if (rc2) return rc2; }
  a.dummy = reinterpret_cast<v2f*>(dummy);

  size_t lds = (size_t)KOUT * 4 + 256 * 8 + (size_t)R3 * 256 * 8 +
               (MODE == kModeReal2x ? (size_t)C * 8 : (MODE == kModeQuad ? (size_t)TWQ * 8 : 0)) +
               (size_t)W * XCH * 8;
  MelWaveArgs m;
  if (mel && mel->mag_kind >= 0) {
    *mel->handled = true;
    m.mel_bins = 0; m.nnz = 0; m.csr_w = nullptr; m.csr_off = nullptr; m.csr_lo = nullptr; m.ln10 = 0.f;
    m.out = mel->out; m.mag_kind = mel->mag_kind;
    int rcm;
    if ((rcm = launch_mel_init(c, &m.gmax))) return rcm;
    a.z = nullptr;
  } else if (mel) {  // CSR of the triangular filter rows restricted to bins < fft_length / 2
    m.mag_kind = -1;
    std::vector<float> cw;
    std::vector<int> off(mel->mel_bins + 1, 0), lo(mel->mel_bins, 0);
    const int half = KOUT / 2;
    for (int b = 0; b < mel->mel_bins; ++b) {
      const float* fr = mel->filters_host + (size_t)b * KOUT;
      int l = half, h = 0;
      for (int k = 0; k < half; ++k)
        if (fr[k] != 0.0f) { if (k < l) l = k; h = k + 1; }
      if (h <= l) { l = 0; h = 0; }
      lo[b] = l;
      for (int k = l; k < h; ++k) cw.push_back(fr[k]);
      off[b + 1] = (int)cw.size();
    }
    if (cw.empty()) cw.push_back(0.0f);
    if (cw.size() > 6144 || mel->mel_bins > 1024) return NXSIG_OK;  // keep the CSR in LDS; denser banks take the two-step path
    *mel->handled = true;
    const void *dw = nullptr, *doff = nullptr, *dlo = nullptr;
    int rcm;
    if ((rcm = ctx_table(c, 0xC5A1ull, cw.data(), cw.size() * sizeof(float), &dw))) return rcm;
    if ((rcm = ctx_table(c, 0xC5A2ull, off.data(), off.size() * sizeof(int), &doff))) return rcm;
    if ((rcm = ctx_table(c, 0xC5A3ull, lo.data(), lo.size() * sizeof(int), &dlo))) return rcm;
    m.mel_bins = mel->mel_bins; m.nnz = (int)cw.size();
    m.csr_w = reinterpret_cast<const float*>(dw); m.csr_off = reinterpret_cast<const int*>(doff); m.csr_lo = reinterpret_cast<const int*>(dlo);
    m.ln10 = (float)std::log(10.0);
    m.out = mel->out;
    if ((rcm = launch_mel_init(c, &m.gmax))) return rcm;
    lds += (size_t)m.nnz * 4 + (size_t)(2 * mel->mel_bins + 1) * 4;
    a.z = nullptr;
  }
  // Work distribution: each workgroup takes a SHORT contiguous chunk (a few units per wave) and the hardware
  // dispatcher hands chunks out in order.  Many short-lived workgroups balance the load across CUs / XCDs
  // dynamically: measured +12 % over a static equal partition with long-lived workgroups (the slowest CU set
  // the kernel time), at the price of re-loading the 12 KB of tables per workgroup from L2.
  const int units_per_wave = (mel && mel->mag_kind >= 0) ? (MODE == kModePair ? 16 : 8)
                             : mel ? (MODE == kModePair ? 16 : 8)
                                 : tune(c, kT_WAVE_UNITS_PER_WAVE, MODE == kModePair ? 2 : (MODE == kModeQuad ? (J == 8 ? 8 : 4) : 8));  // measured optima, input
                                 // from HBM (round 3, tools/bench_configs.py gen512 / gen256 / gen128 with NXSIG_BENCH_ALT_INPUTS=4: 4 units per
                                 // wave 0.646 / 0.631 of 8 TB/s against 0.591 / 0.611 at round 2's 2 / 3; fft_length 128: 8 -> 0.593 against 0.572)
  const int upw_fill = c->tuning.set[kT_WAVE_UNITS_PER_WAVE] ? units_per_wave : fill_units_per_wave(c, a.total_pairs, W, units_per_wave < 1 ? 1 : units_per_wave);
  a.chunk = (int64_t)W * (upw_fill < 1 ? 1 : upw_fill);
  // a launch so small that its workgroups all fit on the chip at once (three per CU; BASELINE config 2 as written: one 60 s
  // stream, 703 workgroups) is one round of start-up latencies: three pairs per wave amortise them better than two (+1.5 ... 2.6 %
  // in interleaved sweeps, tools/sweep_stft.py with SWEEP_B=1; the steady-state optimum of many rounds stays at two)
  if (MODE == kModePair && !mel && upw_fill == 2 && !c->tuning.set[kT_WAVE_UNITS_PER_WAVE] &&
      (a.total_pairs + 2 * W - 1) / (2 * W) <= (int64_t)c->num_cus * 3)
    a.chunk = (int64_t)W * 3;
  // small_chunk > 0 (launch_stft_wave's one-round geometry): the caller sized the chunk so that every CU holds ONE workgroup of W waves
  if (small_chunk > 0) a.chunk = small_chunk;
  // Interior frames [m_lo, m_hi): every one of the KOUT samples the streaming front-end reads lies inside the signal,
  // whatever the padding mode (window_padding :reflect / :same / explicit only touch the first and last few frames).
  // Interior units go to the branch-free software-pipelined kernel; the edge units to the bounds-checked one.
  constexpr int F = MODE == kModePair ? 2 : (MODE == kModeQuad ? 2 * J : 1);  // frames per unit
  const int64_t hop = s.fr.hop, lo = s.fr.lo, M = s.fr.M;
  int64_t m_lo = lo > 0 ? (lo + hop - 1) / hop : 0;
  int64_t m_hi = (s.fr.L + lo - KOUT >= 0) ? (s.fr.L + lo - KOUT) / hop + 1 : 0;
  if (m_hi > M) m_hi = M;
  if (m_lo > M) m_lo = M;
  if (m_hi < m_lo) m_hi = m_lo;
  int64_t u_lo = (m_lo + F - 1) / F;
  int64_t u_hi = m_hi >= M ? a.pairs_per_row : m_hi / F;  // the last (possibly ragged) unit is interior iff frame M-1 is
  if (u_lo > a.pairs_per_row) u_lo = a.pairs_per_row;
  if (u_hi < u_lo) u_hi = u_lo;
  // quad front-ends: staged input (16-byte loads of the unit's contiguous span, re-distributed through LDS); units must then be
  // complete (no phantom frames) and have 3 floats of slack behind the span — 7 when the spans are not 16-byte aligned (the kernel
  // then starts its loads up to 3 floats early, which stays inside the allocation: an unaligned address is not its first)
  int stg = 0;
  const bool stage_aligned = (reinterpret_cast<uintptr_t>(s.x) & 15) == 0 && (s.batch_stride & 3) == 0 && (lo & 3) == 0;
  if (MODE == kModeQuad && !tune(c, kT_NO_STAGE, 0) && s.fr.hop <= KOUT &&
      (stage_aligned || (((F - 1) * s.fr.hop + KOUT + 3) & ~3) + 4 <= 2048)) {
    const int slack = 4;   // the unpadded kernels load 4 floats more than the span (room for the `mis` floats they may start early)
    const int span4 = (((F - 1) * s.fr.hop + KOUT + 3) & ~3) + 4;   // (+ 4: what the unpadded kernels load, see stft_wave_body)
    stg = (span4 + 255) / 256 <= 4 ? 4 : 8;
    int64_t mh = (s.fr.L + lo - (KOUT + 3 + slack) >= 0) ? (s.fr.L + lo - (KOUT + 3 + slack)) / hop + 1 : 0;
    if (mh > M) mh = M;
    if (mh < m_lo) mh = m_lo;
    u_hi = mh / F;  // complete units only
    if (u_hi < u_lo) u_hi = u_lo;
  }
  if (tune(c, kT_WAVE_NO_SPLIT, 0) && !(u_lo == 0 && u_hi == a.pairs_per_row)) u_hi = u_lo = 0;
  const bool scale = s.has_scale != 0;
  const bool npred = s.fr.N < KOUT;
  const int64_t chunk_main = a.chunk;
  auto go = [&](auto kernel, int64_t upr, int64_t split, int64_t add0, int64_t add1) -> int {
    if (upr == 0) return NXSIG_OK;
    // the edge launch has few, slow (bounds-checked) units: one per wave, so that they all run concurrently
    a.chunk = (split < ((int64_t)1 << 61)) ? W : chunk_main;
    a.units_per_row = upr; a.u_split = split; a.u_add0 = add0; a.u_add1 = add1;
    a.total_pairs = upr * s.batch;
    {  // dispatch record: <sink>.<front-end>[.1r][.edge]
      char tag[48];
      const char* fe = MODE == kModePair ? "pair" : (MODE == kModeReal2x ? (C == 1024 ? "real2x" : "real2x.4k") : (J == 2 ? "quad2" : (J == 4 ? "quad4" : "quad8")));
      std::snprintf(tag, sizeof tag, "%s.%s%s%s", SINK == kSinkSpectrum ? "stft" : (SINK == kSinkMel ? "mel" : "mag"), fe,
                    small_chunk > 0 ? ".1r" : "", (split < ((int64_t)1 << 61)) ? ".edge" : "");
      dispatch_note(tag);
    }
    int64_t blocks = (a.total_pairs + a.chunk - 1) / a.chunk;
    if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "stft: too many frames for one launch");
    if (lds > 64 * 1024)
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if constexpr (std::is_invocable_v<decltype(kernel), MelWaveArgs>) {
      m.w = a;
      hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, m);
    } else {
      hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    }
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  int rc = NXSIG_OK;
  if constexpr (SINK == kSinkMag) {
    const int64_t big = (int64_t)1 << 62;
    {
      const int64_t upr = u_hi - u_lo;
      bool done = false;
      if constexpr (MODE == kModeQuad) {
        if (stg == 4) {
          done = true;
          if (!npred) rc = scale ? go(k_stft_mag_wave<C, MODE, false, true, W, J, false, 4>, upr, big, u_lo, u_lo)
                                 : go(k_stft_mag_wave<C, MODE, false, false, W, J, false, 4>, upr, big, u_lo, u_lo);
          else rc = scale ? go(k_stft_mag_wave<C, MODE, false, true, W, J, true, 4>, upr, big, u_lo, u_lo)
                          : go(k_stft_mag_wave<C, MODE, false, false, W, J, true, 4>, upr, big, u_lo, u_lo);
        } else if (stg == 8) {
          done = true;
          if (!npred) rc = scale ? go(k_stft_mag_wave<C, MODE, false, true, W, J, false, 8>, upr, big, u_lo, u_lo)
                                 : go(k_stft_mag_wave<C, MODE, false, false, W, J, false, 8>, upr, big, u_lo, u_lo);
          else rc = scale ? go(k_stft_mag_wave<C, MODE, false, true, W, J, true, 8>, upr, big, u_lo, u_lo)
                          : go(k_stft_mag_wave<C, MODE, false, false, W, J, true, 8>, upr, big, u_lo, u_lo);
        }
      }
      if (!done) {
        if (!npred) rc = scale ? go(k_stft_mag_wave<C, MODE, false, true, W, J, false>, upr, big, u_lo, u_lo)
                               : go(k_stft_mag_wave<C, MODE, false, false, W, J, false>, upr, big, u_lo, u_lo);
        else rc = scale ? go(k_stft_mag_wave<C, MODE, false, true, W, J, true>, upr, big, u_lo, u_lo)
                        : go(k_stft_mag_wave<C, MODE, false, false, W, J, true>, upr, big, u_lo, u_lo);
      }
      if (rc) return rc;
    }
    {
      const int64_t upr = u_lo + (a.pairs_per_row - u_hi);
      rc = scale ? go(k_stft_mag_wave<C, MODE, true, true, W, J>, upr, u_lo, 0, u_hi - u_lo)
                 : go(k_stft_mag_wave<C, MODE, true, false, W, J>, upr, u_lo, 0, u_hi - u_lo);
      if (rc) return rc;
    }
    if (mel->mag_kind == 2) {
      const int64_t n = (int64_t)s.batch * s.fr.M * (KOUT / 2);
      hipLaunchKernelGGL(k_mag_db_pass2, dim3(mag_db_blocks(c, n)), dim3(256), 0, c->stream, mel->out, n, m.gmax);
      NXSIG_HIP_TRY(hipGetLastError());
    }
    return NXSIG_OK;
  }
  if constexpr (SINK == kSinkMel) {
    const int64_t big = (int64_t)1 << 62;
    {
      const int64_t upr = u_hi - u_lo;
      bool done = false;
      if constexpr (MODE == kModeQuad) {
        if (stg == 4) {
          done = true;
          if (!npred) rc = scale ? go(k_stft_mel_wave<C, MODE, false, true, W, J, false, 4>, upr, big, u_lo, u_lo)
                                 : go(k_stft_mel_wave<C, MODE, false, false, W, J, false, 4>, upr, big, u_lo, u_lo);
          else rc = scale ? go(k_stft_mel_wave<C, MODE, false, true, W, J, true, 4>, upr, big, u_lo, u_lo)
                          : go(k_stft_mel_wave<C, MODE, false, false, W, J, true, 4>, upr, big, u_lo, u_lo);
        } else if (stg == 8) {
          done = true;
          if (!npred) rc = scale ? go(k_stft_mel_wave<C, MODE, false, true, W, J, false, 8>, upr, big, u_lo, u_lo)
                                 : go(k_stft_mel_wave<C, MODE, false, false, W, J, false, 8>, upr, big, u_lo, u_lo);
          else rc = scale ? go(k_stft_mel_wave<C, MODE, false, true, W, J, true, 8>, upr, big, u_lo, u_lo)
                          : go(k_stft_mel_wave<C, MODE, false, false, W, J, true, 8>, upr, big, u_lo, u_lo);
        }
      }
      if (!done) {
        if (!npred) rc = scale ? go(k_stft_mel_wave<C, MODE, false, true, W, J, false>, upr, big, u_lo, u_lo)
                               : go(k_stft_mel_wave<C, MODE, false, false, W, J, false>, upr, big, u_lo, u_lo);
        else rc = scale ? go(k_stft_mel_wave<C, MODE, false, true, W, J, true>, upr, big, u_lo, u_lo)
                        : go(k_stft_mel_wave<C, MODE, false, false, W, J, true>, upr, big, u_lo, u_lo);
      }
      if (rc) return rc;
    }
    {
      const int64_t upr = u_lo + (a.pairs_per_row - u_hi);
      rc = scale ? go(k_stft_mel_wave<C, MODE, true, true, W, J>, upr, u_lo, 0, u_hi - u_lo)
                 : go(k_stft_mel_wave<C, MODE, true, false, W, J>, upr, u_lo, 0, u_hi - u_lo);
      if (rc) return rc;
    }
    return launch_mel_finish(c, mel->out, (int64_t)s.batch * s.fr.M * mel->mel_bins, m.gmax);
  }
  if constexpr (SINK == kSinkSpectrum) {
  {  // interior units u_lo .. u_hi-1
    const int64_t upr = u_hi - u_lo, big = (int64_t)1 << 62;
    bool done = false;
    if constexpr (MODE == kModeQuad) {
      if (stg == 4) {
        done = true;
        // hop == fft_length / 4 (the default 75 % overlap): the instantiation whose parked span is padded against bank conflicts
        if (!npred && 4 * s.fr.hop == KOUT && stage_aligned && tune(c, kT_STAGE_PAD, 1))
          rc = scale ? go(k_stft_wave<C, MODE, false, true, W, J, false, 5>, upr, big, u_lo, u_lo)
                     : go(k_stft_wave<C, MODE, false, false, W, J, false, 5>, upr, big, u_lo, u_lo);
        else if (!npred) rc = scale ? go(k_stft_wave<C, MODE, false, true, W, J, false, 4>, upr, big, u_lo, u_lo)
                               : go(k_stft_wave<C, MODE, false, false, W, J, false, 4>, upr, big, u_lo, u_lo);
        else rc = scale ? go(k_stft_wave<C, MODE, false, true, W, J, true, 4>, upr, big, u_lo, u_lo)
                        : go(k_stft_wave<C, MODE, false, false, W, J, true, 4>, upr, big, u_lo, u_lo);
      } else if (stg == 8) {
        done = true;
        if (!npred) rc = scale ? go(k_stft_wave<C, MODE, false, true, W, J, false, 8>, upr, big, u_lo, u_lo)
                               : go(k_stft_wave<C, MODE, false, false, W, J, false, 8>, upr, big, u_lo, u_lo);
        else rc = scale ? go(k_stft_wave<C, MODE, false, true, W, J, true, 8>, upr, big, u_lo, u_lo)
                        : go(k_stft_wave<C, MODE, false, false, W, J, true, 8>, upr, big, u_lo, u_lo);
      }
    }
    if constexpr (MODE == kModeReal2x) {
      // even hop / padding / row stride and an 8-byte aligned signal: the frame's samples travel as 8-byte loads
      if (!done && !npred && !tune(c, kT_NO_AL8, 0) && (reinterpret_cast<uintptr_t>(s.x) & 7) == 0 && (s.batch_stride & 1) == 0 &&
          (hop & 1) == 0 && (lo & 1) == 0) {
        done = true;
        rc = scale ? go(k_stft_wave<C, MODE, false, true, W, J, false, 2>, upr, big, u_lo, u_lo)
                   : go(k_stft_wave<C, MODE, false, false, W, J, false, 2>, upr, big, u_lo, u_lo);
      }
    }
    if constexpr (MODE == kModePair && C == 1024) {
      // Store cache policy by the size of the result (round 3, tools/sweep_stft.py NXSIG_STORE_POLICY 0 1 with SWEEP_B = 1 ... 96
      // streams of 60 s): a spectrum of 0.15 ... 2.2 GB leaves faster through plain non-temporal stores (3 streams: +10 ... 14 %,
      // 4: +14 ... 18 %, 8: +8 %, 12 / 16: +5 %), from 24 streams on the write-through "sc1 nt" form wins as measured in round 2
      // (32: +1 ... 5 %, flat to 96 streams = 8.8 GB); one stream (92 MB, Infinity-Cache resident) shows no preference.
      // NXSIG_STORE_POLICY forces one (0 / 1 / 2).
      const int64_t out_bytes = (int64_t)s.batch * s.fr.M * KOUT * 8;
      const int stp = tune(c, kT_STORE_POLICY, (out_bytes > ((int64_t)150 << 20) && out_bytes <= ((int64_t)2200 << 20)) ? 0 : 1);
      if (!done && !npred && (stp == 0 || stp == 2)) {
        done = true;
        if (stp == 0) rc = scale ? go(k_stft_wave<C, MODE, false, true, W, J, false, 0, 0>, upr, big, u_lo, u_lo)
                                 : go(k_stft_wave<C, MODE, false, false, W, J, false, 0, 0>, upr, big, u_lo, u_lo);
        else rc = scale ? go(k_stft_wave<C, MODE, false, true, W, J, false, 0, 2>, upr, big, u_lo, u_lo)
                        : go(k_stft_wave<C, MODE, false, false, W, J, false, 0, 2>, upr, big, u_lo, u_lo);
      }
    }
    if (!done) {
      if (!npred) rc = scale ? go(k_stft_wave<C, MODE, false, true, W, J, false>, upr, big, u_lo, u_lo)
                             : go(k_stft_wave<C, MODE, false, false, W, J, false>, upr, big, u_lo, u_lo);
      else rc = scale ? go(k_stft_wave<C, MODE, false, true, W, J, true>, upr, big, u_lo, u_lo)
                      : go(k_stft_wave<C, MODE, false, false, W, J, true>, upr, big, u_lo, u_lo);
    }
    if (rc) return rc;
  }
  {  // edge units 0 .. u_lo-1 and u_hi .. pairs_per_row-1
    const int64_t upr = u_lo + (a.pairs_per_row - u_hi);
    rc = scale ? go(k_stft_wave<C, MODE, true, true, W, J>, upr, u_lo, 0, u_hi - u_lo)
               : go(k_stft_wave<C, MODE, true, false, W, J>, upr, u_lo, 0, u_hi - u_lo);
  }
  return rc;
  }
  return NXSIG_OK;
}

int blue_tables_dev(Ctx* c, int K, int P, const float2** chirp, const float2** Bf);  // kernels_generic.hip

template <int C, int SINK = kSinkSpectrum>
static int launch_blue_wave(Ctx* c, const StftLaunch& s, const MelLaunch* mel = nullptr) {
  constexpr int W = 4, R3 = C / 256, XCH = C + C / 16 + 16;
  BlueWaveArgs b;
  b.mel_bins = 0; b.nnz = 0; b.csr_w = nullptr; b.csr_off = nullptr; b.csr_lo = nullptr; b.out = nullptr; b.gmax = nullptr;
  b.mag_kind = -1;
  WaveArgs& a = b.w;
  a.x = s.x; a.batch_stride = s.batch_stride; a.L = s.fr.L; a.lo = s.fr.lo; a.M = s.fr.M;
  a.N = s.fr.N; a.hop = s.fr.hop; a.reflect = s.fr.reflect; a.batch = s.batch;
  a.pairs_per_row = (s.fr.M + 1) / 2;
  a.total_pairs = a.pairs_per_row * s.batch;
  a.div = s.inv_scale_div; a.has_scale = s.has_scale; a.z = reinterpret_cast<v2f*>(s.z);
  a.twR = nullptr; a.dummy = nullptr;
  { int rc = ensure_wave_tables(c, C); if (rc) return rc; }
  Ctx::WaveTables& wt = c->wave_tables[C];
  a.twB = reinterpret_cast<const v2f*>(wt.twB);
  a.twC = reinterpret_cast<const v2f*>(wt.twC);
  b.twBi = reinterpret_cast<const v2f*>(wt.twBi);
  b.twCi = reinterpret_cast<const v2f*>(wt.twCi);
  a.wtab = s.window_padK;
  b.Kb = s.K;
  const float2 *dc = nullptr, *db = nullptr;
  { int rc = blue_tables_dev(c, s.K, C, &dc, &db); if (rc) return rc; }
  b.chirp = reinterpret_cast<const v2f*>(dc);
  b.Bf = reinterpret_cast<const v2f*>(db);
  size_t lds = (size_t)(2 * 256 + 2 * R3 * 256 + C + C / 2) * 8 + (size_t)(C / 2) * 4 + (size_t)W * XCH * 8;
  int sink = kSinkSpectrum;
  if (mel && mel->mag_kind >= 0) {
    sink = kSinkMag;
    *mel->handled = true;
    b.out = mel->out; b.mag_kind = mel->mag_kind;
    int rcm = launch_mel_init(c, &b.gmax);
    if (rcm) return rcm;
  } else if (mel) {  // CSR of the triangular filter rows restricted to bins < fft_length / 2
    sink = kSinkMel;
    std::vector<float> cw;
    std::vector<int> off(mel->mel_bins + 1, 0), lo(mel->mel_bins, 0);
    const int half = s.K / 2;
    for (int mb = 0; mb < mel->mel_bins; ++mb) {
      const float* fr = mel->filters_host + (size_t)mb * s.K;
      int l = half, h = 0;
      for (int k = 0; k < half; ++k)
        if (fr[k] != 0.0f) { if (k < l) l = k; h = k + 1; }
      if (h <= l) { l = 0; h = 0; }
      lo[mb] = l;
      for (int k = l; k < h; ++k) cw.push_back(fr[k]);
      off[mb + 1] = (int)cw.size();
    }
    if (cw.empty()) cw.push_back(0.0f);
    if (cw.size() > 6144 || mel->mel_bins > 1024) return NXSIG_OK;  // two-step path
    *mel->handled = true;
    const void *dw = nullptr, *doff = nullptr, *dlo = nullptr;
    int rcm;
    if ((rcm = ctx_table(c, 0xC5A1ull, cw.data(), cw.size() * sizeof(float), &dw))) return rcm;
    if ((rcm = ctx_table(c, 0xC5A2ull, off.data(), off.size() * sizeof(int), &doff))) return rcm;
    if ((rcm = ctx_table(c, 0xC5A3ull, lo.data(), lo.size() * sizeof(int), &dlo))) return rcm;
    b.mel_bins = mel->mel_bins; b.nnz = (int)cw.size();
    b.csr_w = reinterpret_cast<const float*>(dw); b.csr_off = reinterpret_cast<const int*>(doff); b.csr_lo = reinterpret_cast<const int*>(dlo);
    b.out = mel->out;
    if ((rcm = launch_mel_init(c, &b.gmax))) return rcm;
    lds += (size_t)b.nnz * 4 + (size_t)(2 * mel->mel_bins + 1) * 4;
  }
  const int units_per_wave = fill_units_per_wave(c, a.total_pairs, W, 4);
  a.chunk = (int64_t)W * (units_per_wave < 1 ? 1 : units_per_wave);
  const int64_t blocks = (a.total_pairs + a.chunk - 1) / a.chunk;
  if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "stft: too many frames for one launch");
  auto go = [&](auto kernel) -> int {
    if (lds > 64 * 1024)
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note(SINK == kSinkSpectrum ? "stft.blue" : (SINK == kSinkMel ? "mel.blue" : "mag.blue"));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, b);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  if (sink != SINK) return set_error(NXSIG_ERR_INVALID_ARG, "internal error: sink mismatch");
  int rc = s.has_scale ? go(k_stft_blue_wave<C, true, W, SINK>) : go(k_stft_blue_wave<C, false, W, SINK>);
  if (SINK == kSinkSpectrum || rc) return rc;
  if (sink == kSinkMel) return launch_mel_finish(c, mel->out, (int64_t)s.batch * s.fr.M * mel->mel_bins, b.gmax);
  if (mel->mag_kind == 2) {
    const int64_t n = (int64_t)s.batch * s.fr.M * (s.K / 2);
    hipLaunchKernelGGL(k_mag_db_pass2, dim3(mag_db_blocks(c, n)), dim3(256), 0, c->stream, mel->out, n, b.gmax);
    NXSIG_HIP_TRY(hipGetLastError());
  }
  return NXSIG_OK;
}

}  // namespace nxsig
