// Small natural-order DFT codelets on registers (forward transform, e^{-2 pi i ..}) for the two-pass wave kernels of composite
// fft lengths (kernels_wave_r20.hip: 400 = 20 x 20; kernels_wave_rab.hip: 320 = 16 x 20, 480 = 24 x 20, 640 = 32 x 20, 960 = 32 x 30, ...).
// Coprime factors are joined by the prime-factor (Good-Thomas) index maps — n = (N2 n1 + N1 n2) mod N in, k = (N2 (N2^-1 mod N1) k1
// + N1 (N1^-1 mod N2) k2) mod N out, no twiddles in between; dft32 is one radix-2 step over two dft16 (wave_stft.hpp).  Index maps
// checked against numpy in tools/emulate_wave_fft.py-style scripts (round 5) and end to end by the parity tests of the kernels.
#pragma once
#include "wave_stft.hpp"

namespace nxsig {

// 3-point DFT: X1 = m - i s (x1 - x2), X2 = m + i s (x1 - x2), m = x0 - (x1 + x2) / 2, s = sin(2 pi / 3)
__device__ __forceinline__ void dft3(v2f& x0, v2f& x1, v2f& x2) {
  const float s = 0.86602540378443865f;
  const v2f t = x1 + x2, u = (x1 - x2) * s;
  const v2f m = x0 - t * 0.5f;
  x0 = x0 + t;
  x1 = add_mi(m, u);
  x2 = add_pi(m, u);
}

// 5-point DFT
__device__ __forceinline__ void dft5(v2f& x0, v2f& x1, v2f& x2, v2f& x3, v2f& x4) {
  const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f, s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
  const v2f a1 = x1 + x4, a2 = x2 + x3, b1 = x1 - x4, b2 = x2 - x3;
  const v2f t1 = x0 + a1 * c1 + a2 * c2, t2 = x0 + a1 * c2 + a2 * c1;
  const v2f u1 = b1 * s1 + b2 * s2, u2 = b1 * s2 - b2 * s1;
  x0 = x0 + a1 + a2;
  x1 = add_mi(t1, u1); x4 = add_pi(t1, u1);
  x2 = add_mi(t2, u2); x3 = add_pi(t2, u2);
}

// 6-point DFT, prime-factor 2 x 3: n = (3 n1 + 2 n2) mod 6, k = (3 k1 + 4 k2) mod 6
__device__ __forceinline__ void dft6(v2f* v) {
  v2f A0[3], A1[3];
#pragma unroll
  for (int n2 = 0; n2 < 3; ++n2) {
    const v2f a = v[(2 * n2) % 6], b = v[(3 + 2 * n2) % 6];
    A0[n2] = a + b; A1[n2] = a - b;
  }
  dft3(A0[0], A0[1], A0[2]);
  dft3(A1[0], A1[1], A1[2]);
#pragma unroll
  for (int k2 = 0; k2 < 3; ++k2) { v[(4 * k2) % 6] = A0[k2]; v[(3 + 4 * k2) % 6] = A1[k2]; }
}

// 20-point DFT, prime-factor 4 x 5: n = (5 n1 + 4 n2) mod 20, k = (5 k1 + 16 k2) mod 20
__device__ __forceinline__ void dft20(v2f* v) {
  v2f A[4][5];
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2) {
    v2f c0 = v[(4 * n2) % 20], c1 = v[(5 + 4 * n2) % 20], c2 = v[(10 + 4 * n2) % 20], c3 = v[(15 + 4 * n2) % 20];
    dft4<false>(c0, c1, c2, c3);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2; A[3][n2] = c3;
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    dft5(A[k1][0], A[k1][1], A[k1][2], A[k1][3], A[k1][4]);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) v[(5 * k1 + 16 * k2) % 20] = A[k1][k2];
  }
}

// 24-point DFT, prime-factor 3 x 8: n = (8 n1 + 3 n2) mod 24, k = (16 k1 + 9 k2) mod 24
__device__ __forceinline__ void dft24(v2f* v) {
  v2f A[3][8];
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2) {
    v2f c0 = v[(3 * n2) % 24], c1 = v[(8 + 3 * n2) % 24], c2 = v[(16 + 3 * n2) % 24];
    dft3(c0, c1, c2);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2;
  }
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    dft8<false>(A[k1]);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[(16 * k1 + 9 * k2) % 24] = A[k1][k2];
  }
}

// 30-point DFT, prime-factor 5 x 6: n = (6 n1 + 5 n2) mod 30, k = (6 k1 + 25 k2) mod 30
__device__ __forceinline__ void dft30(v2f* v) {
  v2f A[5][6];
#pragma unroll
  for (int n2 = 0; n2 < 6; ++n2) {
    v2f c0 = v[(5 * n2) % 30], c1 = v[(6 + 5 * n2) % 30], c2 = v[(12 + 5 * n2) % 30], c3 = v[(18 + 5 * n2) % 30], c4 = v[(24 + 5 * n2) % 30];
    dft5(c0, c1, c2, c3, c4);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2; A[3][n2] = c3; A[4][n2] = c4;
  }
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) {
    dft6(A[k1]);
#pragma unroll
    for (int k2 = 0; k2 < 6; ++k2) v[(6 * k1 + 25 * k2) % 30] = A[k1][k2];
  }
}

// 32-point DFT: one radix-2 step over the 16-point transforms of the even and the odd samples, X[k] = E[k] + W_32^k O[k]
__device__ __forceinline__ void dft32(v2f* v) {
  constexpr float kC[16] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                            0.38268343236508977f, 0.19509032201612825f, 0.0f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f,
                            -0.70710678118654752f, -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323043f};
  constexpr float kS[16] = {0.0f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f, 0.83146961230254524f,
                            0.92387953251128674f, 0.98078528040323043f, 1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                            0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f};
  v2f e[16], o[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { e[j] = v[2 * j]; o[j] = v[2 * j + 1]; }
  dft16<false>(e);
  dft16<false>(o);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const v2f t = k == 0 ? o[0] : (k == 8 ? rot90<false>(o[8]) : cmulc<false>(o[k], kC[k], -kS[k]));   // W_32^k = (cos, -sin)
    v[k] = e[k] + t;
    v[k + 16] = e[k] - t;
  }
}

// 25-point DFT, Cooley-Tukey 5 x 5: n = 5 n1 + n2, k = k1 + 5 k2, twiddles W_25^(n2 k1) between the two rounds of dft5
__device__ __forceinline__ void dft25(v2f* v) {
  // (cos, sin)(2 pi j / 25) for j = n2 k1, n2, k1 = 1..4
  constexpr float kC[5][5] = {{1.f, 1.f, 1.f, 1.f, 1.f},
                              {1.f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899666f},
                              {1.f, 0.87630668004386358f, 0.53582679497899666f, 0.06279051952931337f, -0.42577929156507272f},
                              {1.f, 0.72896862742141155f, 0.06279051952931337f, -0.63742398974868975f, -0.99211470131447788f},
                              {1.f, 0.53582679497899666f, -0.42577929156507272f, -0.99211470131447788f, -0.63742398974868975f}};
  constexpr float kS[5][5] = {{0.f, 0.f, 0.f, 0.f, 0.f},
                              {0.f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868873f, 0.84432792550201508f},
                              {0.f, 0.48175367410171532f, 0.84432792550201508f, 0.99802672842827156f, 0.90482705246601958f},
                              {0.f, 0.68454710592868873f, 0.99802672842827156f, 0.77051324277578925f, 0.12533323356430426f},
                              {0.f, 0.84432792550201508f, 0.90482705246601958f, 0.12533323356430426f, -0.77051324277578925f}};
  v2f Y[5][5];
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2) {
    v2f c0 = v[n2], c1 = v[5 + n2], c2 = v[10 + n2], c3 = v[15 + n2], c4 = v[20 + n2];
    dft5(c0, c1, c2, c3, c4);
    Y[0][n2] = c0; Y[1][n2] = c1; Y[2][n2] = c2; Y[3][n2] = c3; Y[4][n2] = c4;
  }
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) {
#pragma unroll
    for (int n2 = 1; n2 < 5; ++n2)
      if (k1 > 0) Y[k1][n2] = cmulc<false>(Y[k1][n2], kC[k1][n2], -kS[k1][n2]);
    dft5(Y[k1][0], Y[k1][1], Y[k1][2], Y[k1][3], Y[k1][4]);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) v[k1 + 5 * k2] = Y[k1][k2];
  }
}

// 40-point DFT, prime-factor 5 x 8: n = (8 n1 + 5 n2) mod 40, k = (16 k1 + 25 k2) mod 40
__device__ __forceinline__ void dft40(v2f* v) {
  v2f A[5][8];
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2) {
    v2f c0 = v[(5 * n2) % 40], c1 = v[(8 + 5 * n2) % 40], c2 = v[(16 + 5 * n2) % 40], c3 = v[(24 + 5 * n2) % 40], c4 = v[(32 + 5 * n2) % 40];
    dft5(c0, c1, c2, c3, c4);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2; A[3][n2] = c3; A[4][n2] = c4;
  }
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) {
    dft8<false>(A[k1]);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[(16 * k1 + 25 * k2) % 40] = A[k1][k2];
  }
}

// 10-point DFT, prime-factor 2 x 5: n = (5 n1 + 2 n2) mod 10, k = (5 k1 + 6 k2) mod 10
__device__ __forceinline__ void dft10(v2f* v) {
  v2f A0[5], A1[5];
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2) {
    const v2f a = v[(2 * n2) % 10], b = v[(5 + 2 * n2) % 10];
    A0[n2] = a + b; A1[n2] = a - b;
  }
  dft5(A0[0], A0[1], A0[2], A0[3], A0[4]);
  dft5(A1[0], A1[1], A1[2], A1[3], A1[4]);
#pragma unroll
  for (int k2 = 0; k2 < 5; ++k2) { v[(6 * k2) % 10] = A0[k2]; v[(5 + 6 * k2) % 10] = A1[k2]; }
}

// 12-point DFT, prime-factor 3 x 4: n = (4 n1 + 3 n2) mod 12, k = (4 k1 + 9 k2) mod 12
__device__ __forceinline__ void dft12(v2f* v) {
  v2f A[3][4];
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    v2f c0 = v[(3 * n2) % 12], c1 = v[(4 + 3 * n2) % 12], c2 = v[(8 + 3 * n2) % 12];
    dft3(c0, c1, c2);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2;
  }
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    dft4<false>(A[k1][0], A[k1][1], A[k1][2], A[k1][3]);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) v[(4 * k1 + 9 * k2) % 12] = A[k1][k2];
  }
}

// 15-point DFT, prime-factor 3 x 5: n = (5 n1 + 3 n2) mod 15, k = (10 k1 + 6 k2) mod 15
__device__ __forceinline__ void dft15(v2f* v) {
  v2f A[3][5];
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2) {
    v2f c0 = v[(3 * n2) % 15], c1 = v[(5 + 3 * n2) % 15], c2 = v[(10 + 3 * n2) % 15];
    dft3(c0, c1, c2);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2;
  }
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    dft5(A[k1][0], A[k1][1], A[k1][2], A[k1][3], A[k1][4]);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) v[(10 * k1 + 6 * k2) % 15] = A[k1][k2];
  }
}

// 48-point DFT, prime-factor 3 x 16: n = (16 n1 + 3 n2) mod 48, k = (16 k1 + 33 k2) mod 48
__device__ __forceinline__ void dft48(v2f* v) {
  v2f A[3][16];
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) {
    v2f c0 = v[(3 * n2) % 48], c1 = v[(16 + 3 * n2) % 48], c2 = v[(32 + 3 * n2) % 48];
    dft3(c0, c1, c2);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2;
  }
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    dft16<false>(A[k1]);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) v[(16 * k1 + 33 * k2) % 48] = A[k1][k2];
  }
}

// ---- round 6: radix 7 (the frames of 44.1 kHz audio: 441 = 21 x 21, 882 = 42 x 21, 1764 = 42 x 42) and the 50- / 60- / 64-point codelets of
// 2400 = 48 x 50, 2880 = 48 x 60, 3840 = 60 x 64.  Index maps checked against numpy (tools/check_codelet_maps.py).
// 7-point DFT: pairs a_j = x_j + x_(7-j), b_j = x_j - x_(7-j); X_k = x0 + sum a_j cos(2 pi j k / 7) -+ i sum b_j sin(2 pi j k / 7)
__device__ __forceinline__ void dft7(v2f& x0, v2f& x1, v2f& x2, v2f& x3, v2f& x4, v2f& x5, v2f& x6) {
  const float c1 = 0.62348980185873359f, c2 = -0.22252093395631434f, c3 = -0.90096886790241903f;
  const float s1 = 0.7818314824680298f, s2 = 0.97492791218182362f, s3 = 0.43388373911755823f;
  const v2f a1 = x1 + x6, a2 = x2 + x5, a3 = x3 + x4, b1 = x1 - x6, b2 = x2 - x5, b3 = x3 - x4;
  const v2f t1 = x0 + a1 * c1 + a2 * c2 + a3 * c3, t2 = x0 + a1 * c2 + a2 * c3 + a3 * c1, t3 = x0 + a1 * c3 + a2 * c1 + a3 * c2;
  const v2f u1 = b1 * s1 + b2 * s2 + b3 * s3, u2 = b1 * s2 - b2 * s3 - b3 * s1, u3 = b1 * s3 - b2 * s1 + b3 * s2;
  x0 = x0 + a1 + a2 + a3;
  x1 = add_mi(t1, u1); x6 = add_pi(t1, u1);
  x2 = add_mi(t2, u2); x5 = add_pi(t2, u2);
  x3 = add_mi(t3, u3); x4 = add_pi(t3, u3);
}

// 14-point DFT, prime-factor 2 x 7: n = (7 n1 + 2 n2) mod 14, k = (7 k1 + 8 k2) mod 14
__device__ __forceinline__ void dft14(v2f* v) {
  v2f A[2][7];
#pragma unroll
  for (int n2 = 0; n2 < 7; ++n2) {
    v2f c0 = v[(0 + 2 * n2) % 14], c1 = v[(7 + 2 * n2) % 14];
    { const v2f t0 = c0 + c1, t1 = c0 - c1; c0 = t0; c1 = t1; }
    A[0][n2] = c0; A[1][n2] = c1;
  }
#pragma unroll
  for (int k1 = 0; k1 < 2; ++k1) {
    dft7(A[k1][0], A[k1][1], A[k1][2], A[k1][3], A[k1][4], A[k1][5], A[k1][6]);
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) v[(7 * k1 + 8 * k2) % 14] = A[k1][k2];
  }
}

// 21-point DFT, prime-factor 3 x 7: n = (7 n1 + 3 n2) mod 21, k = (7 k1 + 15 k2) mod 21
__device__ __forceinline__ void dft21(v2f* v) {
  v2f A[3][7];
#pragma unroll
  for (int n2 = 0; n2 < 7; ++n2) {
    v2f c0 = v[(0 + 3 * n2) % 21], c1 = v[(7 + 3 * n2) % 21], c2 = v[(14 + 3 * n2) % 21];
    dft3(c0, c1, c2);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2;
  }
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    dft7(A[k1][0], A[k1][1], A[k1][2], A[k1][3], A[k1][4], A[k1][5], A[k1][6]);
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) v[(7 * k1 + 15 * k2) % 21] = A[k1][k2];
  }
}

// 28-point DFT, prime-factor 4 x 7: n = (7 n1 + 4 n2) mod 28, k = (21 k1 + 8 k2) mod 28
__device__ __forceinline__ void dft28(v2f* v) {
  v2f A[4][7];
#pragma unroll
  for (int n2 = 0; n2 < 7; ++n2) {
    v2f c0 = v[(0 + 4 * n2) % 28], c1 = v[(7 + 4 * n2) % 28], c2 = v[(14 + 4 * n2) % 28], c3 = v[(21 + 4 * n2) % 28];
    dft4<false>(c0, c1, c2, c3);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2; A[3][n2] = c3;
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    dft7(A[k1][0], A[k1][1], A[k1][2], A[k1][3], A[k1][4], A[k1][5], A[k1][6]);
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) v[(21 * k1 + 8 * k2) % 28] = A[k1][k2];
  }
}

// 42-point DFT, prime-factor 6 x 7: n = (7 n1 + 6 n2) mod 42, k = (7 k1 + 36 k2) mod 42
__device__ __forceinline__ void dft42(v2f* v) {
  v2f A[6][7];
#pragma unroll
  for (int n2 = 0; n2 < 7; ++n2) {
    v2f cc[6];
#pragma unroll
    for (int n1 = 0; n1 < 6; ++n1) cc[n1] = v[(7 * n1 + 6 * n2) % 42];
    dft6(cc);
#pragma unroll
    for (int k1 = 0; k1 < 6; ++k1) A[k1][n2] = cc[k1];
  }
#pragma unroll
  for (int k1 = 0; k1 < 6; ++k1) {
    dft7(A[k1][0], A[k1][1], A[k1][2], A[k1][3], A[k1][4], A[k1][5], A[k1][6]);
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) v[(7 * k1 + 36 * k2) % 42] = A[k1][k2];
  }
}

// 50-point DFT, prime-factor 2 x 25: n = (25 n1 + 2 n2) mod 50, k = (25 k1 + 26 k2) mod 50
__device__ __forceinline__ void dft50(v2f* v) {
  v2f A0[25], A1[25];
#pragma unroll
  for (int n2 = 0; n2 < 25; ++n2) {
    const v2f p = v[(2 * n2) % 50], q = v[(25 + 2 * n2) % 50];
    A0[n2] = p + q; A1[n2] = p - q;
  }
  dft25(A0);
  dft25(A1);
#pragma unroll
  for (int k2 = 0; k2 < 25; ++k2) { v[(26 * k2) % 50] = A0[k2]; v[(25 + 26 * k2) % 50] = A1[k2]; }
}

// 60-point DFT, prime-factor 4 x 15: n = (15 n1 + 4 n2) mod 60, k = (45 k1 + 16 k2) mod 60
__device__ __forceinline__ void dft60(v2f* v) {
  v2f A[4][15];
#pragma unroll
  for (int n2 = 0; n2 < 15; ++n2) {
    v2f c0 = v[(4 * n2) % 60], c1 = v[(15 + 4 * n2) % 60], c2 = v[(30 + 4 * n2) % 60], c3 = v[(45 + 4 * n2) % 60];
    dft4<false>(c0, c1, c2, c3);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2; A[3][n2] = c3;
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    dft15(A[k1]);
#pragma unroll
    for (int k2 = 0; k2 < 15; ++k2) v[(45 * k1 + 16 * k2) % 60] = A[k1][k2];
  }
}

// 64-point DFT, Cooley-Tukey 8 x 8: n = 8 n1 + n2, k = k1 + 8 k2, twiddles W_64^(n2 k1) between the two rounds of dft8
__device__ __forceinline__ void dft64(v2f* v) {
  constexpr float kC[8][8] = {{1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f},
                              {1.0f, 0.99518472667219693f, 0.98078528040323043f, 0.95694033573220882f, 0.92387953251128674f, 0.88192126434835505f, 0.83146961230254524f, 0.77301045336273699f},
                              {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f, 0.55557023301960229f, 0.38268343236508984f, 0.19509032201612833f},
                              {1.0f, 0.95694033573220882f, 0.83146961230254524f, 0.63439328416364549f, 0.38268343236508984f, 0.09801714032956077f, -0.19509032201612819f, -0.4713967368259977f},
                              {1.0f, 0.92387953251128674f, 0.70710678118654757f, 0.38268343236508984f, 6.123233995736766e-17f, -0.38268343236508973f, -0.70710678118654746f, -0.92387953251128674f},
                              {1.0f, 0.88192126434835505f, 0.55557023301960229f, 0.09801714032956077f, -0.38268343236508973f, -0.77301045336273699f, -0.98078528040323043f, -0.95694033573220894f},
                              {1.0f, 0.83146961230254524f, 0.38268343236508984f, -0.19509032201612819f, -0.70710678118654746f, -0.98078528040323043f, -0.92387953251128685f, -0.55557023301960218f},
                              {1.0f, 0.77301045336273699f, 0.19509032201612833f, -0.4713967368259977f, -0.92387953251128674f, -0.95694033573220894f, -0.55557023301960218f, 0.09801714032956009f}};
  constexpr float kS[8][8] = {{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f},
                              {0.0f, 0.098017140329560604f, 0.19509032201612825f, 0.29028467725446233f, 0.38268343236508978f, 0.47139673682599764f, 0.55557023301960218f, 0.63439328416364549f},
                              {0.0f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f, 0.70710678118654746f, 0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f},
                              {0.0f, 0.29028467725446233f, 0.55557023301960218f, 0.77301045336273699f, 0.92387953251128674f, 0.99518472667219682f, 0.98078528040323043f, 0.88192126434835505f},
                              {0.0f, 0.38268343236508978f, 0.70710678118654746f, 0.92387953251128674f, 1.0f, 0.92387953251128674f, 0.70710678118654757f, 0.38268343236508989f},
                              {0.0f, 0.47139673682599764f, 0.83146961230254524f, 0.99518472667219682f, 0.92387953251128674f, 0.63439328416364549f, 0.19509032201612861f, -0.29028467725446211f},
                              {0.0f, 0.55557023301960218f, 0.92387953251128674f, 0.98078528040323043f, 0.70710678118654757f, 0.19509032201612861f, -0.38268343236508967f, -0.83146961230254524f},
                              {0.0f, 0.63439328416364549f, 0.98078528040323043f, 0.88192126434835505f, 0.38268343236508989f, -0.29028467725446211f, -0.83146961230254524f, -0.99518472667219693f}};
  v2f Y[8][8];
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2) {
    v2f cc[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) cc[n1] = v[8 * n1 + n2];
    dft8<false>(cc);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) Y[k1][n2] = cc[k1];
  }
#pragma unroll
  for (int k1 = 0; k1 < 8; ++k1) {
#pragma unroll
    for (int n2 = 1; n2 < 8; ++n2)
      if (k1 > 0) Y[k1][n2] = cmulc<false>(Y[k1][n2], kC[k1][n2], -kS[k1][n2]);
    dft8<false>(Y[k1]);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) v[k1 + 8 * k2] = Y[k1][k2];
  }
}

// ---- round 6, second pass: 2205 = 35 x 63 (the 50 ms frame of 44.1 kHz audio)
// 9-point DFT, Cooley-Tukey 3 x 3: n = 3 n1 + n2, k = k1 + 3 k2, twiddles W_9^(n2 k1) between the two rounds of dft3
__device__ __forceinline__ void dft9(v2f* v) {
  v2f Y[3][3];
#pragma unroll
  for (int n2 = 0; n2 < 3; ++n2) {
    v2f c0 = v[n2], c1 = v[3 + n2], c2 = v[6 + n2];
    dft3(c0, c1, c2);
    Y[0][n2] = c0; Y[1][n2] = c1; Y[2][n2] = c2;
  }
  Y[1][1] = cmulc<false>(Y[1][1], 0.766044443118978f, -0.6427876096865393f);   // W_9^1
  Y[1][2] = cmulc<false>(Y[1][2], 0.17364817766693041f, -0.984807753012208f);   // W_9^2
  Y[2][1] = cmulc<false>(Y[2][1], 0.17364817766693041f, -0.984807753012208f);   // W_9^2
  Y[2][2] = cmulc<false>(Y[2][2], -0.9396926207859083f, -0.3420201433256689f);   // W_9^4
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    dft3(Y[k1][0], Y[k1][1], Y[k1][2]);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) v[k1 + 3 * k2] = Y[k1][k2];
  }
}

// 35-point DFT, prime-factor 5 x 7: n = (7 n1 + 5 n2) mod 35, k = (21 k1 + 15 k2) mod 35
__device__ __forceinline__ void dft35(v2f* v) {
  v2f A[5][7];
#pragma unroll
  for (int n2 = 0; n2 < 7; ++n2) {
    v2f c0 = v[(5 * n2) % 35], c1 = v[(7 + 5 * n2) % 35], c2 = v[(14 + 5 * n2) % 35], c3 = v[(21 + 5 * n2) % 35], c4 = v[(28 + 5 * n2) % 35];
    dft5(c0, c1, c2, c3, c4);
    A[0][n2] = c0; A[1][n2] = c1; A[2][n2] = c2; A[3][n2] = c3; A[4][n2] = c4;
  }
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) {
    dft7(A[k1][0], A[k1][1], A[k1][2], A[k1][3], A[k1][4], A[k1][5], A[k1][6]);
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) v[(21 * k1 + 15 * k2) % 35] = A[k1][k2];
  }
}

// 63-point DFT, prime-factor 9 x 7: n = (7 n1 + 9 n2) mod 63, k = (28 k1 + 36 k2) mod 63
__device__ __forceinline__ void dft63(v2f* v) {
  v2f A[9][7];
#pragma unroll
  for (int n2 = 0; n2 < 7; ++n2) {
    v2f cc[9];
#pragma unroll
    for (int n1 = 0; n1 < 9; ++n1) cc[n1] = v[(7 * n1 + 9 * n2) % 63];
    dft9(cc);
#pragma unroll
    for (int k1 = 0; k1 < 9; ++k1) A[k1][n2] = cc[k1];
  }
#pragma unroll
  for (int k1 = 0; k1 < 9; ++k1) {
    dft7(A[k1][0], A[k1][1], A[k1][2], A[k1][3], A[k1][4], A[k1][5], A[k1][6]);
#pragma unroll
    for (int k2 = 0; k2 < 7; ++k2) v[(28 * k1 + 36 * k2) % 63] = A[k1][k2];
  }
}

template <int N>
__device__ __forceinline__ void dft_n(v2f* v) {
  static_assert(N == 4 || N == 8 || N == 10 || N == 12 || N == 14 || N == 15 || N == 16 || N == 20 || N == 21 || N == 24 || N == 25 || N == 28 || N == 30 || N == 32 ||
                N == 35 || N == 40 || N == 42 || N == 48 || N == 50 || N == 60 || N == 63 || N == 64, "no codelet for this length");
  if constexpr (N == 4) dft4<false>(v[0], v[1], v[2], v[3]);
  else if constexpr (N == 8) dft8<false>(v);
  else if constexpr (N == 10) dft10(v);
  else if constexpr (N == 12) dft12(v);
  else if constexpr (N == 15) dft15(v);
  else if constexpr (N == 16) dft16<false>(v);
  else if constexpr (N == 20) dft20(v);
  else if constexpr (N == 24) dft24(v);
  else if constexpr (N == 25) dft25(v);
  else if constexpr (N == 30) dft30(v);
  else if constexpr (N == 32) dft32(v);
  else if constexpr (N == 40) dft40(v);
  else if constexpr (N == 14) dft14(v);
  else if constexpr (N == 21) dft21(v);
  else if constexpr (N == 28) dft28(v);
  else if constexpr (N == 42) dft42(v);
  else if constexpr (N == 50) dft50(v);
  else if constexpr (N == 60) dft60(v);
  else if constexpr (N == 64) dft64(v);
  else if constexpr (N == 35) dft35(v);
  else if constexpr (N == 63) dft63(v);
  else dft48(v);
}

}  // namespace nxsig
