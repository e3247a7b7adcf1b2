/* Plain-C consumer of the C ABI (what a NIF / cgo / JNI shim would be): includes include/nxsig.h, links libnxsig.so.
 * Built by tests/test_c_abi.py with gcc -std=c99 (CPU suite: compile + link only; GPU suite: run).
 * Computes stft -> istft of a 2-channel chirp with a library-generated Hann window on HOST buffers and on DEVICE
 * buffers, checks both agree bit for bit and that the round trip reproduces the input on the interior; then the multi-GPU
 * entry points with no host language in the loop: a LOCAL group (two members on device 0: shards + assembly by copies; one
 * member: the RCCL all-gather) must reproduce the unsharded spectrum bit for bit; a 2-D fft_nd round trip; the f64 / c128 tier. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nxsig.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != NXSIG_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nxsig_last_error()); return 2; } \
  } while (0)

int main(void) {
  enum { N = 1024, HOP = 256, L = 48000, CH = 2 };
  int ndev = 0;
  if (nxsig_abi_version() != NXSIG_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
  if (nxsig_device_count(&ndev) != NXSIG_OK || ndev < 1) { printf("no GPU: %s\n", nxsig_last_error()); return 77; }
  nxsig_ctx* ctx = NULL;
  CHECK(nxsig_ctx_create(0, &ctx));
  float* w = (float*)malloc(N * sizeof(float));
  CHECK(nxsig_window_f32(NXSIG_WIN_HANN, N, 1, 0.0, 1e-7, w));
  float* x = (float*)malloc((size_t)CH * L * sizeof(float));
  for (int c = 0; c < CH; ++c)
    for (int i = 0; i < L; ++i) x[(size_t)c * L + i] = (float)sin(1e-7 * (c + 1) * (double)i * (double)i) + 0.25f * (float)cos(0.01 * i);
  nxsig_stft_params p;
  memset(&p, 0, sizeof p);
  p.frame_length = N; p.hop = HOP; p.fft_length = N; p.pad_mode = NXSIG_PAD_VALID; p.scaling = NXSIG_SCALE_NONE; p.sampling_rate = 48000.0;
  const int64_t M = nxsig_num_frames(L, N, HOP, NXSIG_PAD_VALID, 0, 0);
  const int64_t out_len = nxsig_ola_length(M, N, HOP);
  if (M != 184 || out_len != 183 * 256 + 1024) { fprintf(stderr, "shape helpers wrong: %lld %lld\n", (long long)M, (long long)out_len); return 2; }
  nxsig_c64* z = (nxsig_c64*)malloc((size_t)CH * M * N * sizeof(nxsig_c64));
  nxsig_c64* y = (nxsig_c64*)malloc((size_t)CH * out_len * sizeof(nxsig_c64));
  int64_t m_out = 0;
  CHECK(nxsig_stft_f32(ctx, x, L, CH, L, w, &p, z, &m_out, NXSIG_HOST));
  CHECK(nxsig_istft_c64(ctx, z, M, CH, w, &p, y, NXSIG_HOST));
  /* the same through device-resident buffers */
  void *xd, *zd, *yd;
  CHECK(nxsig_alloc(ctx, (size_t)CH * L * sizeof(float), &xd));
  CHECK(nxsig_alloc(ctx, (size_t)CH * M * N * sizeof(nxsig_c64), &zd));
  CHECK(nxsig_alloc(ctx, (size_t)CH * out_len * sizeof(nxsig_c64), &yd));
  CHECK(nxsig_upload(ctx, xd, x, (size_t)CH * L * sizeof(float)));
  CHECK(nxsig_stft_f32(ctx, (const float*)xd, L, CH, L, w, &p, (nxsig_c64*)zd, NULL, NXSIG_DEVICE));
  CHECK(nxsig_istft_c64(ctx, (const nxsig_c64*)zd, M, CH, w, &p, (nxsig_c64*)yd, NXSIG_DEVICE));
  nxsig_c64* y2 = (nxsig_c64*)malloc((size_t)CH * out_len * sizeof(nxsig_c64));
  CHECK(nxsig_download(ctx, y2, yd, (size_t)CH * out_len * sizeof(nxsig_c64)));
  if (memcmp(y, y2, (size_t)CH * out_len * sizeof(nxsig_c64)) != 0) { fprintf(stderr, "host and device paths differ\n"); return 3; }
  double worst = 0.0;
  for (int c = 0; c < CH; ++c)
    for (int64_t i = N; i < out_len - N; ++i) {
      const double d = fabs((double)y[(size_t)c * out_len + i].re - (double)x[(size_t)c * L + i]);
      if (d > worst) worst = d;
    }
  /* istft(z * h) in one call == spectrum_mul followed by istft, bit for bit (h = a one-pole low-pass response) */
  {
    nxsig_c64* h = (nxsig_c64*)malloc(N * sizeof(nxsig_c64));
    nxsig_c64* zf = (nxsig_c64*)malloc((size_t)CH * M * N * sizeof(nxsig_c64));
    nxsig_c64* ya = (nxsig_c64*)malloc((size_t)CH * out_len * sizeof(nxsig_c64));
    for (int k = 0; k < N; ++k) {
      const double wk = 2.0 * 3.14159265358979323846 * k / N, dre = 1.0 - 0.9 * cos(wk), dim = 0.9 * sin(wk), dd = dre * dre + dim * dim;
      h[k].re = (float)(0.1 * dre / dd); h[k].im = (float)(-0.1 * dim / dd);
    }
    CHECK(nxsig_spectrum_mul_c64(ctx, z, (int64_t)CH * M, N, h, zf, NXSIG_HOST));
    CHECK(nxsig_istft_c64(ctx, zf, M, CH, w, &p, ya, NXSIG_HOST));
    CHECK(nxsig_istft_filtered_c64(ctx, (const nxsig_c64*)zd, M, CH, w, &p, h, (nxsig_c64*)yd, NXSIG_DEVICE));
    CHECK(nxsig_download(ctx, y2, yd, (size_t)CH * out_len * sizeof(nxsig_c64)));
    if (memcmp(ya, y2, (size_t)CH * out_len * sizeof(nxsig_c64)) != 0) { fprintf(stderr, "fused filter differs from multiply-then-istft\n"); return 8; }
    free(h); free(zf); free(ya);
  }
  /* invalid arguments come back as codes + messages, never as aborts */
  p.scaling = 7;
  if (nxsig_stft_f32(ctx, x, L, CH, L, w, &p, z, NULL, NXSIG_HOST) != NXSIG_ERR_INVALID_ARG || strstr(nxsig_last_error(), "invalid :scaling") == NULL) {
    fprintf(stderr, "bad scaling was not rejected properly\n");
    return 4;
  }
  p.scaling = NXSIG_SCALE_NONE;
  /* ---- sharded over a group: channels axis on two members that share the GPU, then one member with an RCCL communicator */
  {
    const int32_t devs[2] = {0, 0};
    nxsig_group *g2 = NULL, *g1 = NULL;
    nxsig_c64* zs = (nxsig_c64*)malloc((size_t)CH * M * N * sizeof(nxsig_c64));
    const float* xs[2] = {x, NULL};
    nxsig_c64* zo[2] = {zs, NULL};
    int64_t m0, m1, s0, s1;
    CHECK(nxsig_group_create_local(2, devs, &g2));
    if (nxsig_group_world(g2) != 2 || nxsig_group_local_count(g2) != 2 || nxsig_group_has_rccl(g2) != 0) { fprintf(stderr, "group shape\n"); return 6; }
    for (int gather = 0; gather <= 1; ++gather) {
      memset(zs, 0, (size_t)CH * M * N * sizeof(nxsig_c64));
      CHECK(nxsig_stft_sharded_f32(g2, xs, L, CH, L, w, &p, NXSIG_SHARD_CHANNELS, gather, zo, NXSIG_HOST));
      if (memcmp(zs, z, (size_t)CH * M * N * sizeof(nxsig_c64)) != 0) { fprintf(stderr, "channel shards differ from the unsharded result (gather %d)\n", gather); return 6; }
    }
    CHECK(nxsig_shard_frames(M, N, HOP, 2, 1, &m0, &m1, &s0, &s1));
    if (m0 != 92 || m1 != 184 || s0 != 92 * HOP || s1 != 183 * HOP + N) { fprintf(stderr, "shard_frames wrong\n"); return 6; }
    CHECK(nxsig_group_barrier(g2));
    nxsig_group_destroy(g2);
    CHECK(nxsig_group_create_local(1, NULL, &g1));
    if (nxsig_group_has_rccl(g1) != 1) { fprintf(stderr, "one GPU per member must bring RCCL up\n"); return 6; }
    memset(zs, 0, (size_t)CH * M * N * sizeof(nxsig_c64));
    CHECK(nxsig_stft_sharded_f32(g1, xs, L, CH, L, w, &p, NXSIG_SHARD_CHANNELS, 1, zo, NXSIG_HOST));
    if (memcmp(zs, z, (size_t)CH * M * N * sizeof(nxsig_c64)) != 0) { fprintf(stderr, "RCCL assembly differs\n"); return 6; }
    double v[2] = {1.5, -2.0};
    CHECK(nxsig_group_allreduce_f64(g1, v, 2, 0));
    CHECK(nxsig_group_barrier(g1));
    nxsig_group_destroy(g1);
    free(zs);
  }
  /* ---- fft_nd over both axes of a 6 x 10 real tensor (lengths 8 x 16), then ifft_nd back: the zero-padded input returns */
  {
    const int64_t shape[2] = {6, 10}, lengths[2] = {8, 16}, shape2[2] = {8, 16};
    const int32_t axes[2] = {0, 1};
    float t[60];
    nxsig_c64 f[128], b[128];
    for (int i = 0; i < 60; ++i) t[i] = (float)sin(0.37 * i) + (float)(i % 7);
    CHECK(nxsig_fft_nd(ctx, t, 1, shape, 2, axes, lengths, 2, 0, f, NXSIG_HOST));
    CHECK(nxsig_fft_nd(ctx, f, 0, shape2, 2, axes, lengths, 2, 1, b, NXSIG_HOST));
    for (int r = 0; r < 8; ++r)
      for (int c = 0; c < 16; ++c) {
        const double want = (r < 6 && c < 10) ? (double)t[r * 10 + c] : 0.0;
        if (fabs((double)b[r * 16 + c].re - want) > 2e-5 || fabs((double)b[r * 16 + c].im) > 2e-5) { fprintf(stderr, "fft_nd round trip at (%d, %d)\n", r, c); return 7; }
      }
  }
  /* ---- the f64 / c128 tier through the C boundary: stft of the widened signal with an f64 window, then istft; the round trip must
   *      reproduce the samples to double precision (a c64 detour would stop near 1e-7), and fir_f64 with a 3-tap filter must equal
   *      the direct sum */
  double worst64 = 0.0;
  {
    double* w64 = (double*)malloc(N * sizeof(double));
    double* x64 = (double*)malloc((size_t)CH * L * sizeof(double));
    CHECK(nxsig_window_f64(NXSIG_WIN_HANN, N, 1, 0.0, 1e-7, w64));
    for (size_t i = 0; i < (size_t)CH * L; ++i) x64[i] = (double)x[i] + 1e-9 * (double)(i % 97);
    nxsig_c128* z64 = (nxsig_c128*)malloc((size_t)CH * M * N * sizeof(nxsig_c128));
    nxsig_c128* y64 = (nxsig_c128*)malloc((size_t)CH * out_len * sizeof(nxsig_c128));
    CHECK(nxsig_stft_f64(ctx, x64, L, CH, L, w64, 1, &p, z64, NULL, NXSIG_HOST));
    CHECK(nxsig_istft_c128(ctx, z64, M, CH, w64, 1, &p, y64, NXSIG_HOST));
    for (int c = 0; c < CH; ++c)
      for (int64_t i = N; i < out_len - N; ++i) {
        const double d = fabs(y64[(size_t)c * out_len + i].re - x64[(size_t)c * L + i]);
        if (d > worst64) worst64 = d;
      }
    const double h3[3] = {0.25, 0.5, 0.25};
    double* f64o = (double*)malloc((size_t)CH * L * sizeof(double));
    CHECK(nxsig_fir_f64(ctx, x64, L, CH, L, h3, 3, NXSIG_CONV_SAME, f64o, NXSIG_HOST));
    for (int c = 0; c < CH; ++c)
      for (int64_t i = 1; i + 1 < L; ++i) {
        const double* xr = x64 + (size_t)c * L;
        const double d = fabs(f64o[(size_t)c * L + i] - (0.25 * xr[i - 1] + 0.5 * xr[i] + 0.25 * xr[i + 1]));
        if (d > worst64) worst64 = d;
      }
    free(w64); free(x64); free(z64); free(y64); free(f64o);
    if (!(worst64 < 1e-12)) { fprintf(stderr, "f64 tier: error %.3g\n", worst64); return 8; }
  }
  CHECK(nxsig_free(ctx, xd)); CHECK(nxsig_free(ctx, zd)); CHECK(nxsig_free(ctx, yd));
  nxsig_ctx_destroy(ctx);
  printf("c_abi_smoke ok: M=%lld frames, round-trip max abs err on the interior %.3g\n", (long long)m_out, worst);
  return worst < 1e-5 ? 0 : 5;
}
