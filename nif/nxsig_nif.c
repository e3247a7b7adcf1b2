/*
 * nxsig_nif.c — dirty-NIF shim between the BEAM and the C ABI in include/nxsig.h.
 *
 * NOT COMPILED IN THIS IMAGE: erl_nif.h (Erlang/OTP) is absent and there is no elixir/erl/mix toolchain, so this
 * file is the reference-side binding a maintainer adds; nif/Makefile builds it only when
 * `erl` is found.  It is deliberately mechanical: every function unpacks terms, calls ONE nxsig_* entry point and
 * packs the result; all logic lives behind the C ABI where it is tested (tests/ drive the same entry points
 * through ctypes).
 *
 * Conventions:
 *   - every GPU call is a dirty job (ERL_NIF_DIRTY_JOB_IO_BOUND: the scheduler thread waits on the GPU);
 *   - inputs are borrowed binaries (enif_inspect_binary, read-only, valid for the call), host outputs are
 *     BEAM-owned binaries (enif_make_new_binary), device tensors are resource objects whose destructor frees HBM;
 *   - errors come back as {:error, {code, message}}; nothing throws, aborts or longjmps across the boundary;
 *   - the context resource owns one GPU + one HIP stream; libnxsig serialises calls per context internally, so
 *     dirty schedulers may call concurrently from any OS thread.
 */
#include <erl_nif.h>
#include <stdint.h>
#include <string.h>

#include "../include/nxsig.h"

static ErlNifResourceType* CTX_RES;
static ErlNifResourceType* BUF_RES;

typedef struct { nxsig_ctx* ctx; } ctx_res_t;
typedef struct { ctx_res_t* owner; void* dptr; size_t bytes; } buf_res_t;

static void ctx_dtor(ErlNifEnv* env, void* obj) { (void)env; ctx_res_t* r = obj; if (r->ctx) nxsig_ctx_destroy(r->ctx); }
static void buf_dtor(ErlNifEnv* env, void* obj) {
  (void)env;
  buf_res_t* b = obj;
  if (b->dptr && b->owner && b->owner->ctx) nxsig_free(b->owner->ctx, b->dptr);
  if (b->owner) enif_release_resource(b->owner);
}

static ERL_NIF_TERM mk_atom(ErlNifEnv* env, const char* a) { return enif_make_atom(env, a); }
static ERL_NIF_TERM mk_error(ErlNifEnv* env, int code) {
  const char* msg = nxsig_last_error();
  ERL_NIF_TERM m;
  size_t n = msg ? strlen(msg) : 0;
  unsigned char* p = enif_make_new_binary(env, n, &m);
  if (n) memcpy(p, msg, n);
  return enif_make_tuple2(env, mk_atom(env, "error"), enif_make_tuple2(env, enif_make_int(env, code), m));
}
static ERL_NIF_TERM mk_ok(ErlNifEnv* env, ERL_NIF_TERM v) { return enif_make_tuple2(env, mk_atom(env, "ok"), v); }

static int get_ctx(ErlNifEnv* env, ERL_NIF_TERM t, ctx_res_t** out) { return enif_get_resource(env, t, CTX_RES, (void**)out); }

/* {n, hop, k, pad_mode, pad_lo, pad_hi, scaling, sampling_rate} -> nxsig_stft_params */
static int get_params(ErlNifEnv* env, ERL_NIF_TERM t, nxsig_stft_params* p) {
  const ERL_NIF_TERM* e;
  int arity;
  ErlNifSInt64 lo, hi;
  int n, hop, k, pad, scal;
  double fs;
  if (!enif_get_tuple(env, t, &arity, &e) || arity != 8) return 0;
  if (!enif_get_int(env, e[0], &n) || !enif_get_int(env, e[1], &hop) || !enif_get_int(env, e[2], &k) ||
      !enif_get_int(env, e[3], &pad) || !enif_get_int64(env, e[4], &lo) || !enif_get_int64(env, e[5], &hi) ||
      !enif_get_int(env, e[6], &scal) || !enif_get_double(env, e[7], &fs))
    return 0;
  memset(p, 0, sizeof *p);
  p->frame_length = n; p->hop = hop; p->fft_length = k; p->pad_mode = pad; p->pad_lo = lo; p->pad_hi = hi;
  p->scaling = scal; p->sampling_rate = fs;
  return 1;
}

/* ctx_create(device) */
static ERL_NIF_TERM nif_ctx_create(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int dev;
  if (argc != 1 || !enif_get_int(env, argv[0], &dev)) return enif_make_badarg(env);
  nxsig_ctx* c = NULL;
  int rc = nxsig_ctx_create(dev, &c);
  if (rc) return mk_error(env, rc);
  ctx_res_t* r = enif_alloc_resource(CTX_RES, sizeof *r);
  r->ctx = c;
  ERL_NIF_TERM t = enif_make_resource(env, r);
  enif_release_resource(r);
  return mk_ok(env, t);
}

/* window(kind, n, periodic, beta, eps) -> {:ok, f32 binary}   (host-side generator, BinaryBackend rounding) */
static ERL_NIF_TERM nif_window(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int kind, n, per;
  double beta, eps;
  if (argc != 5 || !enif_get_int(env, argv[0], &kind) || !enif_get_int(env, argv[1], &n) || !enif_get_int(env, argv[2], &per) ||
      !enif_get_double(env, argv[3], &beta) || !enif_get_double(env, argv[4], &eps) || n < 0)
    return enif_make_badarg(env);
  ERL_NIF_TERM out;
  float* w = (float*)enif_make_new_binary(env, (size_t)n * 4, &out);
  int rc = nxsig_window_f32(kind, n, per, beta, eps, w);
  return rc ? mk_error(env, rc) : mk_ok(env, out);
}

/* firwin(num_taps, [cutoff], window_kind, beta, pass_zero, scale, sampling_rate) */
static ERL_NIF_TERM nif_firwin(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int taps, kind, pz, sc;
  double beta, fs, cut[64];
  unsigned len;
  if (argc != 7 || !enif_get_int(env, argv[0], &taps) || !enif_get_list_length(env, argv[1], &len) || len > 64 ||
      !enif_get_int(env, argv[2], &kind) || !enif_get_double(env, argv[3], &beta) || !enif_get_int(env, argv[4], &pz) ||
      !enif_get_int(env, argv[5], &sc) || !enif_get_double(env, argv[6], &fs) || taps < 1)
    return enif_make_badarg(env);
  ERL_NIF_TERM head, tail = argv[1];
  for (unsigned i = 0; i < len; ++i) {
    if (!enif_get_list_cell(env, tail, &head, &tail) || !enif_get_double(env, head, &cut[i])) return enif_make_badarg(env);
  }
  ERL_NIF_TERM out;
  float* h = (float*)enif_make_new_binary(env, (size_t)taps * 4, &out);
  int rc = nxsig_firwin_f32(taps, cut, (int)len, kind, beta, pz, sc, fs, h);
  return rc ? mk_error(env, rc) : mk_ok(env, out);
}

/* stft(ctx, x_bin, length, batch, window_bin, params) -> {:ok, z_bin, num_frames, times_bin, freqs_bin} */
static ERL_NIF_TERM nif_stft(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary x, w;
  ErlNifSInt64 length;
  int batch;
  nxsig_stft_params p;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size != (size_t)batch * (size_t)length * 4 || w.size != (size_t)p.frame_length * 4)
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  ERL_NIF_TERM zt, tt, ft;
  nxsig_c64* z = (nxsig_c64*)enif_make_new_binary(env, (size_t)batch * (size_t)m * (size_t)p.fft_length * 8, &zt);
  int rc = nxsig_stft_f32(c->ctx, (const float*)x.data, length, batch, length, (const float*)w.data, &p, z, NULL, NXSIG_HOST);
  if (rc) return mk_error(env, rc);
  float* t = (float*)enif_make_new_binary(env, (size_t)m * 4, &tt);
  float* f = (float*)enif_make_new_binary(env, (size_t)p.fft_length * 4, &ft);
  if ((rc = nxsig_stft_times_f32(p.frame_length, p.sampling_rate, m, t))) return mk_error(env, rc);
  if ((rc = nxsig_fft_frequencies_f32(p.sampling_rate, p.fft_length, 0, f))) return mk_error(env, rc);
  return enif_make_tuple5(env, mk_atom(env, "ok"), zt, enif_make_int64(env, m), tt, ft);
}

/* istft(ctx, z_bin, num_frames, batch, window_bin, params) -> {:ok, y_bin} */
static ERL_NIF_TERM nif_istft(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary z, w;
  ErlNifSInt64 m;
  int batch;
  nxsig_stft_params p;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &z) || !enif_get_int64(env, argv[2], &m) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p))
    return enif_make_badarg(env);
  if (batch < 1 || m < 1 || z.size != (size_t)batch * (size_t)m * (size_t)p.fft_length * 8 || w.size != (size_t)p.frame_length * 4)
    return enif_make_badarg(env);
  int64_t n = nxsig_ola_length(m, p.frame_length, p.hop);
  if (n < 0) return mk_error(env, (int)n);
  ERL_NIF_TERM yt;
  nxsig_c64* y = (nxsig_c64*)enif_make_new_binary(env, (size_t)batch * (size_t)n * 8, &yt);
  int rc = nxsig_istft_c64(c->ctx, (const nxsig_c64*)z.data, m, batch, (const float*)w.data, &p, y, NXSIG_HOST);
  return rc ? mk_error(env, rc) : mk_ok(env, yt);
}

/* fir(ctx, x_bin, length, batch, taps_bin, mode) -> {:ok, y_bin} */
static ERL_NIF_TERM nif_fir(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary x, h;
  ErlNifSInt64 length;
  int batch, mode;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &h) || !enif_get_int(env, argv[5], &mode))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size != (size_t)batch * (size_t)length * 4 || h.size < 4 || h.size % 4) return enif_make_badarg(env);
  int64_t n = nxsig_conv_length(length, (int64_t)(h.size / 4), mode);
  if (n < 0) return mk_error(env, (int)n);
  ERL_NIF_TERM yt;
  float* y = (float*)enif_make_new_binary(env, (size_t)batch * (size_t)n * 4, &yt);
  int rc = nxsig_fir_f32(c->ctx, (const float*)x.data, length, batch, length, (const float*)h.data, (int)(h.size / 4), mode, y, NXSIG_HOST);
  return rc ? mk_error(env, rc) : mk_ok(env, yt);
}

/* ---- device-resident tensors: keep stft -> edit -> istft chains in HBM ---- */
static ERL_NIF_TERM nif_to_device(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary b;
  if (argc != 2 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &b)) return enif_make_badarg(env);
  void* d = NULL;
  int rc = nxsig_alloc(c->ctx, b.size, &d);
  if (rc) return mk_error(env, rc);
  if ((rc = nxsig_upload(c->ctx, d, b.data, b.size))) { nxsig_free(c->ctx, d); return mk_error(env, rc); }
  buf_res_t* r = enif_alloc_resource(BUF_RES, sizeof *r);
  r->owner = c; enif_keep_resource(c); r->dptr = d; r->bytes = b.size;
  ERL_NIF_TERM t = enif_make_resource(env, r);
  enif_release_resource(r);
  return mk_ok(env, t);
}

static ERL_NIF_TERM nif_from_device(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  buf_res_t* b;
  if (argc != 1 || !enif_get_resource(env, argv[0], BUF_RES, (void**)&b)) return enif_make_badarg(env);
  ERL_NIF_TERM out;
  unsigned char* p = enif_make_new_binary(env, b->bytes, &out);
  int rc = nxsig_download(b->owner->ctx, p, b->dptr, b->bytes);
  return rc ? mk_error(env, rc) : mk_ok(env, out);
}

/* stft_dev(ctx, x_buf, length, batch, window_bin, params) -> {:ok, z_buf, num_frames}  (asynchronous on the ctx stream) */
static ERL_NIF_TERM nif_stft_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  buf_res_t* x;
  ErlNifBinary w;
  ErlNifSInt64 length;
  int batch;
  nxsig_stft_params p;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !enif_get_resource(env, argv[1], BUF_RES, (void**)&x) ||
      !enif_get_int64(env, argv[2], &length) || !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) ||
      !get_params(env, argv[5], &p))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x->bytes < (size_t)batch * (size_t)length * 4 || w.size != (size_t)p.frame_length * 4)
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  size_t zbytes = (size_t)batch * (size_t)m * (size_t)p.fft_length * 8;
  void* z = NULL;
  int rc = nxsig_alloc(c->ctx, zbytes, &z);
  if (rc) return mk_error(env, rc);
  rc = nxsig_stft_f32(c->ctx, (const float*)x->dptr, length, batch, length, (const float*)w.data, &p, (nxsig_c64*)z, NULL, NXSIG_DEVICE);
  if (rc) { nxsig_free(c->ctx, z); return mk_error(env, rc); }
  buf_res_t* r = enif_alloc_resource(BUF_RES, sizeof *r);
  r->owner = c; enif_keep_resource(c); r->dptr = z; r->bytes = zbytes;
  ERL_NIF_TERM t = enif_make_resource(env, r);
  enif_release_resource(r);
  return enif_make_tuple3(env, mk_atom(env, "ok"), t, enif_make_int64(env, m));
}

/* istft_dev(ctx, z_buf, num_frames, batch, window_bin, params) -> {:ok, y_buf} */
static ERL_NIF_TERM nif_istft_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  buf_res_t* z;
  ErlNifBinary w;
  ErlNifSInt64 m;
  int batch;
  nxsig_stft_params p;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !enif_get_resource(env, argv[1], BUF_RES, (void**)&z) ||
      !enif_get_int64(env, argv[2], &m) || !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) ||
      !get_params(env, argv[5], &p))
    return enif_make_badarg(env);
  if (batch < 1 || m < 1 || z->bytes < (size_t)batch * (size_t)m * (size_t)p.fft_length * 8) return enif_make_badarg(env);
  int64_t n = nxsig_ola_length(m, p.frame_length, p.hop);
  if (n < 0) return mk_error(env, (int)n);
  size_t ybytes = (size_t)batch * (size_t)n * 8;
  void* y = NULL;
  int rc = nxsig_alloc(c->ctx, ybytes, &y);
  if (rc) return mk_error(env, rc);
  rc = nxsig_istft_c64(c->ctx, (const nxsig_c64*)z->dptr, m, batch, (const float*)w.data, &p, (nxsig_c64*)y, NXSIG_DEVICE);
  if (rc) { nxsig_free(c->ctx, y); return mk_error(env, rc); }
  buf_res_t* r = enif_alloc_resource(BUF_RES, sizeof *r);
  r->owner = c; enif_keep_resource(c); r->dptr = y; r->bytes = ybytes;
  ERL_NIF_TERM t = enif_make_resource(env, r);
  enif_release_resource(r);
  return mk_ok(env, t);
}

/* spectrum_mul_dev(ctx, z_buf, rows, fft_length, h_bin) -> {:ok, z_buf}   (in place: Nx.multiply(z, hfft), guides/filtering.livemd:141) */
static ERL_NIF_TERM nif_spectrum_mul_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  buf_res_t* z;
  ErlNifBinary h;
  ErlNifSInt64 rows;
  int k;
  if (argc != 5 || !get_ctx(env, argv[0], &c) || !enif_get_resource(env, argv[1], BUF_RES, (void**)&z) ||
      !enif_get_int64(env, argv[2], &rows) || !enif_get_int(env, argv[3], &k) || !enif_inspect_binary(env, argv[4], &h))
    return enif_make_badarg(env);
  if (rows < 0 || k < 1 || h.size != (size_t)k * 8 || z->bytes < (size_t)rows * (size_t)k * 8) return enif_make_badarg(env);
  int rc = nxsig_spectrum_mul_c64(c->ctx, (const nxsig_c64*)z->dptr, rows, k, (const nxsig_c64*)h.data, (nxsig_c64*)z->dptr, NXSIG_DEVICE);
  return rc ? mk_error(env, rc) : mk_ok(env, argv[1]);
}

static int load(ErlNifEnv* env, void** priv, ERL_NIF_TERM info) {
  (void)priv; (void)info;
  CTX_RES = enif_open_resource_type(env, NULL, "nxsig_ctx", ctx_dtor, ERL_NIF_RT_CREATE | ERL_NIF_RT_TAKEOVER, NULL);
  BUF_RES = enif_open_resource_type(env, NULL, "nxsig_buf", buf_dtor, ERL_NIF_RT_CREATE | ERL_NIF_RT_TAKEOVER, NULL);
  return (CTX_RES && BUF_RES) ? 0 : -1;
}
static int upgrade(ErlNifEnv* env, void** priv, void** old, ERL_NIF_TERM info) { (void)old; return load(env, priv, info); }

static ErlNifFunc funcs[] = {
    {"ctx_create", 1, nif_ctx_create, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"window", 5, nif_window, 0},
    {"firwin", 7, nif_firwin, 0},
    {"stft", 6, nif_stft, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"istft", 6, nif_istft, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fir", 6, nif_fir, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"to_device", 2, nif_to_device, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"from_device", 1, nif_from_device, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_dev", 6, nif_stft_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"istft_dev", 6, nif_istft_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"spectrum_mul_dev", 5, nif_spectrum_mul_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
};

ERL_NIF_INIT(Elixir.NxSignalAMD.NIF, funcs, load, NULL, upgrade, NULL)
