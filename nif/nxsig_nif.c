/*
 * nxsig_nif.c — dirty-NIF shim between the BEAM and the C ABI in include/nxsig.h.
 *
 * The build image has no Erlang/OTP (no erl, no erl_nif.h), so nif/Makefile builds this file only where `erl` is
 * found.  It is nevertheless compiled and EXECUTED by the test-suite: tests/stub/erl_nif.h declares the subset of the NIF
 * API used here and tests/stub/erl_nif_fake.c is a miniature term runtime, so tests/test_nif_shim.py (CPU: host generators,
 * argument validation; GPU: stft / istft / fir / device chains / sharded calls) calls these functions exactly as the BEAM
 * would (entry->funcs[i].fptr(env, argc, argv)) and compares with the ctypes path bit for bit.
 *
 * The shim is deliberately mechanical: every function unpacks terms, validates sizes BEFORE allocating, calls ONE
 * nxsig_* entry point and packs the result; all logic lives behind the C ABI.
 *
 * Conventions:
 *   - every GPU call is a dirty job (ERL_NIF_DIRTY_JOB_IO_BOUND: the scheduler thread waits on the GPU);
 *   - inputs are borrowed binaries (enif_inspect_binary, read-only, valid for the call); host outputs are BEAM-owned
 *     binaries (enif_alloc_binary, which can FAIL without taking the VM down, then enif_make_binary); device tensors are
 *     resource objects whose destructor frees HBM;
 *   - errors come back as {:error, {code, message}} (code -1 = ArgumentError on the Elixir side); malformed terms are
 *     badarg; nothing throws, aborts or longjmps across the boundary;
 *   - a context resource owns one GPU + one HIP stream; libnxsig serialises calls per context internally, so dirty
 *     schedulers may call concurrently from any OS thread;
 *   - a group resource owns one context per member GPU and the RCCL communicators (nxsig_group_create_local).
 */
#include <erl_nif.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../include/nxsig.h"

/* What this file assumes about <erl_nif.h>, checked at COMPILE time against whichever header it meets — the stand-in of the test
 * suite (tests/stub/erl_nif.h, written from the documentation) or a real OTP one (nif/Makefile, -Werror).  A first `mix compile`
 * on a BEAM box that disagrees with any of these stops here with the reason, not in a dirty scheduler at run time. */
_Static_assert(sizeof(ERL_NIF_TERM) == sizeof(void*), "ERL_NIF_TERM is one machine word");
_Static_assert(sizeof(ErlNifSInt64) == 8 && sizeof(ErlNifUInt64) == 8, "64-bit integer terms (enif_get_int64 / enif_make_int64)");
_Static_assert(ERL_NIF_DIRTY_JOB_CPU_BOUND == 1 && ERL_NIF_DIRTY_JOB_IO_BOUND == 2, "dirty-scheduler flags of ErlNifFunc.flags");
_Static_assert(offsetof(ErlNifBinary, size) == 0 && offsetof(ErlNifBinary, data) == sizeof(size_t), "ErlNifBinary starts {size, data}");
_Static_assert(offsetof(ErlNifFunc, name) == 0 && offsetof(ErlNifFunc, arity) == sizeof(char*) &&
               offsetof(ErlNifFunc, fptr) > offsetof(ErlNifFunc, arity) && offsetof(ErlNifFunc, flags) > offsetof(ErlNifFunc, fptr),
               "ErlNifFunc is {name, arity, fptr, flags}: the funcs[] table below is written positionally");
_Static_assert(offsetof(ErlNifEntry, major) == 0 && offsetof(ErlNifEntry, funcs) > offsetof(ErlNifEntry, num_of_funcs),
               "ErlNifEntry starts {major, minor, name, num_of_funcs, funcs}");
_Static_assert(ERL_NIF_RT_CREATE == 1 && ERL_NIF_RT_TAKEOVER == 2, "resource-type flags");
_Static_assert(ERL_NIF_LATIN1 == 1, "atom encoding passed to enif_get_atom");
#ifdef ERL_NIF_MAJOR_VERSION   /* a real OTP header: dirty NIFs without a build flag need NIF API 2.11 (OTP 20) or later */
_Static_assert(ERL_NIF_MAJOR_VERSION == 2 && ERL_NIF_MINOR_VERSION >= 11, "needs the NIF API of OTP 20 or later (dirty schedulers)");
#endif
_Static_assert(sizeof(nxsig_c64) == 8 && sizeof(float) == 4 && sizeof(double) == 8, "tensor element sizes of the binaries");

static ErlNifResourceType* CTX_RES;
static ErlNifResourceType* BUF_RES;
static ErlNifResourceType* GRP_RES;

typedef struct { nxsig_ctx* ctx; } ctx_res_t;
typedef struct { nxsig_group* grp; } grp_res_t;
/* a device buffer lives on ONE context: a context resource (owner) or member `member` of a group (gowner); the buffer keeps
 * whichever it belongs to alive, so the memory can always be returned to the allocator it came from */
typedef struct { ctx_res_t* owner; grp_res_t* gowner; int member; void* dptr; size_t bytes; } buf_res_t;

static void ctx_dtor(ErlNifEnv* env, void* obj) { (void)env; ctx_res_t* r = obj; if (r->ctx) nxsig_ctx_destroy(r->ctx); }
static void buf_dtor(ErlNifEnv* env, void* obj) {
  (void)env;
  buf_res_t* b = obj;
  if (b->dptr && b->owner && b->owner->ctx) nxsig_free(b->owner->ctx, b->dptr);
  if (b->dptr && b->gowner && b->gowner->grp) {
    nxsig_ctx* c = nxsig_group_ctx(b->gowner->grp, b->member);
    if (c) nxsig_free(c, b->dptr);
  }
  if (b->owner) enif_release_resource(b->owner);
  if (b->gowner) enif_release_resource(b->gowner);
}
static nxsig_ctx* buf_ctx(const buf_res_t* b) {
  if (b->owner) return b->owner->ctx;
  if (b->gowner && b->gowner->grp) return nxsig_group_ctx(b->gowner->grp, b->member);
  return NULL;
}
static void grp_dtor(ErlNifEnv* env, void* obj) { (void)env; grp_res_t* g = obj; if (g->grp) nxsig_group_destroy(g->grp); }

static ERL_NIF_TERM mk_atom(ErlNifEnv* env, const char* a) { return enif_make_atom(env, a); }
static ERL_NIF_TERM mk_error_msg(ErlNifEnv* env, int code, const char* msg) {
  ErlNifBinary b;
  size_t n = msg ? strlen(msg) : 0;
  ERL_NIF_TERM m;
  if (enif_alloc_binary(n, &b)) { if (n) memcpy(b.data, msg, n); m = enif_make_binary(env, &b); }
  else m = mk_atom(env, "enomem");
  return enif_make_tuple2(env, mk_atom(env, "error"), enif_make_tuple2(env, enif_make_int(env, code), m));
}
static ERL_NIF_TERM mk_error(ErlNifEnv* env, int code) { return mk_error_msg(env, code, nxsig_last_error()); }
static ERL_NIF_TERM mk_oom(ErlNifEnv* env) { return mk_error_msg(env, NXSIG_ERR_OOM, "cannot allocate the result binary"); }
static ERL_NIF_TERM mk_ok(ErlNifEnv* env, ERL_NIF_TERM v) { return enif_make_tuple2(env, mk_atom(env, "ok"), v); }

/* result binaries: enif_alloc_binary reports failure instead of aborting the VM; sizes are overflow-checked */
static int mul_size(size_t* acc, uint64_t f) {
  if (f != 0 && *acc > (size_t)-1 / f) return 0;
  *acc *= (size_t)f;
  return 1;
}
static int out_bin(ErlNifBinary* b, uint64_t a, uint64_t c, uint64_t d, uint64_t elem) {
  size_t n = 1;
  if (!mul_size(&n, a) || !mul_size(&n, c) || !mul_size(&n, d) || !mul_size(&n, elem)) return 0;
  if (n > ((size_t)1 << 46)) return 0; /* 64 TiB: far beyond any host; keeps a bad shape from reaching the allocator */
  return enif_alloc_binary(n, b);
}

static int get_ctx(ErlNifEnv* env, ERL_NIF_TERM t, ctx_res_t** out) {
  return enif_get_resource(env, t, CTX_RES, (void**)out) && (*out)->ctx != NULL;
}
static int get_buf(ErlNifEnv* env, ERL_NIF_TERM t, buf_res_t** out) { return enif_get_resource(env, t, BUF_RES, (void**)out); }
static int get_grp(ErlNifEnv* env, ERL_NIF_TERM t, grp_res_t** out) {
  return enif_get_resource(env, t, GRP_RES, (void**)out) && (*out)->grp != NULL;
}
/* Elixir numbers arrive as floats or integers */
static int get_number(ErlNifEnv* env, ERL_NIF_TERM t, double* d) {
  ErlNifSInt64 i;
  if (enif_get_double(env, t, d)) return 1;
  if (enif_get_int64(env, t, &i)) { *d = (double)i; return 1; }
  return 0;
}

/* {n, hop, k, pad_mode, pad_lo, pad_hi, scaling, sampling_rate} -> nxsig_stft_params; geometry is validated here so that
 * no size below is computed from a non-positive length (a negative fft_length reached enif_make_new_binary) */
static int get_params(ErlNifEnv* env, ERL_NIF_TERM t, nxsig_stft_params* p) {
  const ERL_NIF_TERM* e;
  int arity;
  ErlNifSInt64 lo, hi;
  int n, hop, k, pad, scal;
  double fs;
  if (!enif_get_tuple(env, t, &arity, &e) || arity != 8) return 0;
  if (!enif_get_int(env, e[0], &n) || !enif_get_int(env, e[1], &hop) || !enif_get_int(env, e[2], &k) ||
      !enif_get_int(env, e[3], &pad) || !enif_get_int64(env, e[4], &lo) || !enif_get_int64(env, e[5], &hi) ||
      !enif_get_int(env, e[6], &scal) || !get_number(env, e[7], &fs))
    return 0;
  if (n < 1 || k < 1) return 0;
  memset(p, 0, sizeof *p);
  p->frame_length = n; p->hop = hop; p->fft_length = k; p->pad_mode = pad; p->pad_lo = lo; p->pad_hi = hi;
  p->scaling = scal; p->sampling_rate = fs;
  return 1;
}

static ERL_NIF_TERM make_buf(ErlNifEnv* env, ctx_res_t* c, void* d, size_t bytes) {
  buf_res_t* r = enif_alloc_resource(BUF_RES, sizeof *r);
  r->owner = c; enif_keep_resource(c); r->gowner = NULL; r->member = 0; r->dptr = d; r->bytes = bytes;
  ERL_NIF_TERM t = enif_make_resource(env, r);
  enif_release_resource(r);
  return t;
}
/* a buffer on member `member` of a group */
static ERL_NIF_TERM make_gbuf(ErlNifEnv* env, grp_res_t* g, int member, void* d, size_t bytes) {
  buf_res_t* r = enif_alloc_resource(BUF_RES, sizeof *r);
  r->owner = NULL; r->gowner = g; enif_keep_resource(g); r->member = member; r->dptr = d; r->bytes = bytes;
  ERL_NIF_TERM t = enif_make_resource(env, r);
  enif_release_resource(r);
  return t;
}

/* ------------------------------------------------------------------------------------------------ contexts */
static ERL_NIF_TERM nif_device_count(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  (void)argv;
  int n = 0;
  if (argc != 0) return enif_make_badarg(env);
  int rc = nxsig_device_count(&n);
  return rc ? mk_error(env, rc) : mk_ok(env, enif_make_int(env, n));
}

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

static ERL_NIF_TERM nif_sync(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  if (argc != 1 || !get_ctx(env, argv[0], &c)) return enif_make_badarg(env);
  int rc = nxsig_sync(c->ctx);
  return rc ? mk_error(env, rc) : mk_atom(env, "ok");
}

/* last_dispatch(ctx) -> {:ok, binary}: the kernel families of the last compute call on this context ("stft.pair", "istft.wave.deep+
 * istft.edge_chunks" ...; include/nxsig.h: nxsig_ctx_last_dispatch — the per-context copy, because two NIF calls of one Erlang process
 * may run on different scheduler threads).  Diagnostic: NxSignalAMD.last_dispatch/1 */
static ERL_NIF_TERM nif_last_dispatch(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  if (argc != 1 || !get_ctx(env, argv[0], &c)) return enif_make_badarg(env);
  char buf[512];
  int rc = nxsig_ctx_last_dispatch(c->ctx, buf, sizeof buf);
  if (rc) return mk_error(env, rc);
  ErlNifBinary b;
  const size_t n = strlen(buf);
  if (!enif_alloc_binary(n, &b)) return mk_oom(env);
  memcpy(b.data, buf, n);
  return mk_ok(env, enif_make_binary(env, &b));
}

/* ------------------------------------------------------------------------------------------------ host generators */
/* window(kind, n, periodic, beta, eps) -> {:ok, f32 binary}   (NxSignal.Windows.*, BinaryBackend rounding) */
static ERL_NIF_TERM nif_window(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int kind, n, per;
  double beta, eps;
  if (argc != 5 || !enif_get_int(env, argv[0], &kind) || !enif_get_int(env, argv[1], &n) || !enif_get_int(env, argv[2], &per) ||
      !get_number(env, argv[3], &beta) || !get_number(env, argv[4], &eps) || n < 0)
    return enif_make_badarg(env);
  ErlNifBinary b;
  if (!out_bin(&b, (uint64_t)n, 1, 1, 4)) return mk_oom(env);
  int rc = nxsig_window_f32(kind, n, per, beta, eps, (float*)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &b));
}

/* firwin(num_taps, [cutoff], window_kind, beta, pass_zero, scale, sampling_rate) */
static ERL_NIF_TERM nif_firwin(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int taps, kind, pz, sc;
  double beta, fs, cut[64];
  unsigned len;
  if (argc != 7 || !enif_get_int(env, argv[0], &taps) || !enif_get_list_length(env, argv[1], &len) || len > 64 ||
      !enif_get_int(env, argv[2], &kind) || !get_number(env, argv[3], &beta) || !enif_get_int(env, argv[4], &pz) ||
      !enif_get_int(env, argv[5], &sc) || !get_number(env, argv[6], &fs) || taps < 1)
    return enif_make_badarg(env);
  ERL_NIF_TERM head, tail = argv[1];
  for (unsigned i = 0; i < len; ++i)
    if (!enif_get_list_cell(env, tail, &head, &tail) || !get_number(env, head, &cut[i])) return enif_make_badarg(env);
  ErlNifBinary b;
  if (!out_bin(&b, (uint64_t)taps, 1, 1, 4)) return mk_oom(env);
  int rc = nxsig_firwin_f32(taps, cut, (int)len, kind, beta, pz, sc, fs, (float*)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &b));
}

/* fft_frequencies(sampling_rate, fft_length, endpoint) -> {:ok, f32 binary}   (lib/nx_signal.ex:154-166) */
static ERL_NIF_TERM nif_fft_frequencies(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  double fs;
  int k, endpoint;
  if (argc != 3 || !get_number(env, argv[0], &fs) || !enif_get_int(env, argv[1], &k) || !enif_get_int(env, argv[2], &endpoint) || k < 1)
    return enif_make_badarg(env);
  ErlNifBinary b;
  if (!out_bin(&b, (uint64_t)k, 1, 1, 4)) return mk_oom(env);
  int rc = nxsig_fft_frequencies_f32(fs, k, endpoint, (float*)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &b));
}

/* mel_filters(fft_length, mel_bins, sampling_rate, max_mel, mel_frequency_spacing) -> {:ok, f32[mel_bins][fft_length]} (:397-445) */
static ERL_NIF_TERM nif_mel_filters(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int k, bins;
  double fs, max_mel, sp;
  if (argc != 5 || !enif_get_int(env, argv[0], &k) || !enif_get_int(env, argv[1], &bins) || !get_number(env, argv[2], &fs) ||
      !get_number(env, argv[3], &max_mel) || !get_number(env, argv[4], &sp) || k < 1 || bins < 1)
    return enif_make_badarg(env);
  ErlNifBinary b;
  if (!out_bin(&b, (uint64_t)k, (uint64_t)bins, 1, 4)) return mk_oom(env);
  int rc = nxsig_mel_filters_f32(k, bins, fs, max_mel, sp, (float*)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &b));
}

/* sinc(t_bin) -> {:ok, f32 binary}   (lib/nx_signal/waveforms.ex:451-457) */
static ERL_NIF_TERM nif_sinc(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ErlNifBinary t, b;
  if (argc != 1 || !enif_inspect_binary(env, argv[0], &t) || t.size % 4) return enif_make_badarg(env);
  if (!out_bin(&b, t.size / 4, 1, 1, 4)) return mk_oom(env);
  int rc = nxsig_sinc_f32((const float*)t.data, (int64_t)(t.size / 4), (float*)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &b));
}

/* ------------------------------------------------------------------------------------------------ hot path, host tensors */
/* stft(ctx, x_bin, length, batch, window_bin, params) -> {:ok, z_bin, num_frames, times_bin, freqs_bin}
 * stft_c64: the same with x_bin holding c64 samples (interleaved f32 re, im) — lib/nx_signal.ex:94-102 on complex data */
static ERL_NIF_TERM stft_host(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[], size_t elem) {
  ctx_res_t* c;
  ErlNifBinary x, w, zb, tb, fb;
  ErlNifSInt64 length;
  int batch;
  nxsig_stft_params p;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size / elem / (size_t)batch != (size_t)length || x.size % elem || w.size != (size_t)p.frame_length * 4)
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  if (!out_bin(&zb, (uint64_t)batch, (uint64_t)m, (uint64_t)p.fft_length, 8)) return mk_oom(env);
  int rc = elem == 8 ? nxsig_stft_c64(c->ctx, (const nxsig_c64*)x.data, length, batch, length, (const float*)w.data, &p, (nxsig_c64*)zb.data, NULL, NXSIG_HOST)
                     : nxsig_stft_f32(c->ctx, (const float*)x.data, length, batch, length, (const float*)w.data, &p, (nxsig_c64*)zb.data, NULL, NXSIG_HOST);
  if (rc) { enif_release_binary(&zb); return mk_error(env, rc); }
  if (!out_bin(&tb, (uint64_t)m, 1, 1, 4)) { enif_release_binary(&zb); return mk_oom(env); }
  if (!out_bin(&fb, (uint64_t)p.fft_length, 1, 1, 4)) { enif_release_binary(&zb); enif_release_binary(&tb); return mk_oom(env); }
  if ((rc = nxsig_stft_times_f32(p.frame_length, p.sampling_rate, m, (float*)tb.data)) ||
      (rc = nxsig_fft_frequencies_f32(p.sampling_rate, p.fft_length, 0, (float*)fb.data))) {
    enif_release_binary(&zb); enif_release_binary(&tb); enif_release_binary(&fb);
    return mk_error(env, rc);
  }
  return enif_make_tuple5(env, mk_atom(env, "ok"), enif_make_binary(env, &zb), enif_make_int64(env, m), enif_make_binary(env, &tb),
                          enif_make_binary(env, &fb));
}

static ERL_NIF_TERM nif_stft(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) { return stft_host(env, argc, argv, 4); }
static ERL_NIF_TERM nif_stft_c64(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) { return stft_host(env, argc, argv, 8); }

/* istft(ctx, z_bin, num_frames, batch, window_bin, params) -> {:ok, y_bin} */
static ERL_NIF_TERM nif_istft(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary z, w, yb;
  ErlNifSInt64 m;
  int batch;
  nxsig_stft_params p;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &z) || !enif_get_int64(env, argv[2], &m) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p))
    return enif_make_badarg(env);
  if (batch < 1 || m < 1 || z.size % 8 || z.size / 8 / (size_t)batch / (size_t)m != (size_t)p.fft_length ||
      z.size / 8 % ((size_t)batch * (size_t)m) || w.size != (size_t)p.frame_length * 4)
    return enif_make_badarg(env);
  int64_t n = nxsig_ola_length(m, p.frame_length, p.hop);
  if (n < 0) return mk_error(env, (int)n);
  if (!out_bin(&yb, (uint64_t)batch, (uint64_t)n, 1, 8)) return mk_oom(env);
  int rc = nxsig_istft_c64(c->ctx, (const nxsig_c64*)z.data, m, batch, (const float*)w.data, &p, (nxsig_c64*)yb.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&yb); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &yb));
}

/* istft_filtered(ctx, z_bin, num_frames, batch, window_bin, params, h_bin) -> {:ok, y_bin}
 * NxSignal.istft(Nx.multiply(z, hfft), window, opts) in one library call (guides/filtering.livemd:141 + :150-157) */
static ERL_NIF_TERM nif_istft_filtered(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary z, w, h, yb;
  ErlNifSInt64 m;
  int batch;
  nxsig_stft_params p;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &z) || !enif_get_int64(env, argv[2], &m) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p) ||
      !enif_inspect_binary(env, argv[6], &h))
    return enif_make_badarg(env);
  if (batch < 1 || m < 1 || p.fft_length < 1 || z.size % 8 || z.size / 8 / (size_t)batch / (size_t)m != (size_t)p.fft_length ||
      z.size / 8 % ((size_t)batch * (size_t)m) || w.size != (size_t)p.frame_length * 4 || h.size != (size_t)p.fft_length * 8)
    return enif_make_badarg(env);
  int64_t n = nxsig_ola_length(m, p.frame_length, p.hop);
  if (n < 0) return mk_error(env, (int)n);
  if (!out_bin(&yb, (uint64_t)batch, (uint64_t)n, 1, 8)) return mk_oom(env);
  int rc = nxsig_istft_filtered_c64(c->ctx, (const nxsig_c64*)z.data, m, batch, (const float*)w.data, &p, (const nxsig_c64*)h.data,
                                    (nxsig_c64*)yb.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&yb); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &yb));
}

/* fir(ctx, x_bin, length, batch, taps_bin, mode) -> {:ok, y_bin}   (Convolution.convolve(method: :fft), real 1-D rows) */
static ERL_NIF_TERM nif_fir(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary x, h, yb;
  ErlNifSInt64 length;
  int batch, mode;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &h) || !enif_get_int(env, argv[5], &mode))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size % 4 || x.size / 4 / (size_t)batch != (size_t)length || h.size < 4 || h.size % 4 ||
      h.size / 4 > 0x7fffffff)
    return enif_make_badarg(env);
  int64_t n = nxsig_conv_length(length, (int64_t)(h.size / 4), mode);
  if (n < 0) return mk_error(env, (int)n);
  if (!out_bin(&yb, (uint64_t)batch, (uint64_t)n, 1, 4)) return mk_oom(env);
  int rc = nxsig_fir_f32(c->ctx, (const float*)x.data, length, batch, length, (const float*)h.data, (int)(h.size / 4), mode,
                         (float*)yb.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&yb); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &yb));
}

/* as_windowed(ctx, x_bin, length, batch, window_length, stride, pad_mode, pad_lo, pad_hi) -> {:ok, frames_bin, num_frames} */
static ERL_NIF_TERM nif_as_windowed(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary x, ob;
  ErlNifSInt64 length, lo, hi;
  int batch, wl, stride, pad;
  if (argc != 9 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_get_int(env, argv[4], &wl) || !enif_get_int(env, argv[5], &stride) ||
      !enif_get_int(env, argv[6], &pad) || !enif_get_int64(env, argv[7], &lo) || !enif_get_int64(env, argv[8], &hi))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size % 4 || x.size / 4 / (size_t)batch != (size_t)length) return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, wl, stride, pad, lo, hi);
  if (m < 0) return mk_error(env, (int)m);
  if (!out_bin(&ob, (uint64_t)batch, (uint64_t)m, (uint64_t)wl, 4)) return mk_oom(env);
  int rc = nxsig_as_windowed_f32(c->ctx, (const float*)x.data, length, batch, length, wl, stride, pad, lo, hi, (float*)ob.data, NULL, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), enif_make_binary(env, &ob), enif_make_int64(env, m));
}

/* overlap_and_add(ctx, frames_bin, num_frames, batch, frame_length, overlap_length, components) -> {:ok, out_bin} */
static ERL_NIF_TERM nif_overlap_and_add(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary f, ob;
  ErlNifSInt64 m;
  int batch, n, overlap, comps;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &f) || !enif_get_int64(env, argv[2], &m) ||
      !enif_get_int(env, argv[3], &batch) || !enif_get_int(env, argv[4], &n) || !enif_get_int(env, argv[5], &overlap) ||
      !enif_get_int(env, argv[6], &comps))
    return enif_make_badarg(env);
  if (batch < 1 || m < 1 || n < 1 || (comps != 1 && comps != 2) || f.size % ((size_t)4 * (size_t)comps) ||
      f.size / 4 / (size_t)comps / (size_t)batch / (size_t)m != (size_t)n || f.size / 4 / (size_t)comps % ((size_t)batch * (size_t)m))
    return enif_make_badarg(env);
  /* the reference's own check (lib/nx_signal.ex:692-695) must surface as ArgumentError, not badarg: let the library say it */
  int64_t out_len = (overlap >= 0 && overlap < n) ? m * (int64_t)(n - overlap) + overlap : 1;
  if (!out_bin(&ob, (uint64_t)batch, (uint64_t)out_len, (uint64_t)comps, 4)) return mk_oom(env);
  int rc = nxsig_overlap_and_add(c->ctx, (const float*)f.data, m, batch, n, overlap, comps, (float*)ob.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &ob));
}

/* fft(ctx, in_bin, in_is_real, rows, n_in, fft_length, inverse) -> {:ok, c64 binary}   (Nx.fft / Nx.ifft over the last axis) */
static ERL_NIF_TERM nif_fft(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary in, ob;
  ErlNifSInt64 rows;
  int is_real, n_in, k, inv;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &in) || !enif_get_int(env, argv[2], &is_real) ||
      !enif_get_int64(env, argv[3], &rows) || !enif_get_int(env, argv[4], &n_in) || !enif_get_int(env, argv[5], &k) ||
      !enif_get_int(env, argv[6], &inv))
    return enif_make_badarg(env);
  const size_t es = is_real ? 4 : 8;
  if (rows < 1 || n_in < 1 || k < 1 || in.size % es || in.size / es / (size_t)rows != (size_t)n_in || in.size / es % (size_t)rows)
    return enif_make_badarg(env);
  if (!out_bin(&ob, (uint64_t)rows, (uint64_t)k, 1, 8)) return mk_oom(env);
  int rc = nxsig_fft(c->ctx, in.data, is_real, rows, n_in, k, inv, (nxsig_c64*)ob.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &ob));
}

/* ------------------------------------------------------------------------------------------------ f64 / c128 tier
 * The reference computes in the type of its operands (include/nxsig.h "f64 / c128 tier"): f64 / c128 binaries in, the same out. */
/* window_f64(kind, n, periodic, beta, eps) -> {:ok, f64 binary}   (NxSignal.Windows.*(n, type: {:f, 64})) */
static ERL_NIF_TERM nif_window_f64(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int kind, n, per;
  double beta, eps;
  if (argc != 5 || !enif_get_int(env, argv[0], &kind) || !enif_get_int(env, argv[1], &n) || !enif_get_int(env, argv[2], &per) ||
      !get_number(env, argv[3], &beta) || !get_number(env, argv[4], &eps) || n < 0)
    return enif_make_badarg(env);
  ErlNifBinary b;
  if (!out_bin(&b, (uint64_t)n, 1, 1, 8)) return mk_oom(env);
  int rc = nxsig_window_f64(kind, n, per, beta, eps, (double*)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &b));
}

/* firwin_f64(num_taps, [cutoff], window_kind, beta, pass_zero, scale, sampling_rate) -> {:ok, f64 binary} */
static ERL_NIF_TERM nif_firwin_f64(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int taps, kind, pz, sc;
  double beta, fs, cut[64];
  unsigned len;
  if (argc != 7 || !enif_get_int(env, argv[0], &taps) || !enif_get_list_length(env, argv[1], &len) || len > 64 ||
      !enif_get_int(env, argv[2], &kind) || !get_number(env, argv[3], &beta) || !enif_get_int(env, argv[4], &pz) ||
      !enif_get_int(env, argv[5], &sc) || !get_number(env, argv[6], &fs) || taps < 1)
    return enif_make_badarg(env);
  ERL_NIF_TERM head, tail = argv[1];
  for (unsigned i = 0; i < len; ++i)
    if (!enif_get_list_cell(env, tail, &head, &tail) || !get_number(env, head, &cut[i])) return enif_make_badarg(env);
  ErlNifBinary b;
  if (!out_bin(&b, (uint64_t)taps, 1, 1, 8)) return mk_oom(env);
  int rc = nxsig_firwin_f64(taps, cut, (int)len, kind, beta, pz, sc, fs, (double*)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &b));
}

/* fft_frequencies_f64(sampling_rate, fft_length, endpoint) -> {:ok, f64 binary} */
static ERL_NIF_TERM nif_fft_frequencies_f64(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  double fs;
  int k, endpoint;
  if (argc != 3 || !get_number(env, argv[0], &fs) || !enif_get_int(env, argv[1], &k) || !enif_get_int(env, argv[2], &endpoint) || k < 1)
    return enif_make_badarg(env);
  ErlNifBinary b;
  if (!out_bin(&b, (uint64_t)k, 1, 1, 8)) return mk_oom(env);
  int rc = nxsig_fft_frequencies_f64(fs, k, endpoint, (double*)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &b));
}

/* sinc_f64(t_bin) -> {:ok, f64 binary} */
static ERL_NIF_TERM nif_sinc_f64(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ErlNifBinary t, b;
  if (argc != 1 || !enif_inspect_binary(env, argv[0], &t) || t.size % 8) return enif_make_badarg(env);
  if (!out_bin(&b, t.size / 8, 1, 1, 8)) return mk_oom(env);
  int rc = nxsig_sinc_f64((const double*)t.data, (int64_t)(t.size / 8), (double*)b.data);
  if (rc) { enif_release_binary(&b); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &b));
}

/* stft_f64(ctx, x_bin (f64), length, batch, window_bin, window_is_f64, params) -> {:ok, z_bin (c128), num_frames, times_bin, freqs_bin}
 * times / frequencies stay f32 like the reference's (lib/nx_signal.ex:106-111) */
static ERL_NIF_TERM nif_stft_f64(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary x, w, zb, tb, fb;
  ErlNifSInt64 length;
  int batch, wf64;
  nxsig_stft_params p;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !enif_get_int(env, argv[5], &wf64) ||
      !get_params(env, argv[6], &p))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size % 8 || x.size / 8 / (size_t)batch != (size_t)length ||
      w.size != (size_t)p.frame_length * (wf64 ? 8 : 4))
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  if (!out_bin(&zb, (uint64_t)batch, (uint64_t)m, (uint64_t)p.fft_length, 16)) return mk_oom(env);
  int rc = nxsig_stft_f64(c->ctx, (const double*)x.data, length, batch, length, w.data, wf64, &p, (nxsig_c128*)zb.data, NULL, NXSIG_HOST);
  if (rc) { enif_release_binary(&zb); return mk_error(env, rc); }
  if (!out_bin(&tb, (uint64_t)m, 1, 1, 4)) { enif_release_binary(&zb); return mk_oom(env); }
  if (!out_bin(&fb, (uint64_t)p.fft_length, 1, 1, 4)) { enif_release_binary(&zb); enif_release_binary(&tb); return mk_oom(env); }
  if ((rc = nxsig_stft_times_f32(p.frame_length, p.sampling_rate, m, (float*)tb.data)) ||
      (rc = nxsig_fft_frequencies_f32(p.sampling_rate, p.fft_length, 0, (float*)fb.data))) {
    enif_release_binary(&zb); enif_release_binary(&tb); enif_release_binary(&fb);
    return mk_error(env, rc);
  }
  return enif_make_tuple5(env, mk_atom(env, "ok"), enif_make_binary(env, &zb), enif_make_int64(env, m), enif_make_binary(env, &tb),
                          enif_make_binary(env, &fb));
}

/* stft_c128(ctx, x_bin (c128), length, batch, window_bin, window_is_f64, params) -> {:ok, z_bin (c128), num_frames, times_bin, freqs_bin}
 * complex f64 samples (lib/nx_signal.ex:94-102 on a c128 tensor, or c64 samples under an f64 window) */
static ERL_NIF_TERM nif_stft_c128(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary x, w, zb, tb, fb;
  ErlNifSInt64 length;
  int batch, wf64;
  nxsig_stft_params p;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !enif_get_int(env, argv[5], &wf64) ||
      !get_params(env, argv[6], &p))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size % 16 || x.size / 16 / (size_t)batch != (size_t)length ||
      w.size != (size_t)p.frame_length * (wf64 ? 8 : 4))
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  if (!out_bin(&zb, (uint64_t)batch, (uint64_t)m, (uint64_t)p.fft_length, 16)) return mk_oom(env);
  int rc = nxsig_stft_c128(c->ctx, (const nxsig_c128*)x.data, length, batch, length, w.data, wf64, &p, (nxsig_c128*)zb.data, NULL, NXSIG_HOST);
  if (rc) { enif_release_binary(&zb); return mk_error(env, rc); }
  if (!out_bin(&tb, (uint64_t)m, 1, 1, 4)) { enif_release_binary(&zb); return mk_oom(env); }
  if (!out_bin(&fb, (uint64_t)p.fft_length, 1, 1, 4)) { enif_release_binary(&zb); enif_release_binary(&tb); return mk_oom(env); }
  if ((rc = nxsig_stft_times_f32(p.frame_length, p.sampling_rate, m, (float*)tb.data)) ||
      (rc = nxsig_fft_frequencies_f32(p.sampling_rate, p.fft_length, 0, (float*)fb.data))) {
    enif_release_binary(&zb); enif_release_binary(&tb); enif_release_binary(&fb);
    return mk_error(env, rc);
  }
  return enif_make_tuple5(env, mk_atom(env, "ok"), enif_make_binary(env, &zb), enif_make_int64(env, m), enif_make_binary(env, &tb),
                          enif_make_binary(env, &fb));
}

/* istft_c128(ctx, z_bin (c128), num_frames, batch, window_bin, window_is_f64, params) -> {:ok, y_bin (c128)} */
static ERL_NIF_TERM nif_istft_c128(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary z, w, yb;
  ErlNifSInt64 m;
  int batch, wf64;
  nxsig_stft_params p;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &z) || !enif_get_int64(env, argv[2], &m) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !enif_get_int(env, argv[5], &wf64) ||
      !get_params(env, argv[6], &p))
    return enif_make_badarg(env);
  if (batch < 1 || m < 1 || z.size % 16 || z.size / 16 / (size_t)batch / (size_t)m != (size_t)p.fft_length ||
      z.size / 16 % ((size_t)batch * (size_t)m) || w.size != (size_t)p.frame_length * (wf64 ? 8 : 4))
    return enif_make_badarg(env);
  int64_t n = nxsig_ola_length(m, p.frame_length, p.hop);
  if (n < 0) return mk_error(env, (int)n);
  if (!out_bin(&yb, (uint64_t)batch, (uint64_t)n, 1, 16)) return mk_oom(env);
  int rc = nxsig_istft_c128(c->ctx, (const nxsig_c128*)z.data, m, batch, w.data, wf64, &p, (nxsig_c128*)yb.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&yb); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &yb));
}

/* fir_f64(ctx, x_bin (f64), length, batch, taps_bin (f64), mode) -> {:ok, y_bin (f64)} */
static ERL_NIF_TERM nif_fir_f64(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary x, h, yb;
  ErlNifSInt64 length;
  int batch, mode;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &h) || !enif_get_int(env, argv[5], &mode))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size % 8 || x.size / 8 / (size_t)batch != (size_t)length || h.size < 8 || h.size % 8 ||
      h.size / 8 > 0x7fffffff)
    return enif_make_badarg(env);
  int64_t n = nxsig_conv_length(length, (int64_t)(h.size / 8), mode);
  if (n < 0) return mk_error(env, (int)n);
  if (!out_bin(&yb, (uint64_t)batch, (uint64_t)n, 1, 8)) return mk_oom(env);
  int rc = nxsig_fir_f64(c->ctx, (const double*)x.data, length, batch, length, (const double*)h.data, (int)(h.size / 8), mode,
                         (double*)yb.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&yb); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &yb));
}

/* fft_c128(ctx, in_bin, in_is_real, rows, n_in, fft_length, inverse) -> {:ok, c128 binary}   (Nx.fft / Nx.ifft of f64 / c128 rows) */
static ERL_NIF_TERM nif_fft_c128(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary in, ob;
  ErlNifSInt64 rows;
  int is_real, n_in, k, inv;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &in) || !enif_get_int(env, argv[2], &is_real) ||
      !enif_get_int64(env, argv[3], &rows) || !enif_get_int(env, argv[4], &n_in) || !enif_get_int(env, argv[5], &k) ||
      !enif_get_int(env, argv[6], &inv))
    return enif_make_badarg(env);
  const size_t es = is_real ? 8 : 16;
  if (rows < 1 || n_in < 1 || k < 1 || in.size % es || in.size / es / (size_t)rows != (size_t)n_in || in.size / es % (size_t)rows)
    return enif_make_badarg(env);
  if (!out_bin(&ob, (uint64_t)rows, (uint64_t)k, 1, 16)) return mk_oom(env);
  int rc = nxsig_fft_c128(c->ctx, in.data, is_real, rows, n_in, k, inv, (nxsig_c128*)ob.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &ob));
}

/* as_windowed_f64(ctx, x_bin, length, batch, window_length, stride, pad_mode, pad_lo, pad_hi) -> {:ok, frames_bin, num_frames} */
static ERL_NIF_TERM nif_as_windowed_f64(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary x, ob;
  ErlNifSInt64 length, lo, hi;
  int batch, wl, stride, pad;
  if (argc != 9 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_get_int(env, argv[4], &wl) || !enif_get_int(env, argv[5], &stride) ||
      !enif_get_int(env, argv[6], &pad) || !enif_get_int64(env, argv[7], &lo) || !enif_get_int64(env, argv[8], &hi))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size % 8 || x.size / 8 / (size_t)batch != (size_t)length) return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, wl, stride, pad, lo, hi);
  if (m < 0) return mk_error(env, (int)m);
  if (!out_bin(&ob, (uint64_t)batch, (uint64_t)m, (uint64_t)wl, 8)) return mk_oom(env);
  int rc = nxsig_as_windowed_f64(c->ctx, (const double*)x.data, length, batch, length, wl, stride, pad, lo, hi, (double*)ob.data, NULL,
                                 NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), enif_make_binary(env, &ob), enif_make_int64(env, m));
}

/* overlap_and_add_f64(ctx, frames_bin, num_frames, batch, frame_length, overlap_length, components) -> {:ok, out_bin} */
static ERL_NIF_TERM nif_overlap_and_add_f64(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary f, ob;
  ErlNifSInt64 m;
  int batch, n, overlap, comps;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &f) || !enif_get_int64(env, argv[2], &m) ||
      !enif_get_int(env, argv[3], &batch) || !enif_get_int(env, argv[4], &n) || !enif_get_int(env, argv[5], &overlap) ||
      !enif_get_int(env, argv[6], &comps))
    return enif_make_badarg(env);
  if (batch < 1 || m < 1 || n < 1 || (comps != 1 && comps != 2) || f.size % ((size_t)8 * (size_t)comps) ||
      f.size / 8 / (size_t)comps / (size_t)batch / (size_t)m != (size_t)n || f.size / 8 / (size_t)comps % ((size_t)batch * (size_t)m))
    return enif_make_badarg(env);
  int64_t out_len = (overlap >= 0 && overlap < n) ? m * (int64_t)(n - overlap) + overlap : 1;
  if (!out_bin(&ob, (uint64_t)batch, (uint64_t)out_len, (uint64_t)comps, 8)) return mk_oom(env);
  int rc = nxsig_overlap_and_add_f64(c->ctx, (const double*)f.data, m, batch, n, overlap, comps, (double*)ob.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &ob));
}

/* fftconvolve_c64(ctx, a_bin, b_bin, mode) -> {:ok, c64 binary}   (1-D complex case of Convolution.fftconvolve/3) */
static ERL_NIF_TERM nif_fftconvolve_c64(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary a, b, ob;
  int mode;
  if (argc != 4 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &a) || !enif_inspect_binary(env, argv[2], &b) ||
      !enif_get_int(env, argv[3], &mode) || a.size < 8 || b.size < 8 || a.size % 8 || b.size % 8)
    return enif_make_badarg(env);
  int64_t n = nxsig_conv_length((int64_t)(a.size / 8), (int64_t)(b.size / 8), mode);
  if (n < 0) return mk_error(env, (int)n);
  if (!out_bin(&ob, (uint64_t)n, 1, 1, 8)) return mk_oom(env);
  int rc = nxsig_fftconvolve_c64(c->ctx, (const nxsig_c64*)a.data, (int64_t)(a.size / 8), (const nxsig_c64*)b.data, (int64_t)(b.size / 8), mode,
                                 (nxsig_c64*)ob.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &ob));
}

static int get_i64_list(ErlNifEnv* env, ERL_NIF_TERM list, int64_t* out, unsigned cap, unsigned* n) {
  ERL_NIF_TERM head, tail = list;
  if (!enif_get_list_length(env, list, n) || *n > cap) return 0;
  for (unsigned i = 0; i < *n; ++i) {
    ErlNifSInt64 v;
    if (!enif_get_list_cell(env, tail, &head, &tail) || !enif_get_int64(env, head, &v)) return 0;
    out[i] = v;
  }
  return 1;
}

/* fft_nd(ctx, in_bin, in_is_real, shape, axes, lengths, inverse) -> {:ok, c64 binary}   (Transforms.fft_nd / ifft_nd, any axes) */
static ERL_NIF_TERM nif_fft_nd(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary in, ob;
  int is_real, inv;
  int64_t shape[8], axes64[16], lengths[16], osh[8];
  int32_t axes[16];
  unsigned rank, na, nl;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &in) || !enif_get_int(env, argv[2], &is_real) ||
      !get_i64_list(env, argv[3], shape, 8, &rank) || !get_i64_list(env, argv[4], axes64, 16, &na) ||
      !get_i64_list(env, argv[5], lengths, 16, &nl) || !enif_get_int(env, argv[6], &inv) || rank < 1 || na != nl)
    return enif_make_badarg(env);
  size_t n_in = 1, n_out = 1;
  for (unsigned d = 0; d < rank; ++d) { if (shape[d] < 1 || !mul_size(&n_in, (uint64_t)shape[d])) return enif_make_badarg(env); osh[d] = shape[d]; }
  if (in.size % (is_real ? 4 : 8) || in.size / (is_real ? 4 : 8) != n_in) return enif_make_badarg(env);
  for (unsigned i = 0; i < na; ++i) {
    int64_t ax = axes64[i] < 0 ? axes64[i] + (int64_t)rank : axes64[i];
    if (ax < 0 || ax >= (int64_t)rank || lengths[i] < 1) return enif_make_badarg(env);
    axes[i] = (int32_t)ax; osh[ax] = lengths[i];
  }
  for (unsigned d = 0; d < rank; ++d) if (!mul_size(&n_out, (uint64_t)osh[d])) return mk_oom(env);
  if (!out_bin(&ob, n_out, 1, 1, 8)) return mk_oom(env);
  int rc = nxsig_fft_nd(c->ctx, in.data, is_real, shape, (int32_t)rank, axes, lengths, (int32_t)na, inv, (nxsig_c64*)ob.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &ob));
}

/* fftconvolve_nd(ctx, a_bin, a_is_real, a_shape, b_bin, b_is_real, b_shape, mode) -> {:ok, out_bin, out_shape}
 * (Convolution.fftconvolve/3 for operands of equal rank; out is f32 when both are real, else c64) */
typedef int (*conv_nd_fn)(nxsig_ctx*, const void*, int32_t, const int64_t*, const void*, int32_t, const int64_t*, int32_t, int32_t, void*,
                          int64_t*, int32_t);
static ERL_NIF_TERM conv_nd_common(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[], conv_nd_fn fn, const char* rank_msg) {
  ctx_res_t* c;
  ErlNifBinary a, b, ob;
  int a_real, b_real, mode;
  int64_t s1[8], s2[8], osh[8];
  unsigned r1, r2;
  if (argc != 8 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &a) || !enif_get_int(env, argv[2], &a_real) ||
      !get_i64_list(env, argv[3], s1, 8, &r1) || !enif_inspect_binary(env, argv[4], &b) || !enif_get_int(env, argv[5], &b_real) ||
      !get_i64_list(env, argv[6], s2, 8, &r2) || !enif_get_int(env, argv[7], &mode) || r1 < 1)
    return enif_make_badarg(env);
  if (r1 != r2) return mk_error_msg(env, NXSIG_ERR_INVALID_ARG, rank_msg);
  size_t na = 1, nb = 1, nfull = 1;
  for (unsigned d = 0; d < r1; ++d) {
    if (s1[d] < 1 || s2[d] < 1 || !mul_size(&na, (uint64_t)s1[d]) || !mul_size(&nb, (uint64_t)s2[d]) ||
        !mul_size(&nfull, (uint64_t)(s1[d] + s2[d] - 1)))
      return enif_make_badarg(env);
  }
  if (a.size % (a_real ? 4 : 8) || a.size / (a_real ? 4 : 8) != na || b.size % (b_real ? 4 : 8) || b.size / (b_real ? 4 : 8) != nb)
    return enif_make_badarg(env);
  const size_t es = (a_real && b_real) ? 4 : 8;
  if (!out_bin(&ob, nfull, 1, 1, es)) return mk_oom(env);  /* every mode's result fits the full size */
  int rc = fn(c->ctx, a.data, a_real, s1, b.data, b_real, s2, (int32_t)r1, mode, ob.data, osh, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  size_t nres = 1;
  ERL_NIF_TERM dims[8];
  for (unsigned d = 0; d < r1; ++d) { nres *= (size_t)osh[d]; dims[d] = enif_make_int64(env, osh[d]); }
  if (!enif_realloc_binary(&ob, nres * es)) { enif_release_binary(&ob); return mk_oom(env); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), enif_make_binary(env, &ob), enif_make_list_from_array(env, dims, r1));
}
static ERL_NIF_TERM nif_fftconvolve_nd(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  return conv_nd_common(env, argc, argv, nxsig_fftconvolve_nd, "Rank of in1 and in2 must be equal.");  /* convolution.ex:295-296 */
}
/* convolve_direct(ctx, a_bin, a_is_real, a_shape, b_bin, b_is_real, b_shape, mode) -> {:ok, out_bin, out_shape}
 * (Convolution.convolve/3 with its default method: time-domain sums, lib/nx_signal/convolution.ex:95-218) */
static ERL_NIF_TERM nif_convolve_direct(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  return conv_nd_common(env, argc, argv, nxsig_convolve_direct,
                        "NxSignal.convolve/3 requires both inputs to have the same rank or one of them to be a scalar");  /* :112-115 */
}

/* stft_to_mel(ctx, z_bin, rows, fft_length, mel_bins, filters_bin) -> {:ok, f32[rows][mel_bins]}   (lib/nx_signal.ex:486-513) */
static ERL_NIF_TERM nif_stft_to_mel(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary z, f, ob;
  ErlNifSInt64 rows;
  int k, bins;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &z) || !enif_get_int64(env, argv[2], &rows) ||
      !enif_get_int(env, argv[3], &k) || !enif_get_int(env, argv[4], &bins) || !enif_inspect_binary(env, argv[5], &f))
    return enif_make_badarg(env);
  if (rows < 1 || k < 1 || bins < 1 || z.size % 8 || z.size / 8 / (size_t)rows != (size_t)k || z.size / 8 % (size_t)rows ||
      f.size % 4 || f.size / 4 / (size_t)bins != (size_t)k || f.size / 4 % (size_t)bins)
    return enif_make_badarg(env);
  if (!out_bin(&ob, (uint64_t)rows, (uint64_t)bins, 1, 4)) return mk_oom(env);
  int rc = nxsig_stft_to_mel(c->ctx, (const nxsig_c64*)z.data, rows, k, bins, (const float*)f.data, (float*)ob.data, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &ob));
}

/* stft_mel(ctx, x_bin, length, batch, window_bin, params, mel_bins, filters_bin) -> {:ok, f32[batch][M][mel_bins], M}  (fused) */
static ERL_NIF_TERM nif_stft_mel(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary x, w, f, ob;
  ErlNifSInt64 length;
  int batch, bins;
  nxsig_stft_params p;
  if (argc != 8 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p) ||
      !enif_get_int(env, argv[6], &bins) || !enif_inspect_binary(env, argv[7], &f))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || bins < 1 || x.size % 4 || x.size / 4 / (size_t)batch != (size_t)length ||
      w.size != (size_t)p.frame_length * 4 || f.size % 4 || f.size / 4 / (size_t)bins != (size_t)p.fft_length || f.size / 4 % (size_t)bins)
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  if (!out_bin(&ob, (uint64_t)batch, (uint64_t)m, (uint64_t)bins, 4)) return mk_oom(env);
  int rc = nxsig_stft_mel_f32(c->ctx, (const float*)x.data, length, batch, length, (const float*)w.data, &p, bins, (const float*)f.data,
                              (float*)ob.data, NULL, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), enif_make_binary(env, &ob), enif_make_int64(env, m));
}

/* stft_magnitude(ctx, x_bin, length, batch, window_bin, params, kind) -> {:ok, f32[batch][M][fft_length / 2], M}
 * (fused |s| / |s|^2 / dBFS spectrogram of guides/spectrogram.livemd:76-92; kind 0 abs, 1 power, 2 dbfs) */
static ERL_NIF_TERM nif_stft_magnitude(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary x, w, ob;
  ErlNifSInt64 length;
  int batch, kind;
  nxsig_stft_params p;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p) ||
      !enif_get_int(env, argv[6], &kind))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size % 4 || x.size / 4 / (size_t)batch != (size_t)length || w.size != (size_t)p.frame_length * 4 ||
      p.fft_length < 2)
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  if (!out_bin(&ob, (uint64_t)batch, (uint64_t)m, (uint64_t)(p.fft_length / 2), 4)) return mk_oom(env);
  int rc = nxsig_stft_magnitude_f32(c->ctx, (const float*)x.data, length, batch, length, (const float*)w.data, &p, kind, (float*)ob.data, NULL, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), enif_make_binary(env, &ob), enif_make_int64(env, m));
}

/* ------------------------------------------------------------------------------------------------ device-resident tensors
 * keep stft -> edit -> istft chains in HBM (guides/filtering.livemd:137-159).  Device calls are
 * asynchronous on the context's stream; from_device synchronises. */
static ERL_NIF_TERM nif_to_device(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  ErlNifBinary b;
  if (argc != 2 || !get_ctx(env, argv[0], &c) || !enif_inspect_binary(env, argv[1], &b)) return enif_make_badarg(env);
  void* d = NULL;
  int rc = nxsig_alloc(c->ctx, b.size, &d);
  if (rc) return mk_error(env, rc);
  if ((rc = nxsig_upload(c->ctx, d, b.data, b.size))) { nxsig_free(c->ctx, d); return mk_error(env, rc); }
  return mk_ok(env, make_buf(env, c, d, b.size));
}

static ERL_NIF_TERM nif_from_device(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  buf_res_t* b;
  ErlNifBinary ob;
  if (argc != 1 || !get_buf(env, argv[0], &b)) return enif_make_badarg(env);
  nxsig_ctx* bc = buf_ctx(b);
  if (!bc) return enif_make_badarg(env);
  if (!enif_alloc_binary(b->bytes, &ob)) return mk_oom(env);
  int rc = nxsig_download(bc, ob.data, b->dptr, b->bytes);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &ob));
}

static ERL_NIF_TERM nif_buf_size(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  buf_res_t* b;
  if (argc != 1 || !get_buf(env, argv[0], &b)) return enif_make_badarg(env);
  return enif_make_int64(env, (ErlNifSInt64)b->bytes);
}

/* stft_dev(ctx, x_buf, length, batch, window_bin, params) -> {:ok, z_buf, num_frames}; stft_c64_dev: x_buf holds c64 samples */
static ERL_NIF_TERM stft_dev_impl(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[], size_t elem) {
  ctx_res_t* c;
  buf_res_t* x;
  ErlNifBinary w;
  ErlNifSInt64 length;
  int batch;
  nxsig_stft_params p;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !get_buf(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x->owner != c || x->bytes / elem / (size_t)batch < (size_t)length || w.size != (size_t)p.frame_length * 4)
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  size_t zbytes = 8;
  if (!mul_size(&zbytes, (uint64_t)batch) || !mul_size(&zbytes, (uint64_t)m) || !mul_size(&zbytes, (uint64_t)p.fft_length)) return mk_oom(env);
  void* z = NULL;
  int rc = nxsig_alloc(c->ctx, zbytes, &z);
  if (rc) return mk_error(env, rc);
  rc = elem == 8 ? nxsig_stft_c64(c->ctx, (const nxsig_c64*)x->dptr, length, batch, length, (const float*)w.data, &p, (nxsig_c64*)z, NULL, NXSIG_DEVICE)
                 : nxsig_stft_f32(c->ctx, (const float*)x->dptr, length, batch, length, (const float*)w.data, &p, (nxsig_c64*)z, NULL, NXSIG_DEVICE);
  if (rc) { nxsig_free(c->ctx, z); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), make_buf(env, c, z, zbytes), enif_make_int64(env, m));
}
static ERL_NIF_TERM nif_stft_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) { return stft_dev_impl(env, argc, argv, 4); }
static ERL_NIF_TERM nif_stft_c64_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) { return stft_dev_impl(env, argc, argv, 8); }

/* stft_onesided_dev / stft_packed_dev(ctx, x_buf, length, batch, window_bin, params) -> {:ok, z_buf, num_frames}
 * c64[batch][M][fft_length / 2]: bins 0 .. fft_length/2 - 1; packed: the imaginary part of bin 0 carries Re X[fft_length / 2] */
static ERL_NIF_TERM stft_half_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[], int packed) {
  ctx_res_t* c;
  buf_res_t* x;
  ErlNifBinary w;
  ErlNifSInt64 length;
  int batch;
  nxsig_stft_params p;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !get_buf(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p))
    return enif_make_badarg(env);
  if (p.fft_length < 2 || batch < 1 || length < 1 || x->owner != c || x->bytes / 4 / (size_t)batch < (size_t)length || w.size != (size_t)p.frame_length * 4)
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  size_t zbytes = 8;
  if (!mul_size(&zbytes, (uint64_t)batch) || !mul_size(&zbytes, (uint64_t)m) || !mul_size(&zbytes, (uint64_t)(p.fft_length / 2))) return mk_oom(env);
  void* z = NULL;
  int rc = nxsig_alloc(c->ctx, zbytes, &z);
  if (rc) return mk_error(env, rc);
  rc = packed ? nxsig_stft_packed_f32(c->ctx, (const float*)x->dptr, length, batch, length, (const float*)w.data, &p, (nxsig_c64*)z, NULL, NXSIG_DEVICE)
              : nxsig_stft_onesided_f32(c->ctx, (const float*)x->dptr, length, batch, length, (const float*)w.data, &p, (nxsig_c64*)z, NULL, NXSIG_DEVICE);
  if (rc) { nxsig_free(c->ctx, z); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), make_buf(env, c, z, zbytes), enif_make_int64(env, m));
}
static ERL_NIF_TERM nif_stft_onesided_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) { return stft_half_dev(env, argc, argv, 0); }
static ERL_NIF_TERM nif_stft_packed_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) { return stft_half_dev(env, argc, argv, 1); }

/* istft_packed_dev(ctx, z_buf, num_frames, batch, window_bin, params) -> {:ok, y_buf}: z packed c64[batch][M][fft_length / 2],
 * y REAL f32[batch][M * hop + frame_length - hop] */
static ERL_NIF_TERM nif_istft_packed_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  buf_res_t* z;
  ErlNifBinary w;
  ErlNifSInt64 m;
  int batch;
  nxsig_stft_params p;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !get_buf(env, argv[1], &z) || !enif_get_int64(env, argv[2], &m) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p))
    return enif_make_badarg(env);
  if (batch < 1 || m < 1 || p.fft_length < 2 || z->owner != c || w.size != (size_t)p.frame_length * 4 ||
      z->bytes / 8 / (size_t)batch / (size_t)m < (size_t)(p.fft_length / 2))
    return enif_make_badarg(env);
  int64_t n = nxsig_ola_length(m, p.frame_length, p.hop);
  if (n < 0) return mk_error(env, (int)n);
  size_t ybytes = 4;
  if (!mul_size(&ybytes, (uint64_t)batch) || !mul_size(&ybytes, (uint64_t)n)) return mk_oom(env);
  void* y = NULL;
  int rc = nxsig_alloc(c->ctx, ybytes, &y);
  if (rc) return mk_error(env, rc);
  rc = nxsig_istft_packed_f32(c->ctx, (const nxsig_c64*)z->dptr, m, batch, (const float*)w.data, &p, (float*)y, NXSIG_DEVICE);
  if (rc) { nxsig_free(c->ctx, y); return mk_error(env, rc); }
  return mk_ok(env, make_buf(env, c, y, ybytes));
}

/* istft_dev(ctx, z_buf, num_frames, batch, window_bin, params) -> {:ok, y_buf} */
static ERL_NIF_TERM nif_istft_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  buf_res_t* z;
  ErlNifBinary w;
  ErlNifSInt64 m;
  int batch;
  nxsig_stft_params p;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !get_buf(env, argv[1], &z) || !enif_get_int64(env, argv[2], &m) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p))
    return enif_make_badarg(env);
  /* the window binary is read by the library: its size must match (a short window was read out of bounds) */
  if (batch < 1 || m < 1 || z->owner != c || w.size != (size_t)p.frame_length * 4 ||
      z->bytes / 8 / (size_t)batch / (size_t)m < (size_t)p.fft_length)
    return enif_make_badarg(env);
  int64_t n = nxsig_ola_length(m, p.frame_length, p.hop);
  if (n < 0) return mk_error(env, (int)n);
  size_t ybytes = 8;
  if (!mul_size(&ybytes, (uint64_t)batch) || !mul_size(&ybytes, (uint64_t)n)) return mk_oom(env);
  void* y = NULL;
  int rc = nxsig_alloc(c->ctx, ybytes, &y);
  if (rc) return mk_error(env, rc);
  rc = nxsig_istft_c64(c->ctx, (const nxsig_c64*)z->dptr, m, batch, (const float*)w.data, &p, (nxsig_c64*)y, NXSIG_DEVICE);
  if (rc) { nxsig_free(c->ctx, y); return mk_error(env, rc); }
  return mk_ok(env, make_buf(env, c, y, ybytes));
}

/* istft_filtered_dev(ctx, z_buf, num_frames, batch, window_bin, params, h_bin) -> {:ok, y_buf}   (z_buf is left untouched) */
static ERL_NIF_TERM nif_istft_filtered_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  buf_res_t* z;
  ErlNifBinary w, h;
  ErlNifSInt64 m;
  int batch;
  nxsig_stft_params p;
  if (argc != 7 || !get_ctx(env, argv[0], &c) || !get_buf(env, argv[1], &z) || !enif_get_int64(env, argv[2], &m) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p) ||
      !enif_inspect_binary(env, argv[6], &h))
    return enif_make_badarg(env);
  if (batch < 1 || m < 1 || p.fft_length < 1 || z->owner != c || w.size != (size_t)p.frame_length * 4 ||
      h.size != (size_t)p.fft_length * 8 || z->bytes / 8 / (size_t)batch / (size_t)m < (size_t)p.fft_length)
    return enif_make_badarg(env);
  int64_t n = nxsig_ola_length(m, p.frame_length, p.hop);
  if (n < 0) return mk_error(env, (int)n);
  size_t ybytes = 8;
  if (!mul_size(&ybytes, (uint64_t)batch) || !mul_size(&ybytes, (uint64_t)n)) return mk_oom(env);
  void* y = NULL;
  int rc = nxsig_alloc(c->ctx, ybytes, &y);
  if (rc) return mk_error(env, rc);
  rc = nxsig_istft_filtered_c64(c->ctx, (const nxsig_c64*)z->dptr, m, batch, (const float*)w.data, &p, (const nxsig_c64*)h.data,
                                (nxsig_c64*)y, NXSIG_DEVICE);
  if (rc) { nxsig_free(c->ctx, y); return mk_error(env, rc); }
  return mk_ok(env, make_buf(env, c, y, ybytes));
}

/* fir_dev(ctx, x_buf, length, batch, taps_bin, mode) -> {:ok, y_buf, out_length} */
static ERL_NIF_TERM nif_fir_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  buf_res_t* x;
  ErlNifBinary h;
  ErlNifSInt64 length;
  int batch, mode;
  if (argc != 6 || !get_ctx(env, argv[0], &c) || !get_buf(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &h) || !enif_get_int(env, argv[5], &mode))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x->owner != c || x->bytes / 4 / (size_t)batch < (size_t)length || h.size < 4 || h.size % 4 ||
      h.size / 4 > 0x7fffffff)
    return enif_make_badarg(env);
  int64_t n = nxsig_conv_length(length, (int64_t)(h.size / 4), mode);
  if (n < 0) return mk_error(env, (int)n);
  size_t ybytes = 4;
  if (!mul_size(&ybytes, (uint64_t)batch) || !mul_size(&ybytes, (uint64_t)n)) return mk_oom(env);
  void* y = NULL;
  int rc = nxsig_alloc(c->ctx, ybytes, &y);
  if (rc) return mk_error(env, rc);
  rc = nxsig_fir_f32(c->ctx, (const float*)x->dptr, length, batch, length, (const float*)h.data, (int)(h.size / 4), mode, (float*)y, NXSIG_DEVICE);
  if (rc) { nxsig_free(c->ctx, y); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), make_buf(env, c, y, ybytes), enif_make_int64(env, n));
}

/* spectrum_mul_dev(ctx, z_buf, rows, fft_length, h_bin) -> {:ok, z_buf}   (in place: Nx.multiply(z, hfft), guides/filtering.livemd:141) */
static ERL_NIF_TERM nif_spectrum_mul_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res_t* c;
  buf_res_t* z;
  ErlNifBinary h;
  ErlNifSInt64 rows;
  int k;
  if (argc != 5 || !get_ctx(env, argv[0], &c) || !get_buf(env, argv[1], &z) || !enif_get_int64(env, argv[2], &rows) ||
      !enif_get_int(env, argv[3], &k) || !enif_inspect_binary(env, argv[4], &h))
    return enif_make_badarg(env);
  if (rows < 1 || k < 1 || z->owner != c || h.size != (size_t)k * 8 || z->bytes / 8 / (size_t)rows < (size_t)k) return enif_make_badarg(env);
  int rc = nxsig_spectrum_mul_c64(c->ctx, (const nxsig_c64*)z->dptr, rows, k, (const nxsig_c64*)h.data, (nxsig_c64*)z->dptr, NXSIG_DEVICE);
  return rc ? mk_error(env, rc) : mk_ok(env, argv[1]);
}

/* ------------------------------------------------------------------------------------------------ multi-GPU groups
 * One BEAM process drives every GPU of the node: a LOCAL group (ncclCommInitAll).  The vectorized (multichannel) axis of the
 * reference (lib/nx_signal.ex:358-363) is what `axis = 0` shards. */
/* group_create([device_id]) -> {:ok, group} */
static ERL_NIF_TERM nif_group_create(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  unsigned len;
  int32_t devs[64];
  if (argc != 1 || !enif_get_list_length(env, argv[0], &len) || len < 1 || len > 64) return enif_make_badarg(env);
  ERL_NIF_TERM head, tail = argv[0];
  for (unsigned i = 0; i < len; ++i) {
    int d;
    if (!enif_get_list_cell(env, tail, &head, &tail) || !enif_get_int(env, head, &d)) return enif_make_badarg(env);
    devs[i] = d;
  }
  nxsig_group* g = NULL;
  int rc = nxsig_group_create_local((int32_t)len, devs, &g);
  if (rc) return mk_error(env, rc);
  grp_res_t* r = enif_alloc_resource(GRP_RES, sizeof *r);
  r->grp = g;
  ERL_NIF_TERM t = enif_make_resource(env, r);
  enif_release_resource(r);
  return mk_ok(env, t);
}

/* group_info(group) -> {world, local_count, has_rccl} */
static ERL_NIF_TERM nif_group_info(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  if (argc != 1 || !get_grp(env, argv[0], &g)) return enif_make_badarg(env);
  return enif_make_tuple3(env, enif_make_int(env, nxsig_group_world(g->grp)), enif_make_int(env, nxsig_group_local_count(g->grp)),
                          enif_make_int(env, nxsig_group_has_rccl(g->grp)));
}

/* stft_sharded(group, x_bin, length, batch, window_bin, params, axis, gather) -> {:ok, z_bin, num_frames} */
static ERL_NIF_TERM nif_stft_sharded(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  ErlNifBinary x, w, zb;
  ErlNifSInt64 length;
  int batch, axis, gather;
  nxsig_stft_params p;
  if (argc != 8 || !get_grp(env, argv[0], &g) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p) ||
      !enif_get_int(env, argv[6], &axis) || !enif_get_int(env, argv[7], &gather))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size % 4 || x.size / 4 / (size_t)batch != (size_t)length || w.size != (size_t)p.frame_length * 4)
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  if (!out_bin(&zb, (uint64_t)batch, (uint64_t)m, (uint64_t)p.fft_length, 8)) return mk_oom(env);
  const float* xs[64] = {(const float*)x.data};
  nxsig_c64* zs[64] = {(nxsig_c64*)zb.data};
  int rc = nxsig_stft_sharded_f32(g->grp, xs, length, batch, length, (const float*)w.data, &p, axis, gather, zs, NXSIG_HOST);
  if (rc) { enif_release_binary(&zb); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), enif_make_binary(env, &zb), enif_make_int64(env, m));
}

/* istft_sharded(group, z_bin, num_frames, batch, window_bin, params, axis, gather) -> {:ok, y_bin} */
static ERL_NIF_TERM nif_istft_sharded(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  ErlNifBinary z, w, yb;
  ErlNifSInt64 m;
  int batch, axis, gather;
  nxsig_stft_params p;
  if (argc != 8 || !get_grp(env, argv[0], &g) || !enif_inspect_binary(env, argv[1], &z) || !enif_get_int64(env, argv[2], &m) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p) ||
      !enif_get_int(env, argv[6], &axis) || !enif_get_int(env, argv[7], &gather))
    return enif_make_badarg(env);
  if (batch < 1 || m < 1 || p.fft_length < 1 || z.size % 8 || z.size / 8 / (size_t)batch / (size_t)m != (size_t)p.fft_length ||
      z.size / 8 % ((size_t)batch * (size_t)m) || w.size != (size_t)p.frame_length * 4)
    return enif_make_badarg(env);
  int64_t n = nxsig_ola_length(m, p.frame_length, p.hop);
  if (n < 0) return mk_error(env, (int)n);
  if (!out_bin(&yb, (uint64_t)batch, (uint64_t)n, 1, 8)) return mk_oom(env);
  const nxsig_c64* zs[64] = {(const nxsig_c64*)z.data};
  nxsig_c64* ys[64] = {(nxsig_c64*)yb.data};
  int rc = nxsig_istft_sharded_c64(g->grp, zs, m, batch, (const float*)w.data, &p, axis, gather, ys, NXSIG_HOST);
  if (rc) { enif_release_binary(&yb); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &yb));
}

/* stft_mel_sharded(group, x_bin, length, batch, window_bin, params, mel_bins, filters_bin, axis) -> {:ok, f32[batch][M][mel_bins], M}
 * (the sharded log-mel: the members' maxima are all-reduced between the two passes) */
static ERL_NIF_TERM nif_stft_mel_sharded(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  ErlNifBinary x, w, f, ob;
  ErlNifSInt64 length;
  int batch, bins, axis;
  nxsig_stft_params p;
  if (argc != 9 || !get_grp(env, argv[0], &g) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p) ||
      !enif_get_int(env, argv[6], &bins) || !enif_inspect_binary(env, argv[7], &f) || !enif_get_int(env, argv[8], &axis))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || bins < 1 || x.size % 4 || x.size / 4 / (size_t)batch != (size_t)length ||
      w.size != (size_t)p.frame_length * 4 || p.fft_length < 2 || f.size % 4 || f.size / 4 / (size_t)bins != (size_t)p.fft_length ||
      f.size / 4 % (size_t)bins)
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  if (!out_bin(&ob, (uint64_t)batch, (uint64_t)m, (uint64_t)bins, 4)) return mk_oom(env);
  const float* xs[64] = {(const float*)x.data};
  float* os[64] = {(float*)ob.data};
  int rc = nxsig_stft_mel_sharded_f32(g->grp, xs, length, batch, length, (const float*)w.data, &p, bins, (const float*)f.data, axis, os,
                                      NULL, NXSIG_HOST);
  if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), enif_make_binary(env, &ob), enif_make_int64(env, m));
}

/* fir_sharded(group, x_bin, length, batch, taps_bin, mode, axis, gather) -> {:ok, y_bin} */
static ERL_NIF_TERM nif_fir_sharded(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  ErlNifBinary x, h, yb;
  ErlNifSInt64 length;
  int batch, mode, axis, gather;
  if (argc != 8 || !get_grp(env, argv[0], &g) || !enif_inspect_binary(env, argv[1], &x) || !enif_get_int64(env, argv[2], &length) ||
      !enif_get_int(env, argv[3], &batch) || !enif_inspect_binary(env, argv[4], &h) || !enif_get_int(env, argv[5], &mode) ||
      !enif_get_int(env, argv[6], &axis) || !enif_get_int(env, argv[7], &gather))
    return enif_make_badarg(env);
  if (batch < 1 || length < 1 || x.size % 4 || x.size / 4 / (size_t)batch != (size_t)length || h.size < 4 || h.size % 4 ||
      h.size / 4 > 0x7fffffff)
    return enif_make_badarg(env);
  int64_t n = nxsig_conv_length(length, (int64_t)(h.size / 4), mode);
  if (n < 0) return mk_error(env, (int)n);
  if (!out_bin(&yb, (uint64_t)batch, (uint64_t)n, 1, 4)) return mk_oom(env);
  const float* xs[64] = {(const float*)x.data};
  float* ys[64] = {(float*)yb.data};
  int rc = nxsig_fir_sharded_f32(g->grp, xs, length, batch, length, (const float*)h.data, (int)(h.size / 4), mode, axis, gather, ys, NXSIG_HOST);
  if (rc) { enif_release_binary(&yb); return mk_error(env, rc); }
  return mk_ok(env, enif_make_binary(env, &yb));
}

/* ------------------------------------------------------------------------------------------------ device-resident shards
 * NxSignalAMD.Sharded.Tensor: one dense device buffer per group member, the payload never visits the host between
 * group_scatter (Sharded.to_device) and group_gather (Sharded.from_device).  The reference axis being sharded is Nx's
 * vectorized axis (lib/nx_signal.ex:358-363) or, for one long stream, the frame / sample axis. */
#define NXSIG_MAX_MEMBERS 64

/* shard_range(kind, a, b, c, world, rank) -> {x0, x1, y0, y1}
 *   kind 0: nxsig_shard_range(total = a)            -> {begin, end, 0, 0}
 *   kind 1: nxsig_shard_frames(num_frames = a, frame_length = b, hop = c)   -> {m0, m1, s0, s1}
 *   kind 2: nxsig_shard_istft(num_frames = a, frame_length = b, hop = c)    -> {f0, f1, n0, n1}
 *   kind 3: nxsig_shard_fir(length = a, num_taps = b, mode = c)             -> {n0, n1, s0, s1} */
static ERL_NIF_TERM nif_shard_range(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int kind, world, rank;
  ErlNifSInt64 a, b, c;
  if (argc != 6 || !enif_get_int(env, argv[0], &kind) || !enif_get_int64(env, argv[1], &a) || !enif_get_int64(env, argv[2], &b) ||
      !enif_get_int64(env, argv[3], &c) || !enif_get_int(env, argv[4], &world) || !enif_get_int(env, argv[5], &rank))
    return enif_make_badarg(env);
  if (b < -0x7fffffff || b > 0x7fffffff || c < -0x7fffffff || c > 0x7fffffff) return enif_make_badarg(env);
  int64_t r[4] = {0, 0, 0, 0};
  int rc;
  switch (kind) {
    case 0: rc = nxsig_shard_range(a, world, rank, &r[0], &r[1]); break;
    case 1: rc = nxsig_shard_frames(a, (int32_t)b, (int32_t)c, world, rank, &r[0], &r[1], &r[2], &r[3]); break;
    case 2: rc = nxsig_shard_istft(a, (int32_t)b, (int32_t)c, world, rank, &r[0], &r[1], &r[2], &r[3]); break;
    case 3: rc = nxsig_shard_fir(a, (int32_t)b, (int32_t)c, world, rank, &r[0], &r[1], &r[2], &r[3]); break;
    default: return enif_make_badarg(env);
  }
  if (rc) return mk_error(env, rc);
  return mk_ok(env, enif_make_tuple4(env, enif_make_int64(env, r[0]), enif_make_int64(env, r[1]), enif_make_int64(env, r[2]),
                                     enif_make_int64(env, r[3])));
}

/* [{row0, rows, off_bytes, len_bytes}, ...] — one entry per LOCAL member, in member order */
typedef struct { int64_t row0, rows, off, len; } part_t;
static int get_parts(ErlNifEnv* env, ERL_NIF_TERM list, int n, part_t* out) {
  unsigned len;
  if (!enif_get_list_length(env, list, &len) || (int)len != n) return 0;
  ERL_NIF_TERM head, tail = list;
  for (int i = 0; i < n; ++i) {
    const ERL_NIF_TERM* e;
    int arity;
    ErlNifSInt64 v[4];
    if (!enif_get_list_cell(env, tail, &head, &tail) || !enif_get_tuple(env, head, &arity, &e) || arity != 4) return 0;
    for (int k = 0; k < 4; ++k)
      if (!enif_get_int64(env, e[k], &v[k]) || v[k] < 0) return 0;
    out[i].row0 = v[0]; out[i].rows = v[1]; out[i].off = v[2]; out[i].len = v[3];
  }
  return 1;
}
static int get_bufs(ErlNifEnv* env, ERL_NIF_TERM list, grp_res_t* g, int n, buf_res_t** out) {
  unsigned len;
  if (!enif_get_list_length(env, list, &len) || (int)len != n) return 0;
  ERL_NIF_TERM head, tail = list;
  for (int i = 0; i < n; ++i) {
    if (!enif_get_list_cell(env, tail, &head, &tail) || !get_buf(env, head, &out[i])) return 0;
    if (out[i]->gowner != g || out[i]->member != i) return 0;   /* buffer i must live on member i of THIS group */
  }
  return 1;
}
static ERL_NIF_TERM bufs_to_list(ErlNifEnv* env, grp_res_t* g, int n, void** d, const size_t* bytes) {
  ERL_NIF_TERM t[NXSIG_MAX_MEMBERS];
  for (int i = 0; i < n; ++i) t[i] = make_gbuf(env, g, i, d[i], bytes[i]);
  return enif_make_list_from_array(env, t, (unsigned)n);
}
static void free_members(grp_res_t* g, int n, void** d) {
  for (int i = 0; i < n; ++i)
    if (d[i]) { nxsig_free(nxsig_group_ctx(g->grp, i), d[i]); d[i] = NULL; }
}

/* group_scatter(group, bin, batch, row_bytes, parts) -> {:ok, [buf]}: member i receives rows [row0, row0 + rows) x bytes
 * [off, off + len) of the host tensor [batch][row_bytes], packed densely ([rows][len]) on its own device */
static ERL_NIF_TERM nif_group_scatter(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  ErlNifBinary b;
  ErlNifSInt64 batch, row_bytes;
  part_t parts[NXSIG_MAX_MEMBERS];
  if (argc != 5 || !get_grp(env, argv[0], &g) || !enif_inspect_binary(env, argv[1], &b) || !enif_get_int64(env, argv[2], &batch) ||
      !enif_get_int64(env, argv[3], &row_bytes))
    return enif_make_badarg(env);
  const int n = nxsig_group_local_count(g->grp);
  if (n < 1 || n > NXSIG_MAX_MEMBERS || batch < 1 || row_bytes < 1 || (uint64_t)b.size / (uint64_t)batch != (uint64_t)row_bytes ||
      b.size % (size_t)batch || !get_parts(env, argv[4], n, parts))
    return enif_make_badarg(env);
  for (int i = 0; i < n; ++i)
    if (parts[i].row0 > batch || parts[i].rows > batch - parts[i].row0 || parts[i].off > row_bytes || parts[i].len > row_bytes - parts[i].off)
      return enif_make_badarg(env);   /* compared without adding: the four values come from Erlang and may sum past INT64_MAX */
  void* d[NXSIG_MAX_MEMBERS] = {0};
  size_t bytes[NXSIG_MAX_MEMBERS];
  for (int i = 0; i < n; ++i) {
    nxsig_ctx* c = nxsig_group_ctx(g->grp, i);
    bytes[i] = (size_t)parts[i].rows * (size_t)parts[i].len;
    int rc = nxsig_alloc(c, bytes[i] ? bytes[i] : 4, &d[i]);
    for (int64_t r = 0; !rc && r < parts[i].rows && parts[i].len > 0; ++r)
      rc = nxsig_upload(c, (char*)d[i] + (size_t)r * (size_t)parts[i].len,
                        b.data + (size_t)(parts[i].row0 + r) * (size_t)row_bytes + (size_t)parts[i].off, (size_t)parts[i].len);
    if (rc) { free_members(g, n, d); return mk_error(env, rc); }
  }
  return mk_ok(env, bufs_to_list(env, g, n, d, bytes));
}

/* group_gather(group, [buf], batch, row_bytes, parts) -> {:ok, bin}: the inverse of group_scatter for RESULT shards */
static ERL_NIF_TERM nif_group_gather(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  ErlNifBinary ob;
  ErlNifSInt64 batch, row_bytes;
  part_t parts[NXSIG_MAX_MEMBERS];
  buf_res_t* bufs[NXSIG_MAX_MEMBERS];
  if (argc != 5 || !get_grp(env, argv[0], &g) || !enif_get_int64(env, argv[2], &batch) || !enif_get_int64(env, argv[3], &row_bytes))
    return enif_make_badarg(env);
  const int n = nxsig_group_local_count(g->grp);
  if (n < 1 || n > NXSIG_MAX_MEMBERS || batch < 1 || row_bytes < 1 || !get_bufs(env, argv[1], g, n, bufs) || !get_parts(env, argv[4], n, parts))
    return enif_make_badarg(env);
  for (int i = 0; i < n; ++i)
    if (parts[i].row0 > batch || parts[i].rows > batch - parts[i].row0 || parts[i].off > row_bytes || parts[i].len > row_bytes - parts[i].off ||
        (parts[i].len > 0 && (uint64_t)parts[i].rows > (uint64_t)bufs[i]->bytes / (uint64_t)parts[i].len))
      return enif_make_badarg(env);
  if (!out_bin(&ob, (uint64_t)batch, (uint64_t)row_bytes, 1, 1)) return mk_oom(env);
  memset(ob.data, 0, ob.size);   /* positions no shard covers (none for the plans of nxsig_shard_*) read as zero */
  for (int i = 0; i < n; ++i) {
    nxsig_ctx* c = nxsig_group_ctx(g->grp, i);
    int rc = nxsig_sync(c);      /* the shard may still be in flight on the member's stream */
    for (int64_t r = 0; !rc && r < parts[i].rows && parts[i].len > 0; ++r)
      rc = nxsig_download(c, ob.data + (size_t)(parts[i].row0 + r) * (size_t)row_bytes + (size_t)parts[i].off,
                          (const char*)bufs[i]->dptr + (size_t)r * (size_t)parts[i].len, (size_t)parts[i].len);
    if (rc) { enif_release_binary(&ob); return mk_error(env, rc); }
  }
  return mk_ok(env, enif_make_binary(env, &ob));
}

/* output shard sizes (bytes per member) of the sharded calls; gather: every member holds the whole tensor */
static int out_sizes(grp_res_t* g, int n, int axis, int gather, int64_t batch, int64_t items_per_row, int64_t item_bytes, int kind, int64_t a,
                     int32_t b, int32_t c, size_t* bytes) {
  const int world = nxsig_group_world(g->grp);
  for (int i = 0; i < n; ++i) {
    const int rank = nxsig_group_rank(g->grp, i);
    int64_t rows = batch, items = items_per_row;
    if (!gather) {
      int64_t r[4];
      int rc;
      if (axis == NXSIG_SHARD_CHANNELS) { if ((rc = nxsig_shard_range(batch, world, rank, &r[0], &r[1]))) return rc; rows = r[1] - r[0]; }
      else if (kind == 1) { if ((rc = nxsig_shard_frames(a, b, c, world, rank, &r[0], &r[1], &r[2], &r[3]))) return rc; items = r[1] - r[0]; }
      else if (kind == 2) { if ((rc = nxsig_shard_istft(a, b, c, world, rank, &r[0], &r[1], &r[2], &r[3]))) return rc; items = r[3] - r[2]; }
      else { if ((rc = nxsig_shard_fir(a, b, c, world, rank, &r[0], &r[1], &r[2], &r[3]))) return rc; items = r[1] - r[0]; }
    }
    size_t sz = (size_t)item_bytes;
    if (!mul_size(&sz, (uint64_t)rows) || !mul_size(&sz, (uint64_t)items)) return NXSIG_ERR_OOM;
    bytes[i] = sz;
  }
  return 0;
}
/* INPUT shard sizes (bytes per member) the sharded calls will READ, from the same partition rules: kind 1 stft (f32 samples; a =
 * num_frames, b = frame_length, c = hop, items_per_row = length), kind 2 istft (c64 frames of item_bytes = K * 8; a = num_frames),
 * kind 3 fir (f32 samples; a = length, b = taps, c = mode).  A buffer smaller than this would be read out of bounds on the device:
 * the *_sharded_dev NIFs refuse it. */
static int in_sizes(grp_res_t* g, int n, int axis, int64_t batch, int64_t items_per_row, int64_t item_bytes, int kind, int64_t a, int32_t b,
                    int32_t c, size_t* bytes) {
  const int world = nxsig_group_world(g->grp);
  for (int i = 0; i < n; ++i) {
    const int rank = nxsig_group_rank(g->grp, i);
    int64_t rows = batch, items = items_per_row, r[4];
    int rc;
    if (axis == NXSIG_SHARD_CHANNELS) { if ((rc = nxsig_shard_range(batch, world, rank, &r[0], &r[1]))) return rc; rows = r[1] - r[0]; }
    else if (kind == 1) { if ((rc = nxsig_shard_frames(a, b, c, world, rank, &r[0], &r[1], &r[2], &r[3]))) return rc; items = r[3] - r[2]; }
    else if (kind == 2) { if ((rc = nxsig_shard_istft(a, b, c, world, rank, &r[0], &r[1], &r[2], &r[3]))) return rc; items = r[1] - r[0]; }
    else { if ((rc = nxsig_shard_fir(a, b, c, world, rank, &r[0], &r[1], &r[2], &r[3]))) return rc; items = r[3] - r[2]; }
    size_t sz = (size_t)item_bytes;
    if (rows < 0 || items < 0 || !mul_size(&sz, (uint64_t)rows) || !mul_size(&sz, (uint64_t)items)) return NXSIG_ERR_OOM;
    bytes[i] = sz;
  }
  return 0;
}
static int bufs_hold(int n, buf_res_t** b, const size_t* need) {
  for (int i = 0; i < n; ++i)
    if (b[i]->bytes < need[i]) return 0;
  return 1;
}
static int alloc_members(grp_res_t* g, int n, const size_t* bytes, void** d) {
  for (int i = 0; i < n; ++i) {
    int rc = nxsig_alloc(nxsig_group_ctx(g->grp, i), bytes[i] ? bytes[i] : 4, &d[i]);
    if (rc) { free_members(g, n, d); return rc; }
  }
  return 0;
}

/* stft_sharded_dev(group, [x_buf], length, batch, window_bin, params, axis, gather) -> {:ok, [z_buf], num_frames}
 * x_buf i = member i's dense input shard (rows [c0, c1) of the tensor / the span [s0, s1) of every row); the result shards
 * (or, gather = 1, the whole c64[batch][M][K] on every member, assembled by the RCCL all-gather) stay on their devices */
static ERL_NIF_TERM nif_stft_sharded_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  ErlNifBinary w;
  ErlNifSInt64 length;
  int batch, axis, gather;
  nxsig_stft_params p;
  buf_res_t* xb[NXSIG_MAX_MEMBERS];
  if (argc != 8 || !get_grp(env, argv[0], &g) || !enif_get_int64(env, argv[2], &length) || !enif_get_int(env, argv[3], &batch) ||
      !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p) || !enif_get_int(env, argv[6], &axis) ||
      !enif_get_int(env, argv[7], &gather))
    return enif_make_badarg(env);
  const int n = nxsig_group_local_count(g->grp);
  if (n < 1 || n > NXSIG_MAX_MEMBERS || batch < 1 || length < 1 || w.size != (size_t)p.frame_length * 4 || !get_bufs(env, argv[1], g, n, xb))
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  size_t bytes[NXSIG_MAX_MEMBERS];
  void* z[NXSIG_MAX_MEMBERS] = {0};
  int rc = in_sizes(g, n, axis, batch, length, 4, 1, m, p.frame_length, p.hop, bytes);
  if (rc) return rc == NXSIG_ERR_OOM ? mk_oom(env) : mk_error(env, rc);
  if (!bufs_hold(n, xb, bytes)) return enif_make_badarg(env);   /* a shard scattered for another plan: it would be read out of bounds */
  rc = out_sizes(g, n, axis, gather, batch, m, (int64_t)p.fft_length * 8, 1, m, p.frame_length, p.hop, bytes);
  if (rc) return rc == NXSIG_ERR_OOM ? mk_oom(env) : mk_error(env, rc);
  if ((rc = alloc_members(g, n, bytes, z))) return mk_error(env, rc);
  const float* xs[NXSIG_MAX_MEMBERS];
  nxsig_c64* zs[NXSIG_MAX_MEMBERS];
  for (int i = 0; i < n; ++i) { xs[i] = (const float*)xb[i]->dptr; zs[i] = (nxsig_c64*)z[i]; }
  /* channel shards: rows of the full length; frame shards: dense per-member spans (batch_stride 0, see nxsig.h) */
  rc = nxsig_stft_sharded_f32(g->grp, xs, length, batch, axis == NXSIG_SHARD_CHANNELS ? length : 0, (const float*)w.data, &p, axis, gather, zs,
                              NXSIG_DEVICE);
  if (rc) { free_members(g, n, z); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), bufs_to_list(env, g, n, z, bytes), enif_make_int64(env, m));
}

/* istft_sharded_dev(group, [z_buf], num_frames, batch, window_bin, params, axis, gather) -> {:ok, [y_buf]} */
static ERL_NIF_TERM nif_istft_sharded_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  ErlNifBinary w;
  ErlNifSInt64 m;
  int batch, axis, gather;
  nxsig_stft_params p;
  buf_res_t* zb[NXSIG_MAX_MEMBERS];
  if (argc != 8 || !get_grp(env, argv[0], &g) || !enif_get_int64(env, argv[2], &m) || !enif_get_int(env, argv[3], &batch) ||
      !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p) || !enif_get_int(env, argv[6], &axis) ||
      !enif_get_int(env, argv[7], &gather))
    return enif_make_badarg(env);
  const int n = nxsig_group_local_count(g->grp);
  if (n < 1 || n > NXSIG_MAX_MEMBERS || batch < 1 || m < 1 || w.size != (size_t)p.frame_length * 4 || !get_bufs(env, argv[1], g, n, zb))
    return enif_make_badarg(env);
  int64_t out_len = nxsig_ola_length(m, p.frame_length, p.hop);
  if (out_len < 0) return mk_error(env, (int)out_len);
  size_t bytes[NXSIG_MAX_MEMBERS];
  void* y[NXSIG_MAX_MEMBERS] = {0};
  int rc = in_sizes(g, n, axis, batch, m, (int64_t)p.frame_length * 8, 2, m, p.frame_length, p.hop, bytes);
  if (rc) return rc == NXSIG_ERR_OOM ? mk_oom(env) : mk_error(env, rc);
  if (!bufs_hold(n, zb, bytes)) return enif_make_badarg(env);   /* e.g. the frame shards of stft(axis: :frames): istft needs halo frames too */
  rc = out_sizes(g, n, axis, gather, batch, out_len, 8, 2, m, p.frame_length, p.hop, bytes);
  if (rc) return rc == NXSIG_ERR_OOM ? mk_oom(env) : mk_error(env, rc);
  if ((rc = alloc_members(g, n, bytes, y))) return mk_error(env, rc);
  const nxsig_c64* zs[NXSIG_MAX_MEMBERS];
  nxsig_c64* ys[NXSIG_MAX_MEMBERS];
  for (int i = 0; i < n; ++i) { zs[i] = (const nxsig_c64*)zb[i]->dptr; ys[i] = (nxsig_c64*)y[i]; }
  rc = nxsig_istft_sharded_c64(g->grp, zs, m, batch, (const float*)w.data, &p, axis, gather, ys, NXSIG_DEVICE);
  if (rc) { free_members(g, n, y); return mk_error(env, rc); }
  return mk_ok(env, bufs_to_list(env, g, n, y, bytes));
}

/* fir_sharded_dev(group, [x_buf], length, batch, taps_bin, mode, axis, gather) -> {:ok, [y_buf]} */
static ERL_NIF_TERM nif_fir_sharded_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  ErlNifBinary h;
  ErlNifSInt64 length;
  int batch, mode, axis, gather;
  buf_res_t* xb[NXSIG_MAX_MEMBERS];
  if (argc != 8 || !get_grp(env, argv[0], &g) || !enif_get_int64(env, argv[2], &length) || !enif_get_int(env, argv[3], &batch) ||
      !enif_inspect_binary(env, argv[4], &h) || !enif_get_int(env, argv[5], &mode) || !enif_get_int(env, argv[6], &axis) ||
      !enif_get_int(env, argv[7], &gather))
    return enif_make_badarg(env);
  const int n = nxsig_group_local_count(g->grp);
  if (n < 1 || n > NXSIG_MAX_MEMBERS || batch < 1 || length < 1 || h.size < 4 || h.size % 4 || h.size / 4 > 0x7fffffff ||
      !get_bufs(env, argv[1], g, n, xb))
    return enif_make_badarg(env);
  const int taps = (int)(h.size / 4);
  int64_t out_len = nxsig_conv_length(length, taps, mode);
  if (out_len < 0) return mk_error(env, (int)out_len);
  size_t bytes[NXSIG_MAX_MEMBERS];
  void* y[NXSIG_MAX_MEMBERS] = {0};
  int rc = in_sizes(g, n, axis, batch, length, 4, 3, length, taps, mode, bytes);
  if (rc) return rc == NXSIG_ERR_OOM ? mk_oom(env) : mk_error(env, rc);
  if (!bufs_hold(n, xb, bytes)) return enif_make_badarg(env);
  rc = out_sizes(g, n, axis, gather, batch, out_len, 4, 3, length, taps, mode, bytes);
  if (rc) return rc == NXSIG_ERR_OOM ? mk_oom(env) : mk_error(env, rc);
  if ((rc = alloc_members(g, n, bytes, y))) return mk_error(env, rc);
  const float* xs[NXSIG_MAX_MEMBERS];
  float* ys[NXSIG_MAX_MEMBERS];
  for (int i = 0; i < n; ++i) { xs[i] = (const float*)xb[i]->dptr; ys[i] = (float*)y[i]; }
  rc = nxsig_fir_sharded_f32(g->grp, xs, length, batch, axis == NXSIG_SHARD_CHANNELS ? length : 0, (const float*)h.data, taps, mode, axis, gather,
                             ys, NXSIG_DEVICE);
  if (rc) { free_members(g, n, y); return mk_error(env, rc); }
  return mk_ok(env, bufs_to_list(env, g, n, y, bytes));
}

/* stft_mel_sharded_dev(group, [x_buf], length, batch, window_bin, params, mel_bins, filters_bin, axis) -> {:ok, [o_buf], M} */
static ERL_NIF_TERM nif_stft_mel_sharded_dev(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  grp_res_t* g;
  ErlNifBinary w, f;
  ErlNifSInt64 length;
  int batch, bins, axis;
  nxsig_stft_params p;
  buf_res_t* xb[NXSIG_MAX_MEMBERS];
  if (argc != 9 || !get_grp(env, argv[0], &g) || !enif_get_int64(env, argv[2], &length) || !enif_get_int(env, argv[3], &batch) ||
      !enif_inspect_binary(env, argv[4], &w) || !get_params(env, argv[5], &p) || !enif_get_int(env, argv[6], &bins) ||
      !enif_inspect_binary(env, argv[7], &f) || !enif_get_int(env, argv[8], &axis))
    return enif_make_badarg(env);
  const int n = nxsig_group_local_count(g->grp);
  if (n < 1 || n > NXSIG_MAX_MEMBERS || batch < 1 || length < 1 || bins < 1 || w.size != (size_t)p.frame_length * 4 || p.fft_length < 2 ||
      f.size % 4 || f.size / 4 / (size_t)bins != (size_t)p.fft_length || f.size / 4 % (size_t)bins || !get_bufs(env, argv[1], g, n, xb))
    return enif_make_badarg(env);
  int64_t m = nxsig_num_frames(length, p.frame_length, p.hop, p.pad_mode, p.pad_lo, p.pad_hi);
  if (m < 0) return mk_error(env, (int)m);
  size_t bytes[NXSIG_MAX_MEMBERS];
  void* o[NXSIG_MAX_MEMBERS] = {0};
  int rc = in_sizes(g, n, axis, batch, length, 4, 1, m, p.frame_length, p.hop, bytes);
  if (rc) return rc == NXSIG_ERR_OOM ? mk_oom(env) : mk_error(env, rc);
  if (!bufs_hold(n, xb, bytes)) return enif_make_badarg(env);
  rc = out_sizes(g, n, axis, 0, batch, m, (int64_t)bins * 4, 1, m, p.frame_length, p.hop, bytes);
  if (rc) return rc == NXSIG_ERR_OOM ? mk_oom(env) : mk_error(env, rc);
  if ((rc = alloc_members(g, n, bytes, o))) return mk_error(env, rc);
  const float* xs[NXSIG_MAX_MEMBERS];
  float* os[NXSIG_MAX_MEMBERS];
  for (int i = 0; i < n; ++i) { xs[i] = (const float*)xb[i]->dptr; os[i] = (float*)o[i]; }
  rc = nxsig_stft_mel_sharded_f32(g->grp, xs, length, batch, axis == NXSIG_SHARD_CHANNELS ? length : 0, (const float*)w.data, &p, bins,
                                  (const float*)f.data, axis, os, NULL, NXSIG_DEVICE);
  if (rc) { free_members(g, n, o); return mk_error(env, rc); }
  return enif_make_tuple3(env, mk_atom(env, "ok"), bufs_to_list(env, g, n, o, bytes), enif_make_int64(env, m));
}

static int load(ErlNifEnv* env, void** priv, ERL_NIF_TERM info) {
  (void)priv; (void)info;
  CTX_RES = enif_open_resource_type(env, NULL, "nxsig_ctx", ctx_dtor, ERL_NIF_RT_CREATE | ERL_NIF_RT_TAKEOVER, NULL);
  BUF_RES = enif_open_resource_type(env, NULL, "nxsig_buf", buf_dtor, ERL_NIF_RT_CREATE | ERL_NIF_RT_TAKEOVER, NULL);
  GRP_RES = enif_open_resource_type(env, NULL, "nxsig_group", grp_dtor, ERL_NIF_RT_CREATE | ERL_NIF_RT_TAKEOVER, NULL);
  return (CTX_RES && BUF_RES && GRP_RES) ? 0 : -1;
}
static int upgrade(ErlNifEnv* env, void** priv, void** old, ERL_NIF_TERM info) { (void)old; return load(env, priv, info); }

static ErlNifFunc funcs[] = {
    {"device_count", 0, nif_device_count, 0},
    {"ctx_create", 1, nif_ctx_create, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"sync", 1, nif_sync, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"last_dispatch", 1, nif_last_dispatch, ERL_NIF_DIRTY_JOB_IO_BOUND},   /* takes the context mutex: may wait behind a running call */
    {"window", 5, nif_window, 0},
    {"firwin", 7, nif_firwin, 0},
    {"fft_frequencies", 3, nif_fft_frequencies, 0},
    {"mel_filters", 5, nif_mel_filters, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"sinc", 1, nif_sinc, 0},
    {"stft", 6, nif_stft, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_c64", 6, nif_stft_c64, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"istft", 6, nif_istft, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"istft_filtered", 7, nif_istft_filtered, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fir", 6, nif_fir, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"as_windowed", 9, nif_as_windowed, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"overlap_and_add", 7, nif_overlap_and_add, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fft", 7, nif_fft, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fftconvolve_c64", 4, nif_fftconvolve_c64, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fft_nd", 7, nif_fft_nd, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fftconvolve_nd", 8, nif_fftconvolve_nd, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"convolve_direct", 8, nif_convolve_direct, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_to_mel", 6, nif_stft_to_mel, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_mel", 8, nif_stft_mel, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_magnitude", 7, nif_stft_magnitude, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"to_device", 2, nif_to_device, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"from_device", 1, nif_from_device, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"buf_size", 1, nif_buf_size, 0},
    {"stft_dev", 6, nif_stft_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_c64_dev", 6, nif_stft_c64_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_onesided_dev", 6, nif_stft_onesided_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_packed_dev", 6, nif_stft_packed_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"istft_packed_dev", 6, nif_istft_packed_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"istft_dev", 6, nif_istft_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"istft_filtered_dev", 7, nif_istft_filtered_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fir_dev", 6, nif_fir_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"spectrum_mul_dev", 5, nif_spectrum_mul_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"group_create", 1, nif_group_create, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"group_info", 1, nif_group_info, 0},
    {"stft_sharded", 8, nif_stft_sharded, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"istft_sharded", 8, nif_istft_sharded, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fir_sharded", 8, nif_fir_sharded, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_mel_sharded", 9, nif_stft_mel_sharded, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"shard_range", 6, nif_shard_range, 0},
    {"group_scatter", 5, nif_group_scatter, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"group_gather", 5, nif_group_gather, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_sharded_dev", 8, nif_stft_sharded_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"istft_sharded_dev", 8, nif_istft_sharded_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fir_sharded_dev", 8, nif_fir_sharded_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_mel_sharded_dev", 9, nif_stft_mel_sharded_dev, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"window_f64", 5, nif_window_f64, 0},
    {"firwin_f64", 7, nif_firwin_f64, 0},
    {"fft_frequencies_f64", 3, nif_fft_frequencies_f64, 0},
    {"sinc_f64", 1, nif_sinc_f64, 0},
    {"stft_f64", 7, nif_stft_f64, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stft_c128", 7, nif_stft_c128, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"istft_c128", 7, nif_istft_c128, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fir_f64", 6, nif_fir_f64, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"fft_c128", 7, nif_fft_c128, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"as_windowed_f64", 9, nif_as_windowed_f64, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"overlap_and_add_f64", 7, nif_overlap_and_add_f64, ERL_NIF_DIRTY_JOB_IO_BOUND},
};

ERL_NIF_INIT(Elixir.NxSignalAMD.NIF, funcs, load, NULL, upgrade, NULL)
