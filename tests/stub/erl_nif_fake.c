/*
 * TEST INFRASTRUCTURE ONLY — a miniature term runtime behind tests/stub/erl_nif.h, so that the dirty-NIF shim
 * (nif/nxsig_nif.c) can be executed without a BEAM: tests/nif_harness.py builds shim + this file into one shared object,
 * builds argument terms through the fake_* helpers below, calls a NIF by name / arity exactly as the BEAM would
 * (entry->funcs[i].fptr(env, argc, argv)) and reads the result terms back.  Semantics follow the erl_nif documentation:
 * getters return 0 on a type mismatch, binaries made by enif_make_new_binary belong to the environment, resources are
 * reference counted and their destructor runs when the last reference goes (the environment holds one per resource term).
 */
#include "erl_nif.h"

#include <stdlib.h>
#include <string.h>

enum { T_INT = 1, T_DOUBLE, T_ATOM, T_BIN, T_TUPLE, T_NIL, T_CONS, T_RES, T_BADARG };

typedef struct term {
  int tag;
  struct term* next_in_env;
  union {
    int64_t i;
    double d;
    char* atom;
    struct { unsigned char* data; size_t size; } bin;
    struct { int n; ERL_NIF_TERM* e; } tup;
    struct { ERL_NIF_TERM head, tail; } cons;
    void* res;
  } u;
} term;

struct enif_environment_t { term* all; };
struct enif_resource_type_t { char name[64]; ErlNifResourceDtor* dtor; };
typedef struct { long refc; ErlNifResourceType* type; size_t size; } res_hdr;

static term* T(ERL_NIF_TERM t) { return (term*)t; }
static term* new_term(ErlNifEnv* env, int tag) {
  term* t = (term*)calloc(1, sizeof(term));
  t->tag = tag; t->next_in_env = env->all; env->all = t;
  return t;
}

ErlNifEnv* fake_env_new(void) { return (ErlNifEnv*)calloc(1, sizeof(ErlNifEnv)); }
void fake_env_free(ErlNifEnv* env) {
  term* t = env->all;
  while (t) {
    term* n = t->next_in_env;
    if (t->tag == T_ATOM) free(t->u.atom);
    if (t->tag == T_BIN) free(t->u.bin.data);
    if (t->tag == T_TUPLE) free(t->u.tup.e);
    if (t->tag == T_RES) enif_release_resource(t->u.res);
    free(t);
    t = n;
  }
  free(env);
}

/* ---- erl_nif API ---- */
int enif_get_int64(ErlNifEnv* e, ERL_NIF_TERM t, ErlNifSInt64* ip) { (void)e; if (!t || T(t)->tag != T_INT) return 0; *ip = T(t)->u.i; return 1; }
int enif_get_int(ErlNifEnv* e, ERL_NIF_TERM t, int* ip) {
  ErlNifSInt64 v;
  if (!enif_get_int64(e, t, &v) || v < -2147483647 - 1 || v > 2147483647) return 0;
  *ip = (int)v; return 1;
}
int enif_get_double(ErlNifEnv* e, ERL_NIF_TERM t, double* dp) { (void)e; if (!t || T(t)->tag != T_DOUBLE) return 0; *dp = T(t)->u.d; return 1; }  /* integers are NOT floats, as in OTP */
int enif_get_tuple(ErlNifEnv* e, ERL_NIF_TERM t, int* arity, const ERL_NIF_TERM** arr) {
  (void)e; if (!t || T(t)->tag != T_TUPLE) return 0; *arity = T(t)->u.tup.n; *arr = T(t)->u.tup.e; return 1;
}
int enif_get_list_length(ErlNifEnv* e, ERL_NIF_TERM t, unsigned* len) {
  (void)e; unsigned n = 0;
  while (t && T(t)->tag == T_CONS) { ++n; t = T(t)->u.cons.tail; }
  if (!t || T(t)->tag != T_NIL) return 0;
  *len = n; return 1;
}
int enif_get_list_cell(ErlNifEnv* e, ERL_NIF_TERM t, ERL_NIF_TERM* head, ERL_NIF_TERM* tail) {
  (void)e; if (!t || T(t)->tag != T_CONS) return 0; *head = T(t)->u.cons.head; *tail = T(t)->u.cons.tail; return 1;
}
int enif_inspect_binary(ErlNifEnv* e, ERL_NIF_TERM t, ErlNifBinary* b) {
  (void)e; if (!t || T(t)->tag != T_BIN) return 0; b->size = T(t)->u.bin.size; b->data = T(t)->u.bin.data; b->ref_bin = NULL; return 1;
}
int enif_get_atom(ErlNifEnv* e, ERL_NIF_TERM t, char* buf, unsigned len, int enc) {
  (void)e; (void)enc; if (!t || T(t)->tag != T_ATOM) return 0;
  size_t n = strlen(T(t)->u.atom); if (n + 1 > len) return 0;
  memcpy(buf, T(t)->u.atom, n + 1); return (int)n + 1;
}
ERL_NIF_TERM enif_make_atom(ErlNifEnv* env, const char* name) { term* t = new_term(env, T_ATOM); t->u.atom = strdup(name); return (ERL_NIF_TERM)t; }
ERL_NIF_TERM enif_make_int64(ErlNifEnv* env, ErlNifSInt64 i) { term* t = new_term(env, T_INT); t->u.i = i; return (ERL_NIF_TERM)t; }
ERL_NIF_TERM enif_make_int(ErlNifEnv* env, int i) { return enif_make_int64(env, i); }
ERL_NIF_TERM enif_make_double(ErlNifEnv* env, double d) { term* t = new_term(env, T_DOUBLE); t->u.d = d; return (ERL_NIF_TERM)t; }
ERL_NIF_TERM enif_make_tuple_from_array(ErlNifEnv* env, const ERL_NIF_TERM arr[], unsigned cnt) {
  term* t = new_term(env, T_TUPLE);
  t->u.tup.n = (int)cnt; t->u.tup.e = (ERL_NIF_TERM*)malloc(sizeof(ERL_NIF_TERM) * (cnt ? cnt : 1));
  if (cnt) memcpy(t->u.tup.e, arr, sizeof(ERL_NIF_TERM) * cnt);
  return (ERL_NIF_TERM)t;
}
ERL_NIF_TERM enif_make_list_from_array(ErlNifEnv* env, const ERL_NIF_TERM arr[], unsigned cnt) {
  ERL_NIF_TERM l = (ERL_NIF_TERM)new_term(env, T_NIL);
  for (unsigned i = cnt; i > 0; --i) { term* c = new_term(env, T_CONS); c->u.cons.head = arr[i - 1]; c->u.cons.tail = l; l = (ERL_NIF_TERM)c; }
  return l;
}
unsigned char* enif_make_new_binary(ErlNifEnv* env, size_t size, ERL_NIF_TERM* termp) {
  term* t = new_term(env, T_BIN);
  t->u.bin.size = size; t->u.bin.data = (unsigned char*)malloc(size ? size : 1);
  if (!t->u.bin.data) abort();  /* the BEAM aborts the VM when a binary cannot be allocated: the shim must validate sizes first */
  *termp = (ERL_NIF_TERM)t;
  return t->u.bin.data;
}
static long g_live_binaries = 0;  /* allocated by enif_alloc_binary and neither made into a term nor released: a leak */
static size_t g_alloc_limit = (size_t)1 << 40;
void fake_set_alloc_limit(size_t bytes) { g_alloc_limit = bytes; }
int enif_alloc_binary(size_t size, ErlNifBinary* bin) {
  if (size > g_alloc_limit) return 0;
  bin->data = (unsigned char*)malloc(size ? size : 1);
  if (!bin->data) return 0;
  bin->size = size; bin->ref_bin = bin->data; ++g_live_binaries;
  return 1;
}
ERL_NIF_TERM enif_make_binary(ErlNifEnv* env, ErlNifBinary* bin) {
  term* t = new_term(env, T_BIN);
  t->u.bin.size = bin->size; t->u.bin.data = bin->data; --g_live_binaries;
  bin->data = NULL; bin->ref_bin = NULL;
  return (ERL_NIF_TERM)t;
}
void enif_release_binary(ErlNifBinary* bin) { if (bin->ref_bin) { free(bin->ref_bin); --g_live_binaries; bin->data = NULL; bin->ref_bin = NULL; } }
int enif_realloc_binary(ErlNifBinary* bin, size_t size) {
  unsigned char* p = (unsigned char*)realloc(bin->data, size ? size : 1);
  if (!p) return 0;
  bin->data = p; bin->ref_bin = p; bin->size = size;
  return 1;
}
long fake_live_binaries(void) { return g_live_binaries; }
ERL_NIF_TERM enif_make_badarg(ErlNifEnv* env) { return (ERL_NIF_TERM)new_term(env, T_BADARG); }
ERL_NIF_TERM enif_make_tuple2(ErlNifEnv* e, ERL_NIF_TERM a, ERL_NIF_TERM b) { ERL_NIF_TERM v[2] = {a, b}; return enif_make_tuple_from_array(e, v, 2); }
ERL_NIF_TERM enif_make_tuple3(ErlNifEnv* e, ERL_NIF_TERM a, ERL_NIF_TERM b, ERL_NIF_TERM c) { ERL_NIF_TERM v[3] = {a, b, c}; return enif_make_tuple_from_array(e, v, 3); }
ERL_NIF_TERM enif_make_tuple4(ErlNifEnv* e, ERL_NIF_TERM a, ERL_NIF_TERM b, ERL_NIF_TERM c, ERL_NIF_TERM d) { ERL_NIF_TERM v[4] = {a, b, c, d}; return enif_make_tuple_from_array(e, v, 4); }
ERL_NIF_TERM enif_make_tuple5(ErlNifEnv* e, ERL_NIF_TERM a, ERL_NIF_TERM b, ERL_NIF_TERM c, ERL_NIF_TERM d, ERL_NIF_TERM f) { ERL_NIF_TERM v[5] = {a, b, c, d, f}; return enif_make_tuple_from_array(e, v, 5); }

static long g_live_resources = 0, g_dtor_calls = 0;
ErlNifResourceType* enif_open_resource_type(ErlNifEnv* env, const char* mod, const char* name, ErlNifResourceDtor* dtor,
                                            ErlNifResourceFlags flags, ErlNifResourceFlags* tried) {
  (void)env; (void)mod; (void)flags;
  ErlNifResourceType* t = (ErlNifResourceType*)calloc(1, sizeof *t);
  strncpy(t->name, name, sizeof t->name - 1); t->dtor = dtor;
  if (tried) *tried = ERL_NIF_RT_CREATE;
  return t;
}
void* enif_alloc_resource(ErlNifResourceType* type, size_t size) {
  res_hdr* h = (res_hdr*)calloc(1, sizeof(res_hdr) + size);
  h->refc = 1; h->type = type; h->size = size; ++g_live_resources;
  return h + 1;
}
void enif_keep_resource(void* obj) { ++((res_hdr*)obj - 1)->refc; }   /* void in OTP: the shim must not use a result */
void enif_release_resource(void* obj) {
  res_hdr* h = (res_hdr*)obj - 1;
  if (--h->refc == 0) { if (h->type->dtor) { ++g_dtor_calls; h->type->dtor(NULL, obj); } --g_live_resources; free(h); }
}
ERL_NIF_TERM enif_make_resource(ErlNifEnv* env, void* obj) { term* t = new_term(env, T_RES); t->u.res = obj; enif_keep_resource(obj); return (ERL_NIF_TERM)t; }
int enif_get_resource(ErlNifEnv* e, ERL_NIF_TERM t, ErlNifResourceType* type, void** objp) {
  (void)e; if (!t || T(t)->tag != T_RES || ((res_hdr*)T(t)->u.res - 1)->type != type) return 0; *objp = T(t)->u.res; return 1;
}

/* ---- helpers for the Python driver ---- */
ErlNifEntry* nif_init(void);
static int g_loaded = 0;
int fake_load(void) {
  if (g_loaded) return 0;
  ErlNifEntry* en = nif_init();
  ErlNifEnv* env = fake_env_new();
  void* priv = NULL;
  int rc = en->load ? en->load(env, &priv, 0) : 0;
  fake_env_free(env);
  g_loaded = rc == 0;
  return rc;
}
int fake_num_funcs(void) { return nif_init()->num_of_funcs; }
const char* fake_func_name(int i) { return nif_init()->funcs[i].name; }
unsigned fake_func_arity(int i) { return nif_init()->funcs[i].arity; }
unsigned fake_func_flags(int i) { return nif_init()->funcs[i].flags; }
const char* fake_module_name(void) { return nif_init()->name; }
ERL_NIF_TERM fake_call(ErlNifEnv* env, const char* name, int argc, const ERL_NIF_TERM* argv) {
  ErlNifEntry* en = nif_init();
  for (int i = 0; i < en->num_of_funcs; ++i)
    if (!strcmp(en->funcs[i].name, name) && (int)en->funcs[i].arity == argc) return en->funcs[i].fptr(env, argc, argv);
  return 0; /* undefined function */
}
ERL_NIF_TERM fake_binary(ErlNifEnv* env, const void* data, size_t n) {
  ERL_NIF_TERM t; unsigned char* p = enif_make_new_binary(env, n, &t); if (n) memcpy(p, data, n); return t;
}
/* a resource term made in one environment can be re-bound into another (what the BEAM does when a term is sent on) */
ERL_NIF_TERM fake_copy_resource(ErlNifEnv* env, ERL_NIF_TERM t) { return (t && T(t)->tag == T_RES) ? enif_make_resource(env, T(t)->u.res) : 0; }
int fake_tag(ERL_NIF_TERM t) { return t ? T(t)->tag : 0; }
const char* fake_atom_name(ERL_NIF_TERM t) { return T(t)->u.atom; }
const unsigned char* fake_bin_data(ERL_NIF_TERM t) { return T(t)->u.bin.data; }
size_t fake_bin_size(ERL_NIF_TERM t) { return T(t)->u.bin.size; }
int fake_tuple_arity(ERL_NIF_TERM t) { return T(t)->u.tup.n; }
ERL_NIF_TERM fake_tuple_elem(ERL_NIF_TERM t, int i) { return T(t)->u.tup.e[i]; }
ERL_NIF_TERM fake_cons_head(ERL_NIF_TERM t) { return T(t)->u.cons.head; }
ERL_NIF_TERM fake_cons_tail(ERL_NIF_TERM t) { return T(t)->u.cons.tail; }
long fake_live_resources(void) { return g_live_resources; }
long fake_dtor_calls(void) { return g_dtor_calls; }
