/*
 * TEST INFRASTRUCTURE ONLY — a stand-in for Erlang/OTP's <erl_nif.h>, which this image does not have (no BEAM).
 *
 * It declares the subset of the NIF API that nif/nxsig_nif.c uses, with OTP's names, argument orders and result
 * conventions (written from the erl_nif documentation, not copied from OTP), so that the shim can be (a) compiled with
 * -Wall -Werror in the CPU test-suite and (b) EXECUTED against tests/stub/erl_nif_fake.c, a miniature term runtime, to
 * drive the real shim code on the GPU from the tests (tests/nif_harness.py).  A production build uses the real header:
 * nif/Makefile never looks here.
 */
#ifndef NXSIG_TEST_ERL_NIF_H
#define NXSIG_TEST_ERL_NIF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uintptr_t ERL_NIF_TERM;
typedef struct enif_environment_t ErlNifEnv;
typedef int64_t ErlNifSInt64;
typedef uint64_t ErlNifUInt64;

typedef struct {
  size_t size;
  unsigned char* data;
  void* ref_bin;
  void* spare[2];
} ErlNifBinary;

typedef struct enif_resource_type_t ErlNifResourceType;
typedef void ErlNifResourceDtor(ErlNifEnv* env, void* obj);
typedef enum { ERL_NIF_RT_CREATE = 1, ERL_NIF_RT_TAKEOVER = 2 } ErlNifResourceFlags;

typedef struct enif_func_t {
  const char* name;
  unsigned arity;
  ERL_NIF_TERM (*fptr)(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]);
  unsigned flags;
} ErlNifFunc;

#define ERL_NIF_DIRTY_JOB_CPU_BOUND 1
#define ERL_NIF_DIRTY_JOB_IO_BOUND 2

typedef struct enif_entry_t {
  int major, minor;
  const char* name;
  int num_of_funcs;
  ErlNifFunc* funcs;
  int (*load)(ErlNifEnv*, void** priv_data, ERL_NIF_TERM load_info);
  int (*reload)(ErlNifEnv*, void** priv_data, ERL_NIF_TERM load_info);
  int (*upgrade)(ErlNifEnv*, void** priv_data, void** old_priv_data, ERL_NIF_TERM load_info);
  void (*unload)(ErlNifEnv*, void* priv_data);
} ErlNifEntry;

#define ERL_NIF_INIT(NAME, FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD)                                                \
  ErlNifEntry* nif_init(void) {                                                                                  \
    static ErlNifEntry entry = {2, 17, #NAME, (int)(sizeof(FUNCS) / sizeof(FUNCS[0])), FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD}; \
    return &entry;                                                                                               \
  }

/* terms in */
int enif_get_int(ErlNifEnv*, ERL_NIF_TERM, int* ip);
int enif_get_int64(ErlNifEnv*, ERL_NIF_TERM, ErlNifSInt64* ip);
int enif_get_double(ErlNifEnv*, ERL_NIF_TERM, double* dp);
int enif_get_tuple(ErlNifEnv*, ERL_NIF_TERM, int* arity, const ERL_NIF_TERM** array);
int enif_get_list_length(ErlNifEnv*, ERL_NIF_TERM, unsigned* len);
int enif_get_list_cell(ErlNifEnv*, ERL_NIF_TERM, ERL_NIF_TERM* head, ERL_NIF_TERM* tail);
int enif_inspect_binary(ErlNifEnv*, ERL_NIF_TERM, ErlNifBinary* bin);
int enif_get_atom(ErlNifEnv*, ERL_NIF_TERM, char* buf, unsigned len, int encoding);
#define ERL_NIF_LATIN1 1
/* terms out */
ERL_NIF_TERM enif_make_atom(ErlNifEnv*, const char* name);
ERL_NIF_TERM enif_make_int(ErlNifEnv*, int i);
ERL_NIF_TERM enif_make_int64(ErlNifEnv*, ErlNifSInt64 i);
ERL_NIF_TERM enif_make_double(ErlNifEnv*, double d);
ERL_NIF_TERM enif_make_tuple_from_array(ErlNifEnv*, const ERL_NIF_TERM arr[], unsigned cnt);
ERL_NIF_TERM enif_make_list_from_array(ErlNifEnv*, const ERL_NIF_TERM arr[], unsigned cnt);
unsigned char* enif_make_new_binary(ErlNifEnv*, size_t size, ERL_NIF_TERM* termp);
/* enif_alloc_binary can fail (returns 0); enif_make_binary hands the binary over to a term; enif_release_binary drops one
 * that was never made into a term */
int enif_alloc_binary(size_t size, ErlNifBinary* bin);
ERL_NIF_TERM enif_make_binary(ErlNifEnv*, ErlNifBinary* bin);
void enif_release_binary(ErlNifBinary* bin);
int enif_realloc_binary(ErlNifBinary* bin, size_t size);
ERL_NIF_TERM enif_make_badarg(ErlNifEnv*);
ERL_NIF_TERM enif_make_tuple2(ErlNifEnv*, ERL_NIF_TERM, ERL_NIF_TERM);
ERL_NIF_TERM enif_make_tuple3(ErlNifEnv*, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM);
ERL_NIF_TERM enif_make_tuple4(ErlNifEnv*, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM);
ERL_NIF_TERM enif_make_tuple5(ErlNifEnv*, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM, ERL_NIF_TERM);
/* resources */
ErlNifResourceType* enif_open_resource_type(ErlNifEnv*, const char* module_str, const char* name, ErlNifResourceDtor* dtor,
                                            ErlNifResourceFlags flags, ErlNifResourceFlags* tried);
void* enif_alloc_resource(ErlNifResourceType* type, size_t size);
void enif_release_resource(void* obj);
void enif_keep_resource(void* obj);   /* returns nothing in OTP */
ERL_NIF_TERM enif_make_resource(ErlNifEnv*, void* obj);
int enif_get_resource(ErlNifEnv*, ERL_NIF_TERM, ErlNifResourceType* type, void** objp);

#ifdef __cplusplus
}
#endif
#endif
