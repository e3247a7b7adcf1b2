"""TEST INFRASTRUCTURE: runs the dirty-NIF shim (nif/nxsig_nif.c) without a BEAM.

The shim is compiled with gcc -std=c11 -Wall -Wextra -Werror against tests/stub/erl_nif.h and linked with
tests/stub/erl_nif_fake.c (a miniature term runtime) and nx_signal_amd/libnxsig.so into tests/_nif_fake.so.  `call(name,
*args)` converts Python values to terms the way the Elixir wrappers build them (int -> integer, float -> float, bytes /
ndarray -> binary, str -> atom, tuple -> tuple, list -> list, Res -> resource), calls the NIF by name / arity through the
ErlNifEntry table exactly as the BEAM does, and converts the result back.  `{:error, {code, msg}}`, badarg and an undefined
function come back as NifError / BadArg / UndefinedNif exceptions."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "_nif_fake.so")
SRC = [os.path.join(ROOT, "nif", "nxsig_nif.c"), os.path.join(ROOT, "tests", "stub", "erl_nif_fake.c")]
T_INT, T_DOUBLE, T_ATOM, T_BIN, T_TUPLE, T_NIL, T_CONS, T_RES, T_BADARG = range(1, 10)


class NifError(Exception):
    def __init__(self, code, msg):
        super().__init__(f"{code}: {msg}")
        self.code, self.msg = code, msg


class BadArg(Exception):
    pass


class UndefinedNif(Exception):
    pass


class Res:
    """a resource term that outlives the call that made it: it sits in an environment of its own (what the BEAM does when a
    term is kept by a process); dropping the Python object frees that environment, i.e. releases the reference"""

    def __init__(self, env, term):
        self.env, self.term = env, term

    def release(self):
        if self.env is not None:
            _lib.fake_env_free(self.env)
            self.env = self.term = None

    def __del__(self):  # pragma: no cover
        try:
            self.release()
        except Exception:
            pass


def build(force=False):
    deps = SRC + [os.path.join(ROOT, "tests", "stub", "erl_nif.h"), os.path.join(ROOT, "include", "nxsig.h")]
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return SO
    libdir = os.path.join(ROOT, "nx_signal_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-g", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-D_GNU_SOURCE",
           "-I" + os.path.join(ROOT, "tests", "stub"), *SRC, "-o", SO, "-L" + libdir, "-lnxsig", "-Wl,-rpath," + libdir]
    subprocess.check_call(cmd)
    return SO


_lib = None
_keep = None  # environment that owns resource terms handed back to Python


def lib():
    global _lib, _keep
    if _lib is None:
        L = C.CDLL(build())
        vp, sz = C.c_void_p, C.c_size_t
        for name, res, args in [
            ("fake_env_new", vp, []), ("fake_env_free", None, [vp]), ("fake_load", C.c_int, []),
            ("fake_num_funcs", C.c_int, []), ("fake_func_name", C.c_char_p, [C.c_int]), ("fake_func_arity", C.c_uint, [C.c_int]),
            ("fake_func_flags", C.c_uint, [C.c_int]), ("fake_module_name", C.c_char_p, []),
            ("fake_call", vp, [vp, C.c_char_p, C.c_int, C.POINTER(vp)]), ("fake_binary", vp, [vp, vp, sz]),
            ("fake_copy_resource", vp, [vp, vp]), ("fake_tag", C.c_int, [vp]), ("fake_atom_name", C.c_char_p, [vp]),
            ("fake_bin_data", vp, [vp]), ("fake_bin_size", sz, [vp]), ("fake_tuple_arity", C.c_int, [vp]),
            ("fake_tuple_elem", vp, [vp, C.c_int]), ("fake_cons_head", vp, [vp]), ("fake_cons_tail", vp, [vp]), ("fake_live_resources", C.c_long, []), ("fake_dtor_calls", C.c_long, []),
            ("fake_live_binaries", C.c_long, []), ("fake_set_alloc_limit", None, [sz]),
            ("enif_make_int64", vp, [vp, C.c_int64]), ("enif_make_double", vp, [vp, C.c_double]), ("enif_make_atom", vp, [vp, C.c_char_p]),
            ("enif_make_tuple_from_array", vp, [vp, C.POINTER(vp), C.c_uint]), ("enif_make_list_from_array", vp, [vp, C.POINTER(vp), C.c_uint]),
            ("enif_get_int64", C.c_int, [vp, vp, C.POINTER(C.c_int64)]), ("enif_get_double", C.c_int, [vp, vp, C.POINTER(C.c_double)]),
        ]:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        assert L.fake_load() == 0
        _keep = L.fake_env_new()
        _lib = L
    return _lib


def funcs():
    L = lib()
    return {(L.fake_func_name(i).decode(), int(L.fake_func_arity(i))): int(L.fake_func_flags(i)) for i in range(L.fake_num_funcs())}


def _to_term(L, env, v):
    if isinstance(v, Res):
        return L.fake_copy_resource(env, v.term)
    if isinstance(v, bool):
        raise TypeError("pass 0 / 1")
    if isinstance(v, (int, np.integer)):
        return L.enif_make_int64(env, int(v))
    if isinstance(v, (float, np.floating)):
        return L.enif_make_double(env, float(v))
    if isinstance(v, str):
        return L.enif_make_atom(env, v.encode())
    if isinstance(v, np.ndarray):
        v = np.ascontiguousarray(v)
        return L.fake_binary(env, v.ctypes.data_as(C.c_void_p), v.nbytes)
    if isinstance(v, (bytes, bytearray)):
        return L.fake_binary(env, bytes(v), len(v))
    if isinstance(v, (tuple, list)):
        arr = (C.c_void_p * max(len(v), 1))(*[_to_term(L, env, e) for e in v])
        return (L.enif_make_tuple_from_array if isinstance(v, tuple) else L.enif_make_list_from_array)(env, arr, len(v))
    raise TypeError(type(v))


def _from_term(L, t):
    tag = L.fake_tag(t)
    if tag == T_INT:
        v = C.c_int64()
        L.enif_get_int64(None, t, C.byref(v))
        return int(v.value)
    if tag == T_DOUBLE:
        v = C.c_double()
        L.enif_get_double(None, t, C.byref(v))
        return float(v.value)
    if tag == T_ATOM:
        return L.fake_atom_name(t).decode()
    if tag == T_BIN:
        return C.string_at(L.fake_bin_data(t), L.fake_bin_size(t))
    if tag == T_TUPLE:
        return tuple(_from_term(L, L.fake_tuple_elem(t, i)) for i in range(L.fake_tuple_arity(t)))
    if tag == T_NIL:
        return []
    if tag == T_CONS:
        out = []
        while L.fake_tag(t) == T_CONS:
            out.append(_from_term(L, L.fake_cons_head(t)))
            t = L.fake_cons_tail(t)
        return out
    if tag == T_RES:
        env = L.fake_env_new()
        return Res(env, L.fake_copy_resource(env, t))
    if tag == T_BADARG:
        raise BadArg()
    raise TypeError(f"term tag {tag}")


def call(name, *args):
    """one NIF call in a fresh environment (freed afterwards, like a NIF call's process-bound env)"""
    L = lib()
    env = L.fake_env_new()
    try:
        argv = (C.c_void_p * max(len(args), 1))(*[_to_term(L, env, a) for a in args])
        t = L.fake_call(env, name.encode(), len(args), argv)
        if not t:
            raise UndefinedNif(f"{name}/{len(args)}")
        out = _from_term(L, t)
    finally:
        L.fake_env_free(env)
    if isinstance(out, tuple) and out and out[0] == "error":
        code, msg = out[1]
        raise NifError(code, msg.decode("utf-8", "replace") if isinstance(msg, bytes) else str(msg))
    return out


def release_all():
    """kept for symmetry with earlier tests: resources are released when their Res objects die"""
    import gc

    gc.collect()
