"""ctypes wrapper of the C leg of the oracle (oracle/bb_baseline.c).  Test/bench infrastructure only."""
import ctypes as C
import os

import numpy as np

from . import build_baseline

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = build_baseline.OUT if os.path.exists(build_baseline.OUT) else build_baseline.build()
        _lib = C.CDLL(path)
        _lib.bb_stft_f32.restype = C.c_int64
        _lib.bb_stft_f32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                     C.c_void_p, C.c_int32]
        _lib.bb_stft_f32_repeat.restype = C.c_int64
        _lib.bb_stft_f32_repeat.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                            C.c_void_p, C.c_int32, C.c_int32]
        _lib.bb_max_threads.restype = C.c_int
    return _lib


def max_threads() -> int:
    return int(_load().bb_max_threads())


def stft(x, w, hop, K, eps=1.0e-10, threads=1):
    """:valid-padded stft of a mono f32 signal with the BinaryBackend-style recursive radix-2 in double."""
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    N = w.shape[0]
    M = (x.shape[0] - N) // hop + 1
    z = np.empty((M, K), dtype=np.complex64)
    got = lib.bb_stft_f32(x.ctypes.data_as(C.c_void_p), x.shape[0], w.ctypes.data_as(C.c_void_p), N, hop, K, eps,
                          z.ctypes.data_as(C.c_void_p), threads)
    if got != M:
        raise ValueError("bb_stft_f32 failed")
    return z


def stft_repeat(x, w, hop, K, reps, threads, eps=1.0e-10, out=None):
    """`reps` passes over the stream shared out over `threads` threads (all-cores throughput leg); returns (frames done, z of
    pass 0).  `out` lets the caller pass a pre-faulted result buffer."""
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    N = w.shape[0]
    M = (x.shape[0] - N) // hop + 1
    z = np.zeros((M, K), dtype=np.complex64) if out is None else out
    got = lib.bb_stft_f32_repeat(x.ctypes.data_as(C.c_void_p), x.shape[0], w.ctypes.data_as(C.c_void_p), N, hop, K, eps,
                                 z.ctypes.data_as(C.c_void_p), int(reps), int(threads))
    if got != reps * M:
        raise ValueError("bb_stft_f32_repeat failed")
    return int(got), z
