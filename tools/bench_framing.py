"""as_windowed / overlap_and_add as standalone calls (rows a3 / a9: the fused kernels never materialise frames, these entry points do)
on 32 x 60 s, window 1024, stride 256 (tools only)."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

ctx = S.Context(0)
lib = _lib.load()
B, L, N, hop = 32, 2880000, 1024, 256
M = (L - N) // hop + 1
x = ctx.to_device(np.random.default_rng(0).standard_normal((B, L), dtype=np.float32))
fr = ctx.empty((B, M, N), np.float32)
y = ctx.empty((B, M * hop + N - hop), np.float32)
Mo = C.c_int64()


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


t1 = timed(lambda: _lib.check(lib.nxsig_as_windowed_f32(ctx.handle, C.c_void_p(x.ptr), L, B, L, N, hop, _lib.PAD_VALID, 0, 0, C.c_void_p(fr.ptr), C.byref(Mo), _lib.DEVICE)))
t2 = timed(lambda: _lib.check(lib.nxsig_overlap_and_add(ctx.handle, C.c_void_p(fr.ptr), M, B, N, N - hop, 1, C.c_void_p(y.ptr), _lib.DEVICE)))
print(json.dumps({"as_windowed_ms": t1, "as_windowed_GBps": B * M * N * 4 / t1 / 1e6, "overlap_and_add_ms": t2, "overlap_and_add_GBps": B * M * N * 4 / t2 / 1e6}))
