#!/usr/bin/env python
"""n-D fftconvolve rates (Convolution.fftconvolve/3 on operands of equal rank) on device-resident tensors: a batch of images
against one 2-D kernel, and a long complex 1-D pair.  One JSON object per line.  usage: python tools/bench_conv2d.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

lib = _lib.load()
MODES = {"full": 0, "same": 1, "valid": 2}

ctx = S.Context(0)
rng = np.random.default_rng(0)
for shape, kshape, mode in [((16, 512, 512), (1, 31, 31), "same"), ((4, 1024, 1024), (1, 65, 65), "same"), ((8, 1000, 1000), (1, 9, 9), "full"),
                            ((64, 100000), (1, 513), "same")]:
    a = ctx.to_device(rng.standard_normal(shape).astype(np.float32))
    k = ctx.to_device(rng.standard_normal(kshape).astype(np.float32))
    rank = len(shape)
    s1, s2, osh = (C.c_int64 * rank)(*shape), (C.c_int64 * rank)(*kshape), (C.c_int64 * rank)()
    y = ctx.empty(tuple(x + z - 1 for x, z in zip(shape, kshape)), np.float32)
    fn = lambda: _lib.check(lib.nxsig_fftconvolve_nd(ctx.handle, C.c_void_p(a.ptr), 1, s1, C.c_void_p(k.ptr), 1, s2, rank, MODES[mode],
                                                      C.c_void_p(y.ptr), osh, _lib.DEVICE))
    for _ in range(2):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        fn()
    ctx.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    n_out = int(np.prod([int(v) for v in osh]))
    print(json.dumps({"case": f"fftconvolve {shape} * {kshape} mode={mode} (f32, device-resident)", "ms": ms,
                      "output_Msamples_per_s": n_out / ms / 1e3}), flush=True)
