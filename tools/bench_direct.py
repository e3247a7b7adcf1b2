#!/usr/bin/env python
"""Convolution.convolve(method: :direct) — the reference's default method — on device-resident f32 tensors: 1-D streams and image
batches (real x real: the register-window kernel).  One JSON object per line."""
import sys, os, time, json, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd())
import nx_signal_amd as S
from nx_signal_amd import _lib
lib = _lib.load(); ctx = S.Context(0); rng = np.random.default_rng(0)
def crandn(shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


CASES = [((8, 1000000), (1, 257), 1, 1, 1), ((1, 8000000), (1, 31), 0, 1, 1), ((16, 512, 512), (1, 9, 9), 1, 1, 1), ((4, 1024, 1024), (1, 31, 31), 1, 1, 1),
         ((8, 1000000), (1, 257), 1, 0, 1), ((8, 1000000), (1, 257), 1, 0, 0)]   # last two: complex stream x real / complex taps
for shape, kshape, mode, a_real, k_real in CASES:
    a = ctx.to_device(rng.standard_normal(shape).astype(np.float32) if a_real else crandn(shape))
    k = ctx.to_device(rng.standard_normal(kshape).astype(np.float32) if k_real else crandn(kshape))
    rank = len(shape)
    s1, s2, osh = (C.c_int64 * rank)(*shape), (C.c_int64 * rank)(*kshape), (C.c_int64 * rank)()
    y = ctx.empty(tuple(x + z - 1 for x, z in zip(shape, kshape)), np.float32 if (a_real and k_real) else np.complex64)
    fn = lambda: _lib.check(lib.nxsig_convolve_direct(ctx.handle, C.c_void_p(a.ptr), a_real, s1, C.c_void_p(k.ptr), k_real, s2, rank, mode, C.c_void_p(y.ptr), osh, _lib.DEVICE))
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(3): fn()
    ctx.sync(); ms = (time.perf_counter() - t0) / 3 * 1e3
    n_out = int(np.prod([int(v) for v in osh])); macs = n_out * int(np.prod(kshape))
    print(json.dumps({"case": f"convolve direct {shape} {'f32' if a_real else 'c64'} * {kshape} {'f32' if k_real else 'c64'}", "ms": ms, "GMAC_per_s": macs / ms / 1e6, "out_Msamples_per_s": n_out / ms / 1e3}), flush=True)
