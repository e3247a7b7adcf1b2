"""Row-transform rates of nxsig_fft / nxsig_fft_nd on device-resident c64 tensors (tools only).
usage: python tools/bench_rows.py [K ...]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

ctx = S.Context(0)
lib = _lib.load()
rng = np.random.default_rng(0)
for K in [int(a) for a in sys.argv[1:]] or [256, 1024, 2048, 4096, 8192, 16384, 1000, 3000]:
    rows = max(1, (512 << 20) // (K * 8))
    x = ctx.to_device((rng.standard_normal((min(rows, 64), K)) + 1j * rng.standard_normal((min(rows, 64), K))).astype(np.complex64))
    xin = ctx.empty((rows, K), np.complex64)
    for r in range(0, rows, x.shape[0]):
        n = min(x.shape[0], rows - r)
        _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xin.ptr + r * K * 8), x.numpy()[:n].ctypes.data_as(C.c_void_p), n * K * 8))
    out = ctx.empty((rows, K), np.complex64)
    for inv in (0, 1):
        fn = lambda: _lib.check(lib.nxsig_fft(ctx.handle, C.c_void_p(xin.ptr), 0, rows, K, K, inv, C.c_void_p(out.ptr), _lib.DEVICE))
        for _ in range(3):
            fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(10):
            fn()
        ms = ctx.timer_stop() / 10
        print(json.dumps({"case": f"fft rows K={K} inverse={inv}, {rows} rows", "ms": ms, "algorithmic_GBps": rows * K * 16 / (ms * 1e-3) / 1e9}), flush=True)
# 2-D transform over both axes of a 4096 x 4096 tensor
x2 = ctx.to_device((rng.standard_normal((64, 4096)) + 0j).astype(np.complex64))
big = ctx.empty((4096, 4096), np.complex64)
for r in range(0, 4096, 64):
    _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(big.ptr + r * 4096 * 8), x2.numpy().ctypes.data_as(C.c_void_p), 64 * 4096 * 8))
S.transforms.fft_nd(big, axes=[0, 1])
ctx.sync()
import time
t0 = time.perf_counter()
for _ in range(5):
    o = S.transforms.fft_nd(big, axes=[0, 1])
ctx.sync()
ms = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps({"case": "fft_nd 4096 x 4096 c64 over both axes (device-resident)", "ms": ms, "algorithmic_GBps": 4096 * 4096 * 16 * 2 / (ms * 1e-3) / 1e9}))
