#!/usr/bin/env python
"""Shape scan (tools only): istft and fir on many-short-rows and one-long-row shapes, device-resident, HIP-event timing.
Looks for cliffs the BASELINE shapes do not show.  usage: python tools/scan_shapes.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402


def timeit(ctx, fn, reps=10, warm=5):
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


def main():
    ctx = S.Context(0)
    lib = _lib.load()
    rng = np.random.default_rng(3)
    # forward, window_padding :reflect, thousands of short rows (every row has two edge units)
    for N, hop, K, rows, L, pad in ((400, 160, 400, 4096, 16000, _lib.PAD_REFLECT), (400, 160, 400, 4096, 16000, _lib.PAD_VALID), (400, 160, 512, 4096, 16000, _lib.PAD_REFLECT),
                                    (1024, 256, 1024, 2048, 48000, _lib.PAD_REFLECT), (1024, 256, 1024, 2048, 48000, _lib.PAD_VALID), (320, 160, 320, 8192, 16000, _lib.PAD_REFLECT)):
        Lp = L + (N // 2) * 2 if pad == _lib.PAD_REFLECT else L
        M = (Lp - N) // hop + 1
        xd = ctx.empty((rows, L), np.float32)
        x1 = rng.standard_normal(L).astype(np.float32)
        for r in range(rows):   # every row: uninitialised rows may hold Inf / NaN bit patterns, which take the kernels' cold routes
            _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xd.ptr + r * L * 4), x1.ctypes.data_as(C.c_void_p), x1.nbytes))
        zd = ctx.empty((rows, M, K), np.complex64)
        w = S.windows.hann(N)
        p = _lib.StftParams(N, hop, K, pad, 0, 0, _lib.SCALE_NONE, 0, 16000.0)
        fn = lambda: _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, rows, L, w.ctypes.data_as(C.c_void_p), C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))
        ms = timeit(ctx, fn)
        gbs = rows * M * (hop * 4 + K * 8) / (ms * 1e-3) / 1e9
        print(json.dumps({"case": f"stft N={N} hop={hop} K={K} pad={'reflect' if pad else 'valid'}, {rows} rows x {L} samples", "ms": round(ms, 4), "frac_of_8TBps": round(gbs / 8000, 4)}), flush=True)
        del xd, zd
    for N, hop, rows, M in ((1024, 256, 2048, 184), (1024, 256, 1, 400000), (1024, 256, 64, 6000), (512, 128, 4096, 120), (400, 160, 4096, 98),
                            (2048, 512, 512, 300), (256, 64, 8192, 100), (320, 160, 8192, 99)):
        z1 = (rng.standard_normal((min(M, 64), N)) + 1j * rng.standard_normal((min(M, 64), N))).astype(np.complex64)
        zd = ctx.empty((rows, M, N), np.complex64)
        chunk = np.ascontiguousarray(np.tile(z1, ((M + len(z1) - 1) // len(z1), 1))[:M])
        for r in range(rows):
            _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(zd.ptr + r * M * N * 8), chunk.ctypes.data_as(C.c_void_p), chunk.nbytes))
        w = S.windows.hann(N)
        out_len = M * hop + N - hop
        yd = ctx.empty((rows, out_len), np.complex64)
        p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
        fn = lambda: _lib.check(lib.nxsig_istft_c64(ctx.handle, C.c_void_p(zd.ptr), M, rows, w.ctypes.data_as(C.c_void_p), C.byref(p), C.c_void_p(yd.ptr), _lib.DEVICE))
        ms = timeit(ctx, fn)
        gbs = rows * M * (N * 8 + hop * 8) / (ms * 1e-3) / 1e9
        print(json.dumps({"case": f"istft N={N} hop={hop}, {rows} rows x {M} frames", "ms": round(ms, 4), "frac_of_8TBps": round(gbs / 8000, 4)}), flush=True)
        del zd, yd
    for taps, rows, L in ((257, 2048, 48000), (257, 1, 100000000), (101, 4096, 16000), (257, 64, 1000000), (1025, 2048, 48000), (33, 8192, 16000)):
        h = S.filters.firwin(taps, [0.2])
        xd = ctx.empty((rows, L), np.float32)
        x1 = rng.standard_normal(L).astype(np.float32)
        for r in range(rows):
            _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xd.ptr + r * L * 4), x1.ctypes.data_as(C.c_void_p), x1.nbytes))
        yd = ctx.empty((rows, L), np.float32)
        fn = lambda: _lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(xd.ptr), L, rows, L, h.ctypes.data_as(C.c_void_p), taps, _lib.CONV_SAME, C.c_void_p(yd.ptr), _lib.DEVICE))
        ms = timeit(ctx, fn)
        gbs = rows * L * 8 / (ms * 1e-3) / 1e9
        print(json.dumps({"case": f"fir {taps} taps :same, {rows} rows x {L} samples", "ms": round(ms, 4), "frac_of_8TBps": round(gbs / 8000, 4)}), flush=True)
        del xd, yd


if __name__ == "__main__":
    main()
