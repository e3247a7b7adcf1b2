"""Latency of SMALL calls (device-resident, back-to-back launches, HIP events): stft / istft / fir on one utterance-sized input.
usage: python tools/bench_small.py        prints microseconds per call for a list of shapes"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

ctx = S.Context(0)
lib = _lib.load()
rng = np.random.default_rng(3)


def timeit(fn, reps=300):
    for _ in range(20):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps * 1e3


def stft_istft(N, K, hop, L, B):
    w = S.windows.hann(N)
    M = (L - N) // hop + 1
    xd = ctx.to_device(rng.standard_normal((B, L)).astype(np.float32))
    zd = ctx.empty((B, M, K), np.complex64)
    p = _lib.StftParams(N, hop, K, 0, 0, 0, 0, 0, 16000.0)
    wp = w.ctypes.data_as(C.c_void_p)
    f = lambda: _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, B, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))  # noqa: E731
    t_f = timeit(f)
    fam_f = ctx.last_dispatch()
    t_i, fam_i = float("nan"), "-"
    if K == N:
        yd = ctx.empty((B, M * hop + N - hop), np.complex64)
        g = lambda: _lib.check(lib.nxsig_istft_c64(ctx.handle, C.c_void_p(zd.ptr), M, B, wp, C.byref(p), C.c_void_p(yd.ptr), _lib.DEVICE))  # noqa: E731
        t_i = timeit(g)
        fam_i = ctx.last_dispatch()
    print(f"stft N={N} K={K} hop={hop} L={L} B={B} ({B * M} frames): {t_f:7.2f} us [{fam_f}]   istft {t_i:7.2f} us [{fam_i}]", flush=True)


def fir(taps, L, B):
    h = np.ascontiguousarray(S.filters.firwin(taps if taps % 2 else taps + 1, [0.2])[:taps])
    xd = ctx.to_device(rng.standard_normal((B, L)).astype(np.float32))
    yd = ctx.empty((B, L), np.float32)
    f = lambda: _lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(xd.ptr), L, B, L, h.ctypes.data_as(C.c_void_p), taps, _lib.CONV_SAME, C.c_void_p(yd.ptr), _lib.DEVICE))  # noqa: E731
    t = timeit(f)
    print(f"fir {taps} taps L={L} B={B}: {t:7.2f} us [{ctx.last_dispatch()}]", flush=True)


for cfg in [(1024, 1024, 256, 48000, 1), (1024, 1024, 256, 480000, 1), (400, 512, 160, 160000, 1), (400, 512, 160, 160000, 8), (512, 512, 128, 160000, 1),
            (256, 256, 64, 48000, 4), (2048, 2048, 512, 480000, 1), (4096, 4096, 1024, 480000, 1), (400, 400, 160, 160000, 1), (960, 960, 240, 480000, 1),
            (1764, 1764, 441, 441000, 1)]:
    stft_istft(*cfg)
for cfg in [(257, 48000, 1), (257, 480000, 2), (65, 160000, 8), (1025, 480000, 1), (4097, 480000, 1)]:
    fir(*cfg)
