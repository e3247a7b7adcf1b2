"""Throughput of the f64 / c128 tier on device-resident data (tools only): stft N=1024 hop=256, its istft, the 257-tap FIR and
Bluestein / small-N shapes.  Algorithmic bytes: stft 8 hop + 16 K per frame, istft 16 K + 16 hop, fir 16 per sample.
usage: python tools/bench_f64.py"""
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
rng = np.random.Generator(np.random.PCG64(1))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


def stft_case(N, hop, K, B, L):
    M = (L - N) // hop + 1
    xd = ctx.to_device(rng.standard_normal((B, L)))
    zd = ctx.empty((B, M, K), np.complex128)
    w = S.windows.hann(N, type="f64")
    p = _lib.StftParams(N, hop, K, 0, 0, 0, 0, 0, 48000.0)
    ms = timed(lambda: _lib.check(lib.nxsig_stft_f64(ctx.handle, C.c_void_p(xd.ptr), L, B, L, w.ctypes.data_as(C.c_void_p), 1, C.byref(p),
                                                     C.c_void_p(zd.ptr), None, 1)))
    by = B * M * (hop * 8 + K * 16)
    out = {"case": f"stft f64 N={N} hop={hop} fft_length={K}, {B} x {L} samples", "ms": ms, "frames_per_s": B * M / (ms * 1e-3),
           "algorithmic_GBps": by / (ms * 1e-3) / 1e9, "frac_of_8TBps": by / (ms * 1e-3) / 8e12}
    if K == N:
        yd = ctx.empty((B, M * hop + N - hop), np.complex128)
        ms2 = timed(lambda: _lib.check(lib.nxsig_istft_c128(ctx.handle, C.c_void_p(zd.ptr), M, B, w.ctypes.data_as(C.c_void_p), 1,
                                                            C.byref(p), C.c_void_p(yd.ptr), 1)))
        by2 = B * M * (K * 16 + hop * 16)
        print(json.dumps({"case": f"istft c128 N={N} hop={hop}, {B} rows", "ms": ms2, "frames_per_s": B * M / (ms2 * 1e-3),
                          "algorithmic_GBps": by2 / (ms2 * 1e-3) / 1e9, "frac_of_8TBps": by2 / (ms2 * 1e-3) / 8e12}))
    print(json.dumps(out))


def fir_case(taps, B, L):
    xd = ctx.to_device(rng.standard_normal((B, L)))
    yd = ctx.empty((B, L), np.float64)
    h = S.filters.firwin(taps, [0.2], type="f64")
    ms = timed(lambda: _lib.check(lib.nxsig_fir_f64(ctx.handle, C.c_void_p(xd.ptr), L, B, L, h.ctypes.data_as(C.c_void_p), taps, 1,
                                                    C.c_void_p(yd.ptr), 1)))
    by = B * L * 16
    print(json.dumps({"case": f"fir f64 {taps} taps :same, {B} x {L} samples", "ms": ms, "samples_per_s": B * L / (ms * 1e-3),
                      "algorithmic_GBps": by / (ms * 1e-3) / 1e9, "frac_of_8TBps": by / (ms * 1e-3) / 8e12}))


stft_case(1024, 256, 1024, 8, 2880000)
stft_case(512, 128, 512, 8, 2880000)
stft_case(2048, 512, 2048, 8, 2880000)
stft_case(400, 160, 512, 8, 2880000)
stft_case(1000, 250, 1000, 8, 1440000)
fir_case(257, 8, 2880000)
fir_case(1025, 8, 2880000)
