"""STFT-domain filtering chain on config-3 geometry (N=1024 hop=256, 16 x 60 s): spectrum_multiply + istft (two launches,
26 KB of HBM traffic per frame) against nxsig_istft_filtered_c64 (one launch, 10 KB per frame); interleaved rounds, bit-exact
comparison of the outputs (tools only).  usage: python tools/bench_filtered.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

N = int(os.environ.get("SWEEP_N", 1024))
hop, L, B = N // int(os.environ.get("SWEEP_R", 4)), int(os.environ.get("SWEEP_L", 2880000)), int(os.environ.get("SWEEP_B", 16))
M = (L - N) // hop + 1
ctx = S.Context(0)
lib = _lib.load()
w = S.windows.hann(N)
rng = np.random.Generator(np.random.PCG64(1))
x = rng.standard_normal((B, L), dtype=np.float32)
xd = ctx.to_device(x)
zd, _, _ = S.stft(xd, w, ctx=ctx, overlap_length=N - hop, fft_length=N, sampling_rate=48000)
taps = S.filters.firwin(129, [4000], sampling_rate=48000)
h = np.ascontiguousarray(np.fft.fft(np.asarray(taps, np.float64), N).astype(np.complex64))
out_len = M * hop + N - hop
zf = ctx.empty((B, M, N), np.complex64)
y1 = ctx.empty((B, out_len), np.complex64)
y2 = ctx.empty((B, out_len), np.complex64)
p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
wp, hp = w.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p)
V = C.c_void_p


def mul():
    _lib.check(lib.nxsig_spectrum_mul_c64(ctx.handle, V(zd.ptr), B * M, N, hp, V(zf.ptr), 1))


def inv():
    _lib.check(lib.nxsig_istft_c64(ctx.handle, V(zf.ptr), M, B, wp, C.byref(p), V(y1.ptr), 1))


def two_step():
    mul()
    inv()


def fused():
    _lib.check(lib.nxsig_istft_filtered_c64(ctx.handle, V(zd.ptr), M, B, wp, C.byref(p), hp, V(y2.ptr), 1))


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


res = {"spectrum_mul": [], "istft": [], "two_step": [], "fused": []}
for rnd in range(5):
    for k, fn in (("spectrum_mul", mul), ("istft", inv), ("two_step", two_step), ("fused", fused)):
        res[k].append(timed(fn))
same = bool(np.array_equal(y1.numpy().view(np.uint32), y2.numpy().view(np.uint32)))
out = {"workload": f"N={N} hop={hop} {B} x {L} samples, M={M}"}
for k, v in res.items():
    v = sorted(v)
    out[k + "_ms"] = round(v[len(v) // 2], 4)
out["two_step_Mframes_per_s"] = round(B * M / (out["two_step_ms"] * 1e-3) / 1e6, 1)
out["fused_Mframes_per_s"] = round(B * M / (out["fused_ms"] * 1e-3) / 1e6, 1)
out["fused_algorithmic_GBps"] = round(B * M * (N * 8 + hop * 8) / (out["fused_ms"] * 1e-3) / 1e9, 1)
out["speedup"] = round(out["two_step_ms"] / out["fused_ms"], 2)
out["bit_identical_to_two_step"] = same
print(json.dumps(out))
