"""Interleaved A/B sweep of the config-3 iSTFT launch (N=1024 hop=256, 16 x 60 s) over an env knob, several rounds in ONE
process (tools only).  usage: python tools/sweep_istft.py NXSIG_ISTFT_T 0 1"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

knob, vals = sys.argv[1], sys.argv[2:]
N = int(os.environ.get("SWEEP_N", 1024))
hop, L, B = N // int(os.environ.get("SWEEP_R", 4)), int(os.environ.get("SWEEP_L", 2880000)), int(os.environ.get("SWEEP_B", 16))
M = (L - N) // hop + 1
ctx = S.Context(0)
lib = _lib.load()
w = S.windows.hann(N)
rng = np.random.Generator(np.random.PCG64(1))
xd = ctx.empty((B, L), np.float32)
x = rng.standard_normal(L, dtype=np.float32)
for b in range(B):
    xr = np.roll(x, 997 * b)
    _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xd.ptr + b * L * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes))
zd, _, _ = S.stft(xd, w, ctx=ctx, overlap_length=N - hop, fft_length=N, sampling_rate=48000)
yd = ctx.empty((B, M * hop + N - hop), np.complex64)
p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
wp = w.ctypes.data_as(C.c_void_p)


def run(reps=20):
    for _ in range(5):
        _lib.check(lib.nxsig_istft_c64(ctx.handle, C.c_void_p(zd.ptr), M, B, wp, C.byref(p), C.c_void_p(yd.ptr), 1))
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        _lib.check(lib.nxsig_istft_c64(ctx.handle, C.c_void_p(zd.ptr), M, B, wp, C.byref(p), C.c_void_p(yd.ptr), 1))
    return ctx.timer_stop() / reps


res = {v: [] for v in vals}
outs = {}
for rnd in range(5):
    for v in vals:
        ctx.set_tuning(knob, int(v))   # the library reads the environment only at context creation (round 4)
        res[v].append(B * M * (N * 8 + hop * 8) / (run() * 1e-3) / 1e9)
        if rnd == 0:
            outs[v] = yd.numpy()[0, :200000].copy()
for v in vals:
    r = sorted(res[v])
    d = float(np.max(np.abs(outs[v] - outs[vals[0]])) / np.max(np.abs(outs[vals[0]])))
    print(f"{knob}={v:>6s}  median {r[len(r)//2]:7.1f} GB/s   min {r[0]:7.1f}  max {r[-1]:7.1f}   max diff vs first variant {d:.2e}")
