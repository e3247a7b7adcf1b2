"""Interleaved A/B sweep of the headline STFT launch over an env knob, several rounds in ONE process (tools only).
usage: python tools/sweep_stft.py NXSIG_WAVE_UNITS_PER_CU 48 96 192 384"""
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
K = int(os.environ.get("SWEEP_K", N))            # fft_length (>= N: zero-padded frames, e.g. the 400-in-512 speech framing)
hop, L, B = int(os.environ.get("SWEEP_HOP", N // 4)), int(os.environ.get("SWEEP_L", 2880000)), int(os.environ.get("SWEEP_B", 32))
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
zd = ctx.empty((B, M, K), np.complex64)
p = _lib.StftParams(N, hop, K, 0, 0, 0, 0, 0, 48000.0)
wp = w.ctypes.data_as(C.c_void_p)


def run(reps=int(os.environ.get("SWEEP_REPS", 20))):
    for _ in range(3):
        _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, B, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, 1))
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, B, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, 1))
    return ctx.timer_stop() / reps


res = {v: [] for v in vals}
for rnd in range(5):
    for v in vals:
        ctx.set_tuning(knob, int(v))   # the library reads the environment only at context creation (round 4)
        res[v].append(B * M * (hop * 4 + K * 8) / (run() * 1e-3) / 1e9)
for v in vals:
    r = sorted(res[v])
    med = r[len(r) // 2]
    print(f"{knob}={v:>6s}  median {med:7.1f} GB/s   min {r[0]:7.1f}  max {r[-1]:7.1f}   ({B * M * (hop * 4 + K * 8) / med / 1e3:8.2f} us per launch)")
