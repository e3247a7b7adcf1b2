"""Interleaved A/B sweep of the config-5 FIR launch (257 taps :same, 8 ch x 600 s) over an env knob (tools only).
usage: python tools/sweep_fir.py NXSIG_FIR_UNITS_PER_WAVE 8 16 32"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

knob, vals = sys.argv[1], sys.argv[2:]
taps = int(os.environ.get("SWEEP_TAPS", 257))
L, B = int(os.environ.get("SWEEP_L", 28800000)), int(os.environ.get("SWEEP_B", 8))
ctx = S.Context(0)
lib = _lib.load()
h = S.filters.firwin(taps if taps % 2 else taps + 1, [4000.0], sampling_rate=48000)[:taps].copy()
rng = np.random.Generator(np.random.PCG64(1))
xd = ctx.empty((B, L), np.float32)
x = rng.standard_normal(L, dtype=np.float32)
for b in range(B):
    xr = np.roll(x, 997 * b)
    _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xd.ptr + b * L * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes))
yd = ctx.empty((B, L), np.float32)
hp = h.ctypes.data_as(C.c_void_p)


def run(reps=10):
    for _ in range(5):
        _lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(xd.ptr), L, B, L, hp, taps, 1, C.c_void_p(yd.ptr), 1))
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        _lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(xd.ptr), L, B, L, hp, taps, 1, C.c_void_p(yd.ptr), 1))
    return ctx.timer_stop() / reps


res = {v: [] for v in vals}
for rnd in range(5):
    for v in vals:
        ctx.set_tuning(knob, int(v))   # the library reads the environment only at context creation (round 4)
        res[v].append(B * L * 8 / (run() * 1e-3) / 1e9)
for v in vals:
    r = sorted(res[v])
    print(f"{knob}={v:>6s}  median {r[len(r)//2]:7.1f} GB/s   min {r[0]:7.1f}  max {r[-1]:7.1f}")
