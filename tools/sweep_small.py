"""Interleaved sweep of the ONE-ROUND geometry of the pair-mode STFT kernel (BASELINE config 2 as written: one 60 s stream per launch).
usage: python tools/sweep_small.py [W:chunk ...]      W = waves per workgroup (0 = the many-round default), chunk = pairs per workgroup
(0 = ceil(pairs / CUs)).  SWEEP_L / SWEEP_B choose the launch; prints frames/s and microseconds per launch, median of 7 rounds."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

cfgs = sys.argv[1:] or ["0:0", "12:0", "8:0"]
N, hop = 1024, 256
L, B = int(os.environ.get("SWEEP_L", 2880000)), int(os.environ.get("SWEEP_B", 1))
M = (L - N) // hop + 1
ctx = S.Context(0)
lib = _lib.load()
w = S.windows.hann(N)
rng = np.random.Generator(np.random.PCG64(1))
xd = ctx.to_device(rng.standard_normal((B, L), dtype=np.float32))
zd = ctx.empty((B, M, N), np.complex64)
p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
wp = w.ctypes.data_as(C.c_void_p)


def run(reps=200):
    for _ in range(10):
        _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, B, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, 1))
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, B, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, 1))
    return ctx.timer_stop() / reps


ref = None
res = {v: [] for v in cfgs}
for rnd in range(7):
    for v in cfgs:
        ws, ch, *rest = (int(t) for t in v.split(":"))   # W:chunk[:store policy]
        ctx.set_tuning("WAVE_SMALL_W", ws)
        ctx.set_tuning("WAVE_SMALL_CHUNK", ch)
        if rest:
            ctx.set_tuning("STORE_POLICY", rest[0])
        else:
            ctx.clear_tuning("STORE_POLICY")
        res[v].append(run())
        if rnd == 0:   # every geometry must write the same bits
            z = zd.numpy()
            if ref is None:
                ref = z
            elif not np.array_equal(ref.view(np.uint32), z.view(np.uint32)):
                print(f"MISMATCH at {v}")
for v in cfgs:
    r = sorted(res[v])
    ms = r[len(r) // 2]
    print(f"W:chunk={v:>8s}  median {ms * 1e3:7.2f} us  {B * M / (ms * 1e-3) / 1e6:7.1f} M frames/s   min {r[0] * 1e3:7.2f} max {r[-1] * 1e3:7.2f} us", flush=True)
