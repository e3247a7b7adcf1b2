"""Phase time stamps of the pair-mode STFT kernel on ONE launch of BASELINE config 2 as written (diagnostic build only):
    python tools/build_variant.py tools/_ab/libnxsig_trace.so -DNXSIG_TRACE kernels_wave.hip
    python tools/trace_small.py [W:chunk ...]
Every wave stamps wall_clock64 (100 MHz) at: entry, tables staged, first unit windowed, after each unit.  Printed per geometry:
percentiles over waves of each stamp relative to the earliest entry, in microseconds."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nx_signal_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libnxsig_trace.so")
import nx_signal_amd as S  # noqa: E402

cfgs = sys.argv[1:] or ["0:0", "12:0"]
N, hop, L = 1024, 256, int(os.environ.get("SWEEP_L", 2880000))
M = (L - N) // hop + 1
ctx = S.Context(0)
lib = _lib.load()
lib.nxsig_diag_set_trace.restype = C.c_int
lib.nxsig_diag_set_trace.argtypes = [C.c_void_p]
w = S.windows.hann(N)
xd = ctx.to_device(np.random.default_rng(1).standard_normal((1, L)).astype(np.float32))
zd = ctx.empty((1, M, N), np.complex64)
p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
wp = w.ctypes.data_as(C.c_void_p)
NW = 16384
tr = ctx.empty((NW, 8), np.uint64)


def call():
    _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, 1, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, 1))


for v in cfgs:
    ws, ch = (int(t) for t in v.split(":"))
    ctx.set_tuning("WAVE_SMALL_W", ws)
    ctx.set_tuning("WAVE_SMALL_CHUNK", ch)
    assert lib.nxsig_diag_set_trace(None) == 0
    for _ in range(50):
        call()
    ctx.sync()
    _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(tr.ptr), np.zeros((NW, 8), np.uint64).ctypes.data_as(C.c_void_p), NW * 64))
    assert lib.nxsig_diag_set_trace(C.c_void_p(tr.ptr)) == 0
    ctx.timer_start()
    call()
    ms = ctx.timer_stop()
    t = tr.numpy().astype(np.float64)
    W = ws if ws >= 8 else 4
    last = t.max(axis=1)
    nwg = int(np.count_nonzero(t[:, 0] > 0) + W - 1) // W
    lw = last[: nwg * W].reshape(nwg, W)
    ew = t[: nwg * W, 0].reshape(nwg, W)
    ok = (lw > 0).all(axis=1)
    t00 = t[t[:, 0] > 0, 0].min()
    wg_end, wg_first_end, wg_start = (lw[ok].max(axis=1) - t00) / 100, (lw[ok].min(axis=1) - t00) / 100, (ew[ok].min(axis=1) - t00) / 100
    print(f"   per workgroup ({ok.sum()} full ones): start p10/median/p90 {np.percentile(wg_start, 10):.2f}/{np.median(wg_start):.2f}/{np.percentile(wg_start, 90):.2f}  "
          f"last wave ends min/p10/median/p90/max {wg_end.min():.2f}/{np.percentile(wg_end, 10):.2f}/{np.median(wg_end):.2f}/{np.percentile(wg_end, 90):.2f}/{wg_end.max():.2f}  "
          f"spread inside a workgroup (last - first wave end) median {np.median(wg_end - wg_first_end):.2f} max {(wg_end - wg_first_end).max():.2f} us")
    xcd = np.arange(nwg)[ok] % 8
    print("   by XCD (workgroup index mod 8): median start " + " ".join(f"{np.median(wg_start[xcd == k]):.2f}" for k in range(8)) +
          " | median end " + " ".join(f"{np.median(wg_end[xcd == k]):.2f}" for k in range(8)))
    live = t[:, 0] > 0
    t = t[live]
    t0 = t[:, 0].min()
    print(f"== W:chunk={v}: {live.sum()} waves, this launch {ms * 1e3:.2f} us (HIP events)")
    names = ["entry", "tables", "unit0 in", "unit 1 done", "unit 2 done", "unit 3 done", "unit 4 done", "unit 5 done"]
    for i, nm in enumerate(names):
        col = t[:, i]
        col = col[col > 0]
        if col.size == 0:
            continue
        us = (col - t0) / 100.0
        print(f"   {nm:12s} n={col.size:5d}  min {us.min():6.2f}  p10 {np.percentile(us, 10):6.2f}  median {np.median(us):6.2f}  p90 {np.percentile(us, 90):6.2f}  max {us.max():6.2f} us")
