"""The packed one-sided pair against the full two-sided path on config-3 geometry (N = 1024, hop = 256, 16 x 60 s, device-resident):
stft vs stft_packed, istft vs istft_packed, and the round trip (tools only).   usage: python tools/bench_packed.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

N, hop, L, B = 1024, 256, 2880000, int(os.environ.get("SWEEP_B", 16))
M = (L - N) // hop + 1
ctx = S.Context(0)
lib = _lib.load()
w = S.windows.hann(N)
rng = np.random.Generator(np.random.PCG64(1))
x = rng.standard_normal((B, L), dtype=np.float32)
xd = ctx.to_device(x)
out_len = M * hop + N - hop
zf = ctx.empty((B, M, N), np.complex64)
zp = ctx.empty((B, M, N // 2), np.complex64)
yf = ctx.empty((B, out_len), np.complex64)
yp = ctx.empty((B, out_len), np.float32)
p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
wp = w.ctypes.data_as(C.c_void_p)
V = C.c_void_p
calls = {
    "stft (c64[M][1024] out)": (lambda: lib.nxsig_stft_f32(ctx.handle, V(xd.ptr), L, B, L, wp, C.byref(p), V(zf.ptr), None, 1), hop * 4 + N * 8),
    "stft_packed (c64[M][512] out)": (lambda: lib.nxsig_stft_packed_f32(ctx.handle, V(xd.ptr), L, B, L, wp, C.byref(p), V(zp.ptr), None, 1), hop * 4 + N * 4),
    "istft (c64 in, c64 out)": (lambda: lib.nxsig_istft_c64(ctx.handle, V(zf.ptr), M, B, wp, C.byref(p), V(yf.ptr), 1), N * 8 + hop * 8),
    "istft_packed (packed in, f32 out)": (lambda: lib.nxsig_istft_packed_f32(ctx.handle, V(zp.ptr), M, B, wp, C.byref(p), V(yp.ptr), 1), N * 4 + hop * 4),
}


def timed(fn, reps=20):
    for _ in range(5):
        _lib.check(fn())
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        _lib.check(fn())
    return ctx.timer_stop() / reps


res = {k: [] for k in calls}
for _ in range(5):
    for k, (fn, _) in calls.items():
        res[k].append(timed(fn))
for k, (fn, bpf) in calls.items():
    ms = sorted(res[k])[2]
    print(json.dumps({"case": k, "ms": ms, "frames_per_s": B * M / (ms * 1e-3), "algorithmic_bytes_per_frame": bpf,
                      "algorithmic_GBps": B * M * bpf / (ms * 1e-3) / 1e9}), flush=True)
y = yp.numpy()
err = float(np.max(np.abs(y[:, N:-N] - x[:, N:out_len - N])) / np.max(np.abs(x)))
print(json.dumps({"round_trip_max_err": err, "chain_ms_full": sorted(res["stft (c64[M][1024] out)"])[2] + sorted(res["istft (c64 in, c64 out)"])[2],
                  "chain_ms_packed": sorted(res["stft_packed (c64[M][512] out)"])[2] + sorted(res["istft_packed (packed in, f32 out)"])[2]}))
