"""Per-call host latency of the device-resident entry points on a small input (1 s mono 48 kHz, BASELINE config 1):
what a caller pays per call once the tables exist.  usage: python tools/bench_latency.py"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    ctx = S.Context(0)
    N, hop, L = 1024, 256, 48000
    x = np.random.default_rng(0).standard_normal(L).astype(np.float32)
    w = S.windows.hann(N)
    h = S.filters.firwin(257, [4000], sampling_rate=48000)
    xd = ctx.to_device(x)
    M = (L - N) // hop + 1
    zd = ctx.empty((M, N), np.complex64)
    yd = ctx.empty((M * hop + N - hop,), np.complex64)
    fd = ctx.empty((L,), np.float32)
    p = _lib.StftParams(N, hop, N, _lib.PAD_VALID, 0, 0, _lib.SCALE_NONE, 0, 48000.0)
    wp, hp = w.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p)
    calls = {
        "stft": lambda: lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, 1, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE),
        "istft": lambda: lib.nxsig_istft_c64(ctx.handle, C.c_void_p(zd.ptr), M, 1, wp, C.byref(p), C.c_void_p(yd.ptr), _lib.DEVICE),
        "fir": lambda: lib.nxsig_fir_f32(ctx.handle, C.c_void_p(xd.ptr), L, 1, L, hp, 257, _lib.CONV_SAME, C.c_void_p(fd.ptr), _lib.DEVICE),
    }
    out = {}
    for name, fn in calls.items():
        for _ in range(20):
            _lib.check(fn())
        ctx.sync()
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t_issue = (time.perf_counter() - t0) / n
        ctx.sync()
        t_total = (time.perf_counter() - t0) / n
        out[name] = {"host_issue_us": t_issue * 1e6, "sustained_us_per_call": t_total * 1e6}
    print(json.dumps({"case": "1 s mono 48 kHz, N=1024 hop=256 (184 frames), device-resident, via ctypes", **out}))
    # the same through the Python mirror (option parsing, result allocation from the context's caching allocator, times / frequencies)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    mirror = {
        "stft(DeviceBuffer)": lambda: S.stft(xd, w, ctx=ctx, **opts),
        "istft(DeviceBuffer)": lambda: S.istft(zd, w, ctx=ctx, **opts),
        "fir(DeviceBuffer)": lambda: S.filters.fir(xd, h, mode="same"),
        "stft(numpy) incl. PCIe": lambda: S.stft(x, w, ctx=ctx, **opts),
    }
    out2 = {}
    for name, fn in mirror.items():
        for _ in range(20):
            fn()
        ctx.sync()
        n = 500
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        ctx.sync()
        out2[name] = {"us_per_call": (time.perf_counter() - t0) / n * 1e6}
    print(json.dumps({"case": "the same calls through the Python mirror (result buffers allocated per call)", **out2}))


if __name__ == "__main__":
    main()
