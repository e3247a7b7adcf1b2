"""Launch-bound case (BASELINE config 2 as written: ONE 60 s stream per call): back-to-back API calls versus the same
calls captured once into a HIP graph (torch.cuda.CUDAGraph on a stream handed to the library) and replayed.
usage: python tools/bench_graph.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402


def main():
    N, hop, L, calls = 1024, 256, 2880000, 32
    lib = _lib.load()
    ctx = S.Context(0)
    x = np.random.default_rng(0).standard_normal(L).astype(np.float32)
    xd = ctx.to_device(x)
    w = S.windows.hann(N)
    M = (L - N) // hop + 1
    zd = ctx.empty((M, N), np.complex64)
    p = _lib.StftParams(N, hop, N, _lib.PAD_VALID, 0, 0, _lib.SCALE_NONE, 0, 48000.0)
    wp = w.ctypes.data_as(C.c_void_p)
    call = lambda: _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, 1, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))
    side = torch.cuda.Stream()
    ctx.set_stream(side.cuda_stream)
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        for _ in range(3 * calls):
            call()
        e0.record()
        for _ in range(10 * calls):
            call()
        e1.record()
    torch.cuda.synchronize()
    eager_us = e0.elapsed_time(e1) * 1e3 / (10 * calls)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(calls):
            call()
    with torch.cuda.stream(side):
        for _ in range(3):
            g.replay()
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
    torch.cuda.synchronize()
    graph_us = e0.elapsed_time(e1) * 1e3 / (10 * calls)
    ctx.set_stream(None)
    print(json.dumps({"case": "stft N=1024 hop=256, one 60 s stream per call (11 247 frames, 103.7 MB)", "eager_us_per_call": eager_us,
                      "graph_us_per_call": graph_us, "eager_frames_per_s": M / (eager_us * 1e-6), "graph_frames_per_s": M / (graph_us * 1e-6)}))


if __name__ == "__main__":
    main()
