"""The host-tensor path (NXSIG_HOST) of nxsig_stft_f32 on 8 x config 2: pinned-slot pipeline (round 6) against the direct pageable copies
with pre-faulting (round 5), result buffer reused (pages resident) or freshly allocated per call.  usage: python tools/bench_host.py [threads ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

N, hop, L, B = 1024, 256, 2880000, int(os.environ.get("HOST_B", 8))
M = (L - N) // hop + 1
ctx = S.Context(0)
lib = _lib.load()
w = S.windows.hann(N)
x = np.random.default_rng(1).standard_normal((B, L)).astype(np.float32)
p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
wp = w.ctypes.data_as(C.c_void_p)


def call(z):
    _lib.check(lib.nxsig_stft_f32(ctx.handle, x.ctypes.data_as(C.c_void_p), L, B, L, wp, C.byref(p), z.ctypes.data_as(C.c_void_p), None, _lib.HOST))


zr = np.empty((B, M, N), np.complex64)
for knob in [int(a) for a in sys.argv[1:]] or [0, 1, 4, 16]:
    ctx.set_tuning("HOST_PIPE", knob)
    call(zr); call(zr)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); call(zr); t.append(time.perf_counter() - t0)
    tf = []
    for _ in range(3):
        zf = np.empty((B, M, N), np.complex64)
        t0 = time.perf_counter(); call(zf); tf.append(time.perf_counter() - t0)
        del zf
    print(f"HOST_PIPE={knob:2d}: resident result buffer {min(t) * 1e3:7.2f} ms = {B * M / min(t) / 1e6:5.2f} M frames/s ({zr.nbytes / min(t) / 1e9:5.1f} GB/s)   "
          f"fresh buffer per call {min(tf) * 1e3:7.2f} ms = {B * M / min(tf) / 1e6:5.2f} M frames/s", flush=True)
