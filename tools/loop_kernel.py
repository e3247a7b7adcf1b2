"""Keeps one kernel (fir | stft) running back to back for N seconds so that rocm-smi can be sampled beside it (tools only;
profiles/r02/power_clocks.txt).  usage: python tools/loop_kernel.py fir 12 & sleep 5; rocm-smi --showclocks --showpower"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import nx_signal_amd as S
from nx_signal_amd import _lib
which = sys.argv[1]
ctx = S.Context(0); lib = _lib.load()
rng = np.random.Generator(np.random.PCG64(1))
if which == "fir":
    L, B = 28800000, 8
    x = ctx.to_device(rng.standard_normal((B, L), dtype=np.float32)); y = ctx.empty((B, L), np.float32)
    h = S.filters.firwin(257, [4000.0], sampling_rate=48000); hp = h.ctypes.data_as(C.c_void_p)
    fn = lambda: lib.nxsig_fir_f32(ctx.handle, C.c_void_p(x.ptr), L, B, L, hp, 257, 1, C.c_void_p(y.ptr), 1)
else:
    L, B, N, hop = 2880000, 32, 1024, 256
    x = ctx.to_device(rng.standard_normal((B, L), dtype=np.float32)); M = (L - N) // hop + 1
    z = ctx.empty((B, M, N), np.complex64); w = S.windows.hann(N); p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
    fn = lambda: lib.nxsig_stft_f32(ctx.handle, C.c_void_p(x.ptr), L, B, L, w.ctypes.data_as(C.c_void_p), C.byref(p), C.c_void_p(z.ptr), None, 1)
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[2]):
    for _ in range(200): fn()
    ctx.sync(); n += 200
print(which, "launches/s", n / (time.time() - t0))
