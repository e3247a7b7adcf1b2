"""Rates of the complex-sample stft (nxsig_stft_c64) per frame length (tools only).  Round 5: N = 1024 / 2048 / 4096 on the framed row kernels
0.25 / 0.12 / 0.15 -> 0.38 / 0.28 / 0.25 of the roofline (interior frames load without per-element bounds and mirror math); the two-pass A x B kernels (k_stft_rab_c64) for 100 ... 1600: 0.33-0.55 (two-step before: 0.05-0.17)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nx_signal_amd as S
from nx_signal_amd import _lib
ctx = S.Context(0); lib = _lib.load(); rng = np.random.default_rng(1)
def timeit(fn, reps=10, warm=5):
    for _ in range(warm): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop() / reps
for N, hop, K, rows, L in ((1024, 256, 1024, 16, 2880000), (512, 128, 512, 16, 2880000), (2048, 512, 2048, 8, 5760000), (400, 160, 512, 16, 2880000), (1000, 250, 1000, 16, 2880000), (256, 64, 256, 16, 2880000), (4096, 1024, 4096, 8, 5760000)):
    x1 = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
    xd = ctx.empty((rows, L), np.complex64)
    for r in range(rows): _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xd.ptr + r * L * 8), x1.ctypes.data_as(C.c_void_p), x1.nbytes))
    M = (L - N) // hop + 1
    zd = ctx.empty((rows, M, K), np.complex64)
    w = S.windows.hann(N)
    p = _lib.StftParams(N, hop, K, _lib.PAD_VALID, 0, 0, _lib.SCALE_NONE, 0, 48000.0)
    fn = lambda: _lib.check(lib.nxsig_stft_c64(ctx.handle, C.c_void_p(xd.ptr), L, rows, L, w.ctypes.data_as(C.c_void_p), C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))
    ms = timeit(fn)
    print(f"stft c64 N={N} hop={hop} K={K} {rows} x {L}: {ms:.3f} ms {rows*M*(hop*8+K*8)/(ms*1e-3)/8e12:.3f}", flush=True)
    del xd, zd
