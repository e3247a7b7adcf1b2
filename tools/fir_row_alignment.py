"""FIR on rows whose length is / is not a multiple of 4 and 32 samples (tools only): 33 / 101 / 257 / 1025 taps, 2048 one-second rows and
8 long rows.  Round 5: rows of odd length ran at 0.26-0.28 of the roofline against 0.48-0.50 (one grid phase for all rows), 0.48-0.54 with
the per-row phase."""
import sys, os, json, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nx_signal_amd as S
from nx_signal_amd import _lib
ctx = S.Context(0); lib = _lib.load(); rng = np.random.default_rng(1)
def timeit(fn, reps=10, warm=5):
    for _ in range(warm): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop() / reps
for taps in (257, 101, 1025, 33):
    for rows, L in ((2048, 48000), (2048, 48001), (2048, 48002), (8, 12000001)):
        h = S.filters.firwin(taps, [0.2])
        x1 = rng.standard_normal(L).astype(np.float32)
        xd = ctx.empty((rows, L), np.float32)
        for r in range(rows): _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xd.ptr + r * L * 4), x1.ctypes.data_as(C.c_void_p), x1.nbytes))
        yd = ctx.empty((rows, L), np.float32)
        fn = lambda: _lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(xd.ptr), L, rows, L, h.ctypes.data_as(C.c_void_p), taps, _lib.CONV_SAME, C.c_void_p(yd.ptr), _lib.DEVICE))
        ms = timeit(fn)
        print(f"fir {taps} taps, {rows} rows x {L}: {rows*L*8/(ms*1e-3)/8e12:.3f}", flush=True)
        del xd, yd
