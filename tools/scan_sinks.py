"""Fused sinks (log-mel, magnitude, one-sided) on thousands of short rows, :reflect and :valid, aligned and odd row lengths (tools only)."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nx_signal_amd as S
from nx_signal_amd import _lib

ctx = S.Context(0); lib = _lib.load(); rng = np.random.default_rng(2)
def timeit(fn, reps=10, warm=5):
    for _ in range(warm): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop() / reps
for N, hop, K, mb, sr in ((400, 160, 400, 80, 16000.0), (400, 160, 512, 80, 16000.0), (1024, 256, 1024, 128, 48000.0), (512, 128, 512, 64, 16000.0)):
    for rows, L in ((4096, 16000), (4096, 16001), (32, 2048000)):
        for pad in (_lib.PAD_REFLECT, _lib.PAD_VALID):
            Lp = L + (N // 2) * 2 if pad == _lib.PAD_REFLECT else L
            M = (Lp - N) // hop + 1
            x1 = rng.standard_normal(L).astype(np.float32)
            xd = ctx.empty((rows, L), np.float32)
            for r in range(rows): _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xd.ptr + r * L * 4), x1.ctypes.data_as(C.c_void_p), x1.nbytes))
            w = S.windows.hann(N)
            filt = np.ascontiguousarray(S.mel_filters(K, mb, sr), dtype=np.float32)
            od = ctx.empty((rows, M, mb), np.float32)
            gd = ctx.empty((rows, M, K // 2), np.float32)
            p = _lib.StftParams(N, hop, K, pad, 0, 0, _lib.SCALE_NONE, 0, sr)
            fm = lambda: _lib.check(lib.nxsig_stft_mel_f32(ctx.handle, C.c_void_p(xd.ptr), L, rows, L, w.ctypes.data_as(C.c_void_p), C.byref(p), mb, filt.ctypes.data_as(C.c_void_p), C.c_void_p(od.ptr), None, _lib.DEVICE))
            fg = lambda: _lib.check(lib.nxsig_stft_magnitude_f32(ctx.handle, C.c_void_p(xd.ptr), L, rows, L, w.ctypes.data_as(C.c_void_p), C.byref(p), 0, C.c_void_p(gd.ptr), None, _lib.DEVICE))
            tm, tg = timeit(fm), timeit(fg)
            print(json.dumps({"case": f"N={N} hop={hop} K={K} {'reflect' if pad else 'valid'} {rows} x {L}", "mel_Mframes_s": round(rows * M / tm / 1e3, 1), "mag_Mframes_s": round(rows * M / tg / 1e3, 1)}), flush=True)
            del xd, od, gd
