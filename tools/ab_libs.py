"""Interleaved A/B of the SAME launch through TWO builds of libnxsig.so in one process (tools only): both libraries are loaded side by
side (separate contexts and streams on the same GPU), rounds alternate, so box-to-box and warm-up drift cancel and a code change is
separated from the box it happened to be measured on.
    usage: python tools/ab_libs.py tools/_ab/libnxsig_base.so nx_signal_amd/libnxsig.so [stft[N]|istft[N]|fir] [rounds]   (stft2048 = config 4's shard, fir = config 5's)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nx_signal_amd import _lib  # noqa: E402  (signature table + structs only)

paths = sys.argv[1:3]
which = sys.argv[3] if len(sys.argv) > 3 else "stft"
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6


def bind(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        try:
            f = getattr(lib, name)
        except AttributeError:
            continue  # an older build without this symbol
        f.restype, f.argtypes = res, args
    return lib


rng = np.random.Generator(np.random.PCG64(5))
SR = 48000


class Side:
    def __init__(self, path):
        self.lib = bind(path)
        self.ctx = C.c_void_p()
        assert self.lib.nxsig_ctx_create(0, C.byref(self.ctx)) == 0

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.lib.nxsig_alloc(self.ctx, nbytes, C.byref(p)) == 0, self.lib.nxsig_last_error()
        return p

    def upload_rows(self, dptr, rows, n, chunk):
        for r in range(rows):
            xr = np.roll(chunk, 977 * r)
            assert self.lib.nxsig_upload(self.ctx, C.c_void_p(dptr.value + r * n * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes) == 0

    def time(self, fn, reps=20, warm=5):
        for _ in range(warm):
            assert fn() == 0, self.lib.nxsig_last_error()
        self.lib.nxsig_sync(self.ctx)
        self.lib.nxsig_timer_start(self.ctx)
        for _ in range(reps):
            fn()
        ms = C.c_float()
        self.lib.nxsig_timer_stop(self.ctx, C.byref(ms))
        return ms.value / reps


def make(side):
    lib, ctx = side.lib, side.ctx
    if which.startswith("stft"):
        N = int(which[4:] or 1024)
        N, hop, B, L = (N, N // 4, 32, SR * 60) if N != 2048 else (2048, 512, 8, SR * 600)
        M = (L - N) // hop + 1
        x = side.alloc(B * L * 4); z = side.alloc(B * M * N * 8)
        side.upload_rows(x, B, L, rng.standard_normal(L, dtype=np.float32))
        w = np.hanning(N + 1)[:N].astype(np.float32)
        p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, float(SR))
        side.keep = (w, p)
        return (lambda: lib.nxsig_stft_f32(ctx, x, L, B, L, w.ctypes.data_as(C.c_void_p), C.byref(p), z, None, 1)), B * M * (hop * 4 + N * 8)
    if which.startswith("mel"):   # mel1024: N = K = 1024 hop 256, 128 bands, 32 x 60 s @48k; mel400 / mel512: N = 400 hop 160 K = 400 / 512 :reflect, 80 bands, 32 x 10 min @16k
        import nx_signal_amd as S
        K = int(which[3:] or 1024)
        N, hop, B, L, mb, sr, pad = (1024, 256, 32, SR * 60, 128, 48000.0, 0) if K == 1024 else (400, 160, 32, 16000 * 600, 80, 16000.0, 1)
        Lp = L + (N // 2) * 2 if pad else L
        M = (Lp - N) // hop + 1
        x = side.alloc(B * L * 4); o = side.alloc(B * M * mb * 4)
        side.upload_rows(x, B, L, rng.standard_normal(L, dtype=np.float32))
        w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32)
        filt = np.ascontiguousarray(S.mel_filters(K, mb, sr), dtype=np.float32)
        p = _lib.StftParams(N, hop, K, pad, 0, 0, 0, 0, sr)
        side.keep = (w, p, filt)
        return (lambda: lib.nxsig_stft_mel_f32(ctx, x, L, B, L, w.ctypes.data_as(C.c_void_p), C.byref(p), mb, filt.ctypes.data_as(C.c_void_p), o, None, 1)), B * M * (hop * 4 + mb * 4)
    if which.startswith("ishort"):   # ishort400: thousands of short rows (4096 rows x ~100 frames, hop = 0.4 N rounded to 16)
        N = int(which[6:] or 400)
        hop, B, M = max(16, int(0.4 * N) // 16 * 16), 4096, 98
        z = side.alloc(B * M * N * 8); y = side.alloc(B * (M * hop + N - hop) * 8)
        zr = (rng.standard_normal((M, N), dtype=np.float32) + 1j * rng.standard_normal((M, N), dtype=np.float32)).astype(np.complex64)
        for r in range(B):
            assert lib.nxsig_upload(ctx, C.c_void_p(z.value + r * M * N * 8), zr.ctypes.data_as(C.c_void_p), zr.nbytes) == 0
        w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32)
        p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, float(SR))
        side.keep = (w, p)
        return (lambda: lib.nxsig_istft_c64(ctx, z, M, B, w.ctypes.data_as(C.c_void_p), C.byref(p), y, 1)), B * M * (N * 8 + hop * 8)
    if which.startswith("istft"):
        N = int(which[5:] or 1024)
        hop, B, L = N // 4, 16, SR * 60
        M = (L - N) // hop + 1
        x = side.alloc(B * L * 4); z = side.alloc(B * M * N * 8); y = side.alloc(B * (M * hop + N - hop) * 8)
        side.upload_rows(x, B, L, rng.standard_normal(L, dtype=np.float32))
        w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32)
        p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, float(SR))
        assert lib.nxsig_stft_f32(ctx, x, L, B, L, w.ctypes.data_as(C.c_void_p), C.byref(p), z, None, 1) == 0
        side.keep = (w, p)
        return (lambda: lib.nxsig_istft_c64(ctx, z, M, B, w.ctypes.data_as(C.c_void_p), C.byref(p), y, 1)), B * M * (N * 8 + hop * 8)
    B, L = 8, SR * 600
    x = side.alloc(B * L * 4); y = side.alloc(B * L * 4)
    side.upload_rows(x, B, L, rng.standard_normal(L, dtype=np.float32))
    h = np.zeros(257, np.float32)
    assert lib.nxsig_firwin_f32(257, (C.c_double * 1)(4000.0), 1, 4, 0.0, 1, 1, float(SR), h.ctypes.data_as(C.c_void_p)) == 0
    side.keep = (h,)
    return (lambda: lib.nxsig_fir_f32(ctx, x, L, B, L, h.ctypes.data_as(C.c_void_p), 257, 1, y, 1)), B * L * 8


sides = [Side(p) for p in paths]
jobs = [make(s) for s in sides]
res = [[] for _ in sides]
for r in range(rounds):
    for i, (s, (fn, nbytes)) in enumerate(zip(sides, jobs)):
        res[i].append(nbytes / (s.time(fn) * 1e-3) / 1e9)
for pth, v in zip(paths, res):
    v2 = sorted(v[1:]) if len(v) > 2 else sorted(v)   # the first round of each side carries the warm-up transient
    print(json.dumps({"lib": pth, "case": which, "GBps_rounds": [round(a, 1) for a in v], "median_after_first": round(v2[len(v2) // 2], 1),
                      "frac_of_8TBps": round(v2[len(v2) // 2] / 8000, 4)}))
