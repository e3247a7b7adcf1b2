#!/usr/bin/env python
"""Secondary measurements for the other BASELINE configs (not the headline bench.py line): HIP-event timing of
stft N=2048 (config 4 shape, per-GPU shard), istft (config 3) and FIR (config 5) on device-resident data.
Prints one JSON object per line.  usage: python tools/bench_configs.py [stft2048] [istft] [fir] [stft1024]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib  # noqa: E402

PEAK = 8000.0


def timeit(ctx, fn, reps=30, warm=20):
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


def fill_normal(ctx, buf, shape, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    L = shape[-1]
    lib = _lib.load()
    chunk = rng.standard_normal(L, dtype=np.float32)
    for r in range(rows):  # same stream with a per-row roll: cheap to generate, still full-entropy data
        xr = np.roll(chunk, 977 * r)
        _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(buf.ptr + r * L * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes))
    return chunk


def stft_case(ctx, N, hop, L, batch, name, K=None, pad=None):
    lib = _lib.load()
    w = S.windows.hann(N)
    K = K or N
    pad = _lib.PAD_VALID if pad is None else pad
    Lp = L + (N // 2) * 2 if pad == _lib.PAD_REFLECT else L   # :reflect pads N/2 on both sides
    M = (Lp - N) // hop + 1
    xd = ctx.empty((batch, L), np.float32)
    fill_normal(ctx, xd, (batch, L), 7)
    zd = ctx.empty((batch, M, K), np.complex64)
    p = _lib.StftParams(N, hop, K, pad, 0, 0, _lib.SCALE_NONE, 0, 48000.0)
    wp = w.ctypes.data_as(C.c_void_p)
    fn = lambda: _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, batch, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))
    nalt = int(os.environ.get("NXSIG_BENCH_ALT_INPUTS", "1"))
    if nalt > 1:  # rotate over several input buffers so that no input stays resident in the Infinity Cache between launches
        xs = [xd]
        for i in range(nalt - 1):
            xi = ctx.empty((batch, L), np.float32)
            fill_normal(ctx, xi, (batch, L), 100 + i)
            xs.append(xi)
        state = {"i": 0}

        def fn():
            xi = xs[state["i"] % nalt]
            state["i"] += 1
            _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xi.ptr), L, batch, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))
    ms = timeit(ctx, fn)
    bpf = hop * 4 + K * 8
    gbs = batch * M * bpf / (ms * 1e-3) / 1e9
    print(json.dumps({"case": name, "N": N, "hop": hop, "batch": batch, "frames": batch * M, "ms": ms,
                      "frames_per_s": batch * M / (ms * 1e-3), "algorithmic_GBps": gbs, "frac_of_8TBps": gbs / PEAK}), flush=True)
    return xd, zd, w, M


def main():
    which = sys.argv[1:] or ["stft1024", "stft2048", "istft", "fir"]
    ctx = S.Context(0)
    lib = _lib.load()
    if "stft1024" in which:
        stft_case(ctx, 1024, 256, 2880000, 32, "stft N=1024 hop=256, 32 x 60 s (config 2 batched)")
    if "stft2048" in which:
        # config 4 per-GPU shard: 8 channels x 10 min @ 48 kHz would be 7.4 GB of output; use 8 ch x 150 s (1.8 GB out)
        stft_case(ctx, 2048, 512, 7200000, 8, "stft N=2048 hop=512, 8 ch x 150 s (config 4 shard, shortened)")
    if "stft2048full" in which:  # config 4, one GPU's FULL shard: 8 channels x 10 min -> 7.37 GB of spectrum
        stft_case(ctx, 2048, 512, 28800000, 8, "stft N=2048 hop=512, 8 ch x 600 s (config 4: one GPU's full shard)")
    if "reflect1024" in which:
        stft_case(ctx, 1024, 256, 2880000, 32, "stft N=1024 hop=256 window_padding :reflect, 32 x 60 s", pad=_lib.PAD_REFLECT)
    if "speech512" in which:
        stft_case(ctx, 400, 160, 16000 * 600, 32, "stft N=400 hop=160 fft_length=512 (25 ms / 10 ms speech framing), 32 x 10 min @16 kHz", K=512)
    if "speech512r" in which:
        stft_case(ctx, 400, 160, 16000 * 600, 32, "stft N=400 hop=160 fft_length=512 :reflect, 32 x 10 min @16 kHz", K=512, pad=_lib.PAD_REFLECT)
    for name in which:
        if name.startswith("case:"):  # case:N:hop:K[:rows[:samples]] — any stft shape, :valid
            f = [int(v) for v in name.split(":")[1:]]
            n, hp, k = f[0], f[1], f[2]
            rows = f[3] if len(f) > 3 else 32
            Lc = f[4] if len(f) > 4 else (1500 * 1024 * 1024 // (rows * 8 * k)) * hp + n
            stft_case(ctx, n, hp, Lc, rows, f"stft N={n} hop={hp} fft_length={k}, {rows} rows x {Lc} samples", K=k)
    for name in which:
        if name.startswith("gen"):  # e.g. gen512: generic-kernel sizes, ~1.7 GB of output each
            n = int(name[3:])
            b = 16
            Lg = (1700 * 1024 * 1024 // (b * 8 * 4)) // n * n  # hop = n/4 -> 8n bytes out per hop samples
            stft_case(ctx, n, n // 4, Lg, b, f"stft N={n} hop={n // 4}, {b} rows (tuned wave kernel if the size has one, else generic)")
    for name in which:
        if name.startswith("ist") and name != "istft":  # e.g. ist512: generic-path istft sizes
            n, _, hp = name[3:].partition("h")   # ist512 or ist512h160 (any hop)
            n = int(n); hop = int(hp) if hp else n // 4; b = 8
            Mi = (900 * 1024 * 1024 // (b * n * 8))
            w = S.windows.hann(n)
            rng = np.random.Generator(np.random.PCG64(5))
            zrow = (rng.standard_normal((64, n), dtype=np.float32) + 1j * rng.standard_normal((64, n), dtype=np.float32)).astype(np.complex64)
            zd = ctx.empty((b, Mi, n), np.complex64)
            for r in range(b):
                for blk in range(0, Mi, 64):
                    cnt = min(64, Mi - blk)
                    _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(zd.ptr + ((r * Mi + blk) * n) * 8), zrow.ctypes.data_as(C.c_void_p), cnt * n * 8))
            out_len = Mi * hop + n - hop
            yd = ctx.empty((b, out_len), np.complex64)
            p = _lib.StftParams(n, hop, n, 0, 0, 0, 0, 0, 48000.0)
            wp = w.ctypes.data_as(C.c_void_p)
            fn = lambda: _lib.check(lib.nxsig_istft_c64(ctx.handle, C.c_void_p(zd.ptr), Mi, b, wp, C.byref(p), C.c_void_p(yd.ptr), _lib.DEVICE))
            ms = timeit(ctx, fn, reps=10, warm=5)
            gbs = b * Mi * (n * 8 + hop * 8) / (ms * 1e-3) / 1e9
            print(json.dumps({"case": f"istft N={n} hop={hop}, {b} rows", "ms": ms, "frames_per_s": b * Mi / (ms * 1e-3), "algorithmic_GBps": gbs,
                              "frac_of_8TBps": gbs / PEAK}), flush=True)
    if "mel" in which:
        N, hop, L, batch, mb = 1024, 256, 2880000, 32, 128
        w = S.windows.hann(N)
        M = (L - N) // hop + 1
        xd = ctx.empty((batch, L), np.float32)
        fill_normal(ctx, xd, (batch, L), 13)
        filt = S.mel_filters(N, mb, 48000.0)
        od = ctx.empty((batch, M, mb), np.float32)
        zd = ctx.empty((batch, M, N), np.complex64)
        p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
        wp, fp = w.ctypes.data_as(C.c_void_p), filt.ctypes.data_as(C.c_void_p)
        fused = lambda: _lib.check(lib.nxsig_stft_mel_f32(ctx.handle, C.c_void_p(xd.ptr), L, batch, L, wp, C.byref(p), mb, fp, C.c_void_p(od.ptr), None, _lib.DEVICE))

        def two_step():
            _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, batch, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))
            _lib.check(lib.nxsig_stft_to_mel(ctx.handle, C.c_void_p(zd.ptr), batch * M, N, mb, fp, C.c_void_p(od.ptr), _lib.DEVICE))

        for name, fn in (("log-mel fused (stft+mel in one kernel)", fused), ("log-mel two-step (stft, then stft_to_mel)", two_step)):
            ms = timeit(ctx, fn)
            print(json.dumps({"case": f"{name}, N=1024 hop=256, {mb} mel bins, 32 x 60 s", "ms": ms, "frames_per_s": batch * M / (ms * 1e-3),
                              "bytes_per_frame_fused": hop * 4 + mb * 4}), flush=True)
    if "mag" in which:
        N, hop, L, batch = 1024, 256, 2880000, 32
        w = S.windows.hann(N)
        M = (L - N) // hop + 1
        xd = ctx.empty((batch, L), np.float32)
        fill_normal(ctx, xd, (batch, L), 13)
        od = ctx.empty((batch, M, N // 2), np.float32)
        p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
        wp = w.ctypes.data_as(C.c_void_p)
        for kind, kname in ((0, "|s|"), (2, "dBFS")):
            fn = lambda: _lib.check(lib.nxsig_stft_magnitude_f32(ctx.handle, C.c_void_p(xd.ptr), L, batch, L, wp, C.byref(p), kind, C.c_void_p(od.ptr), None, _lib.DEVICE))
            ms = timeit(ctx, fn)
            bpf = hop * 4 + (N // 2) * 4 * (3 if kind == 2 else 1)
            print(json.dumps({"case": f"magnitude spectrogram {kname} fused with the stft, N=1024 hop=256, 32 x 60 s", "ms": ms,
                              "frames_per_s": batch * M / (ms * 1e-3), "bytes_per_frame": bpf,
                              "algorithmic_GBps": batch * M * bpf / (ms * 1e-3) / 1e9}), flush=True)
    if "onesided" in which:
        N, hop, L, batch = 1024, 256, 2880000, 32
        w = S.windows.hann(N)
        M = (L - N) // hop + 1
        xd = ctx.empty((batch, L), np.float32)
        fill_normal(ctx, xd, (batch, L), 13)
        od = ctx.empty((batch, M, N // 2), np.complex64)
        p = _lib.StftParams(N, hop, N, 0, 0, 0, 0, 0, 48000.0)
        wp = w.ctypes.data_as(C.c_void_p)
        fn = lambda: _lib.check(lib.nxsig_stft_onesided_f32(ctx.handle, C.c_void_p(xd.ptr), L, batch, L, wp, C.byref(p), C.c_void_p(od.ptr), None, _lib.DEVICE))
        ms = timeit(ctx, fn)
        bpf = hop * 4 + (N // 2) * 8
        print(json.dumps({"case": "one-sided complex spectrum (bins 0..K/2-1) fused with the stft, N=1024 hop=256, 32 x 60 s", "ms": ms,
                          "frames_per_s": batch * M / (ms * 1e-3), "bytes_per_frame": bpf,
                          "algorithmic_GBps": batch * M * bpf / (ms * 1e-3) / 1e9}), flush=True)
    for msname in ("melspeech", "melspeech400"):
      if msname in which:
          # ASR front-end: 25 ms frames / 10 ms hop at 16 kHz, 512-point (or n_fft = 400) FFT, centred (:reflect), 80 mel bands
          N, hop, K, L, batch, mb = 400, 160, (512 if msname == "melspeech" else 400), 16000 * 600, 32, 80
          w = S.windows.hann(N)
          M = (L + 2 * (N // 2) - N) // hop + 1
          xd = ctx.empty((batch, L), np.float32)
          fill_normal(ctx, xd, (batch, L), 13)
          filt = S.mel_filters(K, mb, 16000.0)
          od = ctx.empty((batch, M, mb), np.float32)
          zd = ctx.empty((batch, M, K), np.complex64)
          p = _lib.StftParams(N, hop, K, _lib.PAD_REFLECT, 0, 0, 0, 0, 16000.0)
          wp, fp = w.ctypes.data_as(C.c_void_p), filt.ctypes.data_as(C.c_void_p)
          fused = lambda: _lib.check(lib.nxsig_stft_mel_f32(ctx.handle, C.c_void_p(xd.ptr), L, batch, L, wp, C.byref(p), mb, fp, C.c_void_p(od.ptr), None, _lib.DEVICE))

          def two_step2():
              _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, batch, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))
              _lib.check(lib.nxsig_stft_to_mel(ctx.handle, C.c_void_p(zd.ptr), batch * M, K, mb, fp, C.c_void_p(od.ptr), _lib.DEVICE))

          for name, fn in (("log-mel fused", fused), ("log-mel two-step", two_step2)):
              ms = timeit(ctx, fn)
              print(json.dumps({"case": f"{name}, N=400 hop=160 fft_length={K} :reflect, {mb} mel bins, 32 x 10 min @16 kHz", "ms": ms,
                                "frames_per_s": batch * M / (ms * 1e-3), "audio_seconds_per_s": batch * 600 / (ms * 1e-3),
                                "bytes_per_frame_fused": hop * 4 + mb * 4}), flush=True)
    if "istft" in which:
        N, hop, L, batch = 1024, 256, 2880000, 16
        w = S.windows.hann(N)
        M = (L - N) // hop + 1
        xd = ctx.empty((batch, L), np.float32)
        fill_normal(ctx, xd, (batch, L), 9)
        zd, _, _ = S.stft(xd, w, overlap_length=N - hop, fft_length=N, sampling_rate=48000)
        out_len = M * hop + N - hop
        yd = ctx.empty((batch, out_len), np.complex64)
        p = _lib.StftParams(N, hop, N, 0, 0, 0, _lib.SCALE_NONE, 0, 48000.0)
        wp = w.ctypes.data_as(C.c_void_p)
        fn = lambda: _lib.check(lib.nxsig_istft_c64(ctx.handle, C.c_void_p(zd.ptr), M, batch, wp, C.byref(p), C.c_void_p(yd.ptr), _lib.DEVICE))
        ms = timeit(ctx, fn, reps=10)
        bpf = N * 8 + hop * 8
        gbs = batch * M * bpf / (ms * 1e-3) / 1e9
        print(json.dumps({"case": "istft N=1024 hop=256, 16 x 60 s (config 3 batched)", "ms": ms, "frames_per_s": batch * M / (ms * 1e-3),
                          "algorithmic_GBps": gbs, "frac_of_8TBps": gbs / PEAK}), flush=True)
    if "inv2048" in which:
        # the inverse of BASELINE config 4's shard: 8 channels x 600 s, N = 2048 hop = 512 (7.4 GB of spectrum in, 1.8 GB out) and the
        # round trip stft -> istft on interior samples (size-independent property)
        N, hop, L, batch = 2048, 512, 28800000, 8
        w = S.windows.hann(N)
        M = (L - N) // hop + 1
        xd = ctx.empty((batch, L), np.float32)
        fill_normal(ctx, xd, (batch, L), 11)
        zd, _, _ = S.stft(xd, w, overlap_length=N - hop, fft_length=N, sampling_rate=48000)
        out_len = M * hop + N - hop
        yd = ctx.empty((batch, out_len), np.complex64)
        p = _lib.StftParams(N, hop, N, 0, 0, 0, _lib.SCALE_NONE, 0, 48000.0)
        wp = w.ctypes.data_as(C.c_void_p)
        fn = lambda: _lib.check(lib.nxsig_istft_c64(ctx.handle, C.c_void_p(zd.ptr), M, batch, wp, C.byref(p), C.c_void_p(yd.ptr), _lib.DEVICE))
        ms = timeit(ctx, fn, reps=10, warm=10)
        bpf = N * 8 + hop * 8
        gbs = batch * M * bpf / (ms * 1e-3) / 1e9
        chk = np.empty(8192, np.complex64)
        ref = np.empty(8192, np.float32)
        _lib.check(lib.nxsig_download(ctx.handle, chk.ctypes.data_as(C.c_void_p), C.c_void_p(yd.ptr + 8 * (3 * out_len + 1234567)), chk.nbytes))
        _lib.check(lib.nxsig_download(ctx.handle, ref.ctypes.data_as(C.c_void_p), C.c_void_p(xd.ptr + 4 * (3 * L + 1234567)), ref.nbytes))
        print(json.dumps({"case": "istft N=2048 hop=512, 8 ch x 600 s (the inverse of config 4's shard)", "ms": ms, "frames_per_s": batch * M / (ms * 1e-3),
                          "algorithmic_GBps": gbs, "frac_of_8TBps": gbs / PEAK,
                          "roundtrip_max_err": float(np.max(np.abs(chk.real - ref)) / np.max(np.abs(ref)))}), flush=True)
    for name in which:
        if name.startswith("fir") and name != "fir":  # e.g. fir1025: other filter lengths on the config 5 shard
            taps = int(name[3:])
            L, batch = 28800000, 8
            h = S.filters.firwin(taps if taps % 2 else taps + 1, [4000], sampling_rate=48000)[:taps]
            h = np.ascontiguousarray(h)
            xd = ctx.empty((batch, L), np.float32)
            fill_normal(ctx, xd, (batch, L), 11)
            yd = ctx.empty((batch, L), np.float32)
            hp = h.ctypes.data_as(C.c_void_p)
            fn = lambda: _lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(xd.ptr), L, batch, L, hp, taps, _lib.CONV_SAME, C.c_void_p(yd.ptr), _lib.DEVICE))
            ms = timeit(ctx, fn, reps=10)
            gbs = batch * L * 8 / (ms * 1e-3) / 1e9
            print(json.dumps({"case": f"fir {taps} taps :same, 8 ch x 10 min @48k", "ms": ms, "samples_per_s": batch * L / (ms * 1e-3),
                              "algorithmic_GBps": gbs, "frac_of_8TBps": gbs / PEAK}), flush=True)
    if "fir" in which:
        L, batch = 28800000, 8  # config 5 per-GPU shard: 8 channels x 10 min
        h = S.filters.firwin(257, [4000], sampling_rate=48000)
        xd = ctx.empty((batch, L), np.float32)
        fill_normal(ctx, xd, (batch, L), 11)
        yd = ctx.empty((batch, L), np.float32)
        hp = h.ctypes.data_as(C.c_void_p)
        fn = lambda: _lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(xd.ptr), L, batch, L, hp, 257, _lib.CONV_SAME, C.c_void_p(yd.ptr), _lib.DEVICE))
        ms = timeit(ctx, fn, reps=10)
        gbs = batch * L * 8 / (ms * 1e-3) / 1e9
        print(json.dumps({"case": "fir 257 taps :same, 8 ch x 10 min @48k (config 5 shard)", "ms": ms, "samples_per_s": batch * L / (ms * 1e-3),
                          "algorithmic_GBps": gbs, "frac_of_8TBps": gbs / PEAK}), flush=True)


if __name__ == "__main__":
    main()
