"""Randomised differential run against the oracle (tools only; the committed tests are tests/test_gpu_*.py): random frame lengths, hops,
paddings, row counts, row lengths and tap counts through stft (real / complex), the fused sinks, istft and fir.
usage: python tools/random_parity.py [seconds=120] [seed=0]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nx_signal_amd as S
from oracle import nx_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
LENGTHS = [32, 192, 288, 576, 1152, 1440, 1536, 1920, 64, 100, 120, 128, 160, 200, 240, 256, 300, 320, 360, 384, 400, 480, 500, 512, 600, 640, 720, 768, 800, 900, 960, 1000, 1024, 1200, 1280, 1600, 2048, 333, 441, 48,
           882, 1764, 2205, 2400, 2880, 3840, 443, 4096]   # round 6: the radix-7 / 50- / 60- / 64-point lengths (441 odd), one Bluestein length, the 4096 front end
ctx = S.Context(0)
t0 = time.time(); n = 0; worst = {}
def note(kind, err, what):
    if err > worst.get(kind, (0, None))[0]: worst[kind] = (err, what)
    assert err < 1e-5 or (kind == "mel" and err < 2e-4), (kind, err, what)
while time.time() - t0 < budget:
    K = int(rng.choice(LENGTHS))
    N = K if rng.random() < 0.6 else int(rng.integers(max(2, K // 3), K + K // 2))
    hop = int(rng.integers(1, N + 1)) if rng.random() < 0.5 else max(1, N // int(rng.choice([2, 3, 4, 8])))
    rows = int(rng.choice([1, 2, 3, 5, 33, 130]))
    L = int(rng.integers(N, N + int(rng.choice([40, 40, 400, 3000])) * hop + 7))   # (long rows: launches of more than one round)
    if rows * L > 6_000_000: L = max(N, 6_000_000 // rows)
    pad = str(rng.choice(["valid", "reflect"])) if L > N else "valid"
    scaling = rng.choice([None, "spectrum", "psd"])
    w = S.windows.hann(N) if rng.random() < 0.5 else S.windows.hamming(N)
    opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=16000)
    what = (K, N, hop, rows, L, pad, scaling)
    kind = rng.choice(["stft", "c64", "mel", "mag", "istft", "istft", "fir", "c128"])
    if kind == "stft":
        x = rng.standard_normal((rows, L)).astype(np.float32)
        if rng.random() < 0.2: x[rng.integers(rows), rng.integers(L)] = np.nan
        z = S.stft(ctx.to_device(x), w, ctx=ctx, **opts)[0].numpy(); zo = O.stft(x, w, **opts)[0]
        ok = np.isfinite(zo)
        assert np.array_equal(np.isfinite(z), ok), ("nan pattern", what)
        if ok.any(): note("stft", float(np.max(np.abs(z[ok] - zo[ok])) / max(1e-30, np.max(np.abs(zo[ok])))), what)
    elif kind == "c64":
        x = (rng.standard_normal((rows, L)) + 1j * rng.standard_normal((rows, L))).astype(np.complex64)
        z = S.stft(ctx.to_device(x), w, ctx=ctx, **opts)[0].numpy(); zo = O.stft(x, w, **opts)[0]
        note("c64", float(np.max(np.abs(z - zo)) / np.max(np.abs(zo))), what)
    elif kind == "c128":   # c128 samples on the f64 tier (round 6): tolerance 1e-11 of the largest magnitude
        if rows * L > 400_000: L = max(N, 400_000 // rows)
        x = rng.standard_normal((rows, L)) + 1j * rng.standard_normal((rows, L))
        w64 = w.astype(np.float64)
        z = S.stft(x, w64, **opts)[0]; zo = O.stft_f64(x, w64, **opts)[0]
        e = float(np.max(np.abs(z - zo)) / np.max(np.abs(zo)))
        assert e < 1e-11, ("c128", e, what)
        note("c128", e, what)
    elif kind == "mel" and K >= 64 and K % 2 == 0:
        x = rng.standard_normal((rows, L)).astype(np.float32)
        o2 = dict(opts); o2["scaling"] = None
        mb = int(rng.choice([20, 40, 80]))
        mel = S.mel_spectrogram(ctx.to_device(x), w, mel_bins=mb, ctx=ctx, **o2).numpy()
        zo = np.stack([O.stft(r, w, **o2)[0] for r in x])
        melo = O.stft_to_mel(zo.reshape(-1, K), 16000, K, mb).reshape(mel.shape)
        note("mel", float(np.max(np.abs(mel - melo))), what)
    elif kind == "mag" and K % 2 == 0:
        x = rng.standard_normal((rows, L)).astype(np.float32)
        g = S.spectrogram(ctx.to_device(x), w, kind="magnitude", ctx=ctx, **opts)[0].numpy()
        zo = O.stft(x, w, **opts)[0]
        note("mag", float(np.max(np.abs(g - np.abs(zo[..., : K // 2]))) / np.max(np.abs(zo))), what)
    elif kind == "istft":
        M = int(rng.integers(1, 60))
        z = (rng.standard_normal((rows, M, K)) + 1j * rng.standard_normal((rows, M, K))).astype(np.complex64)
        wk = S.windows.hann(K)
        o3 = dict(overlap_length=K - min(hop, K), fft_length=K, scaling=scaling, sampling_rate=16000)
        y = S.istft(ctx.to_device(z), wk, ctx=ctx, **o3).numpy(); yo = O.istft(z, wk, **o3)
        note("istft", float(np.max(np.abs(y - yo)) / np.max(np.abs(yo))), (K, o3["overlap_length"], rows, M, scaling))
    elif kind == "fir":
        taps = int(rng.choice([1, 2, 7, 33, 64, 101, 200, 257, 300, 400, 512, 513, 600, 769, 1000, 1025, 1026, 1500, 2049, 3000, 3073, 4097, 4100, 5000, 9001, 16385, 20000]))
        Lf = int(rng.integers(max(2, taps // 4), int(rng.choice([30000, 30000, 200000]))))
        if rows * Lf > 3_000_000: rows = max(1, 3_000_000 // Lf)
        mode = str(rng.choice(["same", "full", "valid"]))
        x = rng.standard_normal((rows, Lf)).astype(np.float32)
        h = (rng.standard_normal(taps) / taps ** 0.5).astype(np.float32)
        y = S.filters.fir(ctx.to_device(x), h, mode=mode, ctx=ctx).numpy()
        nout = {"full": Lf + taps - 1, "same": Lf, "valid": abs(Lf - taps) + 1}[mode]   # the reference's apply_mode (convolution.ex:300-347)
        full = np.stack([np.convolve(r.astype(np.float64), h.astype(np.float64), mode="full") for r in x])
        st = (full.shape[1] - nout) // 2 if mode != "full" else 0
        ref = full[:, st:st + nout]
        assert y.shape == ref.shape, (taps, Lf, mode, y.shape, ref.shape)
        note("fir", float(np.max(np.abs(y - ref)) / np.max(np.abs(ref))), (taps, Lf, rows, mode))
    else:
        continue
    n += 1
print("cases", n, "worst", {k: (float(f"{v[0]:.2e}"), v[1]) for k, v in worst.items()})
