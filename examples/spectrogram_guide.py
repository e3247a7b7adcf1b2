"""The reference's guides/spectrogram.livemd on the MI355X path (cited lines are the guide's).

  generate 3 s of sines at 44.1 kHz (:14-37)  ->  Hann window of `fs * window_duration` samples, stft with the default 50 %
  overlap and fft_length 1024 (:70-78)  ->  Nx.abs, dBFS = 20 log10(|s| / max |s|) (:84-86), bins below fs / 2 (:80-82)

The fused sink `spectrogram(kind: "dbfs")` computes the same tensor without writing the complex spectrum to HBM; both are
checked against the CPU oracle's restatement.  Then the production-size figure: 32 streams x 60 s @ 48 kHz.
Run on the GPU box: python examples/spectrogram_guide.py"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from oracle import nx_oracle as O  # noqa: E402  (checker only)


def main():
    fs, t_max = 44.1e3, 3
    full_n = math.ceil(fs * t_max)
    n = np.arange(full_n // 2, dtype=np.float32)
    sin = lambda freq: np.sin(np.float32(2 * math.pi * freq / fs) * n).astype(np.float32)  # noqa: E731
    data = np.concatenate([sin(440) + sin(1000), sin(220) + sin(3000)]).astype(np.float32)
    for window_duration in (20e-3, 10e-3):   # the guide plots several durations; these fit fft_length 1024
        n_window = math.ceil(fs * window_duration)
        window = S.windows.hann(n_window, is_periodic=True)
        opts = dict(sampling_rate=fs, fft_length=1024)
        s, t, f = S.stft(data, window, **opts)                       # the guide's call (:78)
        max_f = int(np.argmin(np.where(f >= fs / 2, np.arange(f.size), f.size + 1)))
        mag = np.abs(s)
        dbfs = (20 * np.log(mag / mag.max()) / np.log(10)).astype(np.float32)[:, :max_f]
        so, to, fo = O.stft(data, window, **opts)
        mo = np.abs(so.astype(np.complex128))
        ref = (20 * np.log(mo / mo.max()) / np.log(10))[:, :max_f]
        loud = mo[:, :max_f] > 1e-3 * mo.max()                        # dB of near-silent bins amplifies round-off of ~0
        fused, tf, ff = S.spectrogram(data, window, kind="dbfs", **opts)
        print(f"window {n_window} samples: stft {s.shape}, dBFS err {np.max(np.abs(dbfs - ref)[loud]):.2e} dB (two-step), "
              f"{np.max(np.abs(fused[:, :max_f] - ref)[loud]):.2e} dB (fused sink); peak bins at "
              f"{sorted(set(np.round(f[np.argsort(mag[5, :max_f])[-2:]]).astype(int).tolist()))} Hz")
    ctx = S.default_context()
    rng = np.random.default_rng(0)
    B, L, N, hop = 32, 48000 * 60, 1024, 256
    xd = ctx.to_device(rng.standard_normal((B, L), dtype=np.float32))
    w = S.windows.hann(N)
    o2 = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    for name, fn in (("stft (complex spectrum, 8 KB/frame)", lambda: S.stft(xd, w, **o2)[0]),
                     ("spectrogram kind=magnitude (fused)", lambda: S.spectrogram(xd, w, kind="magnitude", **o2)[0]),
                     ("spectrogram kind=dbfs (fused)", lambda: S.spectrogram(xd, w, kind="dbfs", **o2)[0]),
                     ("mel_spectrogram 128 bands (fused)", lambda: S.mel_spectrogram(xd, w, mel_bins=128, **o2))):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            y = fn()
        ctx.sync()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name:38s} {B} x 60 s @ 48 kHz: {dt * 1e3:7.2f} ms per pass ({B * 60 / dt:,.0f} s of audio per second; includes the result allocation)")
    del y


if __name__ == "__main__":
    main()
