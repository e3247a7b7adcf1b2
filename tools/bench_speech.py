#!/usr/bin/env python
"""The speech front-end shapes (25 ms frames, 10 ms hop at 16 kHz, :reflect padding, 80 mel bands) on device-resident data:
fft_length 400 (the non-power-of-two transform some featurizers ask for) beside the default fft_length 512, each as the
plain stft, as stft -> stft_to_mel, and as the fused log-mel call.  One JSON object per line.
usage: python tools/bench_speech.py [batch] [seconds]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402

from bench_configs import fill_normal, timeit  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    secs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    fs, N, hop, mel = 16000, 400, 160, 80
    L = fs * secs
    ctx = S.Context(0)
    xd = ctx.empty((batch, L), np.float32)
    fill_normal(ctx, xd, (batch, L), 5)
    w = S.windows.hann(N)
    for K in (400, 512):
        opts = dict(overlap_length=N - hop, fft_length=K, window_padding="reflect", sampling_rate=fs)
        keep = {}

        def f_stft():
            z, _, _ = S.stft(xd, w, ctx, **opts)
            keep["z"] = z

        def f_two():
            z, _, _ = S.stft(xd, w, ctx, **opts)
            keep["m"] = S.stft_to_mel(z, fs, ctx, fft_length=K, mel_bins=mel)

        def f_fused():
            keep["m"] = S.mel_spectrogram(xd, w, ctx, mel_bins=mel, **opts)

        def f_one():
            keep["z"] = S.stft_onesided(xd, w, ctx, **opts)[0]

        for name, fn in (("stft", f_stft), ("stft_onesided", f_one), ("stft -> stft_to_mel", f_two), ("mel_spectrogram (fused)", f_fused)):
            try:
                ms = timeit(ctx, fn, reps=10, warm=3)
            except Exception as e:  # a call this build does not offer for the shape
                print(json.dumps({"case": name, "fft_length": K, "error": str(e)[:200]}), flush=True)
                continue
            M = (L + 2 * (N // 2) - N) // hop + 1
            frames = batch * M
            bpf = hop * 4 + (K * 8 if name == "stft" else K * 4 if name == "stft_onesided" else mel * 4)
            print(json.dumps({"case": f"{name}, N={N} hop={hop} fft_length={K} :reflect, {batch} x {secs} s @16 kHz", "frames": frames,
                              "ms": ms, "frames_per_s": frames / (ms * 1e-3), "audio_hours_per_s": batch * secs / 3600 / (ms * 1e-3),
                              "algorithmic_GBps": frames * bpf / (ms * 1e-3) / 1e9}), flush=True)
            keep.clear()


if __name__ == "__main__":
    main()
