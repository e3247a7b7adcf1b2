"""The reference's guides/filtering.livemd on the MI355X path, step for step (cited lines are the guide's).

  prepare the data (:12-37)  ->  firwin + hfft (:52-67)  ->  convolve(method: :fft, mode: :same) (:126-128)
  ->  stft(scaling: :spectrum) (:137-138)  ->  z * hfft (:141)  ->  istft (:154-157)

Every step is checked against the CPU oracle's restatement of the reference (tolerance 1e-5 of the peak), then the STFT-domain
chain is repeated at production size (16 streams x 60 s @ 48 kHz, N = 1024, hop = 256) with device-resident tensors: the
reference's three calls, and the same result from stft + istft_filtered (the filter fused into the inverse-STFT kernel).
Run on the GPU box: python examples/filtering_guide.py"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from oracle import nx_oracle as O  # noqa: E402  (checker only)


def nerr(a, b):
    return float(np.max(np.abs(np.asarray(a, np.complex128) - np.asarray(b, np.complex128))) / max(float(np.max(np.abs(b))), 1e-30))


def main():
    # ---- Prepare the data (:12-37)
    fs = 16.0e3
    window_length = 2 ** math.ceil(math.log2(fs * 100.0e-3))          # 2048
    signal_length = math.ceil(3 * fs)
    half_n = np.arange(signal_length // 2, dtype=np.float32)
    sin = lambda freq: np.sin(np.float32(2 * math.pi * freq / fs) * half_n).astype(np.float32)  # noqa: E731
    data = np.concatenate([sin(440) + sin(440 * 5 / 2), sin(220) + sin(220 * 4 / 3 * 4)]).astype(np.float32)

    # ---- Preparing the filter (:52-67)
    # (the guide passes `fc` as a number; firwin/3 itself insists on a list, lib/nx_signal/filters.ex:160-162 — mirrored)
    h = S.filters.firwin(window_length, [600], sampling_rate=fs, window="hann")
    assert np.array_equal(h, O.firwin(window_length, [600], sampling_rate=fs, window="hann")), "firwin is bit-exact with the reference rule"
    stft_window = S.windows.hann(window_length)
    hfft = (np.abs(S.transforms.fft_nd(h.astype(np.float32), axes=[0], lengths=[window_length])) + np.float32(1.0e-10)).astype(np.float32)

    # ---- Direct convolution (:126-128): convolve(data, h, mode: :same, method: :fft)
    data_filtered = S.convolution.convolve(data, h, mode="same", method="fft").astype(data.dtype)
    ref = O.fftconvolve(data, h, mode="same")
    print(f"convolve(method: :fft, mode: :same), {window_length} taps: err {nerr(data_filtered, ref):.2e}")

    # ---- Filtering the data in the STFT domain (:137-157)
    opts = dict(fft_length=window_length, sampling_rate=fs, scaling="spectrum")
    z, t, f = S.stft(data, stft_window, **opts)
    zo, to, fo = O.stft(data, stft_window, **opts)
    z_filtered = S.spectrum_multiply(z, hfft.astype(np.complex64))
    data_out = S.istft(z_filtered, stft_window, **opts).real.astype(data.dtype)
    zfo = (zo.astype(np.complex128) * hfft.astype(np.complex128)).astype(np.complex64)
    out_o = O.istft(zfo, stft_window, **opts).real.astype(data.dtype)
    inner = slice(window_length, -window_length)
    print(f"stft {z.shape}: err {nerr(z, zo):.2e}; times / frequencies identical: {np.array_equal(t, to) and np.array_equal(f, fo)}")
    print(f"stft -> z * hfft -> istft: err on the interior {nerr(data_out[inner], out_o[inner]):.2e}")
    fused = S.istft_filtered(z, hfft.astype(np.complex64), stft_window, **opts).real.astype(data.dtype)
    print(f"istft_filtered == multiply-then-istft bit for bit: {np.array_equal(fused, data_out)}")
    # the low-pass at 600 Hz keeps 440 / 220 Hz and removes 1100 / 1173 Hz
    spec = np.abs(np.fft.rfft(data_out[: signal_length // 2]))
    k = lambda hz: int(round(hz * (signal_length // 2) / fs))  # noqa: E731
    print(f"440 Hz kept / 1100 Hz removed: |Y(440)| / |Y(1100)| = {spec[k(440)] / max(spec[k(1100)], 1e-12):.0f}")

    # ---- the same chain at production size, device-resident
    ctx = S.default_context()
    rng = np.random.default_rng(0)
    B, L, N, hop = 16, 48000 * 60, 1024, 256
    xd = ctx.to_device(rng.standard_normal((B, L), dtype=np.float32))
    w = S.windows.hann(N)
    hf = np.fft.fft(np.asarray(S.filters.firwin(129, [4000], sampling_rate=48000), np.float64), N).astype(np.complex64)
    o2 = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    for name, fn in (("stft -> spectrum_multiply -> istft", lambda: S.istft(S.spectrum_multiply(S.stft(xd, w, **o2)[0], hf), w, **o2)),
                     ("stft -> istft_filtered", lambda: S.istft_filtered(S.stft(xd, w, **o2)[0], hf, w, **o2))):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            y = fn()
        ctx.sync()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name:36s} {B} x 60 s @ 48 kHz: {dt * 1e3:7.2f} ms per pass ({B * 60 / dt:,.0f} s of audio per second; includes the result allocations)")
    del y


if __name__ == "__main__":
    main()
