"""The packed one-sided pair (opt-in extension): `stft_packed` writes bins 0 .. K/2 - 1 with Re X[K/2] in the
imaginary part of bin 0; `istft_packed` inverts exactly that layout into a REAL signal.  Parity: against the oracle's full
two-sided stft (lib/nx_signal.ex:68-130) bin for bin, and against the real part of the oracle's istft (:582-638) on the full
Hermitian spectrum, normalised max error < 1e-5 (the tolerance of the path); plus the chain the pair exists for — the
reference's STFT-domain filtering workflow (guides/filtering.livemd:137-159) — against the oracle's full-spectrum chain."""
import numpy as np
import pytest

from oracle import nx_oracle as O

import nx_signal_amd as S

pytestmark = pytest.mark.gpu


def nerr(got, ref):
    d = np.abs(np.asarray(got).astype(np.complex128) - np.asarray(ref).astype(np.complex128))
    return float(d.max()) / max(float(np.max(np.abs(ref))), 1e-30)


def pack(zfull):
    K = zfull.shape[-1]
    zp = zfull[..., : K // 2].copy()
    zp[..., 0] = zp[..., 0].real + 1j * zfull[..., K // 2].real
    return zp


@pytest.mark.parametrize("K,N,hop,pad,scaling", [
    (1024, 1024, 256, "valid", None), (1024, 1024, 256, "reflect", "spectrum"), (1024, 600, 200, "valid", "psd"), (1024, 1024, 333, "same", None),
    (512, 512, 128, "valid", None), (2048, 2048, 512, "valid", "spectrum"), (400, 400, 160, "valid", None), (256, 200, 80, "reflect", None),
    (4096, 4096, 1024, "valid", None), (64, 64, 16, "valid", None), (1000, 1000, 250, "valid", None),
])
def test_stft_packed_is_the_full_spectrum_without_its_mirror(K, N, hop, pad, scaling):
    L = max(20000, 5 * K)
    x = np.stack([O.synth_signal(L, seed=40 + c) for c in range(3)])
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=16000)
    zp, t, f = S.stft_packed(x, w, **opts)
    zo, to, fo = O.stft(x, w, **opts)
    assert zp.shape == zo.shape[:-1] + (K // 2,) and np.array_equal(t, to) and np.array_equal(f, fo[: K // 2])
    assert nerr(zp, pack(zo)) < 1e-5
    # bins 1 .. K/2 - 1 and Re X[0] carry the same bits as the one-sided (and hence the full) spectrum
    z1, _, _ = S.stft_onesided(x, w, **opts)
    assert np.array_equal(zp[..., 1:].view(np.uint32), z1[..., 1:].view(np.uint32))
    assert np.array_equal(zp[..., 0].real.view(np.uint32), z1[..., 0].real.view(np.uint32))
    # device-resident input
    zd, _, _ = S.stft_packed(S.default_context().to_device(x), w, **opts)
    assert np.array_equal(zd.numpy().view(np.uint32), zp.view(np.uint32))


@pytest.mark.parametrize("N,hop,scaling,M,batch", [
    (1024, 256, None, 61, 3), (1024, 128, "spectrum", 40, 2), (1024, 512, "psd", 33, 2), (1024, 1024, None, 9, 1), (1024, 256, None, 7, 1),
    (1024, 256, None, 2000, 4),                         # several runs per row, odd / even run starts
    (1024, 200, None, 50, 2), (512, 128, None, 77, 2), (2048, 512, "spectrum", 21, 2), (256, 64, None, 100, 3), (400, 100, None, 31, 2),
    (64, 16, None, 40, 1),                               # shapes without a fused kernel: through the full layout
])
def test_istft_packed_equals_the_real_part_of_istft_on_the_hermitian_spectrum(N, hop, scaling, M, batch):
    rng = np.random.default_rng(N + hop + M)
    L = N + hop * (M - 1)
    x = rng.standard_normal((batch, L)).astype(np.float32)
    w = S.windows.hann(N)
    zo, _, _ = O.stft(x, w, overlap_length=N - hop, fft_length=N)           # exactly Hermitian rows (real frames)
    zo = zo.astype(np.complex64)
    opts = dict(overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=8000)
    yo = np.stack([O.istft(zo[b], w, **opts) for b in range(batch)])
    y = S.istft_packed(pack(zo), w, **opts)
    assert y.dtype == np.float32 and y.shape == yo.shape
    assert nerr(y, yo.real) < 1e-5, nerr(y, yo.real)
    assert float(np.max(np.abs(yo.imag))) < 1e-5 * float(np.max(np.abs(yo.real)))   # what the packed form drops is round-off
    yd = S.istft_packed(S.default_context().to_device(pack(zo)), w, **opts)
    assert np.array_equal(yd.numpy().view(np.uint32), y.view(np.uint32))
    y2 = S.istft_packed(pack(zo), w, **opts)
    assert np.array_equal(y2.view(np.uint32), y.view(np.uint32))               # run-to-run bit-stable (no atomics in the overlap-add)


def test_packed_round_trip_sixty_seconds():
    """config 3 in packed form: 60 s mono through stft_packed -> istft_packed comes back to 1.5e-7 on every interior sample"""
    x = O.synth_signal(2880000, seed=1234)
    w = S.windows.hann(1024)
    ctx = S.default_context()
    zp, _, _ = S.stft_packed(ctx.to_device(x), w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    y = S.istft_packed(zp, w, overlap_length=768, fft_length=1024, sampling_rate=48000).numpy()
    assert y.shape == (2880000,)
    assert float(np.max(np.abs(y[1024:-1024] - x[1024:-1024]))) < 2e-6 * float(np.max(np.abs(x)))


def test_packed_filtering_chain_vs_the_reference_chain():
    """guides/filtering.livemd:137-159 — stft -> multiply by the filter's spectrum -> istft -> real part — in packed form.  The
    pointwise product of two packed tensors treats bin 0 as the two reals it is (DC and Nyquist)."""
    fs = 48000
    x = O.synth_signal(fs * 2, seed=77)
    w = S.windows.hann(1024)
    h = S.filters.firwin(129, [3000.0], sampling_rate=fs)
    H = np.fft.fft(h.astype(np.float64), 1024).astype(np.complex64)          # hfft of the guide
    opts = dict(overlap_length=768, fft_length=1024, sampling_rate=fs)
    zo, _, _ = O.stft(x, w, **opts)
    yo = O.istft((zo.astype(np.complex128) * H).astype(np.complex64), w, **opts).real
    zp, _, _ = S.stft_packed(x, w, **opts)
    Hp = pack(H[None, :])[0]
    prod = zp * Hp
    prod[:, 0] = zp[:, 0].real * Hp[0].real + 1j * (zp[:, 0].imag * Hp[0].imag)   # DC x DC, Nyquist x Nyquist (both real)
    y = S.istft_packed(prod.astype(np.complex64), w, **opts)
    assert nerr(y, yo) < 1e-5


def test_istft_packed_non_finite_bins_reach_only_their_own_frames():
    rng = np.random.default_rng(3)
    N, hop, M = 1024, 256, 61
    x = rng.standard_normal((2, N + hop * (M - 1))).astype(np.float32)
    w = S.windows.hann(N)
    zo, _, _ = O.stft(x, w, overlap_length=N - hop, fft_length=N)
    zo = zo.astype(np.complex64)
    zo[0, 20, 5] = np.inf; zo[0, 20, N - 5] = np.inf          # a Hermitian pair of non-finite bins
    zo[1, 33, 0] = np.nan
    yo = np.stack([O.istft(zo[b], w, overlap_length=N - hop, fft_length=N) for b in range(2)]).real
    y = S.istft_packed(pack(zo), w, overlap_length=N - hop, fft_length=N)
    assert np.array_equal(np.isfinite(y), np.isfinite(yo))
    fin = np.isfinite(yo)
    assert nerr(y[fin], yo[fin]) < 1e-5
