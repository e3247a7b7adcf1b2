"""The f64 / c128 tier (include/nxsig.h "f64 / c128 tier", nx_signal_amd/csrc/kernels_f64.hip) against the oracle's f64 section.

The reference computes in the type of its operands (lib/nx_signal.ex:101-102, :609; convolution.ex:276-284).  It holds no f64
vector for this path, so these tests pin the HIP kernels to the oracle's double-precision restatement (numpy's pocketfft in
double: an independent transform) to a tolerance of 1e-12 of the largest magnitude of the expected result — four orders of
magnitude below what a c64 computation could reach, i.e. they fail if anything on the way drops to single precision."""
import numpy as np
import pytest

import nx_signal_amd as S
from oracle import nx_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1.0e-12   # of max |expected|


def close(got, ref, rtol=RTOL):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape and got.dtype == ref.dtype, (got.shape, ref.shape, got.dtype, ref.dtype)
    scale = float(np.max(np.abs(ref))) if ref.size else 0.0
    err = float(np.max(np.abs(got - ref))) if ref.size else 0.0
    assert err <= rtol * max(scale, 1.0e-300), (err, scale)


def sig(shape, seed):
    return np.random.Generator(np.random.PCG64(seed)).standard_normal(shape)


@pytest.mark.parametrize("N,K,hop", [
    (4, 4, 1), (8, 8, 2), (64, 64, 16), (128, 128, 32), (512, 512, 128), (1024, 1024, 256), (2048, 2048, 512), (4096, 4096, 1024),
    (8192, 8192, 4096),                          # power-of-two lengths, even and odd log2
    (400, 512, 160), (1024, 256, 256), (3, 8, 1),  # zero-padded / truncated rows (Nx.fft(length:))
    (100, 100, 25), (400, 400, 160), (1000, 1000, 250), (1025, 4096, 300), (3000, 3000, 1500),   # Bluestein
    (48, 48, 12), (7, 7, 3), (60, 60, 20),       # table DFT: short non-powers of two
    (5000, 5000, 2500), (4100, 16384, 2000),     # table DFT beyond the Bluestein range
])
def test_stft_f64_matches_the_oracle(N, K, hop):
    x = sig((2, max(3 * N + 17, 4 * hop + N)), N + K)
    w = S.windows.hann(N, type="f64") if N > 1 else np.ones(1)
    z, t, f = S.stft(x, w, overlap_length=N - hop, fft_length=K, sampling_rate=16000)
    zr, tr, fr_ = O.stft_f64(x, w, overlap_length=N - hop, fft_length=K, sampling_rate=16000)
    assert z.dtype == np.complex128
    close(z, zr)
    assert t.dtype == np.float32 and np.array_equal(t, tr) and np.array_equal(f, fr_)


@pytest.mark.parametrize("padding", ["valid", "reflect", "same", [(5, 3)], [(-2, -1)]])
@pytest.mark.parametrize("scaling", [None, "spectrum", "psd"])
def test_stft_f64_padding_and_scaling(padding, scaling):
    N, hop = 256, 64
    x = sig((3, 2000), 5)
    for w in (S.windows.hamming(N, type="f64"), S.windows.hamming(N)):   # the scalar of :scaling is formed in the window's type
        z, _, _ = S.stft(x, w, overlap_length=N - hop, window_padding=padding, scaling=scaling, sampling_rate=8000)
        zr, _, _ = O.stft_f64(x, w, overlap_length=N - hop, window_padding=padding, scaling=scaling, sampling_rate=8000)
        close(z, zr)


@pytest.mark.parametrize("N,K,hop", [(64, 64, 16), (512, 512, 128), (1024, 1024, 256), (400, 512, 160), (1024, 256, 256),
                                      (100, 100, 25), (1000, 1000, 250), (48, 48, 12), (7, 7, 3), (5000, 5000, 2500)])
def test_stft_c128_samples_match_the_oracle(N, K, hop):
    """nxsig_stft_c128 (round 6): complex f64 samples — the reference frames, multiplies (c128 x f64 componentwise) and transforms whatever
    tensor it is given (lib/nx_signal.ex:94-102).  All three kernel kinds (radix-2, Bluestein, table DFT), host and device buffers, the
    linearity of the transform (stft(a + i b) = stft(a) + i stft(b)) and the f64 entry's bits left as they were."""
    xr, xi = sig((2, max(3 * N + 17, 4 * hop + N)), N + K), sig((2, max(3 * N + 17, 4 * hop + N)), N + K + 1)
    x = xr + 1j * xi
    w = S.windows.hann(N, type="f64")
    for pad in ("valid", "reflect"):
        opts = dict(overlap_length=N - hop, fft_length=K, sampling_rate=16000, window_padding=pad, scaling="spectrum")
        z, t, f = S.stft(x, w, **opts)
        zo, to, fo = O.stft_f64(x, w, **opts)
        assert z.dtype == np.complex128
        close(z, zo)
        assert np.array_equal(t, to) and np.array_equal(f, fo)
        za, zb = S.stft(xr, w, **opts)[0], S.stft(xi, w, **opts)[0]
        close(z, za + 1j * zb, rtol=1e-13)
    ctx = S.Context(0)
    zd = S.stft(ctx.to_device(x), w, ctx=ctx, overlap_length=N - hop, fft_length=K, sampling_rate=16000)[0]
    assert ctx.last_dispatch().startswith("stft.f64.c128")
    assert np.array_equal(zd.numpy().view(np.uint64), S.stft(x, w, overlap_length=N - hop, fft_length=K, sampling_rate=16000)[0].view(np.uint64))


def test_stft_c64_samples_with_an_f64_window_compute_in_c128_and_non_finite_samples_stay_in_their_frames():
    N, hop = 256, 64
    x = (sig((2, 3000), 9) + 1j * sig((2, 3000), 10)).astype(np.complex64)
    w = S.windows.hann(N, type="f64")
    z = S.stft(x, w, overlap_length=N - hop)[0]
    zo = O.stft_f64(x, w, overlap_length=N - hop)[0]
    assert z.dtype == np.complex128
    close(z, zo)
    assert S.stft(x, w.astype(np.float32), overlap_length=N - hop)[0].dtype == np.complex64     # the c64 path is something else
    xn = x.astype(np.complex128)
    xn[0, 1000] = np.nan
    xn[1, 2000] = complex(0.0, np.inf)
    zn = S.stft(xn, w, overlap_length=N - hop)[0]
    zno = O.stft_f64(xn, w, overlap_length=N - hop)[0]
    assert np.array_equal(np.isfinite(zn).all(axis=-1), np.isfinite(zno).all(axis=-1))
    ok = np.isfinite(zno)
    assert float(np.max(np.abs(zn[ok] - zno[ok]))) <= 1e-12 * float(np.max(np.abs(zno[ok])))


def test_stft_f32_samples_with_an_f64_window_compute_in_double():
    N = 512
    x = sig(6000, 3).astype(np.float32)
    w = S.windows.blackman(N, type="f64")
    z, _, _ = S.stft(x, w)
    zr, _, _ = O.stft_f64(x, w)
    assert z.dtype == np.complex128
    close(z, zr)
    # and the f32 call is something else: a c64 result
    assert S.stft(x, w.astype(np.float32))[0].dtype == np.complex64


def test_stft_f64_eps_cleanup_and_nonfinite_frames():
    N, hop = 128, 32
    w = S.windows.hann(N, type="f64")
    z, _, _ = S.stft(1.0e-14 * sig(2000, 1), w, overlap_length=N - hop)
    assert np.all(z == 0)   # every component is below Nx.fft's eps: exact zeros like the reference's
    x = sig(2000, 2)
    x[700] = np.nan
    x[1500] = np.inf
    z, _, _ = S.stft(x, w, overlap_length=N - hop)
    zr, _, _ = O.stft_f64(x, w, overlap_length=N - hop)
    fin, finr = np.isfinite(z).all(axis=-1), np.isfinite(zr).all(axis=-1)
    assert np.array_equal(fin, finr) and 0 < fin.sum() < fin.size   # only the frames that hold the sample
    close(z[fin], zr[finr])


@pytest.mark.parametrize("N,hop", [(8, 2), (256, 64), (1024, 256), (1024, 1024), (400, 100), (2048, 512), (100, 50)])
@pytest.mark.parametrize("wdt", ["f64", "f32"])
def test_istft_c128_matches_the_oracle(N, hop, wdt):
    w = S.windows.hann(N, type=wdt)
    z = sig((2, 9, N), N) + 1j * sig((2, 9, N), N + 1)
    for scaling in (None, "spectrum", "psd"):
        y = S.istft(z, w, overlap_length=N - hop, scaling=scaling, sampling_rate=100, fft_length=N)
        yr = O.istft_f64(z, w, overlap_length=N - hop, scaling=scaling, sampling_rate=100)
        assert y.dtype == np.complex128
        # where the |w|^2 normaliser is tiny (the first / last samples under a tapered window) the quotient amplifies the
        # transform's last-ulp noise by 1 / den: compare those samples relative to that amplification
        den = O.overlap_and_add_f64(np.broadcast_to(np.abs(w.astype(np.float64)) ** 2, (9, N)).copy(), N - hop)
        ok = den > 1.0e-6
        close(y[..., ok], yr[..., ok])
        amp = np.where(den > 1.0e-10, den, 1.0)
        assert np.all(np.abs(y - yr) * amp <= 1.0e-11 * np.max(np.abs(yr[..., ok])))


def test_stft_istft_round_trip_in_double():
    N, hop = 1024, 256
    x = sig((2, 48000), 9)
    w = S.windows.hann(N, type="f64")
    z, _, _ = S.stft(x, w, overlap_length=N - hop, window_padding="reflect")
    y = S.istft(z, w, overlap_length=N - hop)
    core = y[:, N // 2: N // 2 + x.shape[1]]
    assert np.max(np.abs(core.real - x)) < 1.0e-12 and np.max(np.abs(core.imag)) < 1.0e-12   # c64 would stop at ~1e-6


def test_device_resident_f64_chain():
    N, hop = 512, 128
    ctx = S.default_context()
    x = sig((3, 20000), 4)
    w = S.windows.hann(N, type="f64")
    xd = ctx.to_device(x)
    zd, _, _ = S.stft(xd, w, overlap_length=N - hop)
    assert S.is_device(zd) and zd.dtype == np.complex128
    yd = S.istft(zd, w, overlap_length=N - hop)
    zr, _, _ = O.stft_f64(x, w, overlap_length=N - hop)
    close(zd.numpy(), zr)
    close(yd.numpy()[:, N:-N], O.istft_f64(zr, w, overlap_length=N - hop)[:, N:-N])


@pytest.mark.parametrize("n_in,K", [(16, 16), (1000, 1024), (1024, 512), (100, 100), (777, 777), (30, 30), (8192, 8192), (6, 4)])
def test_fft_nd_last_axis_in_double(n_in, K):
    a = sig((3, 2, n_in), n_in)
    c = a + 1j * sig((3, 2, n_in), K)
    for t in (a, c):
        f = S.transforms.fft_nd(t, axes=[-1], lengths=[K])
        assert f.dtype == np.complex128
        close(f, O._eps_clean(np.fft.fft(t, n=K, axis=-1), O.FFT_EPS).astype(np.complex128))
        b = S.transforms.ifft_nd(t, axes=[2], lengths=[K])
        close(b, O._eps_clean(np.fft.ifft(t, n=K, axis=-1), O.FFT_EPS).astype(np.complex128))
    with pytest.raises(S.NxSignalUnsupported):
        S.transforms.fft_nd(a, axes=[0])


def test_as_windowed_and_overlap_and_add_in_double():
    x = sig((2, 1000), 8)
    for pad in ("valid", "reflect", "same", [(3, 4)]):
        got = S.as_windowed(x, window_length=64, stride=16, padding=pad)
        assert got.dtype == np.float64 and np.array_equal(got, O.as_windowed(x, 64, 16, pad))
    fr = sig((2, 7, 64), 1)
    assert np.array_equal(S.overlap_and_add(fr, overlap_length=48), O.overlap_and_add_f64(fr, 48))
    frc = fr + 1j * sig((2, 7, 64), 2)
    got = S.overlap_and_add(frc, overlap_length=32)
    assert got.dtype == np.complex128 and np.array_equal(got, O.overlap_and_add_f64(frc, 32))


@pytest.mark.parametrize("taps", [1, 2, 3, 129, 257, 1025, 2049, 4097])
@pytest.mark.parametrize("mode", ["full", "same", "valid"])
def test_fftconvolve_f64_matches_the_oracle_and_the_direct_sum(taps, mode):
    x = sig((2, 30000), taps)
    h = sig(taps, 1) / max(taps, 1) ** 0.5
    y = S.convolution.fftconvolve(x, h, mode=mode)
    assert y.dtype == np.float64
    close(y, O.fftconvolve_f64(x, h, mode=mode), 2.0e-12)
    full = np.stack([np.convolve(r, h) for r in x])
    new = {"full": full.shape[1], "same": x.shape[1], "valid": x.shape[1] - taps + 1}[mode]
    st = (full.shape[1] - new) // 2
    close(y, full[:, st:st + new], 2.0e-12)


def test_fftconvolve_f64_rows_with_a_nonfinite_sample_come_out_nan_and_limits():
    x = sig((3, 20000), 2)
    x[1, 777] = np.inf
    h = S.filters.firwin(257, [0.2], type="f64")
    y = S.filters.fir(x, h)
    assert np.isnan(y[1]).all() and np.isfinite(y[0]).all() and np.isfinite(y[2]).all()
    y2 = S.filters.fir(np.where(np.isfinite(x), x, 0.0), h)   # the flag was consumed: the next call is clean
    assert np.isfinite(y2).all()
    close(y2[0], O.fftconvolve_f64(x[0], h, mode="same"), 2.0e-12)
    with pytest.raises(S.NxSignalUnsupported):
        S.convolution.fftconvolve(sig(50000, 1), sig(5000, 2))
    # mixed f32 / f64 operands are computed in double
    y3 = S.convolution.fftconvolve(x[0].astype(np.float32), h)
    close(y3, np.convolve(x[0].astype(np.float32).astype(np.float64), h), 2.0e-12)


def test_f64_limits_are_reported_not_approximated():
    with pytest.raises(S.NxSignalUnsupported):
        S.stft(sig(70000, 1), np.ones(16384), fft_length=16384 * 8)
    with pytest.raises(S.NxSignalUnsupported):   # a c64 spectrum with an f64 window would be inverted in c64 by the reference
        S.istft(np.zeros((3, 8), np.complex64), np.ones(8))
    with pytest.raises(S.ArgumentError):
        S.stft_packed(sig(100, 1), np.ones(8, np.float32))


def test_python_lists_follow_nx_tensor_inference():
    """Nx.tensor([0.1, ...]) is f32: a plain list of Python floats must take the f32 path; only an np.float64 ARRAY is an f64 tensor"""
    xs = [float(v) for v in sig(64, 1)]
    w = S.windows.hann(16)
    assert S.stft(xs, w)[0].dtype == np.complex64 and S.stft(np.asarray(xs), w)[0].dtype == np.complex128
    z = S.stft(xs, w)[0]
    assert S.istft(z.tolist(), w).dtype == np.complex64
    assert S.as_windowed(xs, window_length=8).dtype == np.float32 and S.as_windowed([1, 2, 3, 4, 5], window_length=2).dtype.kind == "i"
    assert S.overlap_and_add([[1.0, 2.0], [3.0, 4.0]], overlap_length=1).dtype == np.float32
    assert S.convolution.fftconvolve(xs, [0.5, 0.25]).dtype == np.float32
    assert S.transforms.fft_nd(xs).dtype == np.complex64 and S.transforms.fft_nd(np.asarray(xs)).dtype == np.complex128
    assert S.waveforms.sinc([0.0, 0.5]).dtype == np.float32
