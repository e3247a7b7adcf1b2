"""Seeded differential fuzzing of the HIP path against the oracle: random frame lengths, hops, fft lengths (powers of
two — tuned and generic —, Bluestein and direct-DFT lengths), padding modes, scalings, batch shapes, ragged tails.
Deterministic (fixed seeds) so a failure reproduces."""
import numpy as np
import pytest

from oracle import nx_oracle as O

import nx_signal_amd as S

pytestmark = pytest.mark.gpu

FFT_LENGTHS = [8, 32, 64, 128, 256, 512, 1024, 2048, 4096, 100, 400, 640, 1000, 48, 3000, 8192, 320, 480, 960, 200, 600, 1200, 1100]
WINDOWS = ["hann", "hamming", "blackman", "bartlett", "triangular", "kaiser", "rectangular"]


def nerr(got, ref):
    d = np.abs(np.asarray(got).astype(np.complex128) - np.asarray(ref).astype(np.complex128))
    return float(d.max()) / max(float(np.max(np.abs(ref))), 1e-30)


def make_window(rng, n):
    name = WINDOWS[rng.integers(len(WINDOWS))]
    if name == "rectangular":
        return S.windows.rectangular(n, type="f32")
    if name in ("bartlett", "triangular"):
        return getattr(S.windows, name)(n)
    return getattr(S.windows, name)(n, is_periodic=bool(rng.integers(2)))


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_stft(seed):
    rng = np.random.default_rng(1000 + seed)
    K = FFT_LENGTHS[rng.integers(len(FFT_LENGTHS))]
    N = int(rng.choice([K, K, K, max(2, K // 2), max(2, int(K * 0.8)), min(K + K // 4, 5000)]))
    hop = int(rng.integers(1, N + 1))
    bshape = [(), (), (2,), (3,), (2, 2)][rng.integers(5)]
    pad = ["valid", "valid", "reflect", "same", [(int(rng.integers(0, N)), int(rng.integers(0, N)))]][rng.integers(5)]
    M_target = int(rng.integers(1, 40))
    L = max(N + (M_target - 1) * hop + int(rng.integers(0, hop)), 2 if pad == "reflect" else 1)
    if pad == "reflect":
        L = max(L, N // 2 + 2)
    scaling = [None, None, "spectrum", "psd"][rng.integers(4)]
    fs = float(rng.choice([100, 8000, 44100, 48000]))
    x = rng.standard_normal(bshape + (L,)).astype(np.float32)
    w = make_window(rng, N)
    opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=fs)
    z, t, f = S.stft(x, w, **opts)
    zo, to, fo = O.stft(x, w, **opts)
    assert z.shape == zo.shape, (opts, L)
    assert np.all(np.isfinite(z.view(np.float32)))
    assert nerr(z, zo) < 1e-5, (K, N, hop, L, pad, scaling, bshape, nerr(z, zo))
    assert np.array_equal(t, to, equal_nan=True) and np.array_equal(f, fo)


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_stft_of_complex_samples(seed):
    """round 5: c64 IQ data through nxsig_stft_c64 — the fused framed row kernels (1024 / 2048 / 4096) and the two-step path (every other
    length), random frame lengths, hops, paddings, scalings, batch shapes; host and device-resident"""
    rng = np.random.default_rng(7000 + seed)
    K = [1024, 1024, 2048, 4096, 512, 256, 100, 400, 1000, 3000, 8192, 48, 960][rng.integers(13)]
    N = int(rng.choice([K, K, max(2, K // 2), max(2, int(K * 0.8)), min(K + K // 4, 5000)]))
    hop = int(rng.integers(1, N + 1))
    bshape = [(), (), (2,), (3,), (2, 2)][rng.integers(5)]
    pad = ["valid", "valid", "reflect", "same", [(int(rng.integers(0, N)), int(rng.integers(0, N)))]][rng.integers(5)]
    M_target = int(rng.integers(1, 30))
    L = max(N + (M_target - 1) * hop + int(rng.integers(0, hop)), 2 if pad == "reflect" else 1)
    if pad == "reflect":
        L = max(L, N // 2 + 2)
    scaling = [None, None, "spectrum", "psd"][rng.integers(4)]
    fs = float(rng.choice([100, 8000, 48000]))
    x = (rng.standard_normal(bshape + (L,)) + 1j * rng.standard_normal(bshape + (L,))).astype(np.complex64)
    w = make_window(rng, N)
    opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=fs)
    zo, to, fo = O.stft(x, w, **opts)
    if seed % 2:
        z, t, f = S.stft(x, w, **opts)
    else:
        zd, t, f = S.stft(S.default_context(0).to_device(x), w, **opts)
        z = zd.numpy()
    assert z.shape == zo.shape, (opts, L)
    assert nerr(z, zo) < 1e-5, (K, N, hop, L, pad, scaling, bshape, nerr(z, zo))
    assert np.array_equal(t, to, equal_nan=True) and np.array_equal(f, fo)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_istft(seed):
    rng = np.random.default_rng(2000 + seed)
    N = int(rng.choice([8, 64, 100, 256, 512, 1024, 1024, 1024, 2048, 300, 4096]))
    hop = int(rng.choice([N, N // 2, N // 4, max(1, N // 8), int(rng.integers(1, N + 1))]))
    if N == 1024 and rng.integers(2):
        hop = int(rng.choice([128, 256, 512, 1024]))  # the tuned kernel's hops
    M = int(rng.integers(1, 70))
    bshape = [(), (2,), (3,)][rng.integers(3)]
    scaling = [None, "spectrum", "psd"][rng.integers(3)]
    z = (rng.standard_normal(bshape + (M, N)) + 1j * rng.standard_normal(bshape + (M, N))).astype(np.complex64)
    w = S.windows.hann(N) if rng.integers(2) else S.windows.hamming(N)
    opts = dict(overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=16000)
    y = S.istft(z, w, **opts)
    yo = O.istft(z, w, **opts)
    assert y.shape == yo.shape
    assert nerr(y, yo) < 1e-5, (N, hop, M, bshape, scaling, nerr(y, yo))


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_fir(seed):
    rng = np.random.default_rng(3000 + seed)
    taps = int(rng.choice([1, 3, 17, 64, 129, 257, 257, 385, 513, 700, 100, 255]))
    L = int(rng.integers(1, 30000))
    bshape = [(), (2,), (5,)][rng.integers(3)]
    mode = ["full", "same", "valid"][rng.integers(3)]
    x = rng.standard_normal(bshape + (L,)).astype(np.float32)
    h = (rng.standard_normal(taps) / np.sqrt(taps)).astype(np.float32)
    if len(bshape) == 0:
        y = S.convolution.convolve(x, h, method="fft", mode=mode)
        rows_x, rows_y = [x], [y]
        if L < taps:  # operands are swapped internally; :same stays centred on in1
            assert y.shape[0] == {"full": L + taps - 1, "same": L, "valid": taps - L + 1}[mode]
    else:
        y = S.filters.fir(x, h, mode=mode)
        rows_x, rows_y = list(x), list(y)
    for xr, yr in zip(rows_x, rows_y):
        full = O.direct_convolve_f64(xr, h)
        n = {"full": L + taps - 1, "same": L, "valid": abs(L - taps) + 1}[mode]
        start = (full.shape[0] - n) // 2 if mode != "full" else 0
        ref = full[start:start + n]
        assert yr.shape == ref.shape, (taps, L, mode)
        assert nerr(yr, ref) < 1e-5, (taps, L, mode, nerr(yr, ref))


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_fir_any_taps_offsets_and_slices(seed):
    """every tap count (the block grid pads it), every output offset (the grid is phased to it), rows of odd lengths and odd
    strides, and arbitrary slices of the full convolution through nxsig_fir_slice_f32 — against direct f64 convolution"""
    import ctypes as C

    from nx_signal_amd import _lib
    rng = np.random.default_rng(7000 + seed)
    taps = int(rng.integers(1, 1100)) if seed % 4 else int(rng.choice([33, 65, 129, 193, 257, 385, 513, 1025]))
    L = int(rng.integers(max(taps, 2000), 160000))
    B = int(rng.choice([1, 2, 3]))
    x = rng.standard_normal((B, L)).astype(np.float32)
    h = (rng.standard_normal(taps) / np.sqrt(taps)).astype(np.float32)
    full = [O.direct_convolve_f64(x[b], h) for b in range(B)]
    mode = ["full", "same", "valid"][rng.integers(3)]
    y = S.filters.fir(x, h, mode=mode)
    n = {"full": L + taps - 1, "same": L, "valid": L - taps + 1}[mode]
    start = {"full": 0, "same": (taps - 1) // 2, "valid": taps - 1}[mode]
    assert y.shape == (B, n)
    for b in range(B):
        assert nerr(y[b], full[b][start:start + n]) < 1e-5, (taps, L, mode)
    # an arbitrary slice of the full convolution, device buffers, row stride larger than the length
    ctx = S.default_context()
    lib = _lib.load()
    stride = L + int(rng.integers(0, 7))
    xs = np.zeros((B, stride), np.float32)
    xs[:, :L] = x
    xd = ctx.to_device(xs)
    o0 = int(rng.integers(0, L + taps - 2))
    olen = int(rng.integers(1, min(L + taps - 1 - o0, 70000) + 1))
    yd = ctx.empty((B, olen), np.float32)
    _lib.check(lib.nxsig_fir_slice_f32(ctx.handle, C.c_void_p(xd.ptr), L, B, stride, h.ctypes.data_as(C.c_void_p), taps, o0, olen,
                                       C.c_void_p(yd.ptr), _lib.DEVICE))
    ys = yd.numpy()
    for b in range(B):
        ref = full[b][o0:o0 + olen]
        assert float(np.max(np.abs(ys[b] - ref))) <= 1e-5 * max(float(np.max(np.abs(full[b]))), 1e-30), (taps, L, o0, olen)
    xd.free()
    yd.free()


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_istft_filtered_and_direct_convolution(seed):
    """the round's new entry points on random geometry: istft_filtered == multiply-then-istft (bits); convolve(method: :direct)
    == the oracle's restatement (bits)"""
    rng = np.random.default_rng(8000 + seed)
    N = int(rng.choice([1024, 1024, 512, 256, 2048, 400, 96, 300]))
    R = int(rng.choice([1, 2, 4, 8])) if N in (1024, 512, 256, 2048) else int(rng.choice([2, 3, 4]))
    hop = max(N // R, 1)
    M = int(rng.integers(1, 70))
    B = int(rng.choice([1, 2, 5]))
    z = (rng.standard_normal((B, M, N)) + 1j * rng.standard_normal((B, M, N))).astype(np.complex64)
    hf = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
    w = [S.windows.hann, S.windows.hamming, S.windows.blackman][rng.integers(3)](N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=8000, scaling=[None, "spectrum", "psd"][rng.integers(3)])
    want = S.istft(S.spectrum_multiply(z, hf), w, **opts)
    got = S.istft_filtered(z, hf, w, **opts)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, hop, M, B)
    rank = int(rng.integers(1, 4))
    s1 = tuple(int(v) for v in rng.integers(1, [400, 24, 9][rank - 1] + 1, size=rank))
    s2 = tuple(int(v) for v in rng.integers(1, [60, 8, 4][rank - 1] + 1, size=rank))
    mode = ["full", "same", "valid"][rng.integers(3)]
    if mode == "valid" and not (all(a >= b for a, b in zip(s1, s2)) or all(a <= b for a, b in zip(s1, s2))):
        mode = "same"
    a = rng.standard_normal(s1).astype(np.float32)
    b = rng.standard_normal(s2).astype(np.float32)
    if seed % 3 == 0:
        b = (b + 1j * rng.standard_normal(s2)).astype(np.complex64)
    d = S.convolution.convolve(a, b, mode=mode)
    e = O.convolve_direct(a, b, mode=mode)
    assert d.shape == e.shape and np.array_equal(np.ascontiguousarray(d).view(np.uint32), np.ascontiguousarray(e).view(np.uint32)), (s1, s2, mode)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_framing_and_ola(seed):
    rng = np.random.default_rng(4000 + seed)
    N = int(rng.integers(1, 200))
    stride = int(rng.integers(1, N + 3))
    L = int(rng.integers(N, 4000))
    pad = ["valid", "reflect", "same", [(int(rng.integers(-3, 50)), int(rng.integers(-3, 50)))]][rng.integers(4)]
    if pad == "reflect" and L < N // 2 + 2:
        L = N // 2 + 2
    x = rng.standard_normal((2, L)).astype(np.float32)
    got = S.as_windowed(x, window_length=N, stride=stride, padding=pad)
    exp = O.as_windowed(x, N, stride, pad)
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(exp).view(np.uint32)), (N, stride, L, pad)
    M = int(rng.integers(1, 50))
    ov = int(rng.integers(0, N))
    fr = rng.standard_normal((2, M, N)).astype(np.float32)
    assert np.array_equal(S.overlap_and_add(fr, overlap_length=ov).view(np.uint32), O.overlap_and_add(fr, ov).view(np.uint32))


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_stft_long_rows_interior_edge_split(seed):
    """longer rows (hundreds to thousands of frames) on the tuned lengths: large interior for the streaming kernels,
    edge units under every padding mode, frame_length above / below fft_length, odd hops, several rows"""
    rng = np.random.default_rng(5000 + seed)
    K = int(rng.choice([128, 256, 512, 1024, 2048, 4096, 400, 1000, 8192]))
    N = int(rng.choice([K, K, max(2, int(K * 0.78)), max(2, K // 2 + 1), K + K // 3]))
    hop = int(rng.choice([max(1, N // 4), max(1, N // 2), max(1, N // 3 + 1), max(1, int(rng.integers(1, N + 1)))]))
    M_target = int(rng.integers(150, 1500)) if K <= 1024 else (int(rng.integers(40, 300)) if K <= 4096 else int(rng.integers(20, 120)))
    pad = ["valid", "reflect", "same", [(int(rng.integers(0, 2 * N)), int(rng.integers(0, 2 * N)))],
           [(-int(rng.integers(0, N // 2 + 1)), int(rng.integers(0, N)))]][rng.integers(5)]
    L = N + (M_target - 1) * hop + int(rng.integers(0, hop))
    batch = int(rng.choice([1, 2, 5]))
    scaling = [None, "spectrum", "psd"][rng.integers(3)]
    x = rng.standard_normal((batch, L)).astype(np.float32)
    w = make_window(rng, N)
    opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=22050)
    if isinstance(pad, list) and L + pad[0][0] + pad[0][1] < N:  # a crop that leaves no room for one frame (1 in 90 000 fresh seeds)
        with pytest.raises(S.ArgumentError, match="does not fit"):
            S.stft(x, w, **opts)
        return
    z, t, f = S.stft(x, w, **opts)
    zo, to, fo = O.stft(x, w, **opts)
    assert z.shape == zo.shape, (opts, L)
    assert nerr(z, zo) < 1e-5, (K, N, hop, L, pad, scaling, batch, nerr(z, zo))
    assert np.array_equal(t, to, equal_nan=True) and np.array_equal(f, fo)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_fused_sinks(seed):
    """log-mel and magnitude sinks of the stft kernels against the two-step oracle chain"""
    rng = np.random.default_rng(6000 + seed)
    K = int(rng.choice([128, 256, 512, 1024, 2048, 4096, 400, 64]))
    N = int(rng.choice([K, K, max(2, int(K * 0.78))]))
    hop = int(rng.choice([max(1, N // 4), max(1, N // 2), max(1, int(rng.integers(1, N + 1)))]))
    pad = ["valid", "reflect", "same"][rng.integers(3)]
    L = N + int(rng.integers(5, 300)) * hop + int(rng.integers(0, hop))
    batch = int(rng.choice([1, 3]))
    scaling = [None, "spectrum"][rng.integers(2)]
    fs = 16000
    x = rng.standard_normal((batch, L)).astype(np.float32)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=fs)
    zo, _, _ = O.stft(x, w, **opts)
    half = K // 2
    mag_ref = np.abs(zo[..., :half].astype(np.complex128)).astype(np.float32)
    mag, _, _ = S.spectrogram(x, w, **opts)
    assert mag.shape == mag_ref.shape
    assert np.max(np.abs(mag - mag_ref)) / float(mag_ref.max()) < 1e-5, (K, N, hop, pad)
    mb = int(rng.choice([8, 40, 80]))
    if K >= 64:
        got = S.mel_spectrogram(x, w, mel_bins=mb, **opts)
        ref = O.stft_to_mel(zo.reshape(-1, K), fs, K, mel_bins=mb).reshape(zo.shape[:-1] + (mb,))
        assert got.shape == ref.shape
        assert np.max(np.abs(got - ref)) < 1e-4, (K, N, hop, pad, mb, float(np.max(np.abs(got - ref))))


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_istft_n400(seed):
    """N = 400 iSTFT: random hops (even: native 20 x 20 kernel; odd: generic path), frame counts, rows, scalings, windows"""
    rng = np.random.default_rng(7000 + seed)
    N = 400
    hop = int(rng.choice([160, 100, 80, 200, int(rng.integers(1, N + 1)), 2 * int(rng.integers(1, N // 2 + 1))]))
    M = int(rng.integers(1, 400))
    bshape = [(), (2,), (3,)][rng.integers(3)]
    scaling = [None, "spectrum", "psd"][rng.integers(3)]
    z = (rng.standard_normal(bshape + (M, N)) + 1j * rng.standard_normal(bshape + (M, N))).astype(np.complex64)
    w = make_window(rng, N)
    opts = dict(overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=16000)
    y = S.istft(z, w, **opts)
    yo = O.istft(z, w, **opts)
    assert y.shape == yo.shape
    assert nerr(y, yo) < 1e-5, (hop, M, bshape, scaling, nerr(y, yo))


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_stft_to_mel_bits(seed):
    """stft_to_mel on random spectra (any fft_length, band count, row count, dynamic range): the oracle's bits — at most one value
    in 100 000 may sit one ulp off (the tiled kernel's square root / logarithm carry ~1e-15 relative error before the same roundings)"""
    rng = np.random.default_rng(7000 + seed)
    K = int(rng.choice([16, 64, 100, 256, 400, 401, 512, 640, 1000, 1024, 2048, 4096, 8192]))
    mb = int(rng.integers(1, min(K // 2, 140) + 1))
    rows = int(rng.integers(1, max(2, 60000 // K)))
    fs = int(rng.choice([8000, 16000, 22050, 48000]))
    z = ((rng.standard_normal((rows, K)) + 1j * rng.standard_normal((rows, K))) * 10.0 ** rng.uniform(-4, 4, (rows, 1))).astype(np.complex64)
    got = S.stft_to_mel(z, fs, fft_length=K, mel_bins=mb)
    ref = O.stft_to_mel(z, fs, K, mel_bins=mb)
    assert got.shape == ref.shape
    assert int(np.sum(got != ref)) <= max(1, got.size // 100_000) and np.max(np.abs(got - ref)) < 2e-6, (K, mb, rows, fs)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_long_rows_and_columns(seed):
    """the tiled four-step (rows of 2^13 ... 2^20 points; zero-padded / truncated, real / complex, both directions) and the one-pass
    column transforms of fft_nd (a power-of-two length along a slower axis, inner size a multiple of 8) against numpy in double"""
    rng = np.random.default_rng(8000 + seed)
    K = 1 << int(rng.integers(13, 21))
    rows = int(rng.integers(1, max(2, (1 << 21) // K)))
    n_in = int(rng.choice([K, K - int(rng.integers(1, 1000)), K + int(rng.integers(1, 1000))]))
    x = rng.standard_normal((rows, n_in)).astype(np.float32)
    if seed & 1:
        x = (x + 1j * rng.standard_normal((rows, n_in))).astype(np.complex64)
    inverse = bool(seed & 2)
    fn, npf = (S.transforms.ifft_nd, np.fft.ifft) if inverse else (S.transforms.fft_nd, np.fft.fft)
    assert nerr(fn(x, lengths=[K]), npf(x.astype(np.complex128), n=K, axis=-1)) < 1e-5, (K, rows, n_in, inverse)
    Kc = 1 << int(rng.integers(4, 11))
    inner = 8 * int(rng.integers(1, 12))
    na = int(rng.choice([Kc, max(1, Kc - int(rng.integers(1, 9))), Kc + int(rng.integers(1, 9))]))
    y = rng.standard_normal((int(rng.integers(1, 4)), na, inner)).astype(np.float32)
    if seed & 4:
        y = (y + 1j * rng.standard_normal(y.shape)).astype(np.complex64)
    got = fn(y, axes=[1], lengths=[Kc])
    assert got.shape == (y.shape[0], Kc, inner)
    assert nerr(got, npf(y.astype(np.complex128), n=Kc, axis=1)) < 1e-5, (y.shape, Kc, inverse)


@pytest.mark.parametrize("seed", range(32))
def test_fuzz_non_finite_samples_follow_the_reference(seed):
    """round 3: random geometry (every front-end / padding / scaling), a few Inf / NaN samples at random places: the frames
    that are non-finite are exactly the oracle's, the finite ones agree as usual (lib/nx_signal.ex:94-102: one Nx.fft row per frame)"""
    rng = np.random.default_rng(7000 + seed)
    K = int(rng.choice([1024, 1024, 512, 256, 128, 2048, 4096, 400, 1000, 300, 64, 8192]))
    N = int(rng.choice([K, K, max(2, int(K * 0.8))]))
    hop = int(rng.choice([max(1, N // 4), max(1, N // 2), int(rng.integers(1, N + 1))]))
    pad = ["valid", "valid", "reflect", "same"][rng.integers(4)]
    B = int(rng.integers(1, 4))
    L = N + hop * int(rng.integers(3, 60)) + int(rng.integers(0, hop))
    x = rng.standard_normal((B, L)).astype(np.float32)
    for _ in range(int(rng.integers(1, 5))):
        x[rng.integers(B), rng.integers(L)] = [np.inf, -np.inf, np.nan][rng.integers(3)]
    w = make_window(rng, N)
    scaling = [None, "spectrum", "psd"][rng.integers(3)]
    opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=8000)
    z, _, _ = S.stft(x, w, **opts)
    zo, _, _ = O.stft(x, w, **opts)
    fin, fino = np.isfinite(z).all(axis=-1), np.isfinite(zo).all(axis=-1)
    assert np.array_equal(fin, fino), (K, N, hop, pad, L, B, np.argwhere(fin != fino)[:6])
    if fin.any():
        assert nerr(z[fin], zo[fin]) < 1e-5, (K, N, hop, pad)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_istft_non_finite_bins_and_packed_pair(seed):
    """round 3: (a) istft with a few non-finite bins: sample-for-sample the oracle's finite pattern (Nx.ifft row by row, :609);
    (b) stft_packed -> istft_packed against the oracle's full-spectrum stft / the real part of its istft"""
    rng = np.random.default_rng(8000 + seed)
    N = int(rng.choice([1024, 1024, 512, 256, 128, 2048, 400, 64]))
    hop = int(rng.choice([N // 4, N // 2, N, N // 8]))
    M = int(rng.integers(2 * (N // hop), 90))
    B = int(rng.integers(1, 4))
    w = S.windows.hann(N) if rng.integers(2) else S.windows.hamming(N)
    scaling = [None, "spectrum", "psd"][rng.integers(3)]
    opts = dict(overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=16000)
    z = (rng.standard_normal((B, M, N)) + 1j * rng.standard_normal((B, M, N))).astype(np.complex64)
    for _ in range(int(rng.integers(1, 4))):
        z[rng.integers(B), rng.integers(M), rng.integers(N)] = [np.inf, np.nan, complex(0, np.inf)][rng.integers(3)]
    y = np.asarray(S.istft(z, w, **opts))
    yo = np.stack([O.istft(z[b], w, **opts) for b in range(B)])
    fin, fino = np.isfinite(y), np.isfinite(yo)
    assert np.array_equal(fin, fino), (N, hop, M, B, np.argwhere(fin != fino)[:6])
    if fin.any():
        assert nerr(y[fin], yo[fin]) < 1e-5
    # (b) the packed pair
    x = rng.standard_normal((B, N + hop * (M - 1))).astype(np.float32)
    zp, _, _ = S.stft_packed(x, w, **opts)
    zo, _, _ = O.stft(x, w, **opts)
    pk = zo[..., : N // 2].copy()
    pk[..., 0] = pk[..., 0].real + 1j * zo[..., N // 2].real
    assert nerr(zp, pk) < 1e-5, (N, hop, M)
    yr = np.asarray(S.istft_packed(pk.astype(np.complex64), w, **opts))
    yro = np.stack([O.istft(zo[b].astype(np.complex64), w, **opts) for b in range(B)]).real
    assert yr.dtype == np.float32 and nerr(yr, yro) < 1e-5, (N, hop, M, nerr(yr, yro))


F64_LENGTHS = [4, 8, 32, 64, 128, 512, 1024, 2048, 4096, 8192, 100, 400, 640, 1000, 3000, 48, 7, 60, 5000]


@pytest.mark.parametrize("seed", range(32))
def test_fuzz_f64_tier(seed):
    """stft / istft / fftconvolve on f64 / c128 tensors (and mixed f32 / f64 operands) against the oracle's f64 section: random
    lengths over all three transform kinds (in-place radix-2^2, Bluestein, table DFT), hops, paddings, scalings, window types"""
    rng = np.random.default_rng(9000 + seed)
    K = F64_LENGTHS[rng.integers(len(F64_LENGTHS))]
    N = int(rng.choice([K, K, max(1, K // 2), max(1, int(K * 0.8)), min(K + K // 4, 8192)]))
    hop = int(rng.integers(1, N + 1))
    bshape = [(), (2,), (3,), (2, 2)][rng.integers(4)]
    pad = ["valid", "reflect", "same", [(int(rng.integers(0, N + 1)), int(rng.integers(0, N + 1)))]][rng.integers(4)]
    M_target = int(rng.integers(1, 12))
    L = N + (M_target - 1) * hop + int(rng.integers(0, hop))
    if pad == "reflect":
        L = max(L, N // 2 + 2)
    scaling = [None, "spectrum", "psd"][rng.integers(3)]
    wtype = ["f64", "f64", "f32"][rng.integers(3)]
    name = ["hann", "hamming", "blackman", "kaiser"][rng.integers(4)]
    w = getattr(S.windows, name)(N, type=wtype, is_periodic=bool(rng.integers(2)))
    if N == 1:
        w = np.ones(1, w.dtype)
    x = rng.standard_normal(bshape + (L,))
    if wtype == "f64" and rng.integers(3) == 0:
        x = x.astype(np.float32)   # f32 samples meet an f64 window: still the f64 tier
    opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=float(rng.choice([100, 16000, 48000])))
    z, t, f = S.stft(x, w, **opts)
    zo, to, fo = O.stft_f64(x, w, **opts)
    assert z.dtype == np.complex128 and z.shape == zo.shape, (opts, L)
    fin = np.isfinite(zo.real) & np.isfinite(zo.imag)
    if not fin.all():   # a window that sums to zero under :scaling divides by zero in the reference as well: same NaN / Inf pattern
        zc = np.ascontiguousarray(z).view(np.float64)
        zoc = np.ascontiguousarray(zo).view(np.float64)
        assert np.array_equal(np.isnan(zc), np.isnan(zoc)) and np.array_equal(np.isposinf(zc), np.isposinf(zoc)) and \
            np.array_equal(np.isneginf(zc), np.isneginf(zoc)), (K, N, hop, L, pad, scaling, wtype)
        return
    assert nerr(z, zo) < 1e-12, (K, N, hop, L, pad, scaling, wtype, nerr(z, zo))
    assert np.array_equal(t, to, equal_nan=True) and np.array_equal(f, fo)
    if K == N:   # invert what was just transformed; compared where the |w|^2 normaliser is well conditioned
        ov = dict(overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=opts["sampling_rate"])
        y = S.istft(zo, w, **ov)
        yo = O.istft_f64(zo, w, overlap_length=N - hop, scaling=scaling, sampling_rate=opts["sampling_rate"])
        den = O.overlap_and_add_f64(np.broadcast_to(np.abs(w.astype(np.float64)) ** 2, (zo.shape[-2], N)).copy(), N - hop)
        ok = den > 1.0e-6
        if ok.any() and np.max(np.abs(yo[..., ok])) > 0:
            assert nerr(y[..., ok], yo[..., ok]) < 1e-11, (N, hop, scaling, wtype)
    taps = int(rng.choice([1, 2, 17, 129, 300, 1025, 3000]))
    xs = rng.standard_normal((2, int(rng.integers(taps, 20000))))
    h = rng.standard_normal(taps) / np.sqrt(taps)
    mode = ["full", "same", "valid"][rng.integers(3)]
    yc = S.convolution.fftconvolve(xs, h, mode=mode)
    assert yc.dtype == np.float64 and nerr(yc, O.fftconvolve_f64(xs, h, mode=mode)) < 1e-11, (taps, mode)
