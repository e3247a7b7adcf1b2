"""GPU parity tests (pytest -m gpu, run on a real MI355X through gpurun).

Every test drives the HIP path through the C ABI (nx_signal_amd -> ctypes -> libnxsig.so) and compares
with the CPU oracle (oracle/nx_oracle.py, the BinaryBackend restatement) on the same seeded inputs.

Tolerance (BASELINE.json north_star / SURVEY §0.9): normalised max error
    max|got - ref| / max|ref| < 1e-5       (fp32 butterflies vs an f64-internal reference land at ~2e-7)
plus rel-L2 < 1e-6.  Index-only operations (as_windowed) are bit-exact; overlap_and_add accumulates in
double and rounds once like the reference, so it is bit-exact too.
"""
import numpy as np
import pytest

from conftest import f32_list, nx_all_close
from oracle import nx_oracle as O

import nx_signal_amd as S

pytestmark = pytest.mark.gpu

TOL_MAX = 1e-5
TOL_L2 = 1e-6


def nerr(got, ref):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    d = np.abs(got.astype(np.complex128) - ref.astype(np.complex128))
    scale = max(float(np.max(np.abs(ref))), 1e-30)
    l2 = float(np.sqrt(np.sum(d ** 2)) / max(np.sqrt(np.sum(np.abs(ref.astype(np.complex128)) ** 2)), 1e-30))
    return float(d.max()) / scale, l2


def assert_close(got, ref, what="", tol_max=TOL_MAX, tol_l2=TOL_L2):
    assert np.all(np.isfinite(np.asarray(got).view(np.float32))), what
    m, l2 = nerr(got, ref)
    assert m < tol_max and l2 < tol_l2, f"{what}: normalised max err {m:.3e}, rel-L2 {l2:.3e}"
    return m, l2


@pytest.fixture(scope="module")
def ctx():
    return S.default_context()


def test_library_loaded_and_gpu_is_gfx950(ctx):
    name = ctx.name()
    assert "gfx950" in name, name


# ------------------------------------------------------------------------------- golden vectors
def test_stft_doctest_golden(golden):
    for v in golden["stft"]:
        z, t, f = S.stft(np.array(v["x"]), S.windows.rectangular(v["window"]["n"]), **v["opts"])
        exp = np.array([[complex(float(c[0]), float(c[1])) for c in row] for row in v["z"]], dtype=np.complex64)
        assert z.dtype == np.complex64 and z.shape == exp.shape
        assert np.array_equal(z, exp), v["src"]  # small integers: exact
        assert np.array_equal(t, f32_list(v["t"])) and np.array_equal(f, f32_list(v["f"]))


def test_stft_istft_roundtrip_doctests(golden):
    for v in golden["stft_istft_roundtrip"]:
        x = np.array(v["x"], dtype=np.float32)
        w = S.windows.hann(v["window"]["n"])
        z, _, _ = S.stft(x, w, **v["opts"])
        y = S.istft(z, w, **v["opts"])
        assert y.dtype == np.complex64 and y.shape == (8,)
        exp = np.array([float(e) for e in v["expect"]], dtype=np.float32)
        assert nx_all_close(y.real, exp, atol=1e-5, rtol=1e-5), (v["src"], y)
        assert np.max(np.abs(y.imag)) < 1e-5
        assert y.real[0] == 0.0  # OLA normaliser guard: first sample comes back 0 with Hann (B9)
        zo, _, _ = O.stft(x, w, **v["opts"])
        assert_close(z, zo, v["src"])
        assert_close(y, O.istft(zo, w, **v["opts"]), v["src"])


def test_as_windowed_golden(golden):
    for v in golden["as_windowed"]:
        pad = v["padding"]
        if isinstance(pad, list):
            pad = [tuple(p) for p in pad]
        got = S.as_windowed(np.array(v["x"]), window_length=v["window_length"], stride=v["stride"], padding=pad)
        assert got.tolist() == v["expect"], v["src"]


def test_overlap_and_add_golden(golden):
    for v in golden["overlap_and_add"]:
        t = np.arange(np.prod(v["shape"])).reshape(v["shape"]) if v.get("iota") else np.array(v["t"])
        got = S.overlap_and_add(t, overlap_length=v["overlap_length"])
        assert got.tolist() == v["expect"], v["src"]


def test_fftconvolve_golden(golden):
    for v in golden["fftconvolve"]:
        got = S.convolution.convolve(np.array(v["a"], np.float32), np.array(v["b"], np.float32), method="fft", mode=v["mode"])
        exp = np.array([float(e) for e in v["expect"]])
        assert got.dtype == np.float32
        assert nx_all_close(got, exp, 1e-4, 1e-4), (v["src"], got)


def test_fftconvolve_complex_golden_and_oracle(golden):
    for v in golden["fftconvolve_complex"]:
        a = np.array([complex(*c) for c in v["a"]], dtype=np.complex64)
        b = np.array([complex(*c) for c in v["b"]], dtype=np.complex64)
        got = S.convolution.convolve(a, b, method="fft", mode=v["mode"])
        assert got.dtype == np.complex64
        assert nx_all_close(got, np.array([complex(*c) for c in v["expect"]]), v["atol"], v["rtol"]), v["src"]
    rng = np.random.default_rng(12)
    for n1, n2 in [(3, 3), (100, 9), (9, 100), (4000, 4000), (1, 7)]:
        a = (rng.standard_normal(n1) + 1j * rng.standard_normal(n1)).astype(np.complex64)
        b = (rng.standard_normal(n2) + 1j * rng.standard_normal(n2)).astype(np.complex64)
        for mode in ("full", "same", "valid"):
            got = S.convolution.fftconvolve(a, b, mode=mode)
            exp = O.fftconvolve(a, b, mode=mode)
            assert_close(got, exp, f"complex {n1}x{n2} {mode}")
    mixed = S.convolution.fftconvolve(np.array([1, 2, 3], np.float32), np.array([1j, 2, 3], np.complex64))
    assert mixed.dtype == np.complex64  # "don't complexify" only when both are real (convolutions_test.exs:392-416)


def test_stft_to_mel_doctest_and_oracle(golden):
    """NxSignal.stft_to_mel/3 (lib/nx_signal.ex:465-483 doctest) on top of the HIP stft"""
    for v in golden["stft_to_mel"]:
        x = np.arange(v["x_iota"])
        w = S.windows.hann(v["window"]["n"])
        z, _, _ = S.stft(x, w, **v["opts"])
        mel = S.stft_to_mel(z, v["opts"]["sampling_rate"], fft_length=v["opts"]["fft_length"], mel_bins=v["mel_bins"])
        exp = np.array([f32_list(row) for row in v["expect"]])
        assert mel.shape == exp.shape and mel.dtype == np.float32
        assert nx_all_close(mel, exp, atol=1e-5, rtol=1e-5), (mel, exp)
    rng = np.random.default_rng(21)
    xs = rng.standard_normal((3, 40000)).astype(np.float32)
    for K, mb, fs in [(1024, 128, 16000), (512, 80, 16000), (2048, 64, 48000)]:
        w = S.windows.hann(K)
        opts = dict(overlap_length=K - K // 4, fft_length=K, sampling_rate=fs)
        zo, _, _ = O.stft(xs, w, **opts)
        ref = O.stft_to_mel(zo.reshape(-1, K), fs, K, mel_bins=mb).reshape(zo.shape[:-1] + (mb,))
        got_same = S.stft_to_mel(zo, fs, fft_length=K, mel_bins=mb)           # same spectrum through both
        assert np.max(np.abs(got_same - ref)) < 2e-6, np.max(np.abs(got_same - ref))
        ctx = S.default_context()
        zd, _, _ = S.stft(ctx.to_device(xs), w, **opts)                       # device-resident chain
        got = S.stft_to_mel(zd, fs, fft_length=K, mel_bins=mb).numpy()
        assert got.shape == ref.shape and np.max(np.abs(got - ref)) < 1e-4


@pytest.mark.parametrize("K,mb,fs,rows", [
    (400, 80, 16000, 1000),    # the speech front-end's transform: 32-frame tiles
    (400, 80, 16000, 1),       # a single frame
    (401, 80, 16000, 33),      # odd fft_length: 200 bins of 401
    (512, 80, 16000, 777), (256, 40, 8000, 1001), (64, 10, 8000, 130), (16, 4, 8000, 5),
    (1024, 128, 48000, 300), (2048, 64, 48000, 70), (4096, 40, 48000, 19),
    (8192, 20, 48000, 9),      # four-frame tiles (the first form of the kernel did not fit the LDS here)
    (1000, 33, 22050, 65),
    (1024, 300, 48000, 40),    # more bands than the tiled kernel's table holds: the first form
])
def test_stft_to_mel_is_bit_identical_to_the_oracle(K, mb, fs, rows):
    """NxSignal.stft_to_mel/3 (lib/nx_signal.ex:486-513) on a given spectrum: Nx.abs in double -> f32 -> ** 2, the band sums in
    double rounded once, log / log(10) in f32 arithmetic on a double log, the global maximum, clamp, (x + 4) / 4 — the tiled
    kernel reproduces the oracle's bits (its square root and logarithm are cheaper sequences with ~1e-15 relative error before the
    same single roundings; a mismatch of one ulp is possible only within ~1e-7 ulp of a rounding boundary, allowed for below)."""
    rng = np.random.default_rng(K + mb + rows)
    z = ((rng.standard_normal((rows, K)) + 1j * rng.standard_normal((rows, K))) * 10.0 ** rng.uniform(-3, 3, (rows, 1))).astype(np.complex64)
    got = S.stft_to_mel(z, fs, fft_length=K, mel_bins=mb)
    ref = O.stft_to_mel(z, fs, K, mel_bins=mb)
    assert got.shape == ref.shape == (rows, mb) and got.dtype == np.float32
    bad = int(np.sum(got != ref))
    assert bad <= max(1, got.size // 100_000), (bad, got.size)
    assert np.max(np.abs(got - ref)) < 2e-6
    zd = S.default_context().to_device(z)                      # device-resident input, device-resident result
    assert np.array_equal(S.stft_to_mel(zd, fs, fft_length=K, mel_bins=mb).numpy(), got)


@pytest.mark.parametrize("mb", [40, 300])   # the tiled kernel / the first form (more bands than the tile's table holds)
def test_stft_to_mel_extreme_magnitudes(mb):
    """bins whose |z|^2 leaves the f32 normal range (the double square-root path), exact zeros, an overflowing bin and a NaN:
    the reference's dense dot turns a non-finite |z|^2 into NaN for the whole tensor (inf x 0 weights, then reduce_max)"""
    K, fs = 512, 16000
    rng = np.random.default_rng(5)
    z = (rng.standard_normal((64, K)) + 1j * rng.standard_normal((64, K))).astype(np.complex64)
    z[3] *= np.float32(1e-18)       # |z|^2 ~ 1e-36: below the fast path's range, denormal squares
    z[4] *= np.float32(1e-25)       # squares flush to zero: the 1e-10 clip decides
    z[5] *= np.float32(3e16)        # |z|^2 ~ 1e33: above the fast path's range, still finite
    z[6] = 0
    z[7, 10:20] = 0
    got = S.stft_to_mel(z, fs, fft_length=K, mel_bins=mb)
    ref = O.stft_to_mel(z, fs, K, mel_bins=mb)
    assert np.all(np.isfinite(got)) and np.array_equal(got, ref)
    z[9] *= np.float32(1e25)        # band energy overflows f32 -> +inf -> every value clamps against max - 8 = inf
    with np.errstate(all="ignore"):
        ref = O.stft_to_mel(z, fs, K, mel_bins=mb)
    got = S.stft_to_mel(z, fs, fft_length=K, mel_bins=mb)
    assert np.all(np.isnan(ref)) and np.array_equal(got, ref, equal_nan=True)
    z[9] = z[8]
    z[60, 17] = np.nan
    got = S.stft_to_mel(z, fs, fft_length=K, mel_bins=mb)
    assert np.all(np.isnan(got))
    z[60, 17] = 1.0
    assert np.all(np.isfinite(S.stft_to_mel(z, fs, fft_length=K, mel_bins=mb)))   # the flag is per call


@pytest.mark.parametrize("K,N,hop,pad,scaling,mb", [
    (1024, 1024, 256, "valid", None, 128),      # fused wave kernel, streaming
    (1024, 1024, 256, "reflect", "psd", 80),    # fused, general loader + scaling
    (1024, 1000, 250, "valid", None, 64),       # fused, N < K
    (1024, 1024, 256, "valid", None, 7),        # fewer bands than lanes
    (512, 512, 128, "valid", None, 80),         # quad front-end (4 frames per transform) + fused mel
    (512, 400, 160, "reflect", None, 80),       # speech front-end: 25 ms frames, 10 ms hop, 512-point FFT, centred
    (512, 400, 160, "valid", "psd", 40),
    (256, 256, 64, "same", None, 40),           # 8 frames per transform
    (128, 128, 32, "valid", "spectrum", 20),    # 16 frames per transform
    (2048, 2048, 512, "valid", "spectrum", 128),  # real-2x front-end
    (2048, 1200, 300, "reflect", None, 64),
    (4096, 4096, 1024, "valid", None, 128),     # 2048-point core
    (400, 400, 160, "valid", None, 80),         # non-power-of-two: two-step path behind the same entry
])
def test_mel_spectrogram_fused_matches_two_step_and_oracle(K, N, hop, pad, scaling, mb):
    rng = np.random.default_rng(K + mb)
    x = rng.standard_normal((3, 30000 + 7)).astype(np.float32)  # odd frame counts per row
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K, sampling_rate=16000, window_padding=pad, scaling=scaling)
    got = S.mel_spectrogram(x, w, mel_bins=mb, **opts)
    z, _, _ = S.stft(x, w, **opts)
    two = S.stft_to_mel(z, 16000, fft_length=K, mel_bins=mb)
    zo, _, _ = O.stft(x, w, **opts)
    ref = O.stft_to_mel(zo.reshape(-1, K), 16000, K, mel_bins=mb).reshape(zo.shape[:-1] + (mb,))
    assert got.shape == two.shape == ref.shape and got.dtype == np.float32
    assert np.max(np.abs(got - two)) < 2e-5, np.max(np.abs(got - two))
    assert np.max(np.abs(got - ref)) < 1e-4, np.max(np.abs(got - ref))
    ctx = S.default_context()
    gd = S.mel_spectrogram(ctx.to_device(x), w, mel_bins=mb, **opts)
    assert np.array_equal(gd.numpy().view(np.uint32), got.view(np.uint32))


def test_fft_rows_golden(golden):
    for v in golden["fft_rows"]:
        z = S.transforms.fft_nd(np.array(v["x"]), axes=[-1], lengths=[v["length"]])
        assert nx_all_close(z[0] + z[1], np.array([complex(*c) for c in v["expect_sum"]]), v["atol"], v["rtol"])
        assert nx_all_close(z[0] - z[1], np.array([complex(*c) for c in v["expect_diff"]]), v["atol"], v["rtol"])


# ------------------------------------------------------------------------------- STFT vs oracle
STFT_CASES = [
    # (L, N, overlap, fft_length, padding, scaling, fs, batch_shape)
    (64, 4, 2, 4, "valid", None, 100, ()),
    (100, 8, 6, 16, "reflect", None, 100, ()),
    (1000, 64, 48, 64, "valid", "spectrum", 8000, ()),
    (1000, 64, 32, 128, "same", "psd", 8000, (3,)),
    (1000, 100, 50, "power_of_two", "valid", None, 100, ()),  # N=100 -> K=128 zero-pad
    (1000, 100, 25, 100, "valid", None, 100, ()),  # non power of two: direct DFT path
    (700, 48, 47, 12, "valid", None, 100, ()),  # fft_length < N: truncation; hop 1
    (4000, 256, 192, 256, "reflect", "psd", 16000, (2, 2)),
    (5000, 512, 384, 512, [(7, 300)], None, 100, ()),
    (5000, 512, 384, 512, [(-5, -9)], None, 100, ()),  # negative explicit padding crops (Nx.pad)
    (9000, 1024, 768, 1024, "valid", None, 48000, ()),
    (9000, 1024, 768, 1024, "reflect", "spectrum", 48000, (2,)),
    (9000, 1024, 512, 2048, "valid", None, 48000, ()),  # zero-padded K=2048 from N=1024
    (20000, 2048, 1536, 2048, "valid", None, 48000, (3,)),
    (20000, 2048, 1536, 2048, "same", "psd", 48000, ()),
    (40000, 4096, 3072, 4096, "valid", None, 48000, ()),
    (40000, 8192, 4096, 8192, "valid", None, 48000, ()),
    (3000, 1000, 500, 1000, "valid", None, 100, ()),  # K=1000 direct DFT
    (5, 4, 2, 4, "reflect", None, 100, ()),  # tiny ragged input
    (4, 4, 0, 4, "valid", None, 100, ()),  # exactly one frame, no overlap
]


@pytest.mark.parametrize("case", STFT_CASES, ids=lambda c: f"L{c[0]}-N{c[1]}-ov{c[2]}-K{c[3]}-{c[4] if isinstance(c[4], str) else 'explicit'}-{c[5]}")
def test_stft_matches_oracle(case):
    L, N, overlap, K, pad, scaling, fs, bshape = case
    rng = np.random.default_rng(1234 + L + N)
    x = rng.standard_normal(bshape + (L,)).astype(np.float32)
    w = S.windows.hann(N)
    opts = dict(overlap_length=overlap, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=fs)
    z, t, f = S.stft(x, w, **opts)
    zo, to, fo = O.stft(x, w, **opts)
    assert z.dtype == np.complex64
    assert_close(z, zo, str(case))
    # M == 1 makes Nx.linspace divide 0/0 (App. A rule 5): both sides give NaN by the same rule
    assert np.array_equal(t, to, equal_nan=True) and np.array_equal(f, fo)


@pytest.mark.parametrize("case", STFT_CASES, ids=lambda c: f"L{c[0]}-N{c[1]}-ov{c[2]}-K{c[3]}-{c[4] if isinstance(c[4], str) else 'explicit'}-{c[5]}")
@pytest.mark.parametrize("where", ["host", "device"])
def test_stft_of_complex_samples_matches_oracle(case, where):
    """c64 IQ data.  The reference frames, multiplies and transforms whatever tensor it is given
    (lib/nx_signal.ex:94-102); every STFT_CASES shape — all padding modes, truncation / zero-padding, every scaling, wave / LDS /
    Bluestein / four-step lengths — with complex samples, host and device-resident, equals the oracle to 1e-5."""
    L, N, overlap, K, pad, scaling, fs, bshape = case
    rng = np.random.default_rng(4321 + L + N)
    x = (rng.standard_normal(bshape + (L,)) + 1j * rng.standard_normal(bshape + (L,))).astype(np.complex64)
    w = S.windows.hann(N)
    opts = dict(overlap_length=overlap, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=fs)
    zo, to, fo = O.stft(x, w, **opts)
    if where == "host":
        z, t, f = S.stft(x, w, **opts)
    else:
        zd, t, f = S.stft(S.default_context(0).to_device(x), w, **opts)
        z = zd.numpy()
    assert z.dtype == np.complex64 and z.shape == zo.shape
    assert_close(z, zo, str(case))
    assert np.array_equal(t, to, equal_nan=True) and np.array_equal(f, fo)
    # a complex signal's spectrum is NOT Hermitian: the result must differ from the real part's transform alone
    if L >= 64:
        zr, _, _ = O.stft(np.ascontiguousarray(x.real), w, **opts)
        assert float(np.max(np.abs(zo - zr))) > 1e-3 * float(np.max(np.abs(zo)))


def test_stft_of_complex_samples_long_stream_and_refusals():
    """60 s of complex IQ at N = 1024 hop = 256 through the fused row kernels (one frame per transform), sampled against the oracle, and
    linearity (size-independent property): stft(a + i b) == stft(a) + i stft(b) to fp32 round-off; the one-sided extensions refuse it"""
    L, N, hop = 2_880_000, 1024, 256
    rng = np.random.default_rng(77)
    a = rng.standard_normal(L, dtype=np.float32)
    b = rng.standard_normal(L, dtype=np.float32)
    x = (a + 1j * b).astype(np.complex64)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    ctx = S.default_context(0)
    z = S.stft(ctx.to_device(x), w, **opts)[0].numpy()
    assert z.shape == (11247, 1024)
    for m in (0, 1, 5000, 11245, 11246):
        zo = O.stft(x[m * hop: m * hop + N], w, **opts)[0][0]
        assert float(np.max(np.abs(z[m] - zo)) / np.max(np.abs(zo))) < 1e-5, m
    za = S.stft(a, w, **opts)[0]
    zb = S.stft(b, w, **opts)[0]
    assert float(np.max(np.abs(z - (za + 1j * zb))) / np.max(np.abs(z))) < 1e-6
    with pytest.raises(S.ArgumentError):
        S.stft_onesided(x[:9000], w, **opts)


@pytest.mark.parametrize("K,N,hop", [(100, 100, 25), (1000, 1000, 250), (400, 400, 160), (65, 65, 13), (3000, 3000, 750),
                                     (4095, 4000, 1000), (640, 400, 160), (300, 500, 100), (4097, 4097, 2000), (7, 7, 3)])
def test_stft_non_power_of_two_lengths(K, N, hop):
    """Bluestein path for 64 < K <= 4096 (incl. zero-padded and truncated frames), direct DFT outside it"""
    rng = np.random.default_rng(K)
    x = rng.standard_normal((2, max(5 * N, 3000))).astype(np.float32)
    w = S.windows.hamming(N)
    for scaling in (None, "spectrum"):
        opts = dict(overlap_length=N - hop, fft_length=K, scaling=scaling, sampling_rate=8000)
        z, _, _ = S.stft(x, w, **opts)
        zo, _, _ = O.stft(x, w, **opts)
        assert_close(z, zo, f"K={K} N={N}")


@pytest.mark.parametrize("n_in,K", [(100, 100), (1000, 1000), (257, 300), (3000, 2999), (66, 66)])
def test_fft_rows_non_power_of_two_bluestein(n_in, K):
    rng = np.random.default_rng(n_in + K)
    a = (rng.standard_normal((3, n_in)) + 1j * rng.standard_normal((3, n_in))).astype(np.complex64)
    assert_close(S.transforms.fft_nd(a, lengths=[K]), O.fft(a, length=K), "fft")
    assert_close(S.transforms.ifft_nd(a, lengths=[K]), O.ifft(a, length=K), "ifft")
    r = rng.standard_normal((3, n_in)).astype(np.float32)
    assert_close(S.transforms.fft_nd(r, lengths=[K]), O.fft(r, length=K), "fft real")


def test_stft_other_windows_and_integer_input():
    x = (np.arange(300) % 17 - 8).astype(np.int32)  # integer tensors are legal (B15)
    for w in (S.windows.hamming(32), S.windows.blackman(32), S.windows.kaiser(32, beta=8.0), S.windows.rectangular(32)):
        z, _, _ = S.stft(x, w, overlap_length=24, fft_length=32)
        zo, _, _ = O.stft(x, w, overlap_length=24, fft_length=32)
        assert_close(z, zo)


def test_stft_config1_one_second_mono():
    """BASELINE config 1: 1 s mono 48 kHz, N=1024 hop=256 Hann -> 184 frames."""
    x = O.synth_signal(48000, seed=1234)
    w = S.windows.hann(1024)
    z, t, f = S.stft(x, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert z.shape == (184, 1024)
    zo, to, fo = O.stft(x, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    m, l2 = assert_close(z, zo, "config1")
    assert np.array_equal(t, to) and np.array_equal(f, fo)
    # real input -> Hermitian spectrum
    assert np.max(np.abs(z[:, 1:] - np.conj(z[:, :0:-1]))) / np.max(np.abs(z)) < 1e-6


def test_stft_config2_sixty_seconds_full_size():
    """BASELINE config 2 at full size (11 247 frames) against the oracle, host and device-resident."""
    x = O.synth_signal(2880000, seed=1234)
    w = S.windows.hann(1024)
    opts = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
    z, _, _ = S.stft(x, w, **opts)
    assert z.shape == (11247, 1024)
    zo, _, _ = O.stft(x, w, **opts)
    assert_close(z, zo, "config2")
    ctx = S.default_context()
    xd = ctx.to_device(x)
    zd, _, _ = S.stft(xd, w, **opts)
    assert isinstance(zd, S.DeviceBuffer) and zd.shape == (11247, 1024)
    assert np.array_equal(zd.numpy().view(np.uint32), z.view(np.uint32))  # host-staged == device-resident, bit for bit


def test_stft_linearity_and_shift_properties_full_size():
    """size-independent properties at config-2 size: linearity and hop-shift covariance."""
    L, N, hop = 2880000, 1024, 256
    a = O.synth_signal(L, seed=1)
    b = O.synth_signal(L, seed=2)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    za, _, _ = S.stft(a, w, **opts)
    zb, _, _ = S.stft(b, w, **opts)
    zab, _, _ = S.stft((a + np.float32(0.5) * b).astype(np.float32), w, **opts)
    m, _ = nerr(zab, za + np.float32(0.5) * zb)
    assert m < 1e-5
    zs, _, _ = S.stft(a[hop:], w, **opts)  # shifting the signal by one hop shifts the frames by one
    m, _ = nerr(zs, za[1:])  # (frames change real/imag lanes in the packed FFT, so equal to rounding, not bitwise)
    assert m < 1e-6


# ------------------------------------------------------------------------------- iSTFT vs oracle
ISTFT_CASES = [
    (64, 4, 2, None, 1000), (100, 8, 6, "spectrum", 1000), (1000, 64, 48, "psd", 8000), (1000, 64, 0, None, 1000),
    (1000, 100, 75, None, 1000),  # non power of two
    (9000, 1024, 768, None, 48000), (9000, 1024, 768, "psd", 48000), (9000, 1024, 1023, None, 48000),
    (20000, 2048, 1536, "spectrum", 48000), (40000, 4096, 2048, None, 48000), (3000, 1000, 900, None, 100),
]


@pytest.mark.parametrize("case", ISTFT_CASES, ids=lambda c: f"L{c[0]}-N{c[1]}-ov{c[2]}-{c[3]}")
def test_istft_matches_oracle(case):
    L, N, overlap, scaling, fs = case
    rng = np.random.default_rng(99 + L + N)
    x = rng.standard_normal((2, L)).astype(np.float32)
    w = S.windows.hann(N)
    zo, _, _ = O.stft(x, w, overlap_length=overlap, fft_length=N, scaling=scaling, sampling_rate=fs)
    # perturb so the spectrum is NOT Hermitian (users edit spectra: guides/filtering.livemd:143)
    zo = (zo * (1 + 0.1j)).astype(np.complex64)
    y = S.istft(zo, w, overlap_length=overlap, fft_length=N, scaling=scaling, sampling_rate=fs)
    yo = O.istft(zo, w, overlap_length=overlap, fft_length=N, scaling=scaling, sampling_rate=fs)
    assert y.dtype == np.complex64 and y.shape == yo.shape
    assert_close(y, yo, str(case))


def test_roundtrip_config3_sixty_seconds():
    """BASELINE config 3: stft -> istft on 60 s mono, chain kept in HBM.

    (a) iSTFT parity proper: the SAME spectrum (the oracle's) through the HIP istft vs the oracle istft,
        max normalised err < 1e-5 on every one of the 2 880 000 samples, edges included.
    (b) the GPU chain stft->istft vs the oracle chain and vs the input x on every well-conditioned sample.
    (c) the first/last ~60 samples divide by an OLA normaliser den[n] << 1 (Hann taper): there ANY 1e-7
        difference in z (the GPU stft is 1.2e-7 from the oracle's) is amplified by 1/sqrt(den) in the reference
        and here alike, so the chain is compared with that conditioning factor divided out.
    """
    x = O.synth_signal(2880000, seed=1234)
    w = S.windows.hann(1024)
    opts = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
    ctx = S.default_context()
    zd, _, _ = S.stft(ctx.to_device(x), w, **opts)  # chain stays in HBM
    yd = S.istft(zd, w, **opts)
    y = yd.numpy()
    assert y.shape == (2880000,) and y.dtype == np.complex64
    zo, _, _ = O.stft(x, w, **opts)
    yo = O.istft(zo, w, **opts)
    # (a)
    y_same = S.istft(ctx.to_device(zo), w, **opts).numpy()
    assert_close(y_same, yo, "config3 istft, same spectrum")
    # (b)
    den = O.overlap_and_add(np.broadcast_to((w * w).astype(np.float32), (11247, 1024)), 768, dtype=np.float32)
    good = den >= 0.02 * den.max()
    assert good.sum() >= 2880000 - 400  # ~140 tapered samples at each end
    assert_close(y[good], yo[good], "config3 chain vs oracle chain (well-conditioned samples)")
    m, _ = nerr(y.real[good], x[good])
    assert m < 1e-5, m
    # (c)
    bad = ~good & (den > 1e-10)
    cond_scaled = np.abs(y[bad] - yo[bad]) * np.sqrt(den[bad]) / np.max(np.abs(yo))
    assert cond_scaled.max() < 1e-5, cond_scaled.max()
    assert y.real[0] == 0.0  # B9: guarded normaliser -> first sample is 0
    y2 = S.istft(zd, w, **opts).numpy()
    assert np.array_equal(y2.view(np.uint32), y.view(np.uint32))  # deterministic OLA: run-to-run bit-stable


def test_as_windowed_and_ola_vs_oracle_exact():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 5000)).astype(np.float32)
    for N, stride, pad in [(64, 16, "valid"), (100, 33, "reflect"), (7, 7, "same"), (128, 1, [(3, 200)])]:
        got = S.as_windowed(x, window_length=N, stride=stride, padding=pad)
        exp = O.as_windowed(x, N, stride, pad)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (N, stride, pad)
    fr = rng.standard_normal((2, 40, 64)).astype(np.float32)
    for ov in (0, 16, 48, 63):
        got = S.overlap_and_add(fr, overlap_length=ov)
        assert np.array_equal(got.view(np.uint32), O.overlap_and_add(fr, ov).view(np.uint32)), ov
    frc = (fr + 1j * fr[::-1]).astype(np.complex64)
    got = S.overlap_and_add(frc, overlap_length=48)
    assert np.array_equal(got.view(np.uint32), O.overlap_and_add(frc, 48).view(np.uint32))


@pytest.mark.parametrize("n_in,K", [(8, 8), (5, 5), (3, 8), (16, 4), (1024, 1024), (1000, 1024), (257, 257), (4096, 8192)])
def test_fft_rows_vs_oracle(n_in, K):
    rng = np.random.default_rng(n_in * 7 + K)
    a = (rng.standard_normal((5, n_in)) + 1j * rng.standard_normal((5, n_in))).astype(np.complex64)
    assert_close(S.transforms.fft_nd(a, lengths=[K]), O.fft(a, length=K), "fft")
    assert_close(S.transforms.ifft_nd(a, lengths=[K]), O.ifft(a, length=K), "ifft")
    r = rng.standard_normal((5, n_in)).astype(np.float32)
    assert_close(S.transforms.fft_nd(r, lengths=[K]), O.fft(r, length=K), "fft real")


# ------------------------------------------------------------------------------- FIR
@pytest.mark.parametrize("L,taps,mode", [(100, 9, "full"), (100, 9, "same"), (100, 9, "valid"), (5000, 257, "same"),
                                         (5000, 256, "same"), (100000, 257, "same"), (100000, 31, "full"),
                                         (40000, 1025, "valid"), (300, 257, "same"), (64, 257, "full")])
def test_fir_matches_reference_fftconvolve(L, taps, mode):
    rng = np.random.default_rng(L + taps)
    x = rng.standard_normal(L).astype(np.float32)
    h = S.filters.firwin(taps if taps % 2 else taps + 1, [0.25])[:taps]
    got = S.convolution.convolve(x, h, method="fft", mode=mode)
    if L * 1.0 * (L + taps) < 4e8 and L <= 5000:
        exp = O.fftconvolve(x, h, mode=mode)  # the reference formulation: one FFT of length L+K-1
        assert_close(got, exp, "vs fftconvolve oracle")
    full = O.direct_convolve_f64(x, h)  # independent check: direct convolution in double
    n = {"full": L + taps - 1, "same": L, "valid": abs(L - taps) + 1}[mode]
    start = (full.shape[0] - n) // 2 if mode != "full" else 0
    assert_close(got, full[start:start + n].astype(np.float32), "vs direct f64")


def test_fir_config5_one_million_samples_batched():
    """BASELINE config 5 shape (257-tap low-pass, 48 kHz) on a 1 M-sample slice x 4 channels vs direct f64."""
    h = S.filters.firwin(257, [4000], sampling_rate=48000)
    x = O.synth_signal(1_000_000, seed=7, channels=4)
    ctx = S.default_context()
    yd = S.filters.fir(ctx.to_device(x), h, mode="same")
    y = yd.numpy()
    assert y.shape == (4, 1_000_000)
    for ch in range(4):
        full = O.direct_convolve_f64(x[ch], h)
        assert_close(y[ch], full[128:128 + 1_000_000].astype(np.float32), f"ch{ch}")


def test_stft_config4_shard_full_size_beyond_4gb():
    """BASELINE config 4, one GPU's shard at FULL size: 8 channels x 10 min @ 48 kHz, N=2048 hop=512 -> 8 x 56 247
    frames = 7.37 GB of spectrum.  Frames sampled across the whole output (incl. byte offsets > 4 GiB) are compared
    with the oracle; guards the 64-bit index arithmetic of the kernels."""
    import ctypes as C

    from nx_signal_amd import _lib

    ch, L, N, hop = 8, 28_800_000, 2048, 512
    M = (L - N) // hop + 1
    assert M == 56247
    ctx = S.default_context()
    lib = _lib.load()
    base = O.synth_signal(L, seed=4321)
    xd = ctx.empty((ch, L), np.float32)
    rows = []
    for c in range(ch):
        xr = np.roll(base, 100_003 * c) * np.float32(1.0 + 0.1 * c)
        rows.append(xr)
        _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xd.ptr + c * L * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes))
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    zd, _, _ = S.stft(xd, w, **opts)
    assert zd.shape == (ch, M, N) and zd.nbytes == ch * M * N * 8 > 6 * 2**30
    ctx.sync()
    for c, m in [(0, 0), (0, M - 1), (3, 12345), (4, 33000), (5, 17), (7, M - 1), (7, M - 2), (6, 50001)]:
        got = np.empty(N, np.complex64)
        off = (c * M + m) * N * 8
        _lib.check(lib.nxsig_download(ctx.handle, got.ctypes.data_as(C.c_void_p), C.c_void_p(zd.ptr + off), got.nbytes))
        ref, _, _ = O.stft(rows[c][m * hop: m * hop + N], w, **opts)
        assert_close(got, ref[0], f"channel {c} frame {m} (byte offset {off})")
    # config 3's direction on the same 7.37 GB spectrum: the inverse reads byte offsets > 4 GiB; round-trip property
    # (Hann, 75 % overlap: x is recovered wherever OLA(w^2) is fully covered) on spans sampled across every channel
    yd = S.istft(zd, w, **opts)
    Ly = (M - 1) * hop + N
    assert yd.shape == (ch, Ly)
    ctx.sync()
    for c, s in [(0, N), (0, Ly - 2 * N - 4096), (3, 9_000_000), (4, 17_000_001), (6, 28_000_000), (7, Ly - 2 * N - 4096),
                 (7, 5 * hop + 3)]:
        got = np.empty(4096, np.complex64)
        off = (c * Ly + s) * 8
        _lib.check(lib.nxsig_download(ctx.handle, got.ctypes.data_as(C.c_void_p), C.c_void_p(yd.ptr + off), got.nbytes))
        assert_close(got, rows[c][s: s + 4096].astype(np.complex64), f"round trip channel {c} sample {s}")
    yd.free()
    zd.free()
    xd.free()


def test_fir_stream_beyond_4gb():
    """FIR over more than 4 GiB of samples in one call (40 channels x 10 min @ 48 kHz = 4.6 GB in, 4.6 GB out; config 5's
    filter): spans sampled across the whole output, including the last channel's end, vs direct convolution in double."""
    import ctypes as C

    from nx_signal_amd import _lib

    ch, L, taps = 40, 28_800_000, 257
    ctx = S.default_context()
    lib = _lib.load()
    base = O.synth_signal(L, seed=99)
    h = S.filters.firwin(taps, [4000], sampling_rate=48000)
    xd = ctx.empty((ch, L), np.float32)
    assert xd.nbytes > 4 * 2**30
    shifts = [(7_919 * c) % L for c in range(ch)]
    for c in range(ch):
        xr = np.roll(base, shifts[c])
        _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xd.ptr + c * L * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes))
    yd = S.filters.fir(xd, h, mode="same")
    assert yd.shape == (ch, L)
    ctx.sync()
    n = 8192
    for c, s in [(0, 0), (0, L - n), (17, 1_234_567), (37, 14_400_001), (38, 31), (39, L - n), (39, 20_000_000)]:
        got = np.empty(n, np.float32)
        off = (c * L + s) * 4
        _lib.check(lib.nxsig_download(ctx.handle, got.ctypes.data_as(C.c_void_p), C.c_void_p(yd.ptr + off), got.nbytes))
        xr = np.roll(base, shifts[c])
        lo, hi = max(0, s - 128), min(L, s + n + 128)
        full = O.direct_convolve_f64(xr[lo:hi], h)  # full[j] = sum_k h[k] x[lo + j - k]; y[i] = full_of_row[i + 128]
        j0 = s + 128 - lo
        assert_close(got, full[j0: j0 + n].astype(np.float32), f"channel {c} sample {s} (byte offset {off})")
    yd.free()
    xd.free()


# ------------------------------------------------------------------------------- STFT-domain filtering chain (8f-3)
def test_spectrum_multiply_bit_exact_and_shapes():
    rng = np.random.default_rng(11)
    z = (rng.standard_normal((3, 17, 256)) + 1j * rng.standard_normal((3, 17, 256))).astype(np.complex64)
    h = (rng.standard_normal(256) + 1j * rng.standard_normal(256)).astype(np.complex64)
    exp = (z.astype(np.complex128) * h.astype(np.complex128)).astype(np.complex64)  # double, one rounding (BinaryBackend)
    got = S.spectrum_multiply(z, h)
    assert got.shape == z.shape and np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    ctx = S.default_context()
    gd = S.spectrum_multiply(ctx.to_device(z), h)
    assert np.array_equal(gd.numpy().view(np.uint32), exp.view(np.uint32))
    with pytest.raises(S.ArgumentError):
        S.spectrum_multiply(z, h[:100])


def test_stft_domain_filtering_chain_stays_on_device():
    """guides/filtering.livemd:137-159: stft(scaling: :spectrum) -> z * hfft -> istft(scaling: :spectrum), every
    intermediate device-resident; compared with the same chain through the oracle."""
    fs, N, hop = 8000, 1024, 256
    x = O.synth_signal(40000, seed=21)
    w = S.windows.hann(N)
    hcoef = S.filters.firwin(101, [1000], sampling_rate=fs)
    hfft = O.fft(hcoef, length=N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=fs, scaling="spectrum")
    ctx = S.default_context()
    zd, _, _ = S.stft(ctx.to_device(x), w, **opts)
    zf = S.spectrum_multiply(zd, hfft)
    yd = S.istft(zf, w, **opts)
    assert S.device.is_device(zd) and S.device.is_device(zf) and S.device.is_device(yd)
    zo, _, _ = O.stft(x, w, **opts)
    zfo = (zo.astype(np.complex128) * hfft.astype(np.complex128)).astype(np.complex64)
    yo = O.istft(zfo, w, **opts)
    y = yd.numpy()
    assert y.shape == yo.shape
    inner = slice(N, -N)  # well-conditioned interior (edges: see the conditioning note in the istft tests)
    assert_close(y[inner], yo[inner], "filtered chain, interior")
    assert nerr(y, yo)[0] < 2e-4


@pytest.mark.parametrize("N,hop,scaling,batch,frames", [
    (1024, 256, None, 3, 41),          # fused: the filter rides in the inverse-STFT kernel
    (1024, 128, "spectrum", 2, 30),
    (1024, 512, "psd", 1, 9),
    (1024, 1024, None, 2, 5),          # hop == N: every sample goes through the double-precision edge fix-up (filtered there too)
    (1024, 256, None, 1, 6),           # fewer frames than the tuned kernel takes: two-step behind the same entry
    (1024, 300, None, 2, 12),          # hop the tuned kernel does not take
    (512, 128, "spectrum", 2, 25),     # other sizes: product materialised once, then the size's own kernel
    (2048, 512, None, 1, 11),
    (4096, 1024, None, 1, 9),
    (400, 160, None, 2, 14),
    (256, 64, None, 3, 33),
    (96, 24, None, 2, 20),             # generic path
])
def test_istft_filtered_is_bit_identical_to_multiply_then_istft(N, hop, scaling, batch, frames):
    """istft_filtered(z, h, w) == istft(spectrum_multiply(z, h), w) bit for bit, host and device inputs; z is left untouched"""
    rng = np.random.default_rng(N + hop)
    z = (rng.standard_normal((batch, frames, N)) + 1j * rng.standard_normal((batch, frames, N))).astype(np.complex64)
    h = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=16000, scaling=scaling)
    want = S.istft(S.spectrum_multiply(z, h), w, **opts)
    got = S.istft_filtered(z, h, w, **opts)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ctx = S.default_context()
    zd = ctx.to_device(z)
    gd = S.istft_filtered(zd, h, w, **opts)
    assert S.device.is_device(gd) and np.array_equal(gd.numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(zd.numpy().view(np.uint32), z.view(np.uint32))
    # and against the oracle's chain
    zf = (z.astype(np.complex128) * h.astype(np.complex128)).astype(np.complex64)
    yo = O.istft(zf, w, **opts)
    inner = slice(N, -N) if yo.shape[-1] > 3 * N else slice(None)
    assert nerr(got[..., inner], yo[..., inner])[0] < 2e-4
    with pytest.raises(S.ArgumentError):
        S.istft_filtered(z, h[: N // 2], w, **opts)


# ------------------------------------------------------------------------------- magnitude spectrogram (8f-2)
@pytest.mark.parametrize("K,N,hop,pad,scaling", [
    (1024, 1024, 256, "valid", None),
    (1024, 1024, 512, "reflect", "spectrum"),   # the guide's call: default 50 % overlap
    (512, 400, 160, "reflect", None),
    (256, 256, 64, "valid", "psd"),
    (128, 128, 32, "same", None),
    (2048, 2048, 512, "valid", None),
    (4096, 3000, 1000, "valid", None),
    (400, 400, 160, "valid", None),             # two-step path behind the same entry point
    (64, 64, 16, "valid", None),
])
def test_spectrogram_magnitude_power_dbfs(K, N, hop, pad, scaling):
    """guides/spectrogram.livemd:76-92: Nx.abs(s) of the first fft_length/2 bins, and 20*log(|s|/max|s|)/log(10)"""
    rng = np.random.default_rng(K + N)
    x = rng.standard_normal((2, 20000 + 5)).astype(np.float32)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K, sampling_rate=16000, window_padding=pad, scaling=scaling)
    zo, to, fo = O.stft(x, w, **opts)
    half = K // 2
    mag_ref = np.abs(zo[..., :half].astype(np.complex128)).astype(np.float32)
    mag, t, f = S.spectrogram(x, w, **opts)
    assert mag.shape == mag_ref.shape and mag.dtype == np.float32
    assert np.array_equal(t, to) and np.array_equal(f, fo[:half])
    scale = float(mag_ref.max())
    assert np.max(np.abs(mag - mag_ref)) / scale < 1e-5
    pw, _, _ = S.spectrogram(x, w, kind="power", **opts)
    assert np.max(np.abs(pw - mag_ref.astype(np.float64) ** 2)) / scale ** 2 < 1e-5
    db, _, _ = S.spectrogram(x, w, kind="dbfs", **opts)
    db_ref = 20.0 * np.log10(mag_ref.astype(np.float64) / scale)
    loud = db_ref > -80.0                      # below that the log amplifies fp32 round-off of near-zero bins
    assert np.max(np.abs(db[loud] - db_ref[loud])) < 1e-2
    assert float(db.max()) == 0.0
    ctx = S.default_context()
    md, _, _ = S.spectrogram(ctx.to_device(x), w, **opts)
    assert np.array_equal(md.numpy().view(np.uint32), mag.view(np.uint32))
    with pytest.raises(S.ArgumentError):
        S.spectrogram(x, w, kind="decibels", **opts)


# ------------------------------------------------------------------------------- fft_nd over several axes, correlate (8f-4)
def test_fft_nd_multi_axis_golden_and_oracle(golden):
    for v in golden["fft_nd"]:
        fn = S.transforms.ifft_nd if v["inverse"] else S.transforms.fft_nd
        kw = {"axes": v["axes"]}
        if v["lengths"] is not None:
            kw["lengths"] = v["lengths"]
        z = fn(np.array(v["x"]), **kw)
        a = np.array(v["expect"], dtype=np.float64)
        exp = (a[..., 0] + 1j * a[..., 1]).astype(np.complex64)
        assert z.shape == exp.shape and z.dtype == np.complex64
        assert nx_all_close(z, exp, max(v["atol"], 1e-6), max(v["rtol"], 1e-6)), (v["src"], z)
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((6, 20, 48)) + 1j * rng.standard_normal((6, 20, 48))).astype(np.complex64)
    got = S.transforms.fft_nd(a, axes=[1, 2, 0], lengths=[32, 64, None])
    assert_close(got, O.fft_nd(a, axes=[1, 2, 0], lengths=[32, 64, None]), "3-axis fold")
    back = S.transforms.ifft_nd(got, axes=[0, 2, 1])
    assert_close(back, O.fft_nd(got, axes=[0, 2, 1], inverse=True), "inverse fold")
    with pytest.raises(S.ArgumentError):
        S.transforms.fft_nd(a, axes=[3])


def test_correlate_fft_method(golden):
    for v in golden["correlate"]:
        got = S.convolution.correlate(np.array(v["a"], dtype=np.float32), np.array(v["b"], dtype=np.float32), method="fft")
        assert np.allclose(got, np.array(v["expect"], dtype=np.float32), rtol=1e-5, atol=1e-5), got
    rng = np.random.default_rng(8)
    x = rng.standard_normal(5000).astype(np.float32)
    k = rng.standard_normal(129).astype(np.float32)
    for mode in ("full", "same", "valid"):
        got = S.convolution.correlate(x, k, method="fft", mode=mode)
        full = O.direct_convolve_f64(x, k[::-1].copy())
        n = {"full": 5000 + 128, "same": 5000, "valid": 5000 - 128}[mode]
        start = (full.shape[0] - n) // 2 if mode != "full" else 0
        assert_close(got, full[start:start + n].astype(np.float32), f"correlate {mode}")
    a = (rng.standard_normal(40) + 1j * rng.standard_normal(40)).astype(np.complex64)
    b = (rng.standard_normal(7) + 1j * rng.standard_normal(7)).astype(np.complex64)
    assert_close(S.convolution.correlate(a, b, method="fft"), O.correlate(a, b), "complex correlate")


# ------------------------------------------------------------------------------- HIP graph capture of the hot path
def test_stft_istft_are_capturable_into_a_hip_graph():
    """After the first (table-building) call the device entry points only launch kernels on the context's stream: no
    allocation, no synchronisation, no host copies.  So a caller can capture them into a HIP graph on a stream handed
    over with nxsig_set_stream and replay the small-batch case (HIP runtime driven through ctypes: no torch needed)."""
    import ctypes as C

    from nx_signal_amd import _lib
    hip = C.CDLL("libamdhip64.so")

    def ok(rc, what):
        assert rc == 0, f"{what} -> hip error {rc}"

    N, hop = 1024, 256
    x = O.synth_signal(48000, seed=3)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    ctx = S.Context(0)
    xd = ctx.to_device(x)
    z_ref, _, _ = S.stft(xd, w, ctx=ctx, **opts)       # warm-up: builds and caches every table
    y_ref = S.istft(z_ref, w, ctx=ctx, **opts)
    z_ref_h, y_ref_h = z_ref.numpy(), y_ref.numpy()
    lib = _lib.load()
    M = z_ref.shape[0]
    zd = ctx.empty((M, N), np.complex64)
    yd = ctx.empty(y_ref.shape, np.complex64)
    p = _lib.StftParams(N, hop, N, _lib.PAD_VALID, 0, 0, _lib.SCALE_NONE, 0, 48000.0)
    wp = w.ctypes.data_as(C.c_void_p)
    stream, graph, gexec = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ok(hip.hipStreamCreate(C.byref(stream)), "hipStreamCreate")
    ctx.set_stream(stream.value)
    try:
        ok(hip.hipStreamBeginCapture(stream, 0), "hipStreamBeginCapture")  # hipStreamCaptureModeGlobal
        _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), 48000, 1, 48000, wp, C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))
        _lib.check(lib.nxsig_istft_c64(ctx.handle, C.c_void_p(zd.ptr), M, 1, wp, C.byref(p), C.c_void_p(yd.ptr), _lib.DEVICE))
        ok(hip.hipStreamEndCapture(stream, C.byref(graph)), "hipStreamEndCapture")
        ok(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")
        for _ in range(3):
            ok(hip.hipGraphLaunch(gexec, stream), "hipGraphLaunch")
        ok(hip.hipStreamSynchronize(stream), "hipStreamSynchronize")
    finally:
        ctx.set_stream(None)
        if gexec.value:
            hip.hipGraphExecDestroy(gexec)
        if graph.value:
            hip.hipGraphDestroy(graph)
        hip.hipStreamDestroy(stream)
    assert np.array_equal(zd.numpy().view(np.uint32), z_ref_h.view(np.uint32))
    assert np.array_equal(yd.numpy().view(np.uint32), y_ref_h.view(np.uint32))


def test_frame_packing_istft_fir_and_packed_pair_are_capturable_into_a_hip_graph():
    """the passes round 3 added behind these entry points are graph-capturable too: the non-finite unit list of the frame-packing
    inverse kernels is reset by a memset NODE (not a host copy), the FIR's row flags are consumed by its own poison pass, the
    packed pair is a kernel + the two fix-up passes.  Replays are bit-identical to the eager calls, and a replay on data that now
    holds a non-finite value takes the fix-up route (the list / the flags are device state, not captured host state)."""
    import ctypes as C

    from nx_signal_amd import _lib
    hip = C.CDLL("libamdhip64.so")

    def ok(rc, what):
        assert rc == 0, f"{what} -> hip error {rc}"

    N, hop, L = 512, 128, 40000
    x = O.synth_signal(L, seed=5)
    w = S.windows.hann(N)
    w1 = S.windows.hann(1024)
    h = S.filters.firwin(257, [0.2])
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    opts1 = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
    ctx = S.Context(0)
    xd = ctx.to_device(x)
    z_ref, _, _ = S.stft(xd, w, ctx=ctx, **opts)                   # warm-up calls: build and cache every table
    y_ref = S.istft(z_ref, w, ctx=ctx, **opts).numpy()
    f_ref = S.filters.fir(xd, h, ctx=ctx).numpy()
    zp_ref, _, _ = S.stft_packed(xd, w1, ctx=ctx, **opts1)
    yp_ref = S.istft_packed(zp_ref, w1, ctx=ctx, **opts1).numpy()
    lib = _lib.load()
    M, M1 = z_ref.shape[0], zp_ref.shape[0]
    zd = ctx.to_device(z_ref.numpy())
    yd = ctx.empty(y_ref.shape, np.complex64)
    fd = ctx.empty(f_ref.shape, np.float32)
    zpd = ctx.empty((M1, 512), np.complex64)
    ypd = ctx.empty(yp_ref.shape, np.float32)
    p = _lib.StftParams(N, hop, N, _lib.PAD_VALID, 0, 0, _lib.SCALE_NONE, 0, 48000.0)
    p1 = _lib.StftParams(1024, 256, 1024, _lib.PAD_VALID, 0, 0, _lib.SCALE_NONE, 0, 48000.0)
    wp, w1p, hp = w.ctypes.data_as(C.c_void_p), w1.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p)
    V = C.c_void_p
    stream, graph, gexec = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ok(hip.hipStreamCreate(C.byref(stream)), "hipStreamCreate")
    ctx.set_stream(stream.value)
    try:
        ok(hip.hipStreamBeginCapture(stream, 0), "hipStreamBeginCapture")
        _lib.check(lib.nxsig_istft_c64(ctx.handle, V(zd.ptr), M, 1, wp, C.byref(p), V(yd.ptr), _lib.DEVICE))
        _lib.check(lib.nxsig_fir_f32(ctx.handle, V(xd.ptr), L, 1, L, hp, 257, _lib.CONV_SAME, V(fd.ptr), _lib.DEVICE))
        _lib.check(lib.nxsig_stft_packed_f32(ctx.handle, V(xd.ptr), L, 1, L, w1p, C.byref(p1), V(zpd.ptr), None, _lib.DEVICE))
        _lib.check(lib.nxsig_istft_packed_f32(ctx.handle, V(zpd.ptr), M1, 1, w1p, C.byref(p1), V(ypd.ptr), _lib.DEVICE))
        ok(hip.hipStreamEndCapture(stream, C.byref(graph)), "hipStreamEndCapture")
        ok(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")
        for _ in range(2):
            ok(hip.hipGraphLaunch(gexec, stream), "hipGraphLaunch")
        ok(hip.hipStreamSynchronize(stream), "hipStreamSynchronize")
        assert np.array_equal(yd.numpy().view(np.uint32), y_ref.view(np.uint32))
        assert np.array_equal(fd.numpy().view(np.uint32), f_ref.view(np.uint32))
        assert np.array_equal(zpd.numpy().view(np.uint32), zp_ref.numpy().view(np.uint32))
        assert np.array_equal(ypd.numpy().view(np.uint32), yp_ref.view(np.uint32))
        # the same graph on data with a non-finite value: the device-side list / flags route it through the fix-up passes
        zbad = z_ref.numpy().copy()
        zbad[40, 7] = np.inf
        _lib.check(lib.nxsig_upload(ctx.handle, V(zd.ptr), zbad.ctypes.data_as(V), zbad.nbytes))
        xbad = x.copy()
        xbad[20000] = np.nan
        _lib.check(lib.nxsig_upload(ctx.handle, V(xd.ptr), xbad.ctypes.data_as(V), xbad.nbytes))
        ok(hip.hipGraphLaunch(gexec, stream), "hipGraphLaunch")
        ok(hip.hipStreamSynchronize(stream), "hipStreamSynchronize")
        yo = O.istft(zbad, w, **opts)
        assert np.array_equal(np.isfinite(yd.numpy()), np.isfinite(yo))
        assert not np.isfinite(fd.numpy()).any()
        zo1, _, _ = O.stft(xbad, w1, **opts1)
        assert np.array_equal(np.isfinite(zpd.numpy()).all(axis=-1), np.isfinite(zo1).all(axis=-1))
    finally:
        ctx.set_stream(None)
        if gexec.value:
            hip.hipGraphExecDestroy(gexec)
        if graph.value:
            hip.hipGraphDestroy(graph)
        hip.hipStreamDestroy(stream)


# ------------------------------------------------------------------------------- concurrency (dirty-scheduler threads)
def test_concurrent_calls_from_several_threads():
    """NIF dirty schedulers are arbitrary OS threads: calls on one shared context (serialised by its mutex) and on
    per-thread contexts must be re-entrant and give the single-threaded result (ctypes releases the GIL)."""
    import threading

    N, hop = 1024, 256
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    shared = S.default_context()
    sigs = [O.synth_signal(30000 + 1000 * i, seed=50 + i) for i in range(6)]
    refs = [S.stft(x, w, ctx=shared, **opts)[0] for x in sigs]
    errors = []

    def worker(i, own_ctx):
        try:
            c = S.Context(0) if own_ctx else shared
            for _ in range(10):
                z, _, _ = S.stft(sigs[i], w, ctx=c, **opts)
                if not np.array_equal(z.view(np.uint32), refs[i].view(np.uint32)):
                    errors.append((i, own_ctx, "mismatch"))
                y = S.istft(z, w, ctx=c, **opts)
                if not np.all(np.isfinite(y.view(np.float32))):
                    errors.append((i, own_ctx, "non-finite"))
        except Exception as e:  # noqa: BLE001
            errors.append((i, own_ctx, repr(e)))

    threads = [threading.Thread(target=worker, args=(i, i % 2 == 0)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_stft_config1_against_committed_fixture():
    """HIP stft / istft on BASELINE config 1 against tests/golden/oracle_c1_fixture.npz (committed data, no live oracle)"""
    import os

    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_c1_fixture.npz"))
    x = O.synth_signal(48000, seed=1234)
    assert np.array_equal(x[:2048], fx["x_head"])
    w = S.windows.hann(1024)
    opts = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
    z, t, f = S.stft(x, w, **opts)
    assert z.shape == tuple(fx["shape_z"])
    assert np.array_equal(t, fx["times"]) and np.array_equal(f, fx["freqs"])
    scale = float(np.max(np.abs(fx["z_frames"])))
    assert np.max(np.abs(z[fx["frames"]] - fx["z_frames"])) / scale < 1e-5
    assert np.max(np.abs(z.sum(axis=0) - fx["z_colsum"])) / scale < 2e-4     # 184-term sums of values with 1e-7 error
    y = S.istft(z, w, **opts)
    ys = float(np.max(np.abs(fx["y_head"])))
    # the first / last hop of an iSTFT under a Hann window divide by a vanishing normaliser: 1e-7 differences between the
    # two spectra are amplified there (see the conditioning note in test_istft_config3_*), so compare past the first hop
    assert np.max(np.abs(y[512:1536] - fx["y_head"][512:])) / ys < 1e-5
    assert np.max(np.abs(y[-1536:-512] - fx["y_tail"][:-512])) / ys < 1e-5


def test_eps_clean_up_is_applied():
    """Nx.fft zeroes every component with |x| <= 1e-10 (SURVEY App. A rule 7) and so do the kernels (round 3; the detailed
    tests live in tests/test_gpu_reference_numerics.py): on a tiny-amplitude signal whose components sit around the threshold
    the zero / non-zero decision matches the oracle except within fp32 round-off of the threshold, and on a full-scale signal
    with mathematically zero bins (where fp32 round-off ~1e-7 of the spectrum's scale keeps them non-zero) the usual
    normalised tolerance holds."""
    w = S.windows.hann(1024)
    opts = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
    n = np.arange(1024 * 6, dtype=np.float64)
    x = (O.synth_signal(1024 * 6, seed=3) * np.float32(2e-12)).astype(np.float32)
    z, _, _ = S.stft(x, w, **opts)
    zo, _, _ = O.stft(x, w, **opts)
    g, r = z.view(np.float32).ravel(), zo.astype(np.complex64).view(np.float32).ravel()
    assert (r == 0).sum() > r.size // 2
    wrong = (g == 0) != (r == 0)
    assert wrong.sum() <= 4 and np.all(np.abs(np.abs(g[wrong].astype(np.float64)) - 1e-10) < 2e-15)
    x = np.cos(2 * np.pi * 8 * n / 1024).astype(np.float32)  # bin-centred tone: most bins are ~0 in the reference
    z, _, _ = S.stft(x, w, **opts)
    zo, _, _ = O.stft(x, w, **opts)
    assert float(np.max(np.abs(z - zo))) <= 1e-5 * float(np.max(np.abs(zo)))
