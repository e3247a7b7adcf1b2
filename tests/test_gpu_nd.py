"""Device-side multi-axis transforms, n-D fftconvolve and row lengths beyond the LDS-resident kernels (SURVEY §8 a13 / a14,
§8f-4; csrc/kernels_nd.hip): NxSignal.Transforms.fft_nd / ifft_nd over any axes (lib/nx_signal/transforms.ex:5-21) and
NxSignal.Convolution.fftconvolve/3 for operands of equal rank (lib/nx_signal/convolution.ex:252-347) against the oracle
and the reference's own test literals; four-step rows (powers of two > 8192) and Bluestein rows (other lengths > 4096)
against an f64 FFT; stft / istft with such fft_lengths against the oracle.  Tolerance: normalised max error < 1e-5."""
import numpy as np
import pytest

from oracle import nx_oracle as O

import nx_signal_amd as S

pytestmark = pytest.mark.gpu


def nerr(got, ref):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.max(np.abs(got.astype(np.complex128) - ref.astype(np.complex128))) / max(float(np.max(np.abs(ref))), 1e-30))


def crandn(rng, *shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


@pytest.mark.parametrize("shape,axes,lengths", [
    ((5, 7, 16), [0], [None]), ((5, 7, 16), [1], [9]), ((5, 7, 16), [1, 2], [None, 32]), ((5, 7, 16), [0, 2], [8, 5]),
    ((5, 7, 16), [2, 0, 1], [16, 5, 7]), ((3, 100, 65), [1], [128]), ((130, 70), [0], [130]), ((130, 70), [0, 1], [256, 100]),
    ((2, 3, 4, 5), [1, 3], [3, 8]), ((64, 64), [-2, -1], [None, None]), ((1, 9), [0], [4]), ((300,), [0], [512]),
    # power-of-two transforms along a slower axis whose inner size is a multiple of 8: the one-pass column kernel
    ((3, 100, 64), [1], [128]), ((2, 300, 24), [1], [256]), ((40, 16), [0], [16]), ((1000, 8), [0], [1024]), ((2, 512, 40), [1, 2], [512, 64]),
    ((4, 33, 128), [1], [32]), ((2, 16, 8, 8), [1], [None]), ((3, 64, 72), [0, 1], [4, 64]),
])
def test_fft_nd_any_axes_host_and_device(shape, axes, lengths):
    rng = np.random.default_rng(11)
    for x in (rng.standard_normal(shape).astype(np.float32), crandn(rng, *shape)):
        for inverse in (False, True):
            fn = S.transforms.ifft_nd if inverse else S.transforms.fft_nd
            want = O.fft_nd(x, axes=axes, lengths=lengths, inverse=inverse)
            got = fn(x, axes=axes, lengths=lengths)
            assert got.dtype == np.complex64 and nerr(got, want) < 1e-5, (shape, axes, lengths, inverse)
            dgot = fn(S.default_context().to_device(x), axes=axes, lengths=lengths)  # stays in HBM
            assert isinstance(dgot, S.DeviceBuffer) and np.array_equal(dgot.numpy().view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("K", [16384, 32768, 65536, 1 << 17, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23, 5000, 12000, 10007, 16385, 100003])
def test_long_rows_four_step_and_bluestein(K):
    rng = np.random.default_rng(K)
    rows = 3 if K <= 70000 else 1
    x = crandn(rng, rows, K - 5)  # zero-padded to K
    r = rng.standard_normal((rows, K + 7)).astype(np.float32)  # truncated to K
    for inverse in (False, True):
        fn = S.transforms.ifft_nd if inverse else S.transforms.fft_nd
        npf = np.fft.ifft if inverse else np.fft.fft
        assert nerr(fn(x, lengths=[K]), npf(x.astype(np.complex128), n=K, axis=-1)) < 1e-5, (K, inverse)
        assert nerr(fn(r, lengths=[K]), npf(r.astype(np.float64), n=K, axis=-1)) < 1e-5, (K, inverse, "real")


@pytest.mark.parametrize("N,K,hop", [(16384, 16384, 4096), (12000, 12000, 3000), (9000, 16384, 4500), (5000, 5000, 1250)])
def test_stft_istft_with_long_transforms(N, K, hop):
    x = O.synth_signal(N + hop * 6 + 17, seed=N)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K, sampling_rate=48000)
    for scaling in (None, "spectrum"):
        z, t, f = S.stft(x, w, scaling=scaling, **opts)
        zo, to, fo = O.stft(x, w, scaling=scaling, **opts)
        assert nerr(z, zo) < 1e-5 and np.array_equal(t, to) and np.array_equal(f, fo)
    if N == K:
        zo, _, _ = O.stft(x, w, **opts)
        y = S.istft(zo, w, **opts)
        yo = O.istft(zo, w, **opts)
        assert nerr(y, yo) < 1e-5


def test_fftconvolve_nd_reference_literals(golden):
    for v in golden["fftconvolve_nd"]:
        def arr(x):
            x = np.array(x)
            return (x[..., 0] + 1j * x[..., 1]).astype(np.complex64) if v.get("complex") else x
        a, b, e = arr(v["a"]), arr(v["b"]), arr(v["expect"])
        for aa, bb in ([(a, b), (b, a)] if v.get("swap_too") else [(a, b)]):
            out = S.convolution.convolve(aa, bb, method="fft", mode=v["mode"])
            assert out.shape == e.shape and out.dtype == (np.complex64 if v.get("complex") else np.float32), v["src"]
            assert np.all(np.abs(out - e) <= 1e-4 + 1e-4 * np.abs(e)), v["src"]  # assert_all_close of the reference


@pytest.mark.parametrize("s1,s2", [((12, 17), (5, 4)), ((5, 4), (12, 17)), ((6, 1, 9), (3, 7, 2)), ((4, 5, 6), (4, 5, 6)),
                                   ((40, 300), (7, 31)), ((2, 3, 4, 5), (2, 1, 3, 2)), ((1, 50), (1, 8)), ((33,), (5,))])
@pytest.mark.parametrize("mode", ["full", "same", "valid"])
def test_fftconvolve_nd_matches_oracle(s1, s2, mode):
    rng = np.random.default_rng(sum(s1) + 3 * sum(s2))
    ok1 = all(x >= y for x, y in zip(s1, s2))
    ok2 = all(y >= x for x, y in zip(s1, s2))
    for cplx in (False, True):
        a = crandn(rng, *s1) if cplx else rng.standard_normal(s1).astype(np.float32)
        b = rng.standard_normal(s2).astype(np.float32)
        if mode == "valid" and not (ok1 or ok2):
            with pytest.raises(S.ArgumentError, match="valid"):
                S.convolution.fftconvolve(a, b, mode=mode)
            continue
        if len(s1) == 1 and not cplx:
            continue  # 1-D real runs the overlap-save kernel (covered elsewhere)
        got = S.convolution.fftconvolve(a, b, mode=mode)
        want = O.fftconvolve(a, b, mode=mode)
        assert got.dtype == want.dtype and nerr(got, want) < 1e-5, (s1, s2, mode, cplx)


def test_fftconvolve_rank_mismatch_and_long_complex():
    with pytest.raises(S.ArgumentError, match="Rank of in1 and in2 must be equal"):
        S.convolution.fftconvolve(np.ones((3, 3), np.float32), np.ones((2, 2, 2), np.float32))
    rng = np.random.default_rng(5)
    a, b = crandn(rng, 20000), crandn(rng, 3000)  # n1 + n2 - 1 = 22 999 > 8192: four-step transforms
    for mode in ("full", "same", "valid"):
        got = S.convolution.fftconvolve(a, b, mode=mode)
        ref = np.convolve(a.astype(np.complex128), b.astype(np.complex128), mode=mode)
        assert nerr(got, ref) < 1e-5


def test_correlate_2d():
    rng = np.random.default_rng(2)
    a, k = rng.standard_normal((9, 11)).astype(np.float32), rng.standard_normal((3, 4)).astype(np.float32)
    got = S.convolution.correlate(a, k, method="fft", mode="same")
    assert nerr(got, O.correlate(a, k, mode="same")) < 1e-5


@pytest.mark.parametrize("K", [1024, 2048, 4096])
def test_wave_core_row_kernels(K):
    """kernels_wave_rows.hip: K = 1024 / 2048 (one core pass) and 4096 (four passes + radix-4) rows, f32 and c64 inputs,
    zero-padded (n_in < K), exact (the 16-byte fast path of the 4096 kernel) and truncated (n_in > K) rows, both directions,
    odd row counts (partial workgroups)"""
    rng = np.random.default_rng(K)
    for n_in in (K - 5, K, K + 7, 3):
        for rows in (1, 37):
            xc = crandn(rng, rows, n_in)
            xr = rng.standard_normal((rows, n_in)).astype(np.float32)
            for inverse in (False, True):
                fn = S.transforms.ifft_nd if inverse else S.transforms.fft_nd
                npf = np.fft.ifft if inverse else np.fft.fft
                assert nerr(fn(xc, lengths=[K]), npf(xc.astype(np.complex128), n=K, axis=-1)) < 1e-5, (K, n_in, rows, inverse)
                assert nerr(fn(xr, lengths=[K]), npf(xr.astype(np.float64), n=K, axis=-1)) < 1e-5, (K, n_in, rows, inverse, "real")


@pytest.mark.parametrize("N,hop", [(2048, 300), (4096, 1000), (1024, 100)])
def test_generic_istft_rides_the_wave_rows(N, hop):
    """hops the fused istft kernels do not take: rows IFFT (wave row kernels with the x scale x window epilogue) + deterministic OLA"""
    rng = np.random.default_rng(N + hop)
    M = 9
    z = crandn(rng, 2, M, N)
    w = S.windows.hann(N)
    for scaling in (None, "spectrum"):
        opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000, scaling=scaling)
        y = S.istft(z, w, **opts)
        for b in range(2):
            assert nerr(y[b], O.istft(z[b], w, **opts)) < 1e-5, (N, hop, scaling, b)


def test_fft_nd_long_inner_axis():
    """an axis-0 transform of a short, very wide tensor: 70 000 tile columns (beyond any 65 535 grid-dimension limit)"""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((4, 70000 * 64 // 64 * 1)).astype(np.float32)[:, : 64 * 1100]  # 4 x 70 400
    got = S.transforms.fft_nd(x, axes=[0], lengths=[8])
    assert nerr(got, np.fft.fft(x.astype(np.float64), n=8, axis=0)) < 1e-5
    wide = rng.standard_normal((2, 4300000)).astype(np.float32)  # inner axis of 4.3 M elements: > 65 535 tiles of 64 rows
    got = S.transforms.fft_nd(wide, axes=[0])
    assert nerr(got, np.fft.fft(wide.astype(np.float64), axis=0)) < 1e-5


# ------------------------------------------------------------------------------- convolve(method: :direct), the reference's default
def _gop(v, key):
    cplx = v.get("complex") and (v.get(key + "_complex") or not (v.get("a_complex") or v.get("b_complex")))
    x = np.array(v[key], dtype=np.float64)
    return (x[..., 0] + 1j * x[..., 1]).astype(np.complex64) if cplx else x.astype(np.float32)


def test_convolve_direct_reference_literals(golden):
    """every convolve/3 literal of test/nx_signal/convolutions_test.exs through the HIP kernel, compared with == like there"""
    for v in golden["convolve_direct"]:
        e = np.array(v["expect"], dtype=np.float64)
        exp = (e[..., 0] + 1j * e[..., 1]).astype(np.complex64) if v.get("complex") else e.astype(np.float32)
        got = S.convolution.convolve(_gop(v, "a"), _gop(v, "b"), mode=v["mode"])  # default method
        assert got.shape == exp.shape and got.dtype == exp.dtype and np.array_equal(got, exp), (v["src"], got)
    for v in golden["convolve_direct_doctest"]:
        assert S.convolution.convolve(np.array(v["a"]), np.array(v["b"])).tolist() == v["expect"]
    for v in golden["correlate"] + golden["correlate_direct"]:
        got = S.convolution.correlate(np.array(v["a"], np.float32), np.array(v["b"], np.float32), mode=v.get("mode", "full"))
        assert np.array_equal(got, np.array(v["expect"], np.float32)), (v["src"], got)
    e = golden["convolve_direct_errors"]
    with pytest.raises(S.ArgumentError, match="For :valid mode"):
        S.convolution.convolve(np.ones(e[0]["a_shape"], np.float32), np.ones(e[0]["b_shape"], np.float32), mode="valid")
    for r1, r2 in e[1]["ranks"]:
        with pytest.raises(S.ArgumentError):
            S.convolution.convolve(np.ones((1,) * r1, np.float32), np.ones((1,) * r2, np.float32))


@pytest.mark.parametrize("s1,s2,mode,cplx", [
    ((5000,), (129,), "full", False), ((5000,), (129,), "same", False), ((5000,), (128,), "same", False), ((5000,), (129,), "valid", False),
    ((129,), (5000,), "valid", False), ((40,), (300,), "same", False), ((64,), (64,), "full", True), ((257,), (31,), "same", True),
    ((37, 53), (5, 7), "full", False), ((37, 53), (6, 4), "same", False), ((5, 7), (37, 53), "valid", True),
    ((9, 10, 11), (3, 4, 2), "same", False), ((3, 4, 5, 6), (2, 2, 3, 1), "full", True), ((1, 200), (1, 17), "valid", False),
])
def test_convolve_direct_matches_oracle_bit_for_bit(s1, s2, mode, cplx):
    """same accumulation order as the oracle's restatement of the BinaryBackend (double, row-major over the window): identical bits;
    and the direct and FFT methods agree to FFT round-off (the reference's own cross-check, convolutions_test.exs:392-416)"""
    rng = np.random.default_rng(sum(s1) + 7 * sum(s2))
    a = rng.standard_normal(s1).astype(np.float32)
    b = rng.standard_normal(s2).astype(np.float32)
    if cplx:
        a = (a + 1j * rng.standard_normal(s1)).astype(np.complex64)
    got = S.convolution.convolve(a, b, mode=mode, method="direct")
    exp = O.convolve_direct(a, b, mode=mode)
    assert got.shape == exp.shape and got.dtype == exp.dtype
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    viaf = S.convolution.convolve(a, b, mode=mode, method="fft")
    assert viaf.shape == got.shape and nerr(viaf, got) < 1e-5


@pytest.mark.parametrize("seed", range(48))
def test_convolve_direct_register_window_kernel_fuzz(seed):
    """the register-window kernels (kernel spanning the last two axes at most; real and complex operands): random ranks 1-4, sizes
    around the 8- / 4-wide blocks and 8-tap chunks, kernels longer than the rows, every mode — bit for bit against the oracle"""
    rng = np.random.default_rng(1000 + seed)
    rank = int(rng.integers(1, 5))
    lead = [int(rng.integers(1, 4)) for _ in range(max(0, rank - 2))]
    if rank == 1:
        sa, sb = [int(rng.integers(1, 400))], [int(rng.integers(1, 70))]
    else:
        sa = lead + [int(rng.integers(1, 40)), int(rng.integers(1, 90))]
        sb = [1] * len(lead) + [int(rng.integers(1, 12)), int(rng.integers(1, 40))]
    if seed % 5 == 0:
        sa, sb = sb, sa                    # the kernel is the larger operand
    mode = ["full", "same", "valid"][seed % 3]
    if mode == "valid" and not (all(x >= y for x, y in zip(sa, sb)) or all(x <= y for x, y in zip(sa, sb))):
        mode = "full"
    a = rng.standard_normal(sa).astype(np.float32)
    b = rng.standard_normal(sb).astype(np.float32)
    kind = (seed // 3) % 4                 # real x real (8 outputs per thread), complex x real, real x complex, complex x complex (4)
    if kind in (1, 3):
        a = (a + 1j * rng.standard_normal(sa)).astype(np.complex64)
    if kind in (2, 3):
        b = (b + 1j * rng.standard_normal(sb)).astype(np.complex64)
    got = S.convolution.convolve(a, b, mode=mode, method="direct")
    exp = O.convolve_direct(a, b, mode=mode)
    assert got.shape == exp.shape and got.dtype == (np.float32 if kind == 0 else np.complex64), (sa, sb, mode, kind)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (sa, sb, mode, kind)


def test_convolve_direct_long_stream_config5_slice():
    """the 257-tap low-pass of config 5 on a 200 000-sample slice: the time-domain method against direct f64 convolution"""
    x = O.synth_signal(200000, seed=5)
    h = S.filters.firwin(257, [4000.0], sampling_rate=48000)
    got = S.convolution.convolve(x, h, mode="same")
    full = O.direct_convolve_f64(x, h)
    assert np.array_equal(got, full[128:128 + 200000].astype(np.float32)) or nerr(got, full[128:128 + 200000].astype(np.float32)) < 1e-6
