"""Pins oracle/nx_oracle.py against every vector the reference's own doctests/tests hold for the
STFT / iSTFT / windows / FIR path (tests/golden/reference_vectors.json, transcribed values).
Bit-exact wherever the reference prints f32 values; the reference's own tolerance otherwise."""
import numpy as np
import pytest

from conftest import f32_list, nx_all_close
from oracle import nx_oracle as O


def _bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_exact(got, expect_strs, what):
    exp = f32_list(expect_strs)
    got = np.asarray(got, dtype=np.float32)
    assert got.shape == exp.shape, what
    # -0.0 vs 0.0: the reference prints "0.0" for +0 and "-0.0" for -0; compare bits
    assert np.array_equal(_bits(got), _bits(exp)), f"{what}: got {got.tolist()} expected {exp.tolist()}"


WINDOW_FNS = {
    "bartlett": O.bartlett,
    "triangular": O.triangular,
    "blackman": O.blackman,
    "hamming": O.hamming,
    "hann": O.hann,
    "kaiser": O.kaiser,
}


def test_windows_bit_exact(golden):
    for v in golden["windows"]:
        if v["fn"] == "rectangular":
            w = O.rectangular(v["n"])
            assert w.dtype == np.int64 and w.tolist() == v["expect"]
            continue
        w = WINDOW_FNS[v["fn"]](v["n"], **v["opts"])
        assert w.dtype == np.float32
        assert_bit_exact(w, v["expect"], f"{v['fn']}({v['n']}, {v['opts']}) @ {v['src']}")


def test_sinc_bit_exact(golden):
    for v in golden["sinc"]:
        assert_bit_exact(O.sinc(f32_list(v["t"])), v["expect"], v["src"])


def test_fft_frequencies_bit_exact(golden):
    for v in golden["fft_frequencies"]:
        assert_bit_exact(O.fft_frequencies(v["sampling_rate"], v["fft_length"]), v["expect"], v["src"])


def test_as_windowed(golden):
    for v in golden["as_windowed"]:
        pad = v["padding"]
        if isinstance(pad, list):
            pad = [tuple(p) for p in pad]
        got = O.as_windowed(np.array(v["x"]), v["window_length"], v["stride"], pad)
        assert got.tolist() == v["expect"], v["src"]


def test_overlap_and_add(golden):
    for v in golden["overlap_and_add"]:
        if v.get("iota"):
            t = np.arange(np.prod(v["shape"])).reshape(v["shape"])
        else:
            t = np.array(v["t"])
        got = O.overlap_and_add(t, v["overlap_length"])
        assert got.tolist() == v["expect"], v["src"]


def _window(spec):
    if spec["fn"] == "rectangular":
        return O.rectangular(spec["n"])
    return WINDOW_FNS[spec["fn"]](spec["n"])


def test_stft_doctest_bit_exact(golden):
    for v in golden["stft"]:
        z, t, f = O.stft(np.array(v["x"]), _window(v["window"]), **v["opts"])
        exp = np.array([[[np.float32(c[0]), np.float32(c[1])] for c in row] for row in v["z"]], dtype=np.float32)
        got = np.stack([z.real, z.imag], axis=-1).astype(np.float32)
        assert np.array_equal(got, exp), v["src"]
        assert_bit_exact(t, v["t"], v["src"] + " times")
        assert_bit_exact(f, v["f"], v["src"] + " freqs")


def test_stft_istft_roundtrip_doctests(golden):
    for v in golden["stft_istft_roundtrip"]:
        x = np.array(v["x"], dtype=np.float32 if v["as_type"] == "f32" else np.int32)
        w = _window(v["window"])
        z, _, _ = O.stft(x, w, **v["opts"])
        y = O.istft(z, w, **v["opts"])
        assert y.dtype == np.complex64 and y.shape == (8,)
        if v["as_type"] == "s32":
            # Nx.as_type(c64 -> s32): real part truncated toward zero (Appendix A rule 10)
            got = np.trunc(y.real).astype(np.int32)
            assert got.tolist() == v["expect"], v["src"]
        else:
            assert_bit_exact(y.real, v["expect"], v["src"])


def test_mel_filters_bit_exact(golden):
    for v in golden["mel_filters"]:
        got = O.mel_filters(v["fft_length"], v["mel_bins"], v["sampling_rate"])
        for r, row in enumerate(v["expect"]):
            assert_bit_exact(got[r], row, f"{v['src']} row {r}")


def test_stft_reflect_k16_via_stft_to_mel(golden):
    """pins stft at fft_length 16 > frame 4 (zero-pad) with :reflect padding through the mel doctest"""
    for v in golden["stft_to_mel"]:
        x = np.arange(v["x_iota"])
        z, _, _ = O.stft(x, _window(v["window"]), **v["opts"])
        assert z.shape == (v["frames"], v["frequencies"])
        mel = O.stft_to_mel(z, v["opts"]["sampling_rate"], v["opts"]["fft_length"], mel_bins=v["mel_bins"])
        exp = np.array([f32_list(row) for row in v["expect"]])
        # mel bins 0..2 (18 values) are reproduced bit-for-bit, which pins the oracle's stft at K=16/:reflect.
        # Bin 3 of three frames lands 1 ulp off in the log10 domain (amplified 4x by (x+4)/4): that is inside
        # mel_filters/log post-processing (SURVEY 8f-1 "next" row, not the hot path) and stays a known gap.
        assert np.array_equal(_bits(mel[:, :3]), _bits(exp[:, :3])), v["src"]
        assert int((_bits(mel) == _bits(exp)).sum()) >= 21
        assert np.max(np.abs(mel - exp)) < 1e-7


def test_fftconvolve(golden):
    for v in golden["fftconvolve"]:
        got = O.fftconvolve(np.array(v["a"]), np.array(v["b"]), mode=v["mode"])
        assert got.dtype == np.float32
        if v["exact"]:
            assert_bit_exact(got, v["expect"], v["src"])
        else:
            assert nx_all_close(got, np.array(v["expect"]), v["atol"], v["rtol"]), v["src"]
    for v in golden["fftconvolve_complex"]:
        a = np.array([complex(*c) for c in v["a"]])
        b = np.array([complex(*c) for c in v["b"]])
        got = O.fftconvolve(a, b, mode=v["mode"])
        assert got.dtype == np.complex64
        assert nx_all_close(got, np.array([complex(*c) for c in v["expect"]]), v["atol"], v["rtol"]), v["src"]


def test_direct_vs_fft_crosscheck(golden):
    """the reference's direct-vs-FFT cross-check pattern"""
    for v in golden["convolve_direct_doctest"]:
        d = O.direct_convolve_f64(v["a"], v["b"])
        assert d.tolist() == v["expect"]
        assert nx_all_close(O.fftconvolve(np.array(v["a"]), np.array(v["b"])), d)


def test_fft_rows(golden):
    for v in golden["fft_rows"]:
        z = O.fft(np.array(v["x"]), length=v["length"])
        s = z[0] + z[1]
        d = z[0] - z[1]
        assert nx_all_close(s, np.array([complex(*c) for c in v["expect_sum"]]), v["atol"], v["rtol"])
        assert nx_all_close(d, np.array([complex(*c) for c in v["expect_diff"]]), v["atol"], v["rtol"])


def _cplx(nested):
    a = np.array(nested, dtype=np.float64)
    return (a[..., 0] + 1j * a[..., 1]).astype(np.complex64)


def test_fft_nd_all_axes(golden):
    """Transforms.fft_nd / ifft_nd over two axes: test/nx_signal/convolutions_test.exs:65-93, transforms_test.exs:12-41"""
    for v in golden["fft_nd"]:
        z = O.fft_nd(np.array(v["x"]), axes=v["axes"], lengths=v["lengths"], inverse=v["inverse"])
        exp = _cplx(v["expect"])
        assert z.shape == exp.shape and z.dtype == np.complex64, v["src"]
        if v["atol"] == 0.0:
            assert np.array_equal(z, exp), (v["src"], z)
        else:
            assert nx_all_close(z, exp, v["atol"], v["rtol"]), (v["src"], z)


def _golden_operand(v, key):
    cplx = v.get("complex") and (v.get(key + "_complex") or not (v.get("a_complex") or v.get("b_complex")))
    x = np.array(v[key], dtype=np.float64)
    return (x[..., 0] + 1j * x[..., 1]).astype(np.complex64) if cplx else x.astype(np.float32)


def _golden_expect(v):
    e = np.array(v["expect"], dtype=np.float64)
    return (e[..., 0] + 1j * e[..., 1]).astype(np.complex64) if v.get("complex") else e.astype(np.float32)


def test_convolve_direct_reference_literals(golden):
    """convolve/3 with its default method (:direct): every literal of test/nx_signal/convolutions_test.exs, compared with ==
    like the reference does (integer-valued data: the double accumulation is exact)"""
    for v in golden["convolve_direct"]:
        got = O.convolve_direct(_golden_operand(v, "a"), _golden_operand(v, "b"), mode=v["mode"])
        exp = _golden_expect(v)
        assert got.shape == exp.shape and got.dtype == exp.dtype and np.array_equal(got, exp), (v["src"], got)
    e = golden["convolve_direct_errors"]
    with pytest.raises(ValueError):
        O.convolve_direct(np.ones(e[0]["a_shape"], np.float32), np.ones(e[0]["b_shape"], np.float32), mode="valid")
    for r1, r2 in e[1]["ranks"]:
        with pytest.raises(ValueError):
            O.convolve_direct(np.ones((1,) * r1, np.float32), np.ones((1,) * r2, np.float32))
    for v in golden["correlate_direct"]:
        got = O.correlate(np.array(v["a"], np.float32), np.array(v["b"], np.float32), mode=v["mode"], method="direct")
        assert np.array_equal(got, np.array(v["expect"], np.float32)), (v["src"], got)


def test_correlate_doctest(golden):
    for v in golden["correlate"]:
        got = O.correlate(np.array(v["a"], dtype=np.float32), np.array(v["b"], dtype=np.float32))
        assert np.allclose(got, np.array(v["expect"], dtype=np.float32), rtol=1e-6, atol=1e-6), got


def test_firwin_scipy_vectors(golden):
    for v in golden["firwin"]:
        opts = dict(v["opts"])
        if isinstance(opts.get("window"), list):
            opts["window"] = tuple(opts["window"])
        h = O.firwin(v["num_taps"], v["cutoff"], **opts)
        assert h.dtype == np.float32
        assert nx_all_close(h, np.array(v["expect"]), atol=v["atol"], rtol=1e-4), (v["src"], h.tolist())


def test_firwin_errors(golden):
    for v in golden["firwin_errors"]:
        with pytest.raises(ValueError, match=v["match"]):
            O.firwin(v["num_taps"], v["cutoff"], **v["opts"])
    with pytest.raises(ValueError, match="cutoff must be a list"):
        O.firwin(5, 0.3)


def test_oracle_config1_regression_fixture():
    """tests/golden/oracle_c1_fixture.npz (made by tests/golden/make_oracle_fixtures.py) pins the oracle itself on
    BASELINE config 1: the reference cannot run here, so this guards the checker against silent drift."""
    import os

    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_c1_fixture.npz"))
    x = O.synth_signal(48000, seed=1234)
    assert np.array_equal(x[:2048], fx["x_head"])
    w = O.hann(1024)
    z, t, f = O.stft(x, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert tuple(fx["shape_z"]) == z.shape == (184, 1024)
    assert np.array_equal(t, fx["times"]) and np.array_equal(f, fx["freqs"])
    scale = float(np.max(np.abs(z)))
    assert np.max(np.abs(z[fx["frames"]] - fx["z_frames"])) / scale < 1e-7
    assert np.max(np.abs(z.sum(axis=0).astype(np.complex64) - fx["z_colsum"])) / scale < 1e-5
    assert np.allclose(np.abs(z).sum(axis=1).astype(np.float32), fx["z_abs_rowsum"], rtol=1e-6)
    y = O.istft(z, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert tuple(fx["shape_y"]) == y.shape
    ys = float(np.max(np.abs(y)))
    assert np.max(np.abs(y[:1536] - fx["y_head"])) / ys < 1e-6 and np.max(np.abs(y[-1536:] - fx["y_tail"])) / ys < 1e-6


def test_fftconvolve_nd_golden(golden):
    """n-D FFT-method convolution of the oracle against the reference's own test literals (convolutions_test.exs:95-142,
    :152-162, :444-453, :489-530), compared like the reference's assert_all_close (atol = rtol = 1e-4)"""
    for v in golden["fftconvolve_nd"]:
        def arr(x):
            x = np.array(x)
            return (x[..., 0] + 1j * x[..., 1]).astype(np.complex64) if v.get("complex") else x
        a, b, e = arr(v["a"]), arr(v["b"]), arr(v["expect"])
        for aa, bb in ([(a, b), (b, a)] if v.get("swap_too") else [(a, b)]):
            out = O.fftconvolve(aa, bb, mode=v["mode"])
            assert out.shape == e.shape, v["src"]
            assert out.dtype == (np.complex64 if v.get("complex") else np.float32)
            assert np.all(np.abs(out - e) <= 1e-4 + 1e-4 * np.abs(e)), v["src"]


@pytest.mark.parametrize("n", [1024, 2048])
def test_oracle_fft_pinned_at_benchmark_size_by_a_long_double_dft(n):
    """The reference's own vectors stop at length 16, so at the sizes the 1e-5 claim is made for the oracle's
    "transform in double, clean, round once to c64" (App. A rule 7) rested on numpy's pocketfft alone.  This pins it independently: a
    direct O(N^2) DFT of 8 Hann-windowed frames in 80-bit long double (exact integer angle reduction, long-double pi), rounded once to
    f32, must equal O.fft bit for bit — any accurate transform in double rounds to the same c64 values."""
    ld = np.longdouble
    if np.finfo(ld).eps >= np.finfo(np.float64).eps:
        pytest.skip("long double is not wider than double on this platform")
    rng = np.random.Generator(np.random.PCG64(2048 + n))
    w = O.hann(n)
    frames = (rng.standard_normal((8, n), dtype=np.float32) * w).astype(np.float32)  # the f32 product of lib/nx_signal.ex:101
    got = O.fft(frames)
    pi = ld("3.14159265358979323846264338327950288")
    m = (np.arange(n, dtype=np.int64)[:, None] * np.arange(n, dtype=np.int64)[None, :]) % n  # k n mod N, exact
    ang = (ld(2) * pi / ld(n)) * m.astype(ld)
    c, s = np.cos(ang), np.sin(ang)
    x = frames.astype(ld)
    re = x @ c.T  # X[k] = sum_n x[n] (cos - i sin)(2 pi k n / N)
    im = -(x @ s.T)
    re = np.where(np.abs(re) <= ld(1e-10), ld(0), re)  # Nx.fft's eps clean-up (rule 7): e.g. Im X[N/2], 1e-19 in long double
    im = np.where(np.abs(im) <= ld(1e-10), ld(0), im)
    ref = (re.astype(np.float64).astype(np.float32) + 1j * im.astype(np.float64).astype(np.float32)).astype(np.complex64)
    # long double -> f32 through double can double-round only within 2^-29 ulp(f32) of a tie: never here
    diff = got.view(np.uint32) != ref.view(np.uint32)
    nd = int(diff.sum())
    if nd:
        ulps = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))[diff]
        assert nd <= 1 and ulps.max() <= 1, f"{nd} of {diff.size} components differ from the long-double DFT (max {ulps.max()} ulp)"
    # and the clean-up threshold is far below anything in these frames
    assert np.min(np.abs(np.concatenate([got.real.ravel(), got.imag.ravel()])[np.concatenate([got.real.ravel(), got.imag.ravel()]) != 0])) > 1e-10
