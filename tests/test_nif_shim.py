"""The dirty-NIF shim (nif/nxsig_nif.c) compiled against the stand-in erl_nif.h and EXECUTED through the miniature term
runtime of tests/stub/ (tests/nif_harness.py): the calls are made by name / arity through the ErlNifEntry table with the
term shapes the Elixir wrappers (elixir/lib/) build, and compared with the ctypes path bit for bit.

CPU part (no GPU): the shim compiles with -Wall -Wextra -Werror; nif.ex and the shim's funcs[] table agree in both
directions; host generators; validation happens BEFORE any allocation and a failing allocation is an error
tuple, not an abort.  GPU part (-m gpu): stft / istft / fir / as_windowed / overlap_and_add / fft / mel / device-resident
chain / sharded calls through the NIF equal the ctypes results; resources are released (HBM freed) when the last term goes."""
import os
import re

import numpy as np
import pytest

import nif_harness as H
from conftest import ROOT, f32_list
from oracle import nx_oracle as O

import nx_signal_amd as S

gpu = pytest.mark.gpu
PARAMS = (1024, 256, 1024, 0, 0, 0, 0, 48000.0)  # {n, hop, k, pad_mode, pad_lo, pad_hi, scaling, sampling_rate}


def f32(b):
    return np.frombuffer(b, np.float32)


def c64(b):
    return np.frombuffer(b, np.complex64)


def test_shim_compiles_warning_free_and_loads():
    H.build(force=True)  # gcc -std=c11 -Wall -Wextra -Werror against tests/stub/erl_nif.h
    assert H.lib().fake_module_name() == b"Elixir.NxSignalAMD.NIF"
    assert len(H.funcs()) >= 28


def test_nif_ex_matches_the_shim_table():
    src = open(os.path.join(ROOT, "elixir", "lib", "nx_signal_amd", "nif.ex")).read()
    stubs = {}
    for m in re.finditer(r"^\s*def (\w+)\(([^)]*)\)", src, re.M):
        name, args = m.group(1), m.group(2).strip()
        if name == "load_nif":
            continue
        stubs[(name, 0 if not args else len(args.split(",")))] = True
    table = H.funcs()
    assert set(stubs) == set(table), set(stubs) ^ set(table)
    # every NIF the Elixir modules call exists with that arity
    called = set()
    for root, _, files in os.walk(os.path.join(ROOT, "elixir", "lib")):
        for f in files:
            text = open(os.path.join(root, f)).read()
            for m in re.finditer(r"NIF\.(\w+)\(", text):
                start = m.end()
                depth, i, commas, empty = 1, start, 0, True
                while depth:
                    ch = text[i]
                    if ch in "([{":
                        depth += 1
                    elif ch in ")]}":
                        depth -= 1
                    elif ch == "," and depth == 1:
                        commas += 1
                    if depth and not ch.isspace():
                        empty = False
                    i += 1
                called.add((m.group(1), 0 if empty else commas + 1))
    assert called <= set(table), called - set(table)
    # GPU entry points are dirty jobs (a scheduler thread must never block on the GPU)
    for (name, ar), flags in table.items():
        if name in ("window", "firwin", "fft_frequencies", "sinc", "buf_size", "group_info", "device_count", "shard_range",
                    "window_f64", "firwin_f64", "fft_frequencies_f64", "sinc_f64"):
            continue
        assert flags in (1, 2), (name, ar, flags)


def test_host_generators_through_the_nif(golden):
    for v in golden["windows"]:
        if v["fn"] == "rectangular":
            continue
        kind = {"bartlett": 1, "triangular": 2, "blackman": 3, "hamming": 4, "hann": 5, "kaiser": 6}[v["fn"]]
        o = v["opts"]
        ok, b = H.call("window", kind, v["n"], 1 if o.get("is_periodic", True) else 0, float(o.get("beta", 12.0)), float(o.get("eps", 1e-7)))
        assert ok == "ok"
        if v["exact"]:
            assert np.array_equal(f32(b).view(np.uint32), f32_list(v["expect"]).view(np.uint32)), v["src"]
    for v in golden["firwin"]:
        o = dict(v["opts"])
        win = o.get("window", "hamming")
        kind, beta = (6, float(win[1])) if isinstance(win, list) else ({"hamming": 4, "hann": 5, "blackman": 3, "bartlett": 1, "rectangular": 0}[win], 0.0)
        ok, b = H.call("firwin", v["num_taps"], [float(c) for c in v["cutoff"]], kind, beta, 1 if o.get("pass_zero", True) else 0,
                       1 if o.get("scale", True) else 0, float(o.get("sampling_rate", 2.0)))
        want = S.filters.firwin(v["num_taps"], v["cutoff"], **{k: (tuple(x) if isinstance(x, list) else x) for k, x in o.items()})
        assert np.array_equal(f32(b).view(np.uint32), want.view(np.uint32)), v["src"]
    for v in golden["firwin_errors"]:
        with pytest.raises(H.NifError) as e:
            o = v["opts"]
            kind = 4 if "window" not in o else 99  # an unknown window name maps to no kind on the Elixir side; the library refuses too
            H.call("firwin", v["num_taps"], [float(c) for c in v["cutoff"]], kind, 0.0, 1 if o.get("pass_zero", True) else 0, 1, float(o.get("sampling_rate", 2.0)))
        assert e.value.code == -1  # -> ArgumentError on the Elixir side
    v = golden["fft_frequencies"][0]
    ok, b = H.call("fft_frequencies", float(v["sampling_rate"]), v["fft_length"], 0)
    assert np.array_equal(f32(b), f32_list(v["expect"]))
    v = golden["mel_filters"][0]
    ok, b = H.call("mel_filters", v["fft_length"], v["mel_bins"], float(v["sampling_rate"]), 3016.0, 200.0 / 3.0)
    assert np.array_equal(f32(b).view(np.uint32), S.mel_filters(v["fft_length"], v["mel_bins"], v["sampling_rate"]).reshape(-1).view(np.uint32))
    ok, b = H.call("sinc", np.array([0.0, 0.25, 1.0], np.float32))
    assert np.array_equal(f32(b), S.waveforms.sinc(np.array([0.0, 0.25, 1.0], np.float32)))
    # integers are accepted where Elixir callers may pass them (sampling_rate: 48000)
    assert H.call("fft_frequencies", 16000, 10, 0)[0] == "ok"
    # the f64 generators (`type: {:f, 64}`): the same bits as the ctypes path
    f64 = lambda b: np.frombuffer(b, np.float64)  # noqa: E731
    ok, b = H.call("window_f64", 6, 33, 1, 9.0, 1e-7)
    assert np.array_equal(f64(b), S.windows.kaiser(33, beta=9.0, type="f64"))
    ok, b = H.call("firwin_f64", 31, [0.2, 0.5], 4, 0.0, 0, 1, 2.0)
    assert np.array_equal(f64(b), S.filters.firwin(31, [0.2, 0.5], pass_zero=False, type="f64"))
    ok, b = H.call("fft_frequencies_f64", 44100.0, 16, 0)
    assert np.array_equal(f64(b), S.fft_frequencies(44100.0, fft_length=16, type="f64"))
    ok, b = H.call("sinc_f64", np.array([0.0, 0.25, 1.0]))
    assert np.array_equal(f64(b), S.waveforms.sinc(np.array([0.0, 0.25, 1.0])))


def test_malformed_terms_are_badarg_not_crashes():
    with pytest.raises(H.BadArg):
        H.call("window", 5, -3, 1, 0.0, 1e-7)
    with pytest.raises(H.BadArg):
        H.call("window", "hann", 8, 1, 0.0, 1e-7)
    with pytest.raises(H.BadArg):
        H.call("firwin", 0, [0.5], 4, 0.0, 1, 1, 2.0)
    with pytest.raises(H.BadArg):
        H.call("stft", 1, b"", 0, 0, b"", PARAMS)  # not a context resource
    with pytest.raises(H.UndefinedNif):
        H.call("stft", 1, 2)
    with pytest.raises(H.BadArg):
        H.call("sinc", b"abc")
    assert H.lib().fake_live_binaries() == 0


def test_result_allocation_failure_is_an_error_tuple():
    L = H.lib()
    L.fake_set_alloc_limit(1024)
    try:
        with pytest.raises(H.NifError) as e:
            H.call("window", 5, 4096, 1, 0.0, 1e-7)
        assert e.value.code == -5 and L.fake_live_binaries() == 0
    finally:
        L.fake_set_alloc_limit(1 << 40)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def nctx():
    ok, ctx = H.call("ctx_create", 0)
    assert ok == "ok"
    yield ctx
    H.release_all()


@gpu
def test_stft_istft_fir_through_the_nif_equal_the_ctypes_path(nctx):
    x = np.stack([O.synth_signal(48000, seed=s) for s in (1234, 1235)])
    w = S.windows.hann(1024)
    ok, zb, m, tb, fb = H.call("stft", nctx, x, 48000, 2, w, PARAMS)
    z, t, f = S.stft(x, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert ok == "ok" and m == 184
    assert np.array_equal(c64(zb).view(np.uint32), z.reshape(-1).view(np.uint32))
    assert np.array_equal(f32(tb), t) and np.array_equal(f32(fb), f)
    zo, _, _ = O.stft(x[0], w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert float(np.max(np.abs(c64(zb)[: zo.size].reshape(zo.shape) - zo)) / np.max(np.abs(zo))) < 1e-5
    ok, yb = H.call("istft", nctx, z, 184, 2, w, PARAMS)
    y = S.istft(z, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert np.array_equal(c64(yb).view(np.uint32), y.reshape(-1).view(np.uint32))
    h = S.filters.firwin(257, [4000.0], sampling_rate=48000)
    for mode, name in ((0, "full"), (1, "same"), (2, "valid")):
        ok, yb = H.call("fir", nctx, x, 48000, 2, h, mode)
        assert np.array_equal(f32(yb).view(np.uint32), S.filters.fir(x, h, mode=name).reshape(-1).view(np.uint32))
    # the reference's ArgumentErrors travel as {:error, {-1, message}}
    with pytest.raises(H.NifError) as e:
        H.call("stft", nctx, x, 48000, 2, w, (1024, 256, 1024, 7, 0, 0, 0, 48000.0))
    assert e.value.code == -1 and "invalid padding mode" in e.value.msg
    with pytest.raises(H.NifError) as e:
        H.call("fir", nctx, x, 48000, 2, h, 9)
    assert e.value.code == -1 and "expected mode to be one of" in e.value.msg
    # shapes that do not match the binaries are badarg before anything is allocated or read
    for bad in ((nctx, x, 48001, 2, w, PARAMS), (nctx, x, 48000, 2, w[:1000], PARAMS), (nctx, x, 48000, 2, w, (1024, 256, 0, 0, 0, 0, 0, 48000.0)),
                (nctx, x, 48000, 2, w, (1024, 256, -5, 0, 0, 0, 0, 48000.0)), (nctx, x, 48000, 0, w, PARAMS)):
        with pytest.raises(H.BadArg):
            H.call("stft", *bad)
    with pytest.raises(H.BadArg):
        H.call("istft", nctx, z, 184, 2, w[:100], PARAMS)
    assert H.lib().fake_live_binaries() == 0


@gpu
def test_stft_of_complex_samples_through_the_nif(nctx):
    """stft_c64 / stft_c64_dev: the term shapes nx_signal_amd.ex builds for c64 tensors (binary of interleaved f32 re, im)"""
    rng = np.random.Generator(np.random.PCG64(5))
    x = (rng.standard_normal((2, 48000)) + 1j * rng.standard_normal((2, 48000))).astype(np.complex64)
    w = S.windows.hann(1024)
    ok, zb, m, tb, fb = H.call("stft_c64", nctx, x, 48000, 2, w, PARAMS)
    z, t, f = S.stft(x, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert ok == "ok" and m == 184
    assert np.array_equal(c64(zb).view(np.uint32), z.reshape(-1).view(np.uint32))
    assert np.array_equal(f32(tb), t) and np.array_equal(f32(fb), f)
    zo, _, _ = O.stft(x[1], w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert float(np.max(np.abs(c64(zb)[zo.size:].reshape(zo.shape) - zo)) / np.max(np.abs(zo))) < 1e-5
    ok, xb = H.call("to_device", nctx, x[0])
    ok, zd, md = H.call("stft_c64_dev", nctx, xb, 48000, 1, w, PARAMS)
    ok, back = H.call("from_device", zd)
    assert md == 184 and np.array_equal(c64(back).view(np.uint32), z[0].reshape(-1).view(np.uint32))
    for bad in ((nctx, x, 48001, 2, w, PARAMS), (nctx, x.real.astype(np.float32), 48000, 2, w, PARAMS), (nctx, x, 48000, 2, w[:1000], PARAMS)):
        with pytest.raises(H.BadArg):
            H.call("stft_c64", *bad)
    with pytest.raises(H.BadArg):
        H.call("stft_c64_dev", nctx, xb, 48000, 2, w, PARAMS)   # the buffer holds one complex row, not two


@gpu
def test_f64_tier_through_the_nif(nctx):
    """stft_f64 / istft_c128 / fir_f64 / fft_c128 / as_windowed_f64 / overlap_and_add_f64 with the term shapes nx_signal_amd.ex builds"""
    rng = np.random.Generator(np.random.PCG64(3))
    N, hop = 256, 64
    x = rng.standard_normal((2, 4000))
    w64 = S.windows.hann(N, type="f64")
    w32 = S.windows.hann(N)
    prm = (N, hop, N, 0, 0, 0, 1, 8000.0)
    for w, flag in ((w64, 1), (w32, 0)):
        ok, zb, m, tb, fb = H.call("stft_f64", nctx, x, x.shape[1], 2, w, flag, prm)
        z, t, f = S.stft(x, w, overlap_length=N - hop, scaling="spectrum", sampling_rate=8000.0)
        assert ok == "ok" and m == z.shape[1]
        assert np.array_equal(np.frombuffer(zb, np.complex128).reshape(z.shape), z) and np.array_equal(f32(tb), t) and np.array_equal(f32(fb), f)
        ok, yb = H.call("istft_c128", nctx, z, m, 2, w, flag, prm)
        assert np.array_equal(np.frombuffer(yb, np.complex128).reshape(2, -1), S.istft(z, w, overlap_length=N - hop, scaling="spectrum", sampling_rate=8000.0))
    xc = x + 1j * rng.standard_normal(x.shape)       # c128 samples (round 6)
    ok, zb, m, tb, fb = H.call("stft_c128", nctx, xc, xc.shape[1], 2, w64, 1, prm)
    zc = S.stft(xc, w64, overlap_length=N - hop, scaling="spectrum", sampling_rate=8000.0)[0]
    assert ok == "ok" and m == zc.shape[1] and np.array_equal(np.frombuffer(zb, np.complex128).reshape(zc.shape), zc)
    with pytest.raises(H.BadArg):
        H.call("stft_c128", nctx, x, x.shape[1], 2, w64, 1, prm)   # f64 rows are half the bytes of c128 rows
    h = S.filters.firwin(129, [0.3], type="f64")
    ok, yb = H.call("fir_f64", nctx, x, x.shape[1], 2, h, 1)
    assert np.array_equal(np.frombuffer(yb, np.float64).reshape(2, -1), S.filters.fir(x, h))
    ok, ob = H.call("fft_c128", nctx, x, 1, 2, x.shape[1], 4096, 0)
    assert np.array_equal(np.frombuffer(ob, np.complex128).reshape(2, 4096), S.transforms.fft_nd(x, lengths=[4096]))
    ok, fb, m = H.call("as_windowed_f64", nctx, x, x.shape[1], 2, 100, 30, 1, 0, 0)
    fr = S.as_windowed(x, window_length=100, stride=30, padding="reflect")
    assert m == fr.shape[1] and np.array_equal(np.frombuffer(fb, np.float64).reshape(fr.shape), fr)
    ok, ob = H.call("overlap_and_add_f64", nctx, fr, m, 2, 100, 70, 1)
    assert np.array_equal(np.frombuffer(ob, np.float64).reshape(2, -1), S.overlap_and_add(fr, overlap_length=70))
    with pytest.raises(H.BadArg):
        H.call("stft_f64", nctx, x.astype(np.float32), x.shape[1], 2, w64, 1, prm)   # f32 bytes where f64 are announced
    with pytest.raises(H.NifError) as e:
        H.call("fir_f64", nctx, x, x.shape[1], 2, rng.standard_normal(5000), 1)
    assert e.value.code == -2   # beyond the tier's 4097 taps: reported, not approximated


@gpu
def test_framing_ola_fft_mel_through_the_nif(nctx, golden):
    for v in golden["as_windowed"]:
        x = np.array(v["x"], np.float32)
        pad = v["padding"]
        mode, lo, hi = ({"valid": 0, "reflect": 1, "same": 2}[pad], 0, 0) if isinstance(pad, str) else (3, pad[0][0], pad[0][1])
        ok, fb, m = H.call("as_windowed", nctx, x, x.size, 1, v["window_length"], v["stride"], mode, lo, hi)
        assert np.array_equal(f32(fb).reshape(m, v["window_length"]), np.array(v["expect"], np.float32)), v["src"]
    # int32 words travel bit-exactly (the gather never touches the payload): values far beyond 2^24
    xi = (np.arange(40, dtype=np.int64) * 100_000_007 % (2 ** 31 - 1)).astype(np.int32)
    ok, fb, m = H.call("as_windowed", nctx, xi.view(np.float32), 40, 1, 8, 4, 0, 0, 0)
    assert np.array_equal(np.frombuffer(fb, np.int32).reshape(m, 8), np.stack([xi[4 * i:4 * i + 8] for i in range(m)]))
    assert np.array_equal(S.as_windowed(xi, window_length=8, stride=4), np.stack([xi[4 * i:4 * i + 8] for i in range(m)]))
    fr = np.arange(3 * 4, dtype=np.float32).reshape(3, 4)
    ok, ob = H.call("overlap_and_add", nctx, fr, 3, 1, 4, 2, 1)
    assert np.array_equal(f32(ob), S.overlap_and_add(fr, overlap_length=2))
    with pytest.raises(H.NifError) as e:
        H.call("overlap_and_add", nctx, fr, 3, 1, 4, 4, 1)
    assert e.value.code == -1 and "overlap_length must be a number less than the window size 4, got: 4" in e.value.msg
    xr = O.synth_signal(6 * 50, seed=5).reshape(6, 50)
    ok, ob = H.call("fft", nctx, xr, 1, 6, 50, 64, 0)
    want = S.transforms.fft_nd(xr, axes=[1], lengths=[64]) if hasattr(S.transforms, "fft_nd") else None
    if want is not None:
        assert np.array_equal(c64(ob).view(np.uint32), np.ascontiguousarray(want).reshape(-1).view(np.uint32))
    a = (O.synth_signal(30, seed=1) + 1j * O.synth_signal(30, seed=2)).astype(np.complex64)
    b = (O.synth_signal(9, seed=3) + 1j * O.synth_signal(9, seed=4)).astype(np.complex64)
    ok, ob = H.call("fftconvolve_c64", nctx, a, b, 0)
    ref = np.convolve(a.astype(np.complex128), b.astype(np.complex128))
    assert float(np.max(np.abs(c64(ob) - ref)) / np.max(np.abs(ref))) < 1e-5
    x = O.synth_signal(16000, seed=8)
    w = S.windows.hann(400)
    p = (400, 160, 512, 1, 0, 0, 0, 16000.0)
    filt = S.mel_filters(512, 80, 16000.0)
    ok, mb, m = H.call("stft_mel", nctx, x, 16000, 1, w, p, 80, filt)
    want = S.mel_spectrogram(x, w, overlap_length=240, fft_length=512, sampling_rate=16000, window_padding="reflect", mel_bins=80)
    assert np.array_equal(f32(mb).view(np.uint32), np.ascontiguousarray(want).reshape(-1).view(np.uint32))
    for kind, name in ((0, "magnitude"), (1, "power"), (2, "dbfs")):
        ok, gb, m2 = H.call("stft_magnitude", nctx, x, 16000, 1, w, p, kind)
        wantm = S.spectrogram(x, w, overlap_length=240, fft_length=512, sampling_rate=16000, window_padding="reflect", kind=name)[0]
        assert m2 == wantm.shape[0] and np.array_equal(f32(gb).view(np.uint32), np.ascontiguousarray(wantm).reshape(-1).view(np.uint32))
    z, _, _ = S.stft(x, w, overlap_length=240, fft_length=512, sampling_rate=16000, window_padding="reflect")
    ok, mb2 = H.call("stft_to_mel", nctx, z, z.shape[0], 512, 80, filt)
    assert np.allclose(f32(mb2).reshape(want.shape), want, atol=1e-4)


@gpu
def test_device_resident_chain_through_the_nif(nctx):
    """guides/filtering.livemd:137-159 without leaving HBM: to_device -> stft_dev -> spectrum_mul_dev -> istft_dev -> from_device"""
    x = O.synth_signal(48000, seed=77)
    w = S.windows.hann(1024)
    live0 = H.lib().fake_live_resources()
    ok, xb = H.call("to_device", nctx, x)
    assert H.call("buf_size", xb) == x.nbytes
    ok, zb, m = H.call("stft_dev", nctx, xb, 48000, 1, w, PARAMS)
    hfft = np.fft.fft(np.r_[S.filters.firwin(65, [3000.0], sampling_rate=48000), np.zeros(1024 - 65)]).astype(np.complex64)
    ok, zb2 = H.call("spectrum_mul_dev", nctx, zb, m, 1024, hfft)
    ok, yb = H.call("istft_dev", nctx, zb2, m, 1, w, PARAMS)
    assert H.call("sync", nctx) == "ok"
    ok, out = H.call("from_device", yb)
    z, _, _ = S.stft(x, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    want = S.istft(S.spectrum_multiply(z, hfft), w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert np.array_equal(c64(out).view(np.uint32), want.view(np.uint32))
    # the same two steps in one call (filter applied inside the inverse-STFT kernel); the spectrum buffer stays as it was
    ok, zb3, m3 = H.call("stft_dev", nctx, xb, 48000, 1, w, PARAMS)
    ok, yb2 = H.call("istft_filtered_dev", nctx, zb3, m3, 1, w, PARAMS, hfft)
    ok, out2 = H.call("from_device", yb2)
    assert np.array_equal(c64(out2).view(np.uint32), want.view(np.uint32))
    ok, zkeep = H.call("from_device", zb3)
    assert np.array_equal(c64(zkeep).view(np.uint32), z.reshape(-1).view(np.uint32))
    ok, out3 = H.call("istft_filtered", nctx, z, m3, 1, w, PARAMS, hfft)
    assert np.array_equal(c64(out3).view(np.uint32), want.view(np.uint32))
    with pytest.raises(H.BadArg):
        H.call("istft_filtered_dev", nctx, zb3, m3, 1, w, PARAMS, hfft[:512])  # filter binary of the wrong size
    del zb3, yb2
    ok, zh, mh = H.call("stft_onesided_dev", nctx, xb, 48000, 1, w, PARAMS)
    ok, zhb = H.call("from_device", zh)
    assert mh == m and np.array_equal(c64(zhb).reshape(m, 512).view(np.uint32), np.ascontiguousarray(z[:, :512]).view(np.uint32))
    del zh
    # the packed one-sided pair on the device: stft_packed_dev -> istft_packed_dev -> one download of a REAL signal
    ok, zp, mp = H.call("stft_packed_dev", nctx, xb, 48000, 1, w, PARAMS)
    ok, zpb = H.call("from_device", zp)
    zpw, _, _ = S.stft_packed(x, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert mp == m and np.array_equal(c64(zpb).view(np.uint32), zpw.reshape(-1).view(np.uint32))
    ok, yp = H.call("istft_packed_dev", nctx, zp, mp, 1, w, PARAMS)
    ok, ypb = H.call("from_device", yp)
    ypw = S.istft_packed(zpw, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    assert np.array_equal(f32(ypb).view(np.uint32), ypw.view(np.uint32))
    assert float(np.max(np.abs(f32(ypb)[1024:-1024] - x[1024:48000 - 1024 - 128]))) < 2e-6 * float(np.max(np.abs(x)))
    with pytest.raises(H.BadArg):
        H.call("istft_packed_dev", nctx, zp, mp * 2, 1, w, PARAMS)   # more frames than the buffer holds
    del zp, yp
    ok, fy, n = H.call("fir_dev", nctx, xb, 48000, 1, S.filters.firwin(257, [4000.0], sampling_rate=48000), 1)
    ok, fo = H.call("from_device", fy)
    assert n == 48000 and np.array_equal(f32(fo).view(np.uint32), S.filters.fir(x, S.filters.firwin(257, [4000.0], sampling_rate=48000)).view(np.uint32))
    # a short window binary is refused before the library reads it
    with pytest.raises(H.BadArg):
        H.call("istft_dev", nctx, zb, m, 1, w[:512], PARAMS)
    with pytest.raises(H.BadArg):
        H.call("stft_dev", nctx, xb, 48000, 2, w, PARAMS)  # the buffer holds one row, not two
    # dropping the last term of a device tensor runs its destructor (HBM freed), the context outlives its buffers
    d0 = H.lib().fake_dtor_calls()
    del xb, zb, zb2, yb, fy
    H.release_all()
    assert H.lib().fake_dtor_calls() - d0 >= 4
    assert H.lib().fake_live_resources() <= live0 + 1


@gpu
def test_sharded_calls_through_the_nif():
    ok, ctx = H.call("ctx_create", 0)
    ok, g = H.call("group_create", [0, 0])
    assert H.call("group_info", g) == (2, 2, 0)
    x = np.stack([O.synth_signal(30000, seed=500 + c) for c in range(3)])
    w = S.windows.hann(1024)
    z, _, _ = S.stft(x, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    for axis, gather in ((0, 0), (0, 1)):
        ok, zb, m = H.call("stft_sharded", g, x, 30000, 3, w, PARAMS, axis, gather)
        assert m == z.shape[1] and np.array_equal(c64(zb).view(np.uint32), z.reshape(-1).view(np.uint32))
    ok, zb, m = H.call("stft_sharded", g, x[0], 30000, 1, w, PARAMS, 1, 1)
    assert float(np.max(np.abs(c64(zb).reshape(z[0].shape) - z[0])) / np.max(np.abs(z[0]))) < 1e-6
    h = S.filters.firwin(257, [4000.0], sampling_rate=48000)
    ok, yb = H.call("fir_sharded", g, x, 30000, 3, h, 1, 0, 0)
    assert np.array_equal(f32(yb).view(np.uint32), S.filters.fir(x, h).reshape(-1).view(np.uint32))
    yfull = S.istft(z, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    for axis, gather in ((0, 0), (0, 1), (1, 0)):
        ok, yb = H.call("istft_sharded", g, z, z.shape[1], 3, w, PARAMS, axis, gather)
        assert np.array_equal(c64(yb).view(np.uint32), yfull.reshape(-1).view(np.uint32))
    ok, g1 = H.call("group_create", [0])
    assert H.call("group_info", g1) == (1, 1, 1)  # one GPU per member: RCCL communicator (ncclCommInitAll)
    ok, zb, m = H.call("stft_sharded", g1, x, 30000, 3, w, PARAMS, 0, 1)
    assert np.array_equal(c64(zb).view(np.uint32), z.reshape(-1).view(np.uint32))
    with pytest.raises(H.NifError) as e:
        H.call("stft_sharded", g, x, 30000, 3, w, (1024, 256, 1024, 1, 0, 0, 0, 48000.0), 0, 0)
    assert e.value.code == -1 and ":valid" in e.value.msg
    # the sharded log-mel: quiet first channel, so only the all-reduced maximum clamps it like the unsharded call
    xq = x.copy()
    xq[0] *= np.float32(1e-3)
    filt = S.mel_filters(1024, 80, 48000.0)
    mel = S.mel_spectrogram(xq, w, overlap_length=768, fft_length=1024, sampling_rate=48000, mel_bins=80)
    for grp in (g, g1):
        ok, mb, m = H.call("stft_mel_sharded", grp, xq, 30000, 3, w, PARAMS, 80, filt, 0)
        assert m == mel.shape[1] and np.array_equal(f32(mb).view(np.uint32), mel.reshape(-1).view(np.uint32))
    ok, mb, m = H.call("stft_mel_sharded", g, xq, 30000, 3, w, PARAMS, 80, filt, 1)
    assert float(np.max(np.abs(f32(mb).reshape(mel.shape) - mel))) < 2e-5
    with pytest.raises(H.BadArg):   # a filterbank that is not [mel_bins][fft_length]
        H.call("stft_mel_sharded", g, xq, 30000, 3, w, PARAMS, 80, np.ascontiguousarray(filt[:, :100]), 0)
    del g, g1, ctx
    H.release_all()


def _plan(H, kind, a, b, c, world, pick):
    return [pick(H.call("shard_range", kind, a, b, c, world, r)[1]) for r in range(world)]


@gpu
@pytest.mark.parametrize("world", [2, 8])
def test_device_resident_shards_through_the_nif(world):
    """NxSignalAMD.Sharded.Tensor, term for term: group_scatter (Sharded.Tensor.to_device) -> stft_sharded_dev ->
    istft_sharded_dev / fir_sharded_dev / stft_mel_sharded_dev -> group_gather (from_device).  Between the first and the last
    call the payload exists only as buffer resources on the members' devices; the result bits equal the unsharded calls'."""
    ok, g = H.call("group_create", [0] * world)
    assert H.call("group_info", g)[:2] == (world, world)
    B, L, N, hop = 5, 40000, 1024, 256
    x = np.stack([O.synth_signal(L, seed=700 + c) for c in range(B)])
    w = S.windows.hann(N)
    z, _, _ = S.stft(x, w, overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    M = z.shape[1]
    y = S.istft(z, w, overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    h = S.filters.firwin(257, [4000.0], sampling_rate=48000)
    yf = S.filters.fir(x, h)

    # ---- channels: rows [c0, c1) per member
    rows = _plan(H, 0, B, 0, 0, world, lambda r: (r[0], r[1] - r[0]))
    def rplan(row_bytes):
        return [(r0, n, 0, row_bytes) for r0, n in rows]
    ok, xb = H.call("group_scatter", g, x, B, L * 4, rplan(L * 4))
    assert len(xb) == world and all(isinstance(b, H.Res) for b in xb)
    assert [H.call("buf_size", b) for b in xb] == [n * L * 4 for _, n in rows]
    ok, zb, m = H.call("stft_sharded_dev", g, xb, L, B, w, PARAMS, 0, 0)
    assert m == M and [H.call("buf_size", b) for b in zb] == [n * M * N * 8 for _, n in rows]
    ok, zbin = H.call("group_gather", g, zb, B, M * N * 8, rplan(M * N * 8))
    assert np.array_equal(c64(zbin).view(np.uint32), z.reshape(-1).view(np.uint32))
    ok, yb = H.call("istft_sharded_dev", g, zb, M, B, w, PARAMS, 0, 0)          # the chain stays on the devices
    ok, ybin = H.call("group_gather", g, yb, B, y.shape[1] * 8, rplan(y.shape[1] * 8))
    assert np.array_equal(c64(ybin).view(np.uint32), y.reshape(-1).view(np.uint32))
    ok, fb = H.call("fir_sharded_dev", g, xb, L, B, h, 1, 0, 0)
    ok, fbin = H.call("group_gather", g, fb, B, L * 4, rplan(L * 4))
    assert np.array_equal(f32(fbin).view(np.uint32), yf.reshape(-1).view(np.uint32))
    # gather: every member ends up with the whole spectrum; member 0's copy is downloaded like a DeviceTensor
    ok, zfull, m = H.call("stft_sharded_dev", g, xb, L, B, w, PARAMS, 0, 1)
    assert all(H.call("buf_size", b) == B * M * N * 8 for b in zfull)
    for b in (zfull[0], zfull[-1]):
        ok, one = H.call("from_device", b)
        assert np.array_equal(c64(one).view(np.uint32), z.reshape(-1).view(np.uint32))
    # the sharded log-mel with its all-reduce, on device shards
    xq = x.copy()
    xq[0] *= np.float32(1e-3)
    filt = S.mel_filters(N, 80, 48000.0)
    mel = S.mel_spectrogram(xq, w, overlap_length=N - hop, fft_length=N, sampling_rate=48000, mel_bins=80)
    ok, xqb = H.call("group_scatter", g, xq, B, L * 4, rplan(L * 4))
    ok, mb, m = H.call("stft_mel_sharded_dev", g, xqb, L, B, w, PARAMS, 80, filt, 0)
    ok, mbin = H.call("group_gather", g, mb, B, M * 80 * 4, rplan(M * 80 * 4))
    assert np.array_equal(f32(mbin).view(np.uint32), mel.reshape(-1).view(np.uint32))

    # ---- frames of long streams (batch > 1: the members' spans differ in length)
    fr = [H.call("shard_range", 1, M, N, hop, world, r)[1] for r in range(world)]      # (m0, m1, s0, s1)
    ok, xs = H.call("group_scatter", g, x, B, L * 4, [(0, B, s0 * 4, (s1 - s0) * 4) for _, _, s0, s1 in fr])
    ok, zs, m = H.call("stft_sharded_dev", g, xs, L, B, w, PARAMS, 1, 0)
    ok, zbin = H.call("group_gather", g, zs, B, M * N * 8, [(0, B, m0 * N * 8, (m1 - m0) * N * 8) for m0, m1, _, _ in fr])
    assert float(np.max(np.abs(c64(zbin).reshape(z.shape) - z)) / np.max(np.abs(z))) < 1e-6
    ir = [H.call("shard_range", 2, M, N, hop, world, r)[1] for r in range(world)]      # (f0, f1, n0, n1)
    ok, zi = H.call("group_scatter", g, z, B, M * N * 8, [(0, B, f0 * N * 8, (f1 - f0) * N * 8) for f0, f1, _, _ in ir])
    ok, yi = H.call("istft_sharded_dev", g, zi, M, B, w, PARAMS, 1, 0)
    ok, ybin = H.call("group_gather", g, yi, B, y.shape[1] * 8, [(0, B, n0 * 8, (n1 - n0) * 8) for _, _, n0, n1 in ir])
    assert np.array_equal(c64(ybin).view(np.uint32), y.reshape(-1).view(np.uint32))
    sr = [H.call("shard_range", 3, L, 257, 1, world, r)[1] for r in range(world)]      # (n0, n1, s0, s1)
    ok, xf = H.call("group_scatter", g, x, B, L * 4, [(0, B, s0 * 4, (s1 - s0) * 4) for _, _, s0, s1 in sr])
    ok, ff = H.call("fir_sharded_dev", g, xf, L, B, h, 1, 1, 0)
    ok, fbin = H.call("group_gather", g, ff, B, L * 4, [(0, B, n0 * 4, (n1 - n0) * 4) for n0, n1, _, _ in sr])
    assert float(np.max(np.abs(f32(fbin).reshape(yf.shape) - yf)) / np.max(np.abs(yf))) < 1e-6

    # ---- misuse: buffers of another group / in the wrong order / too few, malformed plans
    ok, g2 = H.call("group_create", [0] * world)
    with pytest.raises(H.BadArg):
        H.call("stft_sharded_dev", g2, xb, L, B, w, PARAMS, 0, 0)
    with pytest.raises(H.BadArg):
        H.call("stft_sharded_dev", g, list(reversed(xb)), L, B, w, PARAMS, 0, 0)
    with pytest.raises(H.BadArg):
        H.call("stft_sharded_dev", g, xb[:-1], L, B, w, PARAMS, 0, 0)
    with pytest.raises(H.BadArg):
        H.call("group_scatter", g, x, B, L * 4, [(0, B + 1, 0, L * 4)] * world)
    with pytest.raises(H.BadArg):
        H.call("group_gather", g, zb, B, M * N * 8, [(0, B, 0, M * N * 8 + 8)] * world)
    with pytest.raises(H.BadArg):
        H.call("shard_range", 7, 10, 0, 0, world, 0)
    # the buffers keep their group alive: drop the group term first, then the buffers
    live = H.lib().fake_live_resources()
    del g, g2
    H.release_all()
    ok, one = H.call("from_device", zfull[1])
    assert np.array_equal(c64(one).view(np.uint32), z.reshape(-1).view(np.uint32))
    del xb, zb, yb, fb, zfull, xqb, mb, xs, zs, zi, yi, xf, ff
    H.release_all()
    assert H.lib().fake_live_resources() < live


@gpu
def test_fft_nd_and_fftconvolve_nd_through_the_nif(nctx, golden):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((5, 7, 16)).astype(np.float32)
    ok, ob = H.call("fft_nd", nctx, x, 1, [5, 7, 16], [1, -1], [9, 32], 0)
    want = S.transforms.fft_nd(x, axes=[1, 2], lengths=[9, 32])
    assert np.array_equal(c64(ob).view(np.uint32), want.reshape(-1).view(np.uint32))
    with pytest.raises(H.BadArg):
        H.call("fft_nd", nctx, x, 1, [5, 7, 16], [3], [9], 0)
    for v in golden["fftconvolve_nd"]:
        def arr(t):
            t = np.array(t)
            return (t[..., 0] + 1j * t[..., 1]).astype(np.complex64) if v.get("complex") else t.astype(np.float32)
        a, b, e = arr(v["a"]), arr(v["b"]), arr(v["expect"])
        mode = {"full": 0, "same": 1, "valid": 2}[v["mode"]]
        real = 0 if v.get("complex") else 1
        ok, ob, osh = H.call("fftconvolve_nd", nctx, a, real, list(a.shape), b, real, list(b.shape), mode)
        out = (f32(ob) if real else c64(ob)).reshape(osh)
        assert tuple(osh) == e.shape and np.all(np.abs(out - e) <= 1e-4 + 1e-4 * np.abs(e)), v["src"]
    with pytest.raises(H.NifError) as ei:
        H.call("fftconvolve_nd", nctx, x, 1, [5, 7, 16], x, 1, [5, 112], 0)
    assert ei.value.code == -1 and "Rank of in1 and in2 must be equal" in ei.value.msg
    # convolve/3 with its default method (:direct): the reference's literals come back exact
    for v in golden["convolve_direct"]:
        def op(key):
            cplx = v.get("complex") and (v.get(key + "_complex") or not (v.get("a_complex") or v.get("b_complex")))
            t = np.atleast_1d(np.array(v[key], dtype=np.float64))
            return ((t[..., 0] + 1j * t[..., 1]).astype(np.complex64), 0) if cplx else (t.astype(np.float32), 1)
        (a, ar), (b, br) = op("a"), op("b")
        e = np.atleast_1d(np.array(v["expect"], dtype=np.float64))
        e = (e[..., 0] + 1j * e[..., 1]).astype(np.complex64) if v.get("complex") else e.astype(np.float32)
        ok, ob, osh = H.call("convolve_direct", nctx, a, ar, list(a.shape), b, br, list(b.shape), {"full": 0, "same": 1, "valid": 2}[v["mode"]])
        out = (f32(ob) if (ar and br) else c64(ob)).reshape(osh)
        assert tuple(osh) == e.shape and np.array_equal(out, e), v["src"]
    with pytest.raises(H.NifError) as ei:
        H.call("convolve_direct", nctx, np.ones((2, 3), np.float32), 1, [2, 3], np.ones((3, 2), np.float32), 1, [3, 2], 2)
    assert ei.value.code == -1 and "For :valid mode" in ei.value.msg
    assert H.lib().fake_live_binaries() == 0
