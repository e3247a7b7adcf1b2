"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/nxsig.h declares,
the host-side generators match the oracle and the reference's golden vectors bit-for-bit, and option
validation raises ArgumentError the way the reference's deftransforms do."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, f32_list
from oracle import nx_oracle as O

import nx_signal_amd as S
from nx_signal_amd import _lib


def _bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nxsig.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nxsig_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 28
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in nxsig.h but not exported by libnxsig.so"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().nxsig_abi_version() == 1


def test_stft_params_layout_matches_header():
    # int32 x4, int64 x2, int32 x2, double  -> 48 bytes, natural alignment
    assert ctypes.sizeof(_lib.StftParams) == 48
    assert _lib.StftParams.pad_lo.offset == 16 and _lib.StftParams.sampling_rate.offset == 40


WINS = [("bartlett", {}), ("triangular", {}), ("blackman", {"is_periodic": True}), ("blackman", {"is_periodic": False}),
        ("hamming", {"is_periodic": True}), ("hamming", {"is_periodic": False}), ("hann", {"is_periodic": True}),
        ("hann", {"is_periodic": False}), ("kaiser", {"beta": 12.0, "is_periodic": True}),
        ("kaiser", {"beta": 5.0, "is_periodic": False}), ("kaiser", {"beta": 2.5, "is_periodic": True})]


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 16, 33, 257, 1024, 2048])
def test_windows_match_oracle_bit_exact(n):
    for name, opts in WINS:
        got = getattr(S.windows, name)(n, **opts)
        exp = getattr(O, name)(n, **opts)
        assert got.dtype == np.float32
        assert np.array_equal(_bits(got), _bits(exp)), (name, opts, n)


def test_windows_golden(golden):
    for v in golden["windows"]:
        got = getattr(S.windows, v["fn"])(v["n"], **v["opts"])
        if v["fn"] == "rectangular":
            assert got.dtype == np.int64 and got.tolist() == v["expect"]
        else:
            assert np.array_equal(_bits(got), _bits(f32_list(v["expect"]))), v["src"]
    assert S.windows.rectangular(5, type="f32").dtype == np.float32


def test_window_option_quirks():
    with pytest.raises(S.ArgumentError):
        S.windows.bartlett(4, name="x")  # bartlett rejects :name (B11)
    with pytest.raises(S.ArgumentError):
        S.windows.hann(4, bogus=1)
    S.windows.triangular(4, name="x")


def test_sinc_and_fft_frequencies_golden(golden):
    for v in golden["sinc"]:
        assert np.array_equal(_bits(S.waveforms.sinc(f32_list(v["t"]))), _bits(f32_list(v["expect"])))
    for v in golden["fft_frequencies"]:
        got = S.fft_frequencies(v["sampling_rate"], fft_length=v["fft_length"])
        assert np.array_equal(_bits(got), _bits(f32_list(v["expect"])))
    for fs, k in [(48000, 1024), (44100.0, 2048), (8000.0, 16), (1, 4)]:
        assert np.array_equal(_bits(S.fft_frequencies(fs, fft_length=k)), _bits(O.fft_frequencies(fs, k)))


def test_stft_times_match_oracle():
    lib = _lib.load()
    for N, fs, M in [(2, 400, 3), (1024, 48000, 184), (2048, 48000.0, 1000), (4, 1, 3)]:
        out = np.empty(M, np.float32)
        _lib.check(lib.nxsig_stft_times_f32(N, float(fs), M, out.ctypes.data_as(ctypes.c_void_p)))
        assert np.array_equal(_bits(out), _bits(O.stft_times(N, fs, M)))


def test_firwin_golden_and_oracle(golden):
    for v in golden["firwin"]:
        opts = dict(v["opts"])
        if isinstance(opts.get("window"), list):
            opts["window"] = tuple(opts["window"])
        h = S.filters.firwin(v["num_taps"], v["cutoff"], **opts)
        assert np.all(np.abs(h - np.array(v["expect"])) <= v["atol"] + 1e-4 * np.abs(np.array(v["expect"]))), v["src"]
        assert np.array_equal(_bits(h), _bits(O.firwin(v["num_taps"], v["cutoff"], **opts))), v["src"]
    h = S.filters.firwin(257, [4000], sampling_rate=48000)  # BASELINE config 5 filter
    assert np.array_equal(_bits(h), _bits(O.firwin(257, [4000], sampling_rate=48000)))
    assert abs(float(h.astype(np.float64).sum()) - 1.0) < 1e-6  # DC gain 1


def test_firwin_errors(golden):
    for v in golden["firwin_errors"]:
        with pytest.raises(S.ArgumentError, match=v["match"]):
            S.filters.firwin(v["num_taps"], v["cutoff"], **v["opts"])
    with pytest.raises(S.ArgumentError, match="cutoff must be a list"):
        S.filters.firwin(5, 0.3)


def test_mel_filters_golden_and_oracle(golden):
    for v in golden["mel_filters"]:
        got = S.mel_filters(v["fft_length"], v["mel_bins"], v["sampling_rate"])
        exp = np.array([f32_list(row) for row in v["expect"]])
        assert np.array_equal(_bits(got), _bits(exp)), v["src"]  # bit-exact with the reference doctest
    for K, mb, fs in [(16, 4, 8000.0), (400, 80, 16000), (1024, 128, 16000), (2048, 128, 48000)]:
        assert np.array_equal(_bits(S.mel_filters(K, mb, fs)), _bits(O.mel_filters(K, mb, fs)))
    with pytest.raises(S.ArgumentError, match="unknown keys"):
        S.mel_filters(16, 4, 8000.0, bogus=1)


def test_num_frames_matches_oracle():
    lib = _lib.load()
    modes = {"valid": (0, 0, 0), "reflect": (1, 0, 0), "same": (2, 0, 0)}
    for L, N, hop in [(48000, 1024, 256), (2880000, 1024, 256), (10, 4, 2), (7, 6, 1), (8, 4, 1), (100, 7, 3)]:
        for name, (mode, lo, hi) in modes.items():
            assert lib.nxsig_num_frames(L, N, hop, mode, lo, hi) == O.num_frames(L, N, hop, name)
    assert lib.nxsig_num_frames(7, 2, 2, 3, 0, 3) == 5  # as_windowed doctest lib/nx_signal.ex:207-217
    assert lib.nxsig_num_frames(48000, 1024, 256, 0, 0, 0) == 184  # BASELINE config 1
    assert lib.nxsig_num_frames(2880000, 1024, 256, 0, 0, 0) == 11247  # config 2
    assert lib.nxsig_num_frames(28800000, 2048, 512, 0, 0, 0) == 56247  # config 4
    assert lib.nxsig_num_frames(3, 8, 1, 0, 0, 0) == _lib.ERR_INVALID_ARG
    assert lib.nxsig_conv_length(10, 3, 0) == 12 and lib.nxsig_conv_length(10, 3, 1) == 10
    assert lib.nxsig_conv_length(10, 3, 2) == 8 and lib.nxsig_conv_length(3, 10, 2) == 8
    assert lib.nxsig_ola_length(3, 4, 1) == 6


def test_option_validation_raises_argument_error_without_gpu():
    x = np.zeros(16, np.float32)
    w = S.windows.hann(4)
    with pytest.raises(S.ArgumentError, match="invalid :scaling"):
        S.stft(x, w, scaling="power")
    with pytest.raises(S.ArgumentError, match="invalid padding mode"):
        S.stft(x, w, window_padding="zeros")  # documented but raises (B2)
    with pytest.raises(S.ArgumentError, match="unknown keys"):
        S.stft(x, w, hop_length=2)
    with pytest.raises(S.ArgumentError, match="f32 extensions"):   # float64 samples belong to the f64 tier (full c128 spectrum only)
        S.stft_onesided(np.zeros(16), w)
    with pytest.raises(S.NxSignalDeviceError):   # ... which computes on the GPU like everything else: no CPU fallback
        S.stft(np.zeros(16), w)
    with pytest.raises(S.ArgumentError, match="invalid :scaling"):
        S.istft(np.zeros((3, 4), np.complex64), w, scaling="power")
    with pytest.raises(S.ArgumentError, match="overlap_length must be a number less than the window size 4, got: 4"):
        S.overlap_and_add(np.zeros((3, 4), np.float32), overlap_length=4)
    with pytest.raises(S.ArgumentError, match="expected an integer >= 1"):
        S.as_windowed(x, window_length=4, stride=0)
    with pytest.raises(S.ArgumentError, match="padding must be a list of"):
        S.as_windowed(x, window_length=4, padding=[(1.5, 2)])
    with pytest.raises(S.ArgumentError, match="expected mode to be one of"):
        S.convolution.convolve(x, w, mode="middle", method="bogus")  # mode validated first (B12)
    with pytest.raises(S.ArgumentError, match="expected method to be one of"):
        S.convolution.convolve(x, w, method="bogus")


def test_no_gpu_means_loud_failure():
    n = ctypes.c_int()
    rc = _lib.load().nxsig_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(S.NxSignalDeviceError, match="no CPU fallback"):
        S.stft(np.zeros(16, np.float32), S.windows.hann(4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nx_signal_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt.replace("oracle/nx_oracle.py:_kaiser_i0", ""), f


def test_no_launch_path_reads_the_environment():
    """round 4: the switches live in the context (Tuning); the library's sources call getenv in exactly two places — the loop of
    nxsig_ctx_create (tuning_from_env) and, at group creation, the RCCL library path (NXSIG_RCCL_LIB) and the NXSIG_QUIET switch of the
    one-line RCCL announcement (round 6)"""
    import glob
    import re

    calls = []
    for f in glob.glob(os.path.join(ROOT, "nx_signal_amd", "csrc", "*.*")):
        if not f.endswith((".cpp", ".hip", ".hpp", ".h")):
            continue
        for i, line in enumerate(open(f).read().splitlines(), 1):
            code = line.split("//")[0]
            if re.search(r"\bgetenv\s*\(", code):
                calls.append((os.path.basename(f), i))
    assert sorted(f for f, _ in calls) == ["api.cpp", "group.cpp", "group.cpp"], calls


def test_codelet_index_maps_match_numpy():
    """tools/check_codelet_maps.py: the prime-factor / Cooley-Tukey index maps written in small_dft.hpp's comments reproduce numpy's
    transform for every composite codelet (the code beside each comment is pinned by the kernels' -m gpu parity tests)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_codelet_maps.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") >= 20 and "WRONG" not in r.stdout
