"""The dispatch switches of the library ("Environment variables") select kernels the default dispatch does not take:
the generic kernels behind every tuned one, the alternative FIR / mel / long-transform / n-D forms.  A context reads the
NXSIG_<NAME> variables ONCE, at creation, into its switch table; no launch looks at the environment (round 4).  Two kinds of cases:
in-process ones flip switches of a live context through nxsig_ctx_set_tuning (what the sweep tools do), and one fresh interpreter per
switch with the variable set runs the parity tests that reach the switched path — in the default `-m gpu` run, not only in
tools/run_matrix.sh (which runs the FULL suites per switch and stays the thorough form)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FIR = ("tests/test_gpu_parity.py tests/test_gpu_tuned_kernels.py", "fir and not beyond_4gb and not long_filters")
STFT = ("tests/test_gpu_parity.py", "stft_matches_oracle or config1 or other_windows or eps_clean")
ISTFT = ("tests/test_gpu_parity.py tests/test_gpu_tuned_kernels.py",
         "istft_matches_oracle or istft_filtered or istft_wave_all_hops or domain_filtering")
MEL = ("tests/test_gpu_parity.py", "mel")
ND = ("tests/test_gpu_nd.py", "fft_nd or long_rows or fftconvolve_nd or long_transforms or correlate")
DIRECT = ("tests/test_gpu_nd.py", "convolve_direct and not config5")

CASES = [
    ("NXSIG_DISABLE_WAVE=1", ("tests/test_gpu_parity.py", "stft_matches_oracle or istft_matches_oracle or fir_matches or config1 or mel_spectrogram")),
    ("NXSIG_DISABLE_WAVE_ROWS=1", ("tests/test_gpu_parity.py tests/test_gpu_nd.py", "fft_rows or wave_core_row or generic_istft")),
    ("NXSIG_DISABLE_BLUE_WAVE=1", ("tests/test_gpu_parity.py tests/test_gpu_tuned_kernels.py", "non_power_of_two or bluestein")),
    ("NXSIG_DISABLE_R20=1", ("tests/test_gpu_parity.py tests/test_gpu_tuned_kernels.py", "non_power_of_two or r20")),
    ("NXSIG_DISABLE_RAB=1", ("tests/test_gpu_parity.py tests/test_gpu_tuned_kernels.py", "non_power_of_two or composite")),
    ("NXSIG_DISABLE_8K=1", ("tests/test_gpu_tuned_kernels.py", "8192")),
    ("NXSIG_DISABLE_4K=1", ("tests/test_gpu_tuned_kernels.py", "4096")),
    ("NXSIG_DISABLE_FUSED_FILTER=1", ISTFT),
    ("NXSIG_ISTFT_DEEP=0", ISTFT),
    ("NXSIG_ISTFT_REGOLA=0", ISTFT),
    ("NXSIG_ISTFT_REGOLA=1", ISTFT),
    ("NXSIG_ISTFT_HALF_DEEP=0", ("tests/test_gpu_tuned_kernels.py", "half_n512")),
    ("NXSIG_ISTFT_HALF_DEEP=1", ("tests/test_gpu_tuned_kernels.py", "half_n512")),
    ("NXSIG_STORE_POLICY=0", STFT),
    ("NXSIG_STORE_POLICY=2", STFT),
    ("NXSIG_WAVE_NO_SPLIT=1", STFT),
    ("NXSIG_NO_AL8=1", ("tests/test_gpu_tuned_kernels.py", "stft_wave_variants")),
    ("NXSIG_NO_STAGE=1", ("tests/test_gpu_tuned_kernels.py", "stft_wave_variants")),
    ("NXSIG_STAGE_PAD=0", ("tests/test_gpu_tuned_kernels.py tests/test_gpu_parity.py", "stft_wave_variants or stft_matches_oracle")),
    ("NXSIG_FIR32=0", FIR),
    ("NXSIG_FIR32=2", FIR),
    ("NXSIG_FIR_PAD_TAPS=0", FIR),
    ("NXSIG_FIR_PHASE=0", FIR),
    ("NXSIG_FIR_HREG=0", FIR),
    ("NXSIG_FIR_R2K=0", ("tests/test_gpu_parity.py tests/test_gpu_tuned_kernels.py", "fir and not beyond_4gb")),
    ("NXSIG_FIR_R2K=2", ("tests/test_gpu_parity.py tests/test_gpu_tuned_kernels.py", "fir and not beyond_4gb")),
    ("NXSIG_FIR_DLINE=0", ("tests/test_gpu_tuned_kernels.py tests/test_gpu_reference_numerics.py", "long_filters or fir_non_finite")),
    ("NXSIG_FIR_DLINE=2", ("tests/test_gpu_tuned_kernels.py", "long_filters or delay_line")),   # three passes also where the fused inverse pass applies
    ("NXSIG_WAVE_SMALL_W=0", STFT),                                                              # the many-round geometry for launches of one round
    ("NXSIG_WAVE_UNITS_PER_WAVE=4", ("tests/test_gpu_tuned_kernels.py tests/test_gpu_parity.py", "stft_wave_variants or stft_matches_oracle or composite_lengths_native")),
    ("NXSIG_MEL_TILE=0", ("tests/test_gpu_parity.py", "mel and not 8192-20-48000 and not stft_to_mel_is_bit")),
    ("NXSIG_MEL_LDS_KB=150", MEL),
    ("NXSIG_FFT_TILED=0", ND),
    ("NXSIG_FFT_TILE_ELEMS=2048", ND),
    ("NXSIG_FFT_TILE_NT=256", ND),
    ("NXSIG_FFT_COLUMNS=0", ND),
    ("NXSIG_CONV_POW2=0", ND),
    ("NXSIG_DIRECT_FAST=0", DIRECT),
    ("NXSIG_POOL_MAX_MB=0", ("tests/test_gpu_tuned_kernels.py", "device_resident_chain or stft_wave_chunk_seams")),
]


@pytest.mark.parametrize("switch,sel", CASES, ids=[c[0] for c in CASES])
def test_parity_under_a_dispatch_switch(switch, sel):
    files, expr = sel
    name, value = switch.split("=")
    env = dict(os.environ)
    env[name] = value
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", *files.split(), "-k", expr],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = r.stdout[-3000:] + r.stderr[-1000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 2, tail   # the selection reaches real tests under the switch


def test_switches_live_in_the_context_and_the_environment_is_read_once(monkeypatch):
    import numpy as np

    import nx_signal_amd as S
    from nx_signal_amd import _lib
    from oracle import nx_oracle as O

    monkeypatch.setenv("NXSIG_DISABLE_WAVE", "1")
    c1 = S.Context(0)
    monkeypatch.delenv("NXSIG_DISABLE_WAVE")
    c2 = S.Context(0)
    assert c1.get_tuning("NXSIG_DISABLE_WAVE") == (1, True) and c1.get_tuning("DISABLE_WAVE") == (1, True)
    assert c2.get_tuning("NXSIG_DISABLE_WAVE") == (0, False)
    os.environ["NXSIG_DISABLE_WAVE"] = "1"       # after creation: nobody looks
    try:
        assert c2.get_tuning("NXSIG_DISABLE_WAVE") == (0, False)
        rng = np.random.default_rng(3)
        x = rng.standard_normal(60000).astype(np.float32)
        w = S.windows.hann(1024)
        opts = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
        zo, _, _ = O.stft(x, w, **opts)
        z_generic = S.stft(c1.to_device(x), w, ctx=c1, **opts)[0].numpy()   # c1: generic kernels (environment at creation)
        z_tuned = S.stft(c2.to_device(x), w, ctx=c2, **opts)[0].numpy()     # c2: wave kernels
        for z in (z_generic, z_tuned):
            assert np.max(np.abs(z - zo)) / np.max(np.abs(zo)) < 1e-5
        assert not np.array_equal(z_generic.view(np.uint32), z_tuned.view(np.uint32))   # two different kernels really ran
        c2.set_tuning("NXSIG_DISABLE_WAVE", 1)    # flip the live context: the same bits as c1 now
        assert np.array_equal(S.stft(c2.to_device(x), w, ctx=c2, **opts)[0].numpy().view(np.uint32), z_generic.view(np.uint32))
        c2.clear_tuning("DISABLE_WAVE")
        assert np.array_equal(S.stft(c2.to_device(x), w, ctx=c2, **opts)[0].numpy().view(np.uint32), z_tuned.view(np.uint32))
        with pytest.raises(_lib.ArgumentError):
            c2.set_tuning("NXSIG_NO_SUCH_SWITCH", 1)
        c1.clear_tuning()
        assert c1.get_tuning("DISABLE_WAVE") == (0, False)
    finally:
        del os.environ["NXSIG_DISABLE_WAVE"]


@pytest.mark.parametrize("name,values", [("NXSIG_FIR32", (0, 2)), ("NXSIG_FIR_HREG", (0,)), ("NXSIG_FIR_PHASE", (0,)), ("NXSIG_ISTFT_DEEP", (0,)),
                                         ("NXSIG_STORE_POLICY", (0, 2)), ("NXSIG_WAVE_UNITS_PER_WAVE", (1, 3))])
def test_flipping_a_switch_in_process_keeps_parity(name, values):
    """what tools/sweep_*.py do: several values of one switch on ONE context, each result against the oracle"""
    import numpy as np

    import nx_signal_amd as S
    from oracle import nx_oracle as O

    ctx = S.Context(0)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 90000)).astype(np.float32)
    w = S.windows.hann(1024)
    opts = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
    h = S.filters.firwin(257, [4000.0], sampling_rate=48000.0)
    zo = np.stack([O.stft(r, w, **opts)[0] for r in x])
    yo = np.stack([O.istft(z, w, **opts) for z in zo])
    fo = np.stack([O.fftconvolve(r, h, mode="same") for r in x])
    for v in (None,) + tuple(values):
        if v is not None:
            ctx.set_tuning(name, v)
        xd = ctx.to_device(x)
        z = S.stft(xd, w, ctx=ctx, **opts)[0].numpy()
        y = S.istft(ctx.to_device(zo), w, ctx=ctx, **opts).numpy()
        f = S.filters.fir(xd, h, mode="same", ctx=ctx).numpy() if hasattr(S.filters, "fir") else None
        assert np.max(np.abs(z - zo)) / np.max(np.abs(zo)) < 1e-5, (name, v)
        assert np.max(np.abs(y - yo)) / np.max(np.abs(yo)) < 1e-5, (name, v)
        if f is not None:
            assert np.max(np.abs(f - fo)) / np.max(np.abs(fo)) < 1e-5, (name, v)


def test_switch_values_are_range_checked_both_ways_in():
    """ISTFT_RUNS_PER_CU = 0 reached an integer division by zero in the iSTFT launchers, and a non-numeric environment value
    was silently read as 0.  Both ways in are validated now: nxsig_ctx_set_tuning refuses an out-of-range value, the environment parser
    ignores it (and says so on stderr); "true" / "off" mean 1 / 0."""
    import numpy as np

    import nx_signal_amd as S
    from nx_signal_amd import _lib
    from oracle import nx_oracle as O

    ctx = S.Context(0)
    for name, bad in (("NXSIG_ISTFT_RUNS_PER_CU", 0), ("NXSIG_ISTFT_RUNS_PER_CU", -3), ("NXSIG_ISTFT_MIN_RUN", 0), ("NXSIG_WAVE_UNITS_PER_WAVE", 0),
                      ("NXSIG_FIR_UNITS_PER_WAVE", -1), ("NXSIG_STORE_POLICY", 7), ("NXSIG_MEL_LDS_KB", 0), ("NXSIG_DISABLE_WAVE", -1)):
        with pytest.raises(_lib.ArgumentError):
            ctx.set_tuning(name, bad)
        assert ctx.get_tuning(name)[1] is False
    code = (
        "import numpy as np, nx_signal_amd as S\n"
        "from oracle import nx_oracle as O\n"
        "c = S.Context(0)\n"
        "print('T', c.get_tuning('ISTFT_RUNS_PER_CU'), c.get_tuning('ISTFT_MIN_RUN'), c.get_tuning('DISABLE_WAVE'), c.get_tuning('FIR_HREG'), c.get_tuning('ISTFT_DEEP'))\n"
        "x = np.random.default_rng(5).standard_normal(40000).astype(np.float32); w = S.windows.hann(1024)\n"
        "o = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)\n"
        "zo = O.stft(x, w, **o)[0]; y = S.istft(c.to_device(zo), w, ctx=c, **o).numpy(); yo = O.istft(zo, w, **o)\n"
        "print('E', float(np.max(np.abs(y - yo)) / np.max(np.abs(yo))))\n"
    )
    env = dict(os.environ, NXSIG_ISTFT_RUNS_PER_CU="0", NXSIG_ISTFT_MIN_RUN="banana", NXSIG_DISABLE_WAVE="true", NXSIG_FIR_HREG="off", NXSIG_ISTFT_DEEP=" 0 ",
               PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "T (0, False) (0, False) (1, True) (0, True) (0, True)" in r.stdout, r.stdout
    assert "NXSIG_ISTFT_RUNS_PER_CU=0 is outside" in r.stderr and "NXSIG_ISTFT_MIN_RUN=banana is not a number" in r.stderr
    assert float(re.search(r"E ([0-9.e+-]+)", r.stdout).group(1)) < 1e-5
