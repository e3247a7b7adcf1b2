"""The dispatch switches of the library ("Environment variables") select kernels the default dispatch does not take:
the generic kernels behind every tuned one, the alternative FIR / mel / long-transform / n-D forms.  Each switch is read once per
process, so each case here is a fresh interpreter that runs the parity tests that reach the switched path — in the default
`-m gpu` run, not only in tools/run_matrix.sh (which runs the FULL suites per switch and stays the thorough form)."""
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
    ("NXSIG_DISABLE_8K=1", ("tests/test_gpu_tuned_kernels.py", "8192")),
    ("NXSIG_DISABLE_4K=1", ("tests/test_gpu_tuned_kernels.py", "4096")),
    ("NXSIG_DISABLE_FUSED_FILTER=1", ISTFT),
    ("NXSIG_ISTFT_NT_LOADS=0", ISTFT),
    ("NXSIG_ISTFT_DEEP=0", ISTFT),
    ("NXSIG_ISTFT_HALF_DEEP=0", ("tests/test_gpu_tuned_kernels.py", "half_n512")),
    ("NXSIG_ISTFT_HALF_DEEP=1", ("tests/test_gpu_tuned_kernels.py", "half_n512")),
    ("NXSIG_STORE_POLICY=0", STFT),
    ("NXSIG_STORE_POLICY=2", STFT),
    ("NXSIG_WAVE_NO_SPLIT=1", STFT),
    ("NXSIG_NO_AL8=1", ("tests/test_gpu_tuned_kernels.py", "stft_wave_variants")),
    ("NXSIG_NO_STAGE=1", ("tests/test_gpu_tuned_kernels.py", "stft_wave_variants")),
    ("NXSIG_FIR32=0", FIR),
    ("NXSIG_FIR32=2", FIR),
    ("NXSIG_FIR_PAD_TAPS=0", FIR),
    ("NXSIG_FIR_PHASE=0", FIR),
    ("NXSIG_FIR_HREG=0", FIR),
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
