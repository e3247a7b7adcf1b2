"""The examples that walk through the reference's guides run as part of the GPU suite (they assert their own parity checks)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,needles", [
    ("filtering_guide.py", ["istft_filtered == multiply-then-istft bit for bit: True", "times / frequencies identical: True"]),
    ("spectrogram_guide.py", ["peak bins at [431, 991] Hz", "mel_spectrogram 128 bands (fused)"]),
    ("sharded_groups.py", ["identical to the unsharded call: True", "taps :same sharded by channels (64, 960000); identical: True",
                           "sharded istft identical to unsharded: True"]),
])
def test_guide_examples_run_and_check_themselves(script, needles):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    for n in needles:
        assert n in r.stdout, (n, r.stdout[-2000:])
    for line in r.stdout.splitlines():  # every reported error is inside the 1e-5 tolerance (dB errors are in dB: < 1e-2)
        if " err " in line and "dB" not in line:
            for tok in line.replace(";", " ").replace(",", " ").split():
                if "e-" in tok:
                    try:
                        assert float(tok) < 1e-5, line
                    except ValueError:
                        pass
