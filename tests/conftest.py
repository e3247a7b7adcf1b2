"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` runs here (no GPU); `-m gpu` runs on the MI355X box through gpurun.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


def f32_list(vals):
    """parse the reference's shortest-round-trip decimal strings to exact f32 values"""
    return np.array([np.float32(v) for v in vals], dtype=np.float32)


def nx_all_close(a, b, atol=1e-4, rtol=1e-4):
    """Nx.all_close semantics used by the reference's assert_all_close (test/support/nx_signal_case.ex:44-46)"""
    a = np.asarray(a)
    b = np.asarray(b)
    return bool(np.all(np.abs(a - b) <= atol + rtol * np.abs(b)))
