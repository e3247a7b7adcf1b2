"""The drop-in boundary exercised WITHOUT Python in the loop: a plain-C program (tests/c_abi_smoke.c) includes
include/nxsig.h and links libnxsig.so.  CPU suite: it compiles as C99 and links (every symbol it uses resolves);
GPU suite: it runs stft -> istft on host and device buffers."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT
from nx_signal_amd import _lib

SRC = os.path.join(ROOT, "tests", "c_abi_smoke.c")
EXE = os.path.join(ROOT, "tests", "_c_abi_smoke")


def _build():
    gcc = shutil.which("gcc")
    assert gcc, "gcc not found"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [gcc, "-std=c99", "-Wall", "-Werror", "-O1", f"-I{os.path.join(ROOT, 'include')}", SRC, "-o", EXE,
           f"-L{libdir}", "-lnxsig", "-lm", f"-Wl,-rpath,{libdir}"]
    subprocess.check_call(cmd)
    return EXE


def test_header_is_valid_c99_and_program_links():
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "nxsig.h")])
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_c_program_runs_on_gpu():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c_abi_smoke ok" in r.stdout
