"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-side C / C++ of the path (SURVEY §5 "race detection /
sanitizers"): csrc/host_numerics.cpp (windows, firwin, mel filterbank ...) and the C leg of the oracle (oracle/bb_baseline.c)
are rebuilt with -fsanitize=address,undefined -fno-sanitize-recover and driven over a sweep of sizes by tests/san_host.cpp;
the NIF shim + its stand-in runtime are rebuilt the same way and driven through the host generators and the malformed-term
cases.  A finding aborts the child process and fails the test."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_san")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not found")
def test_host_numerics_and_oracle_c_under_asan_ubsan():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "san_host")
    obj = os.path.join(BUILD, "bb_baseline.o")
    subprocess.check_call(["gcc", *SAN, "-ffp-contract=off", "-fopenmp", "-c", os.path.join(ROOT, "oracle", "bb_baseline.c"), "-o", obj])
    subprocess.check_call([
        "g++", "-std=c++17", *SAN, "-ffp-contract=off", "-fopenmp", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
        "-I" + os.path.join(ROOT, "nx_signal_amd", "csrc"), os.path.join(ROOT, "tests", "san_host.cpp"),
        os.path.join(ROOT, "nx_signal_amd", "csrc", "host_numerics.cpp"), obj, "-o", exe, "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sanitized host numerics ok" in r.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not found")
def test_nif_shim_under_asan_ubsan():
    """the shim + the stand-in term runtime with ASan / UBSan: host generators, malformed terms, allocation failure"""
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "nif_fake_san.so")
    libdir = os.path.join(ROOT, "nx_signal_amd")
    subprocess.check_call(["gcc", "-std=c11", *SAN, "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-D_GNU_SOURCE",
                           "-I" + os.path.join(ROOT, "tests", "stub"), os.path.join(ROOT, "nif", "nxsig_nif.c"),
                           os.path.join(ROOT, "tests", "stub", "erl_nif_fake.c"), "-o", so, "-L" + libdir, "-lnxsig",
                           "-Wl,-rpath," + libdir])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, nif_harness as H\n"
        "H.SO = %r\n"
        "H.build = lambda force=False: H.SO\n"
        "for n in (1, 2, 5, 64, 1000):\n"
        "    for kind in range(1, 7):\n"
        "        assert H.call('window', kind, n, 1, 8.0, 1e-7)[0] == 'ok'\n"
        "assert H.call('firwin', 31, [0.2, 0.5], 4, 0.0, 1, 1, 2.0)[0] == 'ok'\n"
        "assert H.call('mel_filters', 64, 8, 8000.0, 3016.0, 200.0 / 3.0)[0] == 'ok'\n"
        "assert H.call('fft_frequencies', 8000, 16, 1)[0] == 'ok'\n"
        "assert H.call('sinc', np.linspace(-3, 3, 50).astype(np.float32))[0] == 'ok'\n"
        "for args in (('window', 5, -1, 1, 0.0, 1e-7), ('sinc', b'abc'), ('firwin', 0, [0.5], 4, 0.0, 1, 1, 2.0), ('stft', 1, b'', 0, 0, b'', (1,))):\n"
        "    try:\n        H.call(*args)\n        raise SystemExit('accepted ' + repr(args))\n    except H.BadArg:\n        pass\n"
        "H.lib().fake_set_alloc_limit(64)\n"
        "try:\n    H.call('window', 5, 4096, 1, 0.0, 1e-7)\n    raise SystemExit('allocation limit ignored')\nexcept H.NifError as e:\n    assert e.code == -5\n"
        "assert H.lib().fake_live_binaries() == 0\n"
        "H.release_all()\n"
        "print('sanitized nif shim ok')\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), so)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitized nif shim ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
