"""Builds libnxsig.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

    python -m nx_signal_amd.build            # incremental
    python -m nx_signal_amd.build --force

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  No torch, no cmake: seven translation units (compiled concurrently) and one link line.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "libnxsig.so")
OBJ = os.path.join(HERE, "csrc", "_obj")

ARCH = "gfx950"
# --offload-compress: the gfx950 code objects are stored compressed in the fat binary (the HIP runtime inflates them when the module is
# registered): libnxsig.so 24.8 MB -> 7.5 MB
COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "--offload-compress", f"-I{os.path.join(ROOT, 'include')}"]
UNITS = [
    # (source, extra flags)
    ("host_numerics.cpp", ["-x", "hip", "-ffp-contract=off"]),  # BinaryBackend rounding: no FMA contraction
    ("api.cpp", ["-x", "hip"]),
    ("group.cpp", ["-x", "hip"]),  # multi-GPU groups: RCCL is dlopen()ed at run time, never linked
    ("kernels_generic.hip", []),
    ("kernels_nd.hip", []),
    ("kernels_wave.hip", []),
    ("kernels_wave_mel.hip", []),
    ("kernels_wave_mag.hip", []),
    ("kernels_wave_r20.hip", []),
    ("kernels_wave_rab.hip", []),   # composite fft lengths A x B (round 5): 320 / 480 / 640 / 960 + the dispatchers
    ("kernels_wave_rab_p1.hip", []),  # 100 ... 384
    ("kernels_wave_rab_p2.hip", []),  # 500 ... 900
    ("kernels_wave_rab_p3.hip", []),  # 1000 ... 1600
    ("kernels_wave_rab_p5.hip", []),  # 192 ... 1920 (12- and 48-point codelets)
    ("kernels_wave_rab_p6.hip", []),  # 882, 1764 (radix 7, round 6)
    ("kernels_wave_rab_p7.hip", []),  # 2400, 2880, 3840 (round 6)
    ("kernels_wave_rab_p4.hip", []),  # inverse only: 128 / 256 / 512 / 1024 at any hop
    ("kernels_wave_8k.hip", []),
    ("kernels_wave_rows.hip", []),
    ("kernels_wave_fir32.hip", []),
    ("kernels_wave_firlong.hip", []),  # 1 026 ... 32 769 taps: uniformly partitioned frequency-domain delay line (round 6)
    ("kernels_wave_packed.hip", []),
    ("kernels_f64.hip", []),  # the f64 / c128 tier (workgroup-per-frame kernels in double)
]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the nxsig product path cannot be built (there is no CPU fallback)")
    return exe


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))] + [
        os.path.join(ROOT, "include", "nxsig.h"),
        os.path.abspath(__file__),
    ]
    objs = []
    cc = hipcc()
    running = []  # stale translation units compile concurrently (the wave kernels dominate: ~1 min)
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [cc, f"--offload-arch={ARCH}", *COMMON, *extra, "-c", s, "-o", o]
            if verbose:
                print("[nxsig build]", " ".join(cmd), flush=True)
            running.append((cmd, o, subprocess.Popen(cmd)))
    failed = []
    for cmd, o, proc in running:
        if proc.wait() != 0:
            failed.append(cmd)
            if os.path.exists(o):
                os.remove(o)
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if force or _stale(OUT, objs):
        cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-ldl", "-lpthread", "-o", OUT]
        if verbose:
            print("[nxsig build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


def build_diag(force: bool = False, verbose: bool = True) -> str:
    """tools/libnxsig_diag.so: the no-math traffic models bench.py times beside the iSTFT / FIR kernels (`mix_ceiling`).
    Measurement infrastructure — libnxsig.so neither links nor loads it."""
    src = os.path.join(ROOT, "tools", "diag_mix.hip")
    out = os.path.join(ROOT, "tools", "libnxsig_diag.so")
    if force or _stale(out, [src]):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", out]
        if verbose:
            print("[nxsig build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_diag(force="--force" in sys.argv))
