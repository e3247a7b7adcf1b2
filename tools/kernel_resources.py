"""Per-kernel register / scratch / LDS usage of one translation unit, from hipcc's kernel-resource-usage remarks.

    python tools/kernel_resources.py nx_signal_amd/csrc/kernels_wave.hip [name-filter]

Compiles the unit for gfx950 (no GPU needed) with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
demangled name, VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include",
               "-Wno-unused-function", "-c", src, "-o", os.path.join(td, "o.o"), "-Rpass-analysis=kernel-resource-usage"]
        txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    rows = []
    for b in blocks:
        name = b.split()[0]

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1

        rows.append((name, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    names = [r[0] for r in rows]
    if filt:
        names = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for r, d in zip(rows, names):
        if flt in d:
            print(f"{d[:150]:150s} vgpr {r[1]:4d} agpr {r[2]:4d} scratch {r[3]:5d} occ {r[4]}")


if __name__ == "__main__":
    main()
