"""Builds a VARIANT of libnxsig.so for A/B and diagnostic runs (tools only): the listed translation units are recompiled with extra
flags, every other object comes from the regular build.
    python tools/build_variant.py tools/_ab/libnxsig_trace.so -DNXSIG_TRACE kernels_wave.hip [more units ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nx_signal_amd import build as B  # noqa: E402

out, flags, units = sys.argv[1], [a for a in sys.argv[2:] if a.startswith("-")], [a for a in sys.argv[2:] if not a.startswith("-")]
B.build(verbose=False)
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
objs = []
for src, extra in B.UNITS:
    o = os.path.join(B.OBJ, os.path.splitext(src)[0] + ".o")
    if src in units:
        o = os.path.abspath(out) + "." + os.path.splitext(src)[0] + ".o"
        subprocess.check_call([B.hipcc(), f"--offload-arch={B.ARCH}", *B.COMMON, *extra, *flags, "-c", os.path.join(B.CSRC, src), "-o", o])
    objs.append(o)
subprocess.check_call([B.hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", *objs, "-ldl", "-lpthread", "-o", out])
print(out)
