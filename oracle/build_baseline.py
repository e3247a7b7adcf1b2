"""Builds oracle/_build/libbb_baseline.so (the C leg of the CPU oracle) with gcc.  Test/bench infrastructure."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "bb_baseline.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libbb_baseline.so")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found")
    subprocess.check_call([gcc, "-O2", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", SRC, "-o", OUT, "-lm"])
    return OUT


if __name__ == "__main__":
    print(build(force=True))
