"""Index maps of the composite codelets of nx_signal_amd/csrc/small_dft.hpp against numpy (tools only, no GPU): every prime-factor
(Good-Thomas) codelet reads v[(N2 n1 + N1 n2) mod N], runs N1-point transforms over n1 and N2-point transforms over n2 with NO twiddles
in between, and writes v[(a k1 + b k2) mod N]; the Cooley-Tukey ones (dft9, dft64) put W_N^(n2 k1) between the rounds.  The maps below
are parsed from the header's comments, so a comment and its code that drift apart fail here or in the kernels' parity tests.
    python tools/check_codelet_maps.py"""
import os
import re
import sys

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nx_signal_amd", "csrc", "small_dft.hpp")
src = open(HDR).read()
rng = np.random.default_rng(1)
bad = 0
# "// 35-point DFT, prime-factor 5 x 7: n = (7 n1 + 5 n2) mod 35, k = (21 k1 + 15 k2) mod 35"
for m in re.finditer(r"// (\d+)-point DFT, prime-factor (\d+) x (\d+): n = \((\d+) n1 \+ (\d+) n2\) mod \d+, k = \((\d+) k1 \+ (\d+) k2\) mod \d+", src):
    N, N1, N2, a1, a2, b1, b2 = map(int, m.groups())
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    A = np.empty((N1, N2), complex)
    for n1 in range(N1):
        for n2 in range(N2):
            A[n1, n2] = x[(a1 * n1 + a2 * n2) % N]
    A = np.fft.fft(A, axis=0)
    A = np.fft.fft(A, axis=1)
    y = np.empty(N, complex)
    for k1 in range(N1):
        for k2 in range(N2):
            y[(b1 * k1 + b2 * k2) % N] = A[k1, k2]
    err = np.max(np.abs(y - np.fft.fft(x)))
    ok = err < 1e-10 and N1 * N2 == N and f"void dft{N}(" in src
    bad += not ok
    print(f"dft{N:<3d} {N1:2d} x {N2:2d}  in ({a1} n1 + {a2} n2)  out ({b1} k1 + {b2} k2)  max err {err:.1e}  {'ok' if ok else 'WRONG'}")
# "// 9-point DFT, Cooley-Tukey 3 x 3: n = 3 n1 + n2, k = k1 + 3 k2"
for m in re.finditer(r"// (\d+)-point DFT, Cooley-Tukey (\d+) x (\d+): n = (\d+) n1 \+ n2, k = k1 \+ (\d+) k2", src):
    N, N1, N2, s1, s2 = map(int, m.groups())
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    Y = np.fft.fft(x.reshape(N1, N2), axis=0)                      # over n1 -> k1
    Y = Y * np.exp(-2j * np.pi * np.outer(np.arange(N1), np.arange(N2)) / N)
    Y = np.fft.fft(Y, axis=1)                                      # over n2 -> k2
    y = np.empty(N, complex)
    for k1 in range(N1):
        for k2 in range(N2):
            y[k1 + s2 * k2] = Y[k1, k2]
    err = np.max(np.abs(y - np.fft.fft(x)))
    ok = err < 1e-10 and s1 == N2 and s2 == N1
    bad += not ok
    print(f"dft{N:<3d} {N1:2d} x {N2:2d}  Cooley-Tukey, twiddles W_{N}^(n2 k1)  max err {err:.1e}  {'ok' if ok else 'WRONG'}")
sys.exit(1 if bad else 0)
