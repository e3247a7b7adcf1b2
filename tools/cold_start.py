"""Cold-start cost of a build of libnxsig.so (tools only): dlopen, context, first stft launch, second launch — in a FRESH process per
library, so the fat-binary registration (and, for --offload-compress builds, the inflation of the code objects) is paid each time.
    usage: python tools/cold_start.py [path-to-libnxsig.so ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, sys, time, os
sys.path.insert(0, %r)
t0 = time.perf_counter()
from nx_signal_amd import _lib
import numpy as np
t_imp = time.perf_counter()
lib = C.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, args) in _lib.SIGNATURES.items():
    f = getattr(lib, name); f.restype, f.argtypes = res, args
t_open = time.perf_counter()
ctx = C.c_void_p(); assert lib.nxsig_ctx_create(0, C.byref(ctx)) == 0
t_ctx = time.perf_counter()
hip = C.CDLL("libamdhip64.so")
n, N, hop = 480000, 1024, 256
M = (n - N) // hop + 1
x = C.c_void_p(); z = C.c_void_p(); w = C.c_void_p()
hip.hipMalloc(C.byref(x), C.c_size_t(n * 4)); hip.hipMalloc(C.byref(z), C.c_size_t(M * N * 8)); hip.hipMalloc(C.byref(w), C.c_size_t(N * 4))
hip.hipMemset(x, 0, C.c_size_t(n * 4)); hip.hipMemset(w, 0, C.c_size_t(N * 4)); hip.hipDeviceSynchronize()
p = _lib.StftParams(frame_length=N, hop=hop, fft_length=N, pad_mode=0, pad_lo=0, pad_hi=0, scaling=0, reserved=0, sampling_rate=48000.0)
mout = C.c_int64(0)
def once():
    t = time.perf_counter()
    rc = lib.nxsig_stft_f32(ctx, x, n, 1, n, w, C.byref(p), z, C.byref(mout), 1)
    assert rc == 0, rc
    assert lib.nxsig_sync(ctx) == 0
    return time.perf_counter() - t
a = once(); b = once()
print("%%-44s dlopen %%6.1f ms  ctx %%6.1f ms  first stft %%7.1f ms  second %%6.3f ms  size %%5.1f MB" %% (
    os.path.basename(sys.argv[1]), (t_open - t_imp) * 1e3, (t_ctx - t_open) * 1e3, a * 1e3, b * 1e3, os.path.getsize(sys.argv[1]) / 1e6))
''' % ROOT

for path in sys.argv[1:] or [os.path.join(ROOT, "nx_signal_amd", "libnxsig.so")]:
    for _ in range(3):
        subprocess.run([sys.executable, "-c", CHILD, path], check=False)
