"""Device-to-host rate of DeviceBuffer.numpy() (nxsig_download) into a freshly allocated array (tools only).
NXSIG_NO_PREFAULT=1 shows the plain hipMemcpy for comparison."""
import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import nx_signal_amd as S
ctx = S.Context(0)
z = ctx.empty((8, 11247, 1024), np.complex64)
for _ in range(3):
    t0 = time.perf_counter(); a = z.numpy(); dt = time.perf_counter() - t0
    print(f"download {a.nbytes/1e6:.0f} MB into a fresh array: {dt*1e3:.1f} ms = {a.nbytes/dt/1e9:.1f} GB/s")
    del a
