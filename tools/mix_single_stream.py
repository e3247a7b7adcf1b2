import ctypes as C, os, sys, numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT)
import nx_signal_amd as S
from nx_signal_amd import _lib
d=C.CDLL(os.path.join(ROOT,"tools","libnxsig_diag.so"))
d.nxdiag_stft_mix.argtypes=[C.c_void_p]*4+[C.c_long,C.c_long,C.c_int,C.c_int]
ctx=S.Context(0); lib=_lib.load()
L=2880000; M=(L-1024)//256+1
xd=ctx.to_device(np.random.default_rng(1).standard_normal((1,L)).astype(np.float32))
zd=ctx.empty((1,M,1024),np.complex64)
tab=ctx.to_device(np.linspace(0.5,1.5,3072,dtype=np.float32))
lib.nxsig_get_stream.restype=C.c_void_p
st=C.c_void_p(lib.nxsig_get_stream(ctx.handle))
w=S.windows.hann(1024); p=_lib.StftParams(1024,256,1024,0,0,0,0,0,48000.0)
def t(fn,reps=200):
    for _ in range(20): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop()/reps*1e3
for rnd in range(3):
    for ppw in (1,2,3,4):
        print(f"no-math model, 4-wave workgroups, {ppw} pairs per wave: {t(lambda: d.nxdiag_stft_mix(st,C.c_void_p(xd.ptr),C.c_void_p(zd.ptr),C.c_void_p(tab.ptr),1,L,256,ppw)):.2f} us")
    k=lambda: lib.nxsig_stft_f32(ctx.handle,C.c_void_p(xd.ptr),L,1,L,w.ctypes.data_as(C.c_void_p),C.byref(p),C.c_void_p(zd.ptr),None,1)
    print(f"kernel (one-round geometry): {t(k):.2f} us")
    ctx.set_tuning("WAVE_SMALL_W",0); print(f"kernel (many-round geometry): {t(k):.2f} us"); ctx.clear_tuning("WAVE_SMALL_W")
