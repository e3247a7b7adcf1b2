"""NxSignal.Waveforms.sinc — lib/nx_signal/waveforms.ex:451-457 (the only waveform on the FIR path)."""
from __future__ import annotations

import numpy as np

from . import _lib


def sinc(t):
    a = t if isinstance(t, np.ndarray) else np.asarray(t, dtype=np.float32)   # Nx.tensor(number / list) is f32; an np.float64 array is f64
    if a.dtype == np.float64:   # f64 tensor: evaluated in double (the pi() of the reference's defn stays the f32 constant)
        a = np.ascontiguousarray(a)
        out = np.empty_like(a)
        _lib.check(_lib.load().nxsig_sinc_f64(a.ctypes.data_as(_lib.C.c_void_p), a.size, out.ctypes.data_as(_lib.C.c_void_p)))
        return out
    a = np.ascontiguousarray(np.asarray(t, dtype=np.float32))
    out = np.empty_like(a)
    _lib.check(_lib.load().nxsig_sinc_f32(a.ctypes.data_as(_lib.C.c_void_p), a.size, out.ctypes.data_as(_lib.C.c_void_p)))
    return out
