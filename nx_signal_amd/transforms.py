"""NxSignal.Transforms.fft_nd / ifft_nd over the last axis — lib/nx_signal/transforms.ex:5-21 (the 1-axis
case is how fftconvolve reaches Nx.fft; multi-axis folds are outside the hot path)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArgumentError, NxSignalUnsupported
from .device import default_context


def _run(tensor, inverse, opts, ctx):
    axes = opts.get("axes", [-1])
    lengths = opts.get("lengths") or [None] * len(axes)
    a = np.asarray(tensor)
    if len(axes) != 1 or axes[0] not in (-1, a.ndim - 1):
        raise NxSignalUnsupported("fft_nd: only the last axis is built (1-D hot path)")
    if a.dtype in (np.float64, np.complex128):
        raise ArgumentError("fft_nd: f64/c128 is outside this path; cast to float32/complex64")
    is_real = not np.iscomplexobj(a)
    x = np.ascontiguousarray(a.astype(np.float32 if is_real else np.complex64))
    n_in = x.shape[-1]
    K = int(lengths[0]) if lengths[0] is not None else n_in
    rows = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
    out = np.empty(x.shape[:-1] + (K,), dtype=np.complex64)
    c = ctx or default_context()
    _lib.check(_lib.load().nxsig_fft(c.handle, x.ctypes.data_as(C.c_void_p), int(is_real), rows, n_in, K, int(inverse),
                                     out.ctypes.data_as(C.c_void_p), _lib.HOST))
    return out


def fft_nd(tensor, ctx=None, **opts):
    return _run(tensor, False, opts, ctx)


def ifft_nd(tensor, ctx=None, **opts):
    return _run(tensor, True, opts, ctx)
