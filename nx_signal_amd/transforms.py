"""NxSignal.Transforms.fft_nd / ifft_nd — lib/nx_signal/transforms.ex:5-21: a fold of row FFTs over the listed axes
(the 1-axis case is how fftconvolve reaches Nx.fft; SURVEY §8f-4 for the multi-axis fold).  The whole fold runs in HBM
(nxsig_fft_nd: tiled transpose kernels bring an axis to the back and return it; four-step / Bluestein rows beyond the
LDS-resident sizes), for host tensors and for device-resident ones alike."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArgumentError
from .device import DeviceBuffer, default_context, device_view, is_device


def _run(tensor, inverse, opts, ctx):
    """Enum.zip_reduce(axes, lengths, tensor, &Nx.fft(&3, axis: &1, length: &2)) — transforms.ex:9-11 / :18-20"""
    unknown = [k for k in opts if k not in ("axes", "lengths")]
    if unknown:
        raise ArgumentError(f"unknown keys {unknown} in fft_nd options, the allowed keys are: ['axes', 'lengths']")
    axes = list(opts.get("axes", [-1]))
    lengths = opts.get("lengths") or [None] * len(axes)
    if len(lengths) != len(axes):
        raise ArgumentError("fft_nd: :axes and :lengths must have the same size")
    dev = is_device(tensor)
    if dev:
        ptr, shape, dt = device_view(tensor)
        c = ctx or getattr(tensor, "ctx", None) or default_context()
        if dt not in (np.dtype(np.float32), np.dtype(np.complex64)):
            raise ArgumentError("fft_nd: device input must be float32 or complex64")
        is_real = dt == np.dtype(np.float32)
    else:
        from . import _as_tensor   # Python numbers / lists follow Nx.tensor's inference (f32 / c64)
        a = np.asarray(_as_tensor(tensor))
        if a.dtype in (np.float64, np.complex128):
            return _run_f64(a, inverse, axes, lengths, ctx)
        a = np.ascontiguousarray(a.astype(np.complex64 if np.iscomplexobj(a) else np.float32))
        shape, is_real = a.shape, not np.iscomplexobj(a)
        c = ctx or default_context()
    rank = len(shape)
    if rank == 0:
        raise ArgumentError("fft_nd: expected a tensor of rank >= 1")
    out_shape = list(shape)
    ax_n, len_n = [], []
    for axis, length in zip(axes, lengths):
        ax = int(axis)
        if ax < -rank or ax >= rank:
            raise ArgumentError(f"fft_nd: axis {axis} is out of bounds for a tensor of rank {rank}")
        ax %= rank
        K = int(length) if length is not None else out_shape[ax]
        if K < 1:
            raise ArgumentError("fft_nd: lengths must be positive")
        out_shape[ax] = K
        ax_n.append(ax)
        len_n.append(K)
    lib = _lib.load()
    sh = (C.c_int64 * rank)(*[int(s) for s in shape])
    axs = (C.c_int32 * max(len(ax_n), 1))(*ax_n)
    lns = (C.c_int64 * max(len(len_n), 1))(*len_n)
    if dev:
        out = c.empty(tuple(out_shape), np.complex64)
        _lib.check(lib.nxsig_fft_nd(c.handle, C.c_void_p(ptr), int(is_real), sh, rank, axs, lns, len(ax_n), int(inverse),
                                    C.c_void_p(out.ptr), _lib.DEVICE))
        return out
    out = np.empty(tuple(out_shape), np.complex64)
    _lib.check(lib.nxsig_fft_nd(c.handle, a.ctypes.data_as(C.c_void_p), int(is_real), sh, rank, axs, lns, len(ax_n), int(inverse),
                                out.ctypes.data_as(C.c_void_p), _lib.HOST))
    return out


def _run_f64(a, inverse, axes, lengths, ctx):
    """f64 tier (nxsig_fft_c128): Nx.fft / Nx.ifft in c128 over the LAST axis; other axes are not built in double"""
    if a.ndim == 0:
        raise ArgumentError("fft_nd: expected a tensor of rank >= 1")
    if len(axes) != 1 or int(axes[0]) % a.ndim != a.ndim - 1:
        raise _lib.NxSignalUnsupported("fft_nd: f64 / c128 tensors are transformed over the last axis only")
    a = np.ascontiguousarray(a)
    n_in = int(a.shape[-1])
    K = int(lengths[0]) if lengths[0] is not None else n_in
    if K < 1:
        raise ArgumentError("fft_nd: lengths must be positive")
    rows = int(np.prod(a.shape[:-1], dtype=np.int64)) if a.ndim > 1 else 1
    c = ctx or default_context()
    out = np.empty(a.shape[:-1] + (K,), np.complex128)
    _lib.check(_lib.load().nxsig_fft_c128(c.handle, a.ctypes.data_as(C.c_void_p), int(a.dtype == np.float64), rows, n_in, K, int(inverse),
                                          out.ctypes.data_as(C.c_void_p), _lib.HOST))
    return out


def fft_nd(tensor, ctx=None, **opts):
    return _run(tensor, False, opts, ctx)


def ifft_nd(tensor, ctx=None, **opts):
    return _run(tensor, True, opts, ctx)
