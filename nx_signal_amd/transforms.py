"""NxSignal.Transforms.fft_nd / ifft_nd — lib/nx_signal/transforms.ex:5-21: a fold of row FFTs over the listed axes
(the 1-axis case is how fftconvolve reaches Nx.fft; SURVEY §8f-4 for the multi-axis fold)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArgumentError, NxSignalUnsupported
from .device import default_context


def _rows(x, inverse, K, ctx):
    """FFT of length K (zero-padded / truncated) over the last axis of a contiguous f32 / c64 host array"""
    is_real = not np.iscomplexobj(x)
    n_in = x.shape[-1]
    rows = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
    out = np.empty(x.shape[:-1] + (K,), dtype=np.complex64)
    c = ctx or default_context()
    _lib.check(_lib.load().nxsig_fft(c.handle, x.ctypes.data_as(C.c_void_p), int(is_real), rows, n_in, K, int(inverse),
                                     out.ctypes.data_as(C.c_void_p), _lib.HOST))
    return out


def _run(tensor, inverse, opts, ctx):
    """Enum.zip_reduce(axes, lengths, tensor, &Nx.fft(&3, axis: &1, length: &2)) — transforms.ex:9-11 / :18-20: one row
    FFT per listed axis, in list order; an axis other than the last is brought to the back by a host transpose."""
    unknown = [k for k in opts if k not in ("axes", "lengths")]
    if unknown:
        raise ArgumentError(f"unknown keys {unknown} in fft_nd options, the allowed keys are: ['axes', 'lengths']")
    axes = list(opts.get("axes", [-1]))
    lengths = opts.get("lengths") or [None] * len(axes)
    a = np.asarray(tensor)
    if a.ndim == 0:
        raise ArgumentError("fft_nd: expected a tensor of rank >= 1")
    if len(lengths) != len(axes):
        raise ArgumentError("fft_nd: :axes and :lengths must have the same size")
    if a.dtype in (np.float64, np.complex128):
        raise ArgumentError("fft_nd: f64/c128 is outside this path; cast to float32/complex64")
    acc = a.astype(np.complex64 if np.iscomplexobj(a) else np.float32)
    for axis, length in zip(axes, lengths):
        ax = int(axis)
        if ax < -a.ndim or ax >= a.ndim:
            raise ArgumentError(f"fft_nd: axis {axis} is out of bounds for a tensor of rank {a.ndim}")
        ax %= a.ndim
        x = np.ascontiguousarray(np.moveaxis(acc, ax, -1))
        K = int(length) if length is not None else x.shape[-1]
        if K < 1:
            raise ArgumentError("fft_nd: lengths must be positive")
        acc = np.moveaxis(_rows(x, inverse, K, ctx), -1, ax)
    return np.ascontiguousarray(acc)


def fft_nd(tensor, ctx=None, **opts):
    return _run(tensor, False, opts, ctx)


def ifft_nd(tensor, ctx=None, **opts):
    return _run(tensor, True, opts, ctx)
