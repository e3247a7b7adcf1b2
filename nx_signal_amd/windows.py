"""NxSignal.Windows — lib/nx_signal/windows.ex.  Host-generated with the BinaryBackend rounding rules
(libnxsig host code, not device cosf): the tables are a few KB and must match the reference bit-for-bit."""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import ArgumentError


def _validate(opts, allowed, fn):
    unknown = [k for k in opts if k not in allowed]
    if unknown:
        raise ArgumentError(f"unknown keys {unknown} in {fn} options, the allowed keys are: {list(allowed)}")
    out = dict(allowed)
    out.update(opts)
    return out


def _is_f64(t):
    return t in ("f64", np.float64) or (not isinstance(t, str) and t is not None and np.dtype(t) == np.float64)


def _gen(kind, n, periodic=True, beta=0.0, eps=1.0e-7, t=None):
    """t = f64: the generator evaluated in double (`type: {:f, 64}`), else f32"""
    if not isinstance(n, (int, np.integer)):
        raise ArgumentError(f"window length must be an integer, got: {n!r}")
    if _is_f64(t):
        out = np.empty(int(n), dtype=np.float64)
        _lib.check(_lib.load().nxsig_window_f64(kind, int(n), int(bool(periodic)), float(beta), float(eps),
                                                out.ctypes.data_as(_lib.C.c_void_p)))
        return out
    if t not in (None, "f32", np.float32):
        raise ArgumentError(f"window type must be f32 or f64 (got {t!r})")
    out = np.empty(int(n), dtype=np.float32)
    _lib.check(_lib.load().nxsig_window_f32(kind, int(n), int(bool(periodic)), float(beta), float(eps),
                                            out.ctypes.data_as(_lib.C.c_void_p)))
    return out


def rectangular(n, **opts):
    """windows.ex:33-36 — default type s64 (quirk B11)."""
    o = _validate(opts, {"type": "s64"}, "rectangular")
    w = _gen(_lib.WIN_RECTANGULAR, n)
    if o["type"] in ("s64", np.int64):
        return w.astype(np.int64)
    if o["type"] in ("f32", np.float32):
        return w
    return w.astype(o["type"])


def bartlett(n, **opts):
    """windows.ex:57-78 — rejects :name (quirk B11)."""
    o = _validate(opts, {"type": "f32"}, "bartlett")
    return _gen(_lib.WIN_BARTLETT, n, t=o["type"])


def triangular(n, **opts):
    """windows.ex:98-126."""
    o = _validate(opts, {"name": None, "type": "f32"}, "triangular")
    return _gen(_lib.WIN_TRIANGULAR, n, t=o["type"])


def blackman(n, **opts):
    """windows.ex:160-202."""
    o = _validate(opts, {"name": None, "is_periodic": True, "type": "f32"}, "blackman")
    return _gen(_lib.WIN_BLACKMAN, n, o["is_periodic"], t=o["type"])


def hamming(n, **opts):
    """windows.ex:225-250."""
    o = _validate(opts, {"name": None, "is_periodic": True, "type": "f32"}, "hamming")
    return _gen(_lib.WIN_HAMMING, n, o["is_periodic"], t=o["type"])


def hann(n, **opts):
    """windows.ex:278-305."""
    o = _validate(opts, {"name": None, "is_periodic": True, "type": "f32"}, "hann")
    return _gen(_lib.WIN_HANN, n, o["is_periodic"], t=o["type"])


def kaiser(n, **opts):
    """windows.ex:341-369."""
    o = _validate(opts, {"name": None, "eps": 1.0e-7, "beta": 12.0, "is_periodic": True, "type": "f32"}, "kaiser")
    return _gen(_lib.WIN_KAISER, n, o["is_periodic"], o["beta"], o["eps"], t=o["type"])
