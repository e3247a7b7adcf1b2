"""NxSignal.Convolution — lib/nx_signal/convolution.ex:38-58, :87-93, :95-218, :252-347.

`method: :direct` (the reference's default): time-domain sums on the device (nxsig_convolve_direct: one thread per output,
products accumulated in double in the BinaryBackend's window order) — exact on the integer-valued data of the reference's
tests, O(output x kernel) work, meant for short kernels.  The FFT method: a real stream against a real 1-D filter runs the overlap-save kernel (any length, batched over leading
axes); complex 1-D operands one transform of up to 2^26 points; n-D operands of equal rank the device-side fft_nd fold
(nxsig_fftconvolve_nd: transforms over the axes where neither dimension is 1, broadcast product, inverse, `centered` slice)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArgumentError, NxSignalUnsupported
from .device import default_context, device_view, is_device

_MODES = {"full": _lib.CONV_FULL, "same": _lib.CONV_SAME, "valid": _lib.CONV_VALID}


def convolve(in1, in2, ctx=None, **opts):
    allowed = {"mode": "full", "method": "direct"}
    unknown = [k for k in opts if k not in allowed]
    if unknown:
        raise ArgumentError(f"unknown keys {unknown} in convolve options, the allowed keys are: {list(allowed)}")
    o = dict(allowed)
    o.update(opts)
    if o["mode"] not in _MODES:  # convolution.ex:41-44 (mode is validated before method, quirk B12)
        raise ArgumentError(f"expected mode to be one of [:full, :same, :valid], got: {o['mode']!r}")
    if o["method"] not in ("direct", "fft"):  # :46-49
        raise ArgumentError(f"expected method to be one of [:direct, :fft], got: {o['method']!r}")
    if o["method"] == "direct":
        return _convolve_direct(in1, in2, o["mode"], ctx)
    return fftconvolve(in1, in2, ctx=ctx, mode=o["mode"])


def _convolve_direct(in1, in2, mode, ctx):
    """direct_convolve — lib/nx_signal/convolution.ex:95-218 (rank checks :97-115, :valid operand order :120-135)"""
    if is_device(in1) or is_device(in2):
        raise NxSignalUnsupported("convolve(method: :direct) takes host tensors; device-resident streams use method='fft'")
    a, b = np.asarray(in1), np.asarray(in2)
    if a.ndim != b.ndim:
        if a.ndim == 0 or b.ndim == 0:
            raise ArgumentError(f"Incompatible ranks: {{{a.ndim}, {b.ndim}}}")
        raise ArgumentError("NxSignal.convolve/3 requires both inputs to have the same rank or one of them to be a scalar, "
                            f"got {a.ndim} and {b.ndim}")
    for t in (a, b):
        if t.dtype in (np.float64, np.complex128):
            raise ArgumentError("convolve: f64 / c128 is outside this path (f32 / c64); cast explicitly")
    scalar = a.ndim == 0
    if scalar:
        a, b = a.reshape(1), b.reshape(1)
    if a.ndim > 8:
        raise NxSignalUnsupported("convolve: rank > 8")
    a_real, b_real = not np.iscomplexobj(a), not np.iscomplexobj(b)
    ac = np.ascontiguousarray(a.astype(np.float32 if a_real else np.complex64))
    bc = np.ascontiguousarray(b.astype(np.float32 if b_real else np.complex64))
    rank = ac.ndim
    s1 = (C.c_int64 * rank)(*ac.shape)
    s2 = (C.c_int64 * rank)(*bc.shape)
    osh = (C.c_int64 * rank)()
    out = np.empty([x + y - 1 for x, y in zip(ac.shape, bc.shape)], np.float32 if (a_real and b_real) else np.complex64)
    c = ctx or default_context()
    _lib.check(_lib.load().nxsig_convolve_direct(c.handle, ac.ctypes.data_as(C.c_void_p), int(a_real), s1, bc.ctypes.data_as(C.c_void_p),
                                                 int(b_real), s2, rank, _MODES[mode], out.ctypes.data_as(C.c_void_p), osh, _lib.HOST))
    shape = tuple(int(v) for v in osh)
    res = out.reshape(-1)[: int(np.prod(shape))].reshape(shape).copy()
    return res.reshape(()) if scalar else res


def correlate(in1, in2, ctx=None, **opts):
    """NxSignal.Convolution.correlate/3 — lib/nx_signal/convolution.ex:87-93: convolve(in1, conj(reverse(in2)), opts) with the
    kernel reversed along every axis; :method and :mode go to convolve (default method :direct, like the reference)."""
    k = np.asarray(in2)
    k = k[tuple(slice(None, None, -1) for _ in range(k.ndim))]
    if np.iscomplexobj(k):
        k = np.conj(k)
    return convolve(in1, np.ascontiguousarray(k), ctx=ctx, **opts)


def _fftconvolve_nd(a, b, mode, ctx):
    """equal-rank n-D operands (host): nxsig_fftconvolve_nd"""
    lib = _lib.load()
    for t in (a, b):
        if t.dtype in (np.float64, np.complex128):
            raise ArgumentError("fftconvolve: f64 / c128 is outside this path (f32 / c64); cast explicitly")
    a_real, b_real = not np.iscomplexobj(a), not np.iscomplexobj(b)
    ac = np.ascontiguousarray(a.astype(np.float32 if a_real else np.complex64))
    bc = np.ascontiguousarray(b.astype(np.float32 if b_real else np.complex64))
    rank = ac.ndim
    s1 = (C.c_int64 * rank)(*ac.shape)
    s2 = (C.c_int64 * rank)(*bc.shape)
    osh = (C.c_int64 * rank)()
    full = [x + y - 1 for x, y in zip(ac.shape, bc.shape)]
    out = np.empty(full, np.float32 if (a_real and b_real) else np.complex64)  # every mode's result fits
    c = ctx or default_context()
    _lib.check(lib.nxsig_fftconvolve_nd(c.handle, ac.ctypes.data_as(C.c_void_p), int(a_real), s1, bc.ctypes.data_as(C.c_void_p),
                                        int(b_real), s2, rank, _MODES[mode], out.ctypes.data_as(C.c_void_p), osh, _lib.HOST))
    shape = tuple(int(v) for v in osh)
    return out.reshape(-1)[: int(np.prod(shape))].reshape(shape).copy()


def _fftconvolve_f64(a, h, mode, ctx):
    """f64 tier: real operands of which at least one is f64 — both transforms in c128 (nxsig_fir_f64).  A mixed f32 / f64
    pair is computed entirely in double (the reference rounds the f32 operand's spectrum to c64 before promoting it: ours is
    the more accurate of the two, equal within the f32 operand's rounding)."""
    lib = _lib.load()
    x = np.ascontiguousarray(a.astype(np.float64))
    h64 = np.ascontiguousarray(h.astype(np.float64))
    if x.ndim == 1 and x.shape[0] < h64.shape[0]:
        if mode == "same":
            full = _fftconvolve_f64(h64, x, "full", ctx)
            start = (full.shape[0] - a.shape[0]) // 2
            return full[start:start + a.shape[0]]
        x, h64 = h64, x
    c = ctx or default_context()
    L = int(x.shape[-1])
    batch = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
    n_out = _lib.check(lib.nxsig_conv_length(L, h64.size, _MODES[mode]))
    y = np.empty(x.shape[:-1] + (n_out,), dtype=np.float64)
    _lib.check(lib.nxsig_fir_f64(c.handle, x.ctypes.data_as(C.c_void_p), L, batch, L, h64.ctypes.data_as(C.c_void_p), h64.size,
                                 _MODES[mode], y.ctypes.data_as(C.c_void_p), _lib.HOST))
    return y


def fftconvolve(in1, in2, ctx=None, **opts):
    """1-D real case of fftconvolve: the longer operand streams through HBM, the shorter one is the FIR kernel.
    in1 may carry leading batch axes (independent channels) when it is the signal."""
    allowed = {"mode": "full", "method": "direct"}  # :method accepted and ignored (quirk B12)
    unknown = [k for k in opts if k not in allowed]
    if unknown:
        raise ArgumentError(f"unknown keys {unknown} in fftconvolve options, the allowed keys are: {list(allowed)}")
    mode = opts.get("mode", "full")
    if mode not in _MODES:
        raise ArgumentError(f"expected mode to be one of [:full, :same, :valid], got: {mode!r}")
    lib = _lib.load()
    from . import _as_tensor   # Python numbers / lists follow Nx.tensor's inference (f32); np.float64 arrays are f64
    in1, in2 = _as_tensor(in1), _as_tensor(in2)
    dev = is_device(in1)
    h = np.asarray(in2) if not is_device(in2) else None
    if h is None:
        raise ArgumentError("fftconvolve: the second operand (filter taps) must be a host tensor")
    if np.iscomplexobj(h) or (not dev and np.iscomplexobj(np.asarray(in1))):
        if dev:
            raise NxSignalUnsupported("complex fftconvolve takes host tensors")
        a = np.asarray(in1)
        if a.ndim != 1 or h.ndim != 1:
            if a.ndim != h.ndim:
                raise ArgumentError("Rank of in1 and in2 must be equal.")
            return _fftconvolve_nd(a, h, mode, ctx)
        if a.dtype == np.complex128 or h.dtype == np.complex128:
            raise ArgumentError("fftconvolve: complex128 is outside this path (f32/c64); cast to complex64 explicitly")
        ac = np.ascontiguousarray(a.astype(np.complex64))
        bc = np.ascontiguousarray(h.astype(np.complex64))
        n_out = _lib.check(lib.nxsig_conv_length(ac.size, bc.size, _MODES[mode]))
        out = np.empty(n_out, dtype=np.complex64)
        c = ctx or default_context()
        _lib.check(lib.nxsig_fftconvolve_c64(c.handle, ac.ctypes.data_as(C.c_void_p), ac.size, bc.ctypes.data_as(C.c_void_p),
                                             bc.size, _MODES[mode], out.ctypes.data_as(C.c_void_p), _lib.HOST))
        return out
    if h.ndim != 1:
        if dev:
            raise NxSignalUnsupported("n-D fftconvolve takes host tensors")
        a = np.asarray(in1)
        if a.ndim != h.ndim:  # convolution.ex:295-296
            raise ArgumentError("Rank of in1 and in2 must be equal.")
        return _fftconvolve_nd(a, h, mode, ctx)
    h32 = np.ascontiguousarray(h.astype(np.float32))
    if dev:
        ptr, shape, dt = device_view(in1)
        if dt != np.float32:
            raise ArgumentError("fftconvolve: device input must be float32")
        c = ctx or getattr(in1, "ctx", None) or default_context()
        L = int(shape[-1])
        batch = int(np.prod(shape[:-1], dtype=np.int64)) if len(shape) > 1 else 1
        n_out = _lib.check(lib.nxsig_conv_length(L, h32.size, _MODES[mode]))
        y = c.empty(tuple(shape[:-1]) + (n_out,), np.float32)
        _lib.check(lib.nxsig_fir_f32(c.handle, C.c_void_p(ptr), L, batch, L, h32.ctypes.data_as(C.c_void_p), h32.size,
                                     _MODES[mode], C.c_void_p(y.ptr), _lib.DEVICE))
        return y
    a = np.asarray(in1)
    if a.ndim != 1 and a.ndim != h.ndim and a.ndim < 1:
        raise ArgumentError("Rank of in1 and in2 must be equal.")
    if a.dtype == np.float64 or h.dtype == np.float64:
        return _fftconvolve_f64(a, h, mode, ctx)
    x = np.ascontiguousarray(a.astype(np.float32))
    if x.ndim == 1 and x.shape[0] < h32.shape[0]:
        x, h32 = h32, x  # convolution commutes; keep the longer operand as the stream
        if mode == "same":
            # :same is centred on in1 (convolution.ex:304-306): compute full and slice like the reference
            full = fftconvolve(x, h32, ctx=ctx, mode="full")
            new = a.shape[0]
            start = (full.shape[0] - new) // 2
            return full[start:start + new]
    c = ctx or default_context()
    L = int(x.shape[-1])
    batch = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
    n_out = _lib.check(lib.nxsig_conv_length(L, h32.size, _MODES[mode]))
    y = np.empty(x.shape[:-1] + (n_out,), dtype=np.float32)
    _lib.check(lib.nxsig_fir_f32(c.handle, x.ctypes.data_as(C.c_void_p), L, batch, L, h32.ctypes.data_as(C.c_void_p),
                                 h32.size, _MODES[mode], y.ctypes.data_as(C.c_void_p), _lib.HOST))
    return y
