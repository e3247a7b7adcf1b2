"""NxSignal.Filters.firwin/3 (lib/nx_signal/filters.ex:147-279) and the new `fir` (BASELINE config 5)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, convolution
from ._lib import ArgumentError

_WINDOWS = {
    "hamming": _lib.WIN_HAMMING, "hann": _lib.WIN_HANN, "blackman": _lib.WIN_BLACKMAN,
    "bartlett": _lib.WIN_BARTLETT, "rectangular": _lib.WIN_RECTANGULAR,
}


def firwin(num_taps, cutoff, **opts):
    """Window-method FIR design.  `cutoff` must be a list (quirk B13); window one of "hamming", "hann",
    "blackman", "bartlett", "rectangular" or ("kaiser", beta)."""
    allowed = {"window": "hamming", "pass_zero": True, "scale": True, "sampling_rate": 2.0, "type": "f32"}
    unknown = [k for k in opts if k not in allowed]
    if unknown:
        raise ArgumentError(f"unknown keys {unknown} in firwin options, the allowed keys are: {list(allowed)}")
    o = dict(allowed)
    o.update(opts)
    if not isinstance(cutoff, (list, tuple)):  # filters.ex:160-162
        raise ArgumentError(f"cutoff must be a list of frequencies, got: {cutoff!r}")
    win, beta = o["window"], 0.0
    if isinstance(win, (tuple, list)) and len(win) == 2 and win[0] == "kaiser":
        kind, beta = _lib.WIN_KAISER, float(win[1])
    elif isinstance(win, str) and win in _WINDOWS:
        kind = _WINDOWS[win]
    else:  # filters.ex:274-277
        raise ArgumentError(
            f"unknown window {win!r}, supported: :hamming, :hann, :blackman, :bartlett, :rectangular, {{:kaiser, beta}}"
        )
    f64 = o["type"] in ("f64", np.float64)
    if not f64 and o["type"] not in ("f32", np.float32):
        raise ArgumentError("firwin: type must be f32 or f64")
    cut = (C.c_double * len(cutoff))(*[float(c) for c in cutoff])
    out = np.empty(int(num_taps), dtype=np.float64 if f64 else np.float32)
    entry = _lib.load().nxsig_firwin_f64 if f64 else _lib.load().nxsig_firwin_f32
    _lib.check(entry(int(num_taps), cut, len(cutoff), kind, beta, int(bool(o["pass_zero"])),
                                            int(bool(o["scale"])), float(o["sampling_rate"]),
                                            out.ctypes.data_as(C.c_void_p)))
    return out


def fir(x, taps, mode="same", ctx=None):
    """FIR-filter a real stream (batched over leading axes) with real `taps` by overlap-save block FFT
    convolution on the GPU.  Equals Convolution.convolve(x, taps, method: :fft, mode:) of the reference
    (guides/filtering.livemd:126-128) to fp32 rounding; the reference has no streaming form."""
    return convolution.convolve(x, taps, mode=mode, method="fft", ctx=ctx)
