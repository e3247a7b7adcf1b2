"""nx_signal_amd — MI355X-native STFT / iSTFT / FIR hot path behind the NxSignal API.

Python host mirror of the reference's function-level interface (elixir-nx/nx_signal v0.3.0):

    NxSignal.stft/3, istft/3, as_windowed/2, overlap_and_add/2, fft_frequencies/2   -> this module
    NxSignal.Windows.*            -> nx_signal_amd.windows
    NxSignal.Filters.firwin/3     -> nx_signal_amd.filters  (+ the new `fir`)
    NxSignal.Convolution.*        -> nx_signal_amd.convolution (FFT method, 1-D)
    NxSignal.Waveforms.sinc/1     -> nx_signal_amd.waveforms
    NxSignal.Transforms.fft_nd    -> nx_signal_amd.transforms (last axis)

Same option names, defaults, return shapes and failure cases (ArgumentError) as the reference; atoms
become strings ("valid", "reflect", "spectrum", ...), tensors become numpy arrays (host) or DeviceBuffers
(HBM-resident, returned when the input is device-resident).  All arithmetic runs in hand-written HIP
kernels behind the C ABI in include/nxsig.h; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np

from . import _lib
from ._lib import ArgumentError, NxSignalDeviceError, NxSignalLibraryError, NxSignalUnsupported, StftParams
from .device import Context, DeviceBuffer, default_context, device_view, is_device

__all__ = [
    "stft", "istft", "as_windowed", "overlap_and_add", "fft_frequencies", "mel_filters", "stft_to_mel", "mel_spectrogram",
    "spectrum_multiply", "istft_filtered", "stft_onesided", "spectrogram",
    "Context", "DeviceBuffer", "default_context", "ArgumentError",
    "NxSignalDeviceError", "NxSignalLibraryError", "NxSignalUnsupported",
]

_PAD_ATOMS = {"valid": _lib.PAD_VALID, "reflect": _lib.PAD_REFLECT, "same": _lib.PAD_SAME}
_SCALING = {None: _lib.SCALE_NONE, "spectrum": _lib.SCALE_SPECTRUM, "psd": _lib.SCALE_PSD}


def _validate(opts: dict, allowed: dict, fn: str) -> dict:
    """Keyword.validate!/keyword! — unknown keys raise ArgumentError."""
    unknown = [k for k in opts if k not in allowed]
    if unknown:
        raise ArgumentError(f"unknown keys {unknown} in {fn} options, the allowed keys are: {list(allowed)}")
    out = dict(allowed)
    out.update(opts)
    return out


def _as_tensor(x):
    """What Nx.tensor/1 would make of a host value: numpy arrays keep their dtype (np.float64 IS f64: the f64 tier); plain Python
    numbers and (nested) lists follow Nx's inference — floats become f32, complex numbers c64, integers stay integers."""
    if isinstance(x, np.ndarray) or is_device(x):
        return x
    a = np.asarray(x)
    if a.dtype == np.float64:
        return a.astype(np.float32)
    if a.dtype == np.complex128:
        return a.astype(np.complex64)
    return a


def _host_f32(x, what: str) -> np.ndarray:
    a = np.asarray(x)
    if a.dtype == np.float32:
        return np.ascontiguousarray(a)
    if a.dtype.kind in "iub":  # integer tensors are legal inputs
        return np.ascontiguousarray(a.astype(np.float32))
    if a.dtype == np.float64:
        raise ArgumentError(
            f"{what}: float64 input would produce an f64 / c128 result in the reference; this entry point computes f32 / c64 — "
            "cast to float32 explicitly (stft, istft, as_windowed, overlap_and_add, fft_nd over the last axis and 1-D fftconvolve "
            "take float64 / complex128 and compute in double)"
        )
    raise ArgumentError(f"{what}: unsupported dtype {a.dtype}")


def _window_host(window) -> np.ndarray:
    """the window as the host array handed to the library: f32, or f64 when the caller's is (the f64 tier, include/nxsig.h)"""
    w = np.asarray(window)
    if w.ndim != 1:
        raise ArgumentError(f"window must be a rank-1 tensor, got shape {w.shape}")
    if w.dtype == np.float64:
        return np.ascontiguousarray(w)
    return np.ascontiguousarray(w.astype(np.float32))


def _pad_args(padding, fn="as_windowed"):
    if isinstance(padding, str):
        if padding not in _PAD_ATOMS:  # lib/nx_signal.ex:325-329 (":zeros" and nil raise despite the docs, B2)
            raise ArgumentError(
                "invalid padding mode specified, padding must be one of :valid, :same, or a padding configuration, "
                f"got: {padding!r}"
            )
        return _PAD_ATOMS[padding], 0, 0
    if isinstance(padding, (list, tuple)):
        ok = len(padding) == 1 and len(padding[0]) == 2 and all(isinstance(v, (int, np.integer)) for v in padding[0])
        if not ok:  # :319-323
            raise ArgumentError(
                "padding must be a list of {high, low} tuples, where each element is an integer. " f"Got: {padding!r}"
            )
        return _lib.PAD_EXPLICIT, int(padding[0][0]), int(padding[0][1])
    raise ArgumentError(
        "invalid padding mode specified, padding must be one of :valid, :same, or a padding configuration, "
        f"got: {padding!r}"
    )


def _ctx_of(obj, ctx):
    if ctx is not None:
        return ctx
    if isinstance(obj, DeviceBuffer):
        return obj.ctx
    return default_context()


def _as_ptr(arr: np.ndarray):
    return arr.ctypes.data_as(C.c_void_p)


def _resolve_fft_length(fft_length, n: int) -> int:
    if fft_length in (None, "power_of_two"):
        return int(_lib.load().nxsig_next_pow2(int(n)))
    if isinstance(fft_length, (int, np.integer)) and fft_length >= 1:
        return int(fft_length)
    raise ArgumentError(f"expected :fft_length to be a positive integer or :power_of_two, got: {fft_length!r}")


# ------------------------------------------------------------------------------------------------
def fft_frequencies(sampling_rate, **opts):
    """NxSignal.fft_frequencies/2 — lib/nx_signal.ex:154-166."""
    o = _validate(opts, {"fft_length": None, "type": "f32", "name": "frequencies", "endpoint": False}, "fft_frequencies")
    if o["fft_length"] is None:
        raise ArgumentError("missing :fft_length option")
    K = int(o["fft_length"])
    if o["type"] in ("f64", np.float64):
        out = np.empty(K, dtype=np.float64)
        _lib.check(_lib.load().nxsig_fft_frequencies_f64(float(sampling_rate), K, int(bool(o["endpoint"])), _as_ptr(out)))
        return out
    if o["type"] not in ("f32", np.float32):
        raise ArgumentError(f"fft_frequencies: type must be f32 or f64, got {o['type']!r}")
    out = np.empty(K, dtype=np.float32)
    _lib.check(_lib.load().nxsig_fft_frequencies_f32(float(sampling_rate), K, int(bool(o["endpoint"])), _as_ptr(out)))
    return out


def as_windowed(tensor, ctx: Context | None = None, **opts):
    """NxSignal.as_windowed/2 — lib/nx_signal.ex:249-364.  (..., L) -> (..., M, window_length)."""
    o = _validate(opts, {"window_length": None, "padding": "valid", "stride": 1}, "as_windowed")
    if o["window_length"] is None:
        raise ArgumentError("missing :window_length option")
    stride = o["stride"]
    if not (isinstance(stride, (int, np.integer)) and stride >= 1):  # :282-284
        raise ArgumentError(f"expected an integer >= 1 or a list of integers, got: {stride!r}")
    mode, lo, hi = _pad_args(o["padding"])
    N = int(o["window_length"])
    lib = _lib.load()
    M = C.c_int64()
    tensor = _as_tensor(tensor)
    if is_device(tensor):
        ptr, shape, dt = device_view(tensor)
        if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ArgumentError("as_windowed: device input must be float32 or float64")
        c = _ctx_of(tensor, ctx)
        L = shape[-1]
        batch = int(np.prod(shape[:-1], dtype=np.int64)) if len(shape) > 1 else 1
        m = _lib.check(lib.nxsig_num_frames(L, N, int(stride), mode, lo, hi))
        out = c.empty(shape[:-1] + (m, N), dt)
        entry = lib.nxsig_as_windowed_f64 if dt == np.dtype(np.float64) else lib.nxsig_as_windowed_f32
        _lib.check(entry(c.handle, C.c_void_p(ptr), L, batch, L, N, int(stride), mode, lo, hi, C.c_void_p(out.ptr), C.byref(M), _lib.DEVICE))
        return out
    src = np.asarray(tensor)
    if src.dtype == np.float64:   # f64 tier: 8-byte words through the same gather
        x = np.ascontiguousarray(src)
        c = _ctx_of(None, ctx)
        L = x.shape[-1]
        batch = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
        m = _lib.check(lib.nxsig_num_frames(L, N, int(stride), mode, lo, hi))
        out = np.empty(x.shape[:-1] + (m, N), dtype=np.float64)
        _lib.check(lib.nxsig_as_windowed_f64(c.handle, _as_ptr(x), L, batch, L, N, int(stride), mode, lo, hi, _as_ptr(out), C.byref(M),
                                             _lib.HOST))
        return out
    if src.dtype.kind in "iub":
        # framing is a pure gather (doctests :182-246 keep s64): 32-bit words travel through the kernel untouched, so integer
        # tensors stay EXACT (no rounding through f32); wider integers go through int32 when every value fits
        if src.size and (int(src.min()) < -2 ** 31 or int(src.max()) > 2 ** 31 - 1):
            raise ArgumentError("as_windowed: integer values beyond 32 bits are not supported by the MI355X path")
        x = np.ascontiguousarray(src.astype(np.int32)).view(np.float32)
    else:
        x = _host_f32(src, "as_windowed")
    c = _ctx_of(None, ctx)
    L = x.shape[-1]
    batch = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
    m = _lib.check(lib.nxsig_num_frames(L, N, int(stride), mode, lo, hi))
    out = np.empty(x.shape[:-1] + (m, N), dtype=np.float32)
    _lib.check(lib.nxsig_as_windowed_f32(c.handle, _as_ptr(x), L, batch, L, N, int(stride), mode, lo, hi, _as_ptr(out),
                                         C.byref(M), _lib.HOST))
    if src.dtype.kind in "iub":
        return out.view(np.int32).astype(src.dtype)
    return out


def _resolve_stft_opts(window, opts):
    """option parsing / defaults of NxSignal.stft/3 (lib/nx_signal.ex:71-85) -> (StftParams, N, hop, K)"""
    o = _validate(
        opts,
        {"overlap_length": None, "window": None, "scaling": None, "window_padding": "valid", "sampling_rate": 100,
         "fft_length": "power_of_two"},
        "stft",
    )
    w = _window_host(window)
    N = int(w.shape[0])
    if o["sampling_rate"] is None:
        raise ArgumentError("missing sampling_rate option")  # :81
    fs = float(o["sampling_rate"])
    overlap = N // 2 if o["overlap_length"] is None else int(o["overlap_length"])  # :83
    hop = N - overlap
    if o["scaling"] not in _SCALING:  # :124-126
        raise ArgumentError(f"invalid :scaling, expected one of :spectrum, :psd or nil, got: {o['scaling']!r}")
    mode, lo, hi = _pad_args(o["window_padding"])
    K = _resolve_fft_length(o["fft_length"], N)
    return StftParams(N, hop, K, mode, lo, hi, _SCALING[o["scaling"]], 0, fs), N, hop, K


def stft(data, window, ctx: Context | None = None, **opts):
    """NxSignal.stft/3 — lib/nx_signal.ex:68-130.  Returns (z c64[..., M, K], times f32[M], frequencies f32[K]).

    Defaults follow the code, not the doc: window_padding "valid", sampling_rate 100,
    fft_length "power_of_two", overlap_length div(N, 2); the :window key is accepted and ignored.

    Device-resident inputs (DeviceBuffer, torch tensors, __cuda_array_interface__ objects) return a DeviceBuffer and the call is
    ASYNCHRONOUS on the context's own HIP stream: it is ordered after earlier calls on the same context only.  Data produced
    on another stream (e.g. torch's current stream) must be complete before the call, and the result must not be consumed on
    another stream before `ctx.sync()` — or hand the library that stream with `ctx.set_stream(ptr)`.
    """
    return _stft(data, window, ctx, opts, False)


def stft_onesided(data, window, ctx: Context | None = None, **opts):
    """Extension (not in the reference API): `stft(data, window, **opts)` restricted to the bins 0 .. fft_length/2 - 1 — the slice
    the reference's own downstream code keeps for real signals (stft_to_mel, lib/nx_signal.ex:493-496; the spectrogram guide) —
    written straight from the transform (half the output traffic; the same bits as `stft(...)[0][..., :K // 2]`).
    Returns (z c64[..., M, K // 2], times f32[M], frequencies f32[K // 2])."""
    return _stft(data, window, ctx, opts, True)


def stft_packed(data, window, ctx: Context | None = None, **opts):
    """Extension (not in the reference API): the one-sided spectrum of `stft_onesided` with NOTHING of a real frame's spectrum
    lost — the imaginary part of bin 0 (zero for a real frame) carries Re X[fft_length / 2], the Nyquist bin.  The layout
    `istft_packed` inverts: together they run the reference's stft -> z * H -> istft chain (guides/filtering.livemd:137-159) at
    4 KB + 1 KB of HBM traffic per 1024-point frame instead of 8 + 2.  Returns (z c64[..., M, K // 2], times, frequencies[:K // 2])."""
    return _stft(data, window, ctx, opts, True, packed=True)


def istft_packed(data, window, ctx: Context | None = None, **opts):
    """Inverse of `stft_packed`: c64[..., M, K // 2] (packed) -> REAL f32[..., M * hop + overlap], equal to
    `istft(full Hermitian spectrum).real` to fp32 round-off.  Options as for `istft` (fft_length, if given, is the FULL length)."""
    return _istft(data, window, ctx, opts, None, packed=True)


def _stft(data, window, ctx, opts, onesided, packed=False):
    data, window = _as_tensor(data), _as_tensor(window)
    p, N, hop, K = _resolve_stft_opts(window, opts)
    w = _window_host(window)
    fs = float(p.sampling_rate)
    mode, lo, hi = p.pad_mode, p.pad_lo, p.pad_hi
    lib = _lib.load()
    M = C.c_int64()
    Kout = K // 2 if onesided else K
    if onesided and K < 2:
        raise ArgumentError("stft_onesided: fft_length >= 2 required")
    entry = lib.nxsig_stft_packed_f32 if packed else (lib.nxsig_stft_onesided_f32 if onesided else lib.nxsig_stft_f32)
    if packed and K % 2:
        raise ArgumentError("stft_packed: fft_length must be even")
    # f64 tier: f64 samples or an f64 window make the reference compute in f64 / c128 (Nx.multiply promotes, :101-102)
    ddt = np.dtype(device_view(data)[2]) if is_device(data) else np.asarray(data).dtype
    data_f64 = ddt == np.float64
    # c128 samples, or c64 samples under an f64 window (the product of :101 promotes to c128): nxsig_stft_c128
    data_c128 = ddt == np.complex128 or (ddt == np.complex64 and w.dtype == np.float64)
    if data_f64 or data_c128 or w.dtype == np.float64:
        if onesided:
            raise ArgumentError("stft_onesided / stft_packed are f32 extensions; the f64 tier returns the full c128 spectrum")
        return _stft_f64(data, w, ctx, p, N, hop, K, data_f64, cplx=data_c128)
    # complex samples (c64 IQ data): the reference frames, multiplies and transforms whatever tensor it is given (lib/nx_signal.ex:94-102):
    # one transform per frame, nxsig_stft_c64
    data_c64 = (device_view(data)[2] == np.dtype(np.complex64)) if is_device(data) else (np.asarray(data).dtype == np.complex64)
    if data_c64:
        if onesided:
            raise ArgumentError("stft_onesided / stft_packed need real samples (a complex signal's spectrum is not Hermitian)")
        entry = lib.nxsig_stft_c64
    if is_device(data):
        ptr, shape, dt = device_view(data)
        if dt != np.float32 and not data_c64:
            raise ArgumentError("stft: device input must be float32, float64 or complex64")
        c = _ctx_of(data, ctx)
        L = shape[-1]
        batch = int(np.prod(shape[:-1], dtype=np.int64)) if len(shape) > 1 else 1
        m = _lib.check(lib.nxsig_num_frames(L, N, hop, mode, lo, hi))
        z = c.empty(shape[:-1] + (m, Kout), np.complex64)
        _lib.check(entry(c.handle, C.c_void_p(ptr), L, batch, L, _as_ptr(w), C.byref(p), C.c_void_p(z.ptr),
                                      C.byref(M), _lib.DEVICE))
    else:
        x = np.ascontiguousarray(np.asarray(data)) if data_c64 else _host_f32(data, "stft")
        if x.ndim < 1:
            raise ArgumentError("stft expects a tensor of rank >= 1")
        c = _ctx_of(None, ctx)
        L = x.shape[-1]
        batch = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
        m = _lib.check(lib.nxsig_num_frames(L, N, hop, mode, lo, hi))
        z = np.empty(x.shape[:-1] + (m, Kout), dtype=np.complex64)
        _lib.check(entry(c.handle, _as_ptr(x), L, batch, L, _as_ptr(w), C.byref(p), _as_ptr(z), C.byref(M),
                                      _lib.HOST))
    times = np.empty(m, dtype=np.float32)
    _lib.check(lib.nxsig_stft_times_f32(N, fs, m, _as_ptr(times)))
    freqs = fft_frequencies(fs, fft_length=K)
    return z, times, (freqs[:Kout] if onesided else freqs)


def _stft_f64(data, w, ctx, p, N, hop, K, data_f64, cplx=False):
    """stft of f64 samples and / or with an f64 window: c128 spectrum (nxsig_stft_f64; complex samples: nxsig_stft_c128); times /
    frequencies stay f32 like the reference's (their linspace calls do not take the data type, lib/nx_signal.ex:106-111)"""
    lib = _lib.load()
    M = C.c_int64()
    fs = float(p.sampling_rate)
    mode, lo, hi = p.pad_mode, p.pad_lo, p.pad_hi
    wflag = int(w.dtype == np.float64)
    entry = lib.nxsig_stft_c128 if cplx else lib.nxsig_stft_f64
    wide_t = np.complex128 if cplx else np.float64
    if is_device(data):
        ptr, shape, ddt = device_view(data)
        if np.dtype(ddt) != np.dtype(wide_t):
            raise NxSignalUnsupported("stft: an f64 window with device-resident f32 / c64 samples is not built; widen the samples first")
        c = _ctx_of(data, ctx)
        L = shape[-1]
        batch = int(np.prod(shape[:-1], dtype=np.int64)) if len(shape) > 1 else 1
        m = _lib.check(lib.nxsig_num_frames(L, N, hop, mode, lo, hi))
        z = c.empty(shape[:-1] + (m, K), np.complex128)
        _lib.check(entry(c.handle, C.c_void_p(ptr), L, batch, L, _as_ptr(w), wflag, C.byref(p), C.c_void_p(z.ptr),
                         C.byref(M), _lib.DEVICE))
    else:
        a = np.asarray(data)
        if a.dtype.kind not in ("fiubc" if cplx else "fiub"):
            raise ArgumentError(f"stft: unsupported dtype {a.dtype}")
        x = np.ascontiguousarray(a.astype(wide_t))   # f32 / c64 / integer samples widen exactly
        if x.ndim < 1:
            raise ArgumentError("stft expects a tensor of rank >= 1")
        c = _ctx_of(None, ctx)
        L = x.shape[-1]
        batch = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
        m = _lib.check(lib.nxsig_num_frames(L, N, hop, mode, lo, hi))
        z = np.empty(x.shape[:-1] + (m, K), dtype=np.complex128)
        _lib.check(entry(c.handle, _as_ptr(x), L, batch, L, _as_ptr(w), wflag, C.byref(p), _as_ptr(z), C.byref(M), _lib.HOST))
    times = np.empty(m, dtype=np.float32)
    _lib.check(lib.nxsig_stft_times_f32(N, fs, m, _as_ptr(times)))
    return z, times, fft_frequencies(fs, fft_length=K)


def istft(data, window, ctx: Context | None = None, **opts):
    """NxSignal.istft/3 — lib/nx_signal.ex:582-638.  c64[..., M, K] -> c64[..., M*hop + overlap] (complex, B8)."""
    return _istft(data, window, ctx, opts, None)


def istft_filtered(data, h, window, ctx: Context | None = None, **opts):
    """Extension (not in the reference API): `istft(Nx.multiply(z, h), window, opts)` — the last two steps of the reference's
    STFT-domain filtering workflow (guides/filtering.livemd:141, :150-157) in one call.  Bit-identical to
    `istft(spectrum_multiply(z, h), window, **opts)`; for 1024-point frames the filter is applied inside the inverse-STFT
    kernel and the filtered spectrogram never exists in HBM.  h: c64[fft_length] (host)."""
    hh = np.ascontiguousarray(np.asarray(h).astype(np.complex64))
    if hh.ndim != 1:
        raise ArgumentError("istft_filtered: h must be a rank-1 tensor of fft_length bins")
    return _istft(data, window, ctx, opts, hh)


def _istft(data, window, ctx, opts, hh, packed=False):
    data, window = _as_tensor(data), _as_tensor(window)
    o = _validate(opts, {"fft_length": None, "overlap_length": None, "scaling": None, "sampling_rate": 1000}, "istft")
    w = _window_host(window)
    N = int(w.shape[0])
    overlap = N // 2 if o["overlap_length"] is None else int(o["overlap_length"])  # :594-601
    if o["scaling"] not in _SCALING:  # :622-624
        raise ArgumentError(f"invalid :scaling, expected one of :spectrum, :psd or nil, got: {o['scaling']!r}")
    if o["scaling"] == "psd" and o["sampling_rate"] is None:  # :605
        raise ArgumentError(":sampling_rate is mandatory if scaling is :psd")
    fs = float(o["sampling_rate"]) if o["sampling_rate"] is not None else 0.0
    if overlap >= N:  # overlap_and_add check, :692-695
        raise ArgumentError(f"overlap_length must be a number less than the window size {N}, got: {N}")
    hop = N - overlap
    lib = _lib.load()
    wide = False   # f64 tier: a c128 spectrum is inverted in c128 (Nx.ifft :609)
    if is_device(data):
        ptr, shape, dt = device_view(data)
        if dt not in (np.dtype(np.complex64), np.dtype(np.complex128)):
            raise ArgumentError("istft: device input must be complex64 or complex128")
        wide = dt == np.dtype(np.complex128)
        c = _ctx_of(data, ctx)
    else:
        zin = np.asarray(data)
        wide = zin.dtype in (np.complex128, np.float64)
        zin = np.ascontiguousarray(zin.astype(np.complex128 if wide else np.complex64))
        shape = zin.shape
        c = _ctx_of(None, ctx)
    if wide and (packed or hh is not None):
        raise ArgumentError("istft_packed / istft_filtered are f32 extensions; the f64 tier takes the full c128 spectrum")
    if w.dtype == np.float64 and not wide:
        # the reference would invert in c64 (Nx.ifft rounds to c64) and only then promote the frames to c128: not modelled
        raise NxSignalUnsupported("istft: a c64 spectrum with an f64 window is not built; pass the spectrum as complex128")
    if len(shape) < 2:
        raise ArgumentError("istft expects a tensor of shape {..., frames, frequencies}")
    Mf, Kin = int(shape[-2]), int(shape[-1])
    if packed:
        Kin *= 2   # the packed rows hold fft_length / 2 complex values
    K = _resolve_fft_length(o["fft_length"], Kin)
    if K != Kin:
        raise NxSignalUnsupported("istft: fft_length different from the spectrum's last axis (ifft pad/truncate) is not built")
    if hh is not None and int(hh.shape[0]) != K:
        raise ArgumentError("istft_filtered: h must have fft_length bins")
    batch = int(np.prod(shape[:-2], dtype=np.int64)) if len(shape) > 2 else 1
    p = StftParams(N, hop, K, 0, 0, 0, _SCALING[o["scaling"]], 0, fs)
    out_len = _lib.check(lib.nxsig_ola_length(Mf, N, hop))

    def call(zp, yp, mem):
        if wide:
            return lib.nxsig_istft_c128(c.handle, zp, Mf, batch, _as_ptr(w), int(w.dtype == np.float64), C.byref(p), yp, mem)
        if packed:
            return lib.nxsig_istft_packed_f32(c.handle, zp, Mf, batch, _as_ptr(w), C.byref(p), yp, mem)
        if hh is None:
            return lib.nxsig_istft_c64(c.handle, zp, Mf, batch, _as_ptr(w), C.byref(p), yp, mem)
        return lib.nxsig_istft_filtered_c64(c.handle, zp, Mf, batch, _as_ptr(w), C.byref(p), _as_ptr(hh), yp, mem)

    odt = np.complex128 if wide else (np.float32 if packed else np.complex64)
    if is_device(data):
        y = c.empty(tuple(shape[:-2]) + (out_len,), odt)
        _lib.check(call(C.c_void_p(ptr), C.c_void_p(y.ptr), _lib.DEVICE))
        return y
    y = np.empty(tuple(shape[:-2]) + (out_len,), dtype=odt)
    _lib.check(call(_as_ptr(zin), _as_ptr(y), _lib.HOST))
    return y


def overlap_and_add(tensor, ctx: Context | None = None, **opts):
    """NxSignal.overlap_and_add/2 — lib/nx_signal.ex:684-736.  {..., M, N} -> {..., M*hop + overlap}."""
    o = _validate(opts, {"overlap_length": None, "type": None}, "overlap_and_add")
    if o["overlap_length"] is None:
        raise ArgumentError("missing :overlap_length option")
    overlap = int(o["overlap_length"])
    lib = _lib.load()
    tensor = _as_tensor(tensor)
    dev = is_device(tensor)
    if dev:
        ptr, shape, dt = device_view(tensor)
        c = _ctx_of(tensor, ctx)
        src_dtype = dt
    else:
        src = np.asarray(tensor)
        src_dtype = src.dtype
        if src.dtype in (np.float64, np.complex128):
            arr = np.ascontiguousarray(src)   # f64 tier
        elif src.dtype.kind == "c":
            arr = np.ascontiguousarray(src.astype(np.complex64))
        else:
            if src.dtype.kind in "iu" and src.size and src.ndim >= 2:
                # the reference adds integers exactly; here sums are formed in double and stored as f32, exact only below 2^24:
                # refuse loudly instead of returning rounded integers
                cover = -(-int(src.shape[-1]) // max(int(src.shape[-1]) - overlap, 1))
                if int(np.abs(src).max()) * cover >= 2 ** 24:
                    raise ArgumentError("overlap_and_add: integer sums beyond 2^24 cannot be represented exactly by the f32 "
                                        "MI355X path; cast to a float type explicitly")
            arr = _host_f32(src, "overlap_and_add")
        shape, dt = arr.shape, arr.dtype
    if len(shape) < 2:
        raise ArgumentError("overlap_and_add expects a tensor of shape {..., M, N}")
    Mf, N = int(shape[-2]), int(shape[-1])
    if overlap >= N:  # :692-695 — the message prints the window size twice (quirk B10)
        raise ArgumentError(f"overlap_length must be a number less than the window size {N}, got: {N}")
    if not dev:
        c = _ctx_of(None, ctx)
    comps = 2 if np.dtype(dt).kind == "c" else 1
    entry = lib.nxsig_overlap_and_add_f64 if np.dtype(dt) in (np.dtype(np.float64), np.dtype(np.complex128)) else lib.nxsig_overlap_and_add
    batch = int(np.prod(shape[:-2], dtype=np.int64)) if len(shape) > 2 else 1
    out_len = Mf * (N - overlap) + overlap
    out_shape = tuple(shape[:-2]) + (out_len,)
    if dev:
        out = c.empty(out_shape, dt)
        _lib.check(entry(c.handle, C.c_void_p(ptr), Mf, batch, N, overlap, comps, C.c_void_p(out.ptr), _lib.DEVICE))
        return out
    out = np.empty(out_shape, dtype=dt)
    _lib.check(entry(c.handle, _as_ptr(arr), Mf, batch, N, overlap, comps, _as_ptr(out), _lib.HOST))
    target = o["type"] if o["type"] is not None else src_dtype  # code default: the input type (:685, B10)
    return out.astype(target) if np.dtype(target) != out.dtype else out


def mel_filters(fft_length, mel_bins, sampling_rate, **opts):
    """NxSignal.mel_filters/4 — lib/nx_signal.ex:397-445.  f32[mels: mel_bins][frequencies: fft_length]."""
    o = _validate(opts, {"max_mel": 3016, "mel_frequency_spacing": 200 / 3, "type": "f32"}, "mel_filters")
    if o["type"] not in ("f32", np.float32):
        raise ArgumentError("mel_filters: only type f32 is built")
    return _mel_filters_table(int(fft_length), int(mel_bins), float(sampling_rate), float(o["max_mel"]),
                              float(o["mel_frequency_spacing"])).copy()


@functools.lru_cache(maxsize=16)
def _mel_filters_table(fft_length: int, mel_bins: int, sampling_rate: float, max_mel: float, spacing: float) -> np.ndarray:
    """the filterbank is a pure function of its five parameters: generated once per combination (read-only copy)"""
    out = np.empty((mel_bins, fft_length), dtype=np.float32)
    _lib.check(_lib.load().nxsig_mel_filters_f32(fft_length, mel_bins, sampling_rate, max_mel, spacing, _as_ptr(out)))
    out.setflags(write=False)
    return out


def _mel_filters_for(K, mb, fs, fopts):
    o = _validate(dict(fopts), {"max_mel": 3016, "mel_frequency_spacing": 200 / 3}, "mel_filters")
    return _mel_filters_table(int(K), int(mb), float(fs), float(o["max_mel"]), float(o["mel_frequency_spacing"]))


def stft_to_mel(z, sampling_rate, ctx: Context | None = None, **opts):
    """NxSignal.stft_to_mel/3 — lib/nx_signal.ex:486-513.  c64[..., frames, K] -> f32[..., frames, mel_bins]
    (the global maximum of the reference's `reduce_max` runs over the whole tensor)."""
    o = _validate(opts, {"fft_length": None, "mel_bins": 128, "max_mel": None, "mel_frequency_spacing": None, "type": "f32"},
                  "stft_to_mel")
    if o["fft_length"] is None:
        raise ArgumentError("missing :fft_length option")
    K, mb = int(o["fft_length"]), int(o["mel_bins"])
    fopts = {k: o[k] for k in ("max_mel", "mel_frequency_spacing") if o[k] is not None}
    filt = _mel_filters_for(K, mb, sampling_rate, fopts)
    lib = _lib.load()
    if is_device(z):
        ptr, shape, dt = device_view(z)
        if dt != np.complex64:
            raise ArgumentError("stft_to_mel: device input must be complex64")
        if shape[-1] != K:
            raise ArgumentError("stft_to_mel: the last axis must be fft_length")
        c = _ctx_of(z, ctx)
        rows = int(np.prod(shape[:-1], dtype=np.int64))
        out = c.empty(tuple(shape[:-1]) + (mb,), np.float32)
        _lib.check(lib.nxsig_stft_to_mel(c.handle, C.c_void_p(ptr), rows, K, mb, _as_ptr(filt), C.c_void_p(out.ptr), _lib.DEVICE))
        return out
    zin = np.ascontiguousarray(np.asarray(z).astype(np.complex64))
    if zin.shape[-1] != K:
        raise ArgumentError("stft_to_mel: the last axis must be fft_length")
    c = _ctx_of(None, ctx)
    rows = int(np.prod(zin.shape[:-1], dtype=np.int64))
    out = np.empty(zin.shape[:-1] + (mb,), dtype=np.float32)
    _lib.check(lib.nxsig_stft_to_mel(c.handle, _as_ptr(zin), rows, K, mb, _as_ptr(filt), _as_ptr(out), _lib.HOST))
    return out


def spectrum_multiply(z, h, ctx: Context | None = None):
    """The `Nx.multiply(z, hfft)` step of the reference's STFT-domain filtering workflow (guides/filtering.livemd:137-159:
    stft -> z * H -> istft), kept on the device.  z c64[..., frames, K] (numpy or DeviceBuffer), h c64[K]
    (host) -> same kind and shape as z.  Components are computed in double and rounded once like Nx.BinaryBackend."""
    hh = np.ascontiguousarray(np.asarray(h).astype(np.complex64))
    if hh.ndim != 1:
        raise ArgumentError("spectrum_multiply: h must be a rank-1 tensor of fft_length bins")
    K = int(hh.shape[0])
    lib = _lib.load()
    if is_device(z):
        ptr, shape, dt = device_view(z)
        if dt != np.complex64 or shape[-1] != K:
            raise ArgumentError("spectrum_multiply: z must be complex64 with fft_length bins on the last axis")
        c = _ctx_of(z, ctx)
        out = c.empty(tuple(shape), np.complex64)
        rows = int(np.prod(shape[:-1], dtype=np.int64))
        _lib.check(lib.nxsig_spectrum_mul_c64(c.handle, C.c_void_p(ptr), rows, K, _as_ptr(hh), C.c_void_p(out.ptr), _lib.DEVICE))
        return out
    zin = np.ascontiguousarray(np.asarray(z).astype(np.complex64))
    if zin.shape[-1] != K:
        raise ArgumentError("spectrum_multiply: z must have fft_length bins on the last axis")
    c = _ctx_of(None, ctx)
    out = np.empty_like(zin)
    rows = int(np.prod(zin.shape[:-1], dtype=np.int64))
    _lib.check(lib.nxsig_spectrum_mul_c64(c.handle, _as_ptr(zin), rows, K, _as_ptr(hh), _as_ptr(out), _lib.HOST))
    return out


def mel_spectrogram(data, window, ctx: Context | None = None, **opts):
    """Extension (not in the reference API): `stft_to_mel(stft(data, window, ...)[0], sampling_rate, ...)` fused in one
    kernel — the spectrum never goes to HBM.  Takes the stft options plus :mel_bins (128), :max_mel,
    :mel_frequency_spacing.  Returns f32[..., frames, mel_bins], equal to the two-step result to fp32 rounding."""
    mel_keys = {"mel_bins": 128, "max_mel": None, "mel_frequency_spacing": None}
    mo = {k: opts.pop(k) for k in list(opts) if k in mel_keys}
    o = _validate(
        opts,
        {"overlap_length": None, "window": None, "scaling": None, "window_padding": "valid", "sampling_rate": 100,
         "fft_length": "power_of_two"},
        "mel_spectrogram",
    )
    w = _window_host(window)
    N = int(w.shape[0])
    fs = float(o["sampling_rate"])
    overlap = N // 2 if o["overlap_length"] is None else int(o["overlap_length"])
    hop = N - overlap
    if o["scaling"] not in _SCALING:
        raise ArgumentError(f"invalid :scaling, expected one of :spectrum, :psd or nil, got: {o['scaling']!r}")
    mode, lo, hi = _pad_args(o["window_padding"])
    K = _resolve_fft_length(o["fft_length"], N)
    mb = int(mo.get("mel_bins", 128))
    fopts = {k: mo[k] for k in ("max_mel", "mel_frequency_spacing") if mo.get(k) is not None}
    filt = _mel_filters_for(K, mb, fs, fopts)
    p = StftParams(N, hop, K, mode, lo, hi, _SCALING[o["scaling"]], 0, fs)
    lib = _lib.load()
    M = C.c_int64()
    if is_device(data):
        ptr, shape, dt = device_view(data)
        if dt != np.float32:
            raise ArgumentError("mel_spectrogram: device input must be float32")
        c = _ctx_of(data, ctx)
        L = shape[-1]
        batch = int(np.prod(shape[:-1], dtype=np.int64)) if len(shape) > 1 else 1
        m = _lib.check(lib.nxsig_num_frames(L, N, hop, mode, lo, hi))
        out = c.empty(tuple(shape[:-1]) + (m, mb), np.float32)
        _lib.check(lib.nxsig_stft_mel_f32(c.handle, C.c_void_p(ptr), L, batch, L, _as_ptr(w), C.byref(p), mb, _as_ptr(filt),
                                          C.c_void_p(out.ptr), C.byref(M), _lib.DEVICE))
        return out
    x = _host_f32(data, "mel_spectrogram")
    c = _ctx_of(None, ctx)
    L = x.shape[-1]
    batch = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
    m = _lib.check(lib.nxsig_num_frames(L, N, hop, mode, lo, hi))
    out = np.empty(x.shape[:-1] + (m, mb), dtype=np.float32)
    _lib.check(lib.nxsig_stft_mel_f32(c.handle, _as_ptr(x), L, batch, L, _as_ptr(w), C.byref(p), mb, _as_ptr(filt), _as_ptr(out),
                                      C.byref(M), _lib.HOST))
    return out


def _stft_times(m: int, N: int, fs: float) -> np.ndarray:
    times = np.empty(m, dtype=np.float32)
    _lib.check(_lib.load().nxsig_stft_times_f32(N, fs, m, _as_ptr(times)))
    return times


_MAG_KINDS = {"magnitude": _lib.MAG_ABS, "power": _lib.MAG_POWER, "dbfs": _lib.MAG_DBFS}


def spectrogram(data, window, ctx: Context | None = None, **opts):
    """Extension (not in the reference API): the magnitude spectrogram guides/spectrogram.livemd:76-92
    derives from `NxSignal.stft/3` — `Nx.abs(s)` of the bins below fft_length / 2, optionally in dBFS
    (`20 * log(|s| / reduce_max(|s|)) / log(10)`) — fused with the STFT so the complex spectrum never goes to HBM.
    Takes the stft options plus :kind ("magnitude" (default), "power", "dbfs").
    Returns {spec f32[..., frames, fft_length/2], t, f[:fft_length/2]}."""
    kind = opts.pop("kind", "magnitude")
    if kind not in _MAG_KINDS:
        raise ArgumentError(f"invalid :kind, expected one of {list(_MAG_KINDS)}, got: {kind!r}")
    o = _validate(
        opts,
        {"overlap_length": None, "window": None, "scaling": None, "window_padding": "valid", "sampling_rate": 100,
         "fft_length": "power_of_two"},
        "spectrogram",
    )
    w = _window_host(window)
    N = int(w.shape[0])
    fs = float(o["sampling_rate"])
    overlap = N // 2 if o["overlap_length"] is None else int(o["overlap_length"])
    hop = N - overlap
    if o["scaling"] not in _SCALING:
        raise ArgumentError(f"invalid :scaling, expected one of :spectrum, :psd or nil, got: {o['scaling']!r}")
    mode, lo, hi = _pad_args(o["window_padding"])
    K = _resolve_fft_length(o["fft_length"], N)
    half = K // 2
    p = StftParams(N, hop, K, mode, lo, hi, _SCALING[o["scaling"]], 0, fs)
    lib = _lib.load()
    M = C.c_int64()
    f = fft_frequencies(fs, fft_length=K)[:half]
    if is_device(data):
        ptr, shape, dt = device_view(data)
        if dt != np.float32:
            raise ArgumentError("spectrogram: device input must be float32")
        c = _ctx_of(data, ctx)
        L = shape[-1]
        batch = int(np.prod(shape[:-1], dtype=np.int64)) if len(shape) > 1 else 1
        m = _lib.check(lib.nxsig_num_frames(L, N, hop, mode, lo, hi))
        out = c.empty(tuple(shape[:-1]) + (m, half), np.float32)
        _lib.check(lib.nxsig_stft_magnitude_f32(c.handle, C.c_void_p(ptr), L, batch, L, _as_ptr(w), C.byref(p), _MAG_KINDS[kind],
                                                C.c_void_p(out.ptr), C.byref(M), _lib.DEVICE))
        return out, _stft_times(m, N, fs), f
    x = _host_f32(data, "spectrogram")
    c = _ctx_of(None, ctx)
    L = x.shape[-1]
    batch = int(np.prod(x.shape[:-1], dtype=np.int64)) if x.ndim > 1 else 1
    m = _lib.check(lib.nxsig_num_frames(L, N, hop, mode, lo, hi))
    out = np.empty(x.shape[:-1] + (m, half), dtype=np.float32)
    _lib.check(lib.nxsig_stft_magnitude_f32(c.handle, _as_ptr(x), L, batch, L, _as_ptr(w), C.byref(p), _MAG_KINDS[kind], _as_ptr(out),
                                            C.byref(M), _lib.HOST))
    return out, _stft_times(m, N, fs), f


from . import convolution, filters, transforms, waveforms, windows  # noqa: E402,F401
