"""ctypes binding of libnxsig.so — the ONLY way the Python host mirror reaches the GPU.

There is deliberately no CPU fallback: if the shared library is missing or no GPU is present the
product path raises (NxSignalLibraryError / NxSignalDeviceError).  Signatures mirror include/nxsig.h.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnxsig.so")

# status codes (include/nxsig.h)
OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE, ERR_OOM = 0, -1, -2, -3, -4, -5
HOST, DEVICE = 0, 1
PAD_VALID, PAD_REFLECT, PAD_SAME, PAD_EXPLICIT = 0, 1, 2, 3
SCALE_NONE, SCALE_SPECTRUM, SCALE_PSD = 0, 1, 2
MAG_ABS, MAG_POWER, MAG_DBFS = 0, 1, 2
WIN_RECTANGULAR, WIN_BARTLETT, WIN_TRIANGULAR, WIN_BLACKMAN, WIN_HAMMING, WIN_HANN, WIN_KAISER = range(7)
CONV_FULL, CONV_SAME, CONV_VALID = 0, 1, 2
SHARD_CHANNELS, SHARD_FRAMES = 0, 1


class ArgumentError(ValueError):
    """Mirror of Elixir's ArgumentError, which the reference raises for every invalid option."""


class NxSignalLibraryError(RuntimeError):
    """libnxsig.so is missing / not loadable (build it: python -m nx_signal_amd.build)."""


class NxSignalDeviceError(RuntimeError):
    """HIP failure or no GPU: the hot path has no CPU fallback."""


class NxSignalUnsupported(NotImplementedError):
    """valid in the reference but not built in this implementation yet."""


class StftParams(C.Structure):
    _fields_ = [
        ("frame_length", C.c_int32),
        ("hop", C.c_int32),
        ("fft_length", C.c_int32),
        ("pad_mode", C.c_int32),
        ("pad_lo", C.c_int64),
        ("pad_hi", C.c_int64),
        ("scaling", C.c_int32),
        ("reserved", C.c_int32),
        ("sampling_rate", C.c_double),
    ]


_p = C.c_void_p
_i32, _i64, _f64, _sz = C.c_int32, C.c_int64, C.c_double, C.c_size_t
_pf = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/nxsig.h declares is listed here (tests check both ways)
SIGNATURES = {
    "nxsig_abi_version": (C.c_int, []),
    "nxsig_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "nxsig_ctx_create": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "nxsig_ctx_destroy": (None, [_p]),
    "nxsig_ctx_set_tuning": (C.c_int, [_p, C.c_char_p, _i32]),
    "nxsig_ctx_get_tuning": (C.c_int, [_p, C.c_char_p, C.POINTER(_i32), C.POINTER(_i32)]),
    "nxsig_ctx_clear_tuning": (C.c_int, [_p, C.c_char_p]),
    "nxsig_last_error": (C.c_char_p, []),
    "nxsig_last_dispatch": (C.c_char_p, []),
    "nxsig_ctx_last_dispatch": (_i32, [_p, C.c_char_p, _sz]),
    "nxsig_device_name": (C.c_int, [_p, C.c_char_p, _sz]),
    "nxsig_alloc": (C.c_int, [_p, _sz, C.POINTER(_p)]),
    "nxsig_free": (C.c_int, [_p, _p]),
    "nxsig_upload": (C.c_int, [_p, _p, _p, _sz]),
    "nxsig_download": (C.c_int, [_p, _p, _p, _sz]),
    "nxsig_sync": (C.c_int, [_p]),
    "nxsig_set_stream": (C.c_int, [_p, _p]),
    "nxsig_get_stream": (_p, [_p]),
    "nxsig_timer_start": (C.c_int, [_p]),
    "nxsig_timer_stop": (C.c_int, [_p, _pf]),
    "nxsig_next_pow2": (_i32, [_i32]),
    "nxsig_num_frames": (_i64, [_i64, _i32, _i32, _i32, _i64, _i64]),
    "nxsig_ola_length": (_i64, [_i64, _i32, _i32]),
    "nxsig_conv_length": (_i64, [_i64, _i64, _i32]),
    "nxsig_window_f32": (C.c_int, [_i32, _i32, _i32, _f64, _f64, _p]),
    "nxsig_sinc_f32": (C.c_int, [_p, _i64, _p]),
    "nxsig_firwin_f32": (C.c_int, [_i32, C.POINTER(_f64), _i32, _i32, _f64, _i32, _i32, _f64, _p]),
    "nxsig_fft_frequencies_f32": (C.c_int, [_f64, _i32, _i32, _p]),
    "nxsig_stft_times_f32": (C.c_int, [_i32, _f64, _i64, _p]),
    "nxsig_stft_f32": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, C.POINTER(StftParams), _p, C.POINTER(_i64), _i32]),
    "nxsig_stft_c64": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, C.POINTER(StftParams), _p, C.POINTER(_i64), _i32]),
    "nxsig_stft_onesided_f32": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, C.POINTER(StftParams), _p, C.POINTER(_i64), _i32]),
    "nxsig_istft_c64": (C.c_int, [_p, _p, _i64, _i32, _p, C.POINTER(StftParams), _p, _i32]),
    "nxsig_stft_packed_f32": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, C.POINTER(StftParams), _p, C.POINTER(_i64), _i32]),
    "nxsig_istft_packed_f32": (C.c_int, [_p, _p, _i64, _i32, _p, C.POINTER(StftParams), _p, _i32]),
    "nxsig_istft_filtered_c64": (C.c_int, [_p, _p, _i64, _i32, _p, C.POINTER(StftParams), _p, _p, _i32]),
    "nxsig_as_windowed_f32": (C.c_int, [_p, _p, _i64, _i32, _i64, _i32, _i32, _i32, _i64, _i64, _p, C.POINTER(_i64), _i32]),
    "nxsig_overlap_and_add": (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _i32, _p, _i32]),
    "nxsig_fft": (C.c_int, [_p, _p, _i32, _i64, _i32, _i32, _i32, _p, _i32]),
    "nxsig_fir_f32": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _i32, _i32, _p, _i32]),
    "nxsig_fftconvolve_c64": (C.c_int, [_p, _p, _i64, _p, _i64, _i32, _p, _i32]),
    "nxsig_mel_filters_f32": (C.c_int, [_i32, _i32, _f64, _f64, _f64, _p]),
    "nxsig_stft_to_mel": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p, _i32]),
    "nxsig_spectrum_mul_c64": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _i32]),
    "nxsig_stft_magnitude_f32": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, C.POINTER(StftParams), _i32, _p, C.POINTER(_i64), _i32]),
    "nxsig_fft_nd": (C.c_int, [_p, _p, _i32, C.POINTER(_i64), _i32, C.POINTER(_i32), C.POINTER(_i64), _i32, _i32, _p, _i32]),
    "nxsig_fftconvolve_nd": (C.c_int, [_p, _p, _i32, C.POINTER(_i64), _p, _i32, C.POINTER(_i64), _i32, _i32, _p, C.POINTER(_i64), _i32]),
    "nxsig_convolve_direct": (C.c_int, [_p, _p, _i32, C.POINTER(_i64), _p, _i32, C.POINTER(_i64), _i32, _i32, _p, C.POINTER(_i64), _i32]),
    "nxsig_fir_slice_f32": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _i32, _i64, _i64, _p, _i32]),
    # f64 / c128 tier
    "nxsig_window_f64": (C.c_int, [_i32, _i32, _i32, _f64, _f64, _p]),
    "nxsig_sinc_f64": (C.c_int, [_p, _i64, _p]),
    "nxsig_firwin_f64": (C.c_int, [_i32, C.POINTER(_f64), _i32, _i32, _f64, _i32, _i32, _f64, _p]),
    "nxsig_fft_frequencies_f64": (C.c_int, [_f64, _i32, _i32, _p]),
    "nxsig_stft_f64": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _i32, C.POINTER(StftParams), _p, C.POINTER(_i64), _i32]),
    "nxsig_stft_c128": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _i32, C.POINTER(StftParams), _p, C.POINTER(_i64), _i32]),
    "nxsig_istft_c128": (C.c_int, [_p, _p, _i64, _i32, _p, _i32, C.POINTER(StftParams), _p, _i32]),
    "nxsig_fft_c128": (C.c_int, [_p, _p, _i32, _i64, _i32, _i32, _i32, _p, _i32]),
    "nxsig_as_windowed_f64": (C.c_int, [_p, _p, _i64, _i32, _i64, _i32, _i32, _i32, _i64, _i64, _p, C.POINTER(_i64), _i32]),
    "nxsig_overlap_and_add_f64": (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _i32, _p, _i32]),
    "nxsig_fir_f64": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _i32, _i32, _p, _i32]),
    "nxsig_fir_slice_f64": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _i32, _i64, _i64, _p, _i32]),
    "nxsig_timer_lap": (C.c_int, [_p]),
    "nxsig_timer_laps": (C.c_int, [_p, _pf, _i32, C.POINTER(_i32)]),
    # multi-GPU groups
    "nxsig_shard_range": (C.c_int, [_i64, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "nxsig_shard_frames": (C.c_int, [_i64, _i32, _i32, _i32, _i32] + [C.POINTER(_i64)] * 4),
    "nxsig_shard_istft": (C.c_int, [_i64, _i32, _i32, _i32, _i32] + [C.POINTER(_i64)] * 4),
    "nxsig_shard_fir": (C.c_int, [_i64, _i32, _i32, _i32, _i32] + [C.POINTER(_i64)] * 4),
    "nxsig_rendezvous_publish": (C.c_int, [C.c_char_p, _p, _sz]),
    "nxsig_rendezvous_fetch": (C.c_int, [C.c_char_p, _p, _sz, _i32, _i32]),
    "nxsig_group_create_local": (C.c_int, [_i32, C.POINTER(_i32), C.POINTER(_p)]),
    "nxsig_group_create_rank": (C.c_int, [_i32, _i32, _i32, C.c_char_p, _i32, C.POINTER(_p)]),
    "nxsig_group_destroy": (None, [_p]),
    "nxsig_group_world": (_i32, [_p]),
    "nxsig_group_local_count": (_i32, [_p]),
    "nxsig_group_rank": (_i32, [_p, _i32]),
    "nxsig_group_ctx": (_p, [_p, _i32]),
    "nxsig_group_has_rccl": (_i32, [_p]),
    "nxsig_rccl_info": (C.c_int, [C.POINTER(_i32), C.c_char_p, _sz]),
    "nxsig_mem_info": (C.c_int, [_p, C.POINTER(_sz), C.POINTER(_sz)]),
    "nxsig_group_barrier": (C.c_int, [_p]),
    "nxsig_group_allreduce_f64": (C.c_int, [_p, C.POINTER(_f64), _i32, _i32]),
    "nxsig_group_allgather": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_i64), C.POINTER(_p)]),
    "nxsig_stft_sharded_f32": (C.c_int, [_p, C.POINTER(_p), _i64, _i32, _i64, _p, C.POINTER(StftParams), _i32, _i32, C.POINTER(_p), _i32]),
    "nxsig_istft_sharded_c64": (C.c_int, [_p, C.POINTER(_p), _i64, _i32, _p, C.POINTER(StftParams), _i32, _i32, C.POINTER(_p), _i32]),
    "nxsig_fir_sharded_f32": (C.c_int, [_p, C.POINTER(_p), _i64, _i32, _i64, _p, _i32, _i32, _i32, _i32, C.POINTER(_p), _i32]),
    "nxsig_stft_mel_f32": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, C.POINTER(StftParams), _i32, _p, _p, C.POINTER(_i64), _i32]),
    "nxsig_stft_mel_sharded_f32": (C.c_int, [_p, C.POINTER(_p), _i64, _i32, _i64, _p, C.POINTER(StftParams), _i32, _p, _i32, C.POINTER(_p),
                                             C.POINTER(_i64), _i32]),
}

_lib = None


def load() -> C.CDLL:
    """Loads libnxsig.so (once).  Raises NxSignalLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NxSignalLibraryError(
            f"{LIB_PATH} not found. Build the HIP extension first: `python -m nx_signal_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the nxsig hot path."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NxSignalLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift; tests pin it
        fn.restype = res
        fn.argtypes = args
    if lib.nxsig_abi_version() != 1:
        raise NxSignalLibraryError("libnxsig.so ABI version mismatch")
    _lib = lib
    return lib


def last_dispatch() -> str:
    """Kernel families the calling thread's last compute call launched ('+'-separated, include/nxsig.h: nxsig_last_dispatch)."""
    msg = load().nxsig_last_dispatch()
    return msg.decode() if msg else ""


def last_error() -> str:
    msg = load().nxsig_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int) -> int:
    """Turns a negative nxsig_status into the matching Python exception."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == ERR_INVALID_ARG:
        raise ArgumentError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NxSignalUnsupported(msg)
    if rc == ERR_OOM:
        raise MemoryError(msg)
    raise NxSignalDeviceError(f"nxsig status {rc}: {msg}")
