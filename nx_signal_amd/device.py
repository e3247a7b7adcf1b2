"""GPU context and device-resident buffers for the Python host mirror.

`Context` wraps a nxsig_ctx (one GPU, one HIP stream).  `DeviceBuffer` is the Python analogue of the
Erlang resource object the NIF shim hands to Elixir: an HBM allocation with a shape and dtype, so that
chains such as stft -> edit -> istft stay on the GPU.  Any object exposing
`data_ptr()` (torch) or `__cuda_array_interface__` is accepted as device input as well.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib


class Context:
    def __init__(self, device: int = 0, _borrowed=None):
        lib = _lib.load()
        self._lib = lib
        self.device = int(device)
        self._owned = _borrowed is None
        if _borrowed is not None:  # a context owned by a Group (nxsig_group_ctx)
            self._h = C.c_void_p(_borrowed)
            return
        h = C.c_void_p()
        _lib.check(lib.nxsig_ctx_create(int(device), C.byref(h)))
        self._h = h

    @property
    def handle(self):
        if self._h is None:
            raise _lib.NxSignalDeviceError("context already destroyed")
        return self._h

    def name(self) -> str:
        buf = C.create_string_buffer(256)
        _lib.check(self._lib.nxsig_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def last_dispatch(self) -> str:
        """kernel families the last compute call on this context launched, e.g. "stft.pair" or "istft.wave.deep+istft.edge_chunks"
        (include/nxsig.h: nxsig_ctx_last_dispatch; DESIGN.md section 3 names the families)"""
        buf = C.create_string_buffer(512)
        _lib.check(self._lib.nxsig_ctx_last_dispatch(self.handle, buf, 512))
        return buf.value.decode()

    def sync(self):
        _lib.check(self._lib.nxsig_sync(self.handle))

    def set_tuning(self, name: str, value: int):
        """one dispatch / geometry switch of this context (INTEGRATION.md "Switches"); the environment is only read at creation"""
        _lib.check(self._lib.nxsig_ctx_set_tuning(self.handle, name.encode(), int(value)))

    def get_tuning(self, name: str):
        """(value, is_set) of a switch"""
        v, st = C.c_int32(), C.c_int32()
        _lib.check(self._lib.nxsig_ctx_get_tuning(self.handle, name.encode(), C.byref(v), C.byref(st)))
        return int(v.value), bool(st.value)

    def clear_tuning(self, name=None):
        _lib.check(self._lib.nxsig_ctx_clear_tuning(self.handle, None if name is None else name.encode()))

    def timer_start(self):
        _lib.check(self._lib.nxsig_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float()
        _lib.check(self._lib.nxsig_timer_stop(self.handle, C.byref(ms)))
        return float(ms.value)

    def timer_lap(self):
        """records one event of a per-launch stopwatch series on the context's stream"""
        _lib.check(self._lib.nxsig_timer_lap(self.handle))

    def timer_laps(self):
        """synchronises and returns the intervals (ms) between consecutive timer_lap() events; clears the series"""
        buf = (C.c_float * 4096)()
        n = C.c_int32()
        _lib.check(self._lib.nxsig_timer_laps(self.handle, buf, 4096, C.byref(n)))
        return [float(buf[i]) for i in range(n.value)]

    def set_stream(self, hip_stream_ptr):
        _lib.check(self._lib.nxsig_set_stream(self.handle, C.c_void_p(hip_stream_ptr)))

    def empty(self, shape, dtype) -> "DeviceBuffer":
        return DeviceBuffer.empty(self, shape, dtype)

    def to_device(self, arr) -> "DeviceBuffer":
        return DeviceBuffer.from_numpy(self, arr)

    def close(self):
        if self._h is not None:
            if self._owned:
                self._lib.nxsig_ctx_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_default = {}
_default_lock = threading.Lock()


def default_context(device: int = 0) -> Context:
    with _default_lock:
        ctx = _default.get(device)
        if ctx is None:
            ctx = _default[device] = Context(device)
        return ctx


class DeviceBuffer:
    """HBM allocation owned by a Context (freed on garbage collection)."""

    def __init__(self, ctx: Context, ptr: int, shape, dtype, owner=True):
        self.ctx, self.ptr, self.shape, self.dtype = ctx, int(ptr), tuple(int(s) for s in shape), np.dtype(dtype)
        self._owner = owner

    @property
    def nbytes(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    @classmethod
    def empty(cls, ctx: Context, shape, dtype) -> "DeviceBuffer":
        shape = tuple(int(s) for s in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _lib.check(ctx._lib.nxsig_alloc(ctx.handle, nbytes, C.byref(p)))
        return cls(ctx, p.value, shape, dtype)

    @classmethod
    def from_numpy(cls, ctx: Context, arr) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        buf = cls.empty(ctx, arr.shape, arr.dtype)
        _lib.check(ctx._lib.nxsig_upload(ctx.handle, C.c_void_p(buf.ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return buf

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        _lib.check(self.ctx._lib.nxsig_download(self.ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), out.nbytes))
        return out

    def reshape(self, *shape) -> "DeviceBuffer":
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        if int(np.prod(shape, dtype=np.int64)) != int(np.prod(self.shape, dtype=np.int64)):
            raise _lib.ArgumentError(f"cannot reshape {self.shape} to {shape}")
        view = DeviceBuffer(self.ctx, self.ptr, shape, self.dtype, owner=False)
        view._keep = self
        return view

    def free(self):
        if self._owner and self.ptr:
            self.ctx._lib.nxsig_free(self.ctx.handle, C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):  # pragma: no cover
        try:
            if self.ctx._h is not None:
                self.free()
        except Exception:
            pass

    def __repr__(self):
        return f"DeviceBuffer(shape={self.shape}, dtype={self.dtype}, ptr=0x{self.ptr:x})"


def is_device(obj) -> bool:
    return isinstance(obj, DeviceBuffer) or hasattr(obj, "data_ptr") or hasattr(obj, "__cuda_array_interface__")


def device_view(obj):
    """(ptr, shape, dtype) of a device-resident object (DeviceBuffer, torch tensor, CUDA array interface)."""
    if isinstance(obj, DeviceBuffer):
        return obj.ptr, obj.shape, obj.dtype
    if hasattr(obj, "__cuda_array_interface__"):
        cai = obj.__cuda_array_interface__
        if cai.get("strides") is not None:
            raise _lib.ArgumentError("device inputs must be contiguous")
        return int(cai["data"][0]), tuple(cai["shape"]), np.dtype(cai["typestr"])
    if hasattr(obj, "data_ptr"):  # torch tensor
        if not obj.is_contiguous():
            raise _lib.ArgumentError("device inputs must be contiguous")
        import torch  # only reached when the caller already uses torch

        dt = {torch.float32: np.float32, torch.complex64: np.complex64}.get(obj.dtype)
        if dt is None:
            raise _lib.ArgumentError(f"unsupported device dtype {obj.dtype}")
        return int(obj.data_ptr()), tuple(obj.shape), np.dtype(dt)
    raise _lib.ArgumentError("not a device-resident object")
