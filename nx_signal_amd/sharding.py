"""Multi-GPU sharding of the STFT / FIR path behind the C ABI — no torch, no Python-side collective.

Frames of a stream are independent given their samples and channels (the reference's vectorized axes,
lib/nx_signal.ex:358-363) are fully independent, so the path shards with NO exchange step: rank r owns a contiguous
block of channels, or a contiguous range of frames of one long stream plus an (N - hop)-sample input halo that it
reads redundantly.  The only collective is the OPTIONAL final assembly (`gather=True`): an RCCL all-gather over xGMI
issued by libnxsig.so itself (csrc/group.cpp: ncclCommInitAll / ncclCommInitRank, ncclGroupStart/End, ncclAllGather or
one ncclBroadcast per rank for unequal shards).  Outputs stay sharded and device-resident by default (assembling
config 4 would move 51.6 GB into every GPU: 30x the compute time).

`Group.local(n)`  — one process drives n GPUs (what the Elixir host does through the NIF);
`Group.ranked()`  — one process per GPU under a launcher that sets RANK / WORLD_SIZE / LOCAL_RANK (bench.py).

The shard plans are pure index arithmetic computed by the C library (nxsig_shard_range / _frames / _fir) and are
unit-tested without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .device import Context, DeviceBuffer

CHANNELS, FRAMES = _lib.SHARD_CHANNELS, _lib.SHARD_FRAMES
_AXES = {"channels": CHANNELS, "frames": FRAMES, "samples": FRAMES}


def _four(fn, *args):
    out = [C.c_int64() for _ in range(4)]
    _lib.check(fn(*args, *[C.byref(o) for o in out]))
    return tuple(int(o.value) for o in out)


def split_range(n: int, world: int, rank: int):
    """Contiguous near-equal split of range(n): the first n % world ranks get one extra item."""
    b, e = C.c_int64(), C.c_int64()
    try:
        _lib.check(_lib.load().nxsig_shard_range(int(n), int(world), int(rank), C.byref(b), C.byref(e)))
    except _lib.ArgumentError as exc:
        raise ValueError(str(exc)) from exc
    return int(b.value), int(e.value)


def shard_channels(n_channels: int, world: int, rank: int):
    """Channel block [c0, c1) owned by `rank` (config 4/5: 64 channels -> 8 contiguous channels per GPU)."""
    return split_range(n_channels, world, rank)


def shard_frames(num_frames: int, frame_length: int, hop: int, world: int, rank: int):
    """Frame range [m0, m1) owned by `rank` of a :valid-framed stream and the input span [s0, s1) it needs:
    s0 = m0*hop, s1 = (m1-1)*hop + frame_length — neighbouring shards overlap by frame_length - hop samples (halo)."""
    return _four(_lib.load().nxsig_shard_frames, int(num_frames), int(frame_length), int(hop), int(world), int(rank))


def shard_istft(num_frames: int, frame_length: int, hop: int, world: int, rank: int):
    """iSTFT over frame ranges: `rank` needs input frames [f0, f1) (its share plus ceil(N / hop) - 1 halo frames in front,
    recomputed instead of exchanged, widened to multiples of 8 frames) and keeps output samples [n0, n1); returns (f0, f1, n0, n1)."""
    return _four(_lib.load().nxsig_shard_istft, int(num_frames), int(frame_length), int(hop), int(world), int(rank))


def shard_fir(length: int, num_taps: int, world: int, rank: int, mode: str = "same"):
    """Output range [n0, n1) of a FIR (convolution mode `mode`) owned by `rank` and the input span [s0, s1) (clamped to
    the signal) it needs: num_taps - 1 samples of halo split between history and look-ahead by the mode's offset."""
    m = {"full": _lib.CONV_FULL, "same": _lib.CONV_SAME, "valid": _lib.CONV_VALID}[mode]
    return _four(_lib.load().nxsig_shard_fir, int(length), int(num_taps), m, int(world), int(rank))


def rendezvous_path(env=None) -> str:
    """File through which rank 0 hands the ncclUniqueId to the other ranks of ONE node: unique per launch (the launcher's
    pid and its rendezvous port), removed by rank 0 once the communicator is up."""
    env = os.environ if env is None else env
    d = env.get("NXSIG_RDZV_DIR")
    if d is None:
        # a directory only this user can write to (0700, ownership checked): nobody else can plant an id file there.  Every rank
        # of a launch must arrive at the SAME path, so if that directory cannot be had (it exists with other permissions, /tmp is
        # read-only ...) all ranks fall back to /tmp itself, where the C side still creates the file exclusively with mode 0600
        # and validates its content
        d = os.path.join("/tmp", "nxsig-%d" % os.getuid())
        try:
            os.makedirs(d, mode=0o700, exist_ok=True)
            st = os.stat(d)
            if st.st_uid != os.getuid() or (st.st_mode & 0o077):
                d = "/tmp"
        except OSError:
            d = "/tmp"
    # unique per LAUNCH: the launcher's pid and port, the elastic run id and restart count (a restarted worker group of the same
    # launcher must not meet the id file of the group that died), and an optional nonce a launcher may hand to all its ranks
    return os.path.join(d, "nxsig_rdzv_%s_%s_%s_%s_%d" % (env.get("MASTER_PORT", "0"), env.get("TORCHELASTIC_RUN_ID", "none"),
                                                            env.get("TORCHELASTIC_RESTART_COUNT", "0"), env.get("NXSIG_RDZV_NONCE", "0"), os.getppid()))


class Group:
    """nxsig_group: one context + stream per member GPU, RCCL communicators for the assembly."""

    def __init__(self, handle):
        self._lib = _lib.load()
        self._h = handle
        self.world = int(self._lib.nxsig_group_world(handle))
        self.local_count = int(self._lib.nxsig_group_local_count(handle))
        self.ranks = [int(self._lib.nxsig_group_rank(handle, i)) for i in range(self.local_count)]
        self.contexts = [Context(_borrowed=self._lib.nxsig_group_ctx(handle, i)) for i in range(self.local_count)]
        self.has_rccl = bool(self._lib.nxsig_group_has_rccl(handle))

    @classmethod
    def local(cls, n: int, devices=None) -> "Group":
        lib = _lib.load()
        h = C.c_void_p()
        ids = None if devices is None else (C.c_int32 * n)(*[int(d) for d in devices])
        _lib.check(lib.nxsig_group_create_local(int(n), ids, C.byref(h)))
        return cls(h)

    @classmethod
    def ranked(cls, world=None, rank=None, device=None, path=None, timeout_ms=120000) -> "Group":
        world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        path = rendezvous_path() if path is None else path
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.nxsig_group_create_rank(world, rank, device, path.encode(), int(timeout_ms), C.byref(h)))
        return cls(h)

    @property
    def handle(self):
        if self._h is None:
            raise _lib.NxSignalDeviceError("group already destroyed")
        return self._h

    def barrier(self):
        _lib.check(self._lib.nxsig_group_barrier(self.handle))

    def allreduce(self, values, op="max"):
        """element-wise max / sum of a few host doubles over the PROCESSES of the group"""
        v = (C.c_double * len(values))(*[float(x) for x in values])
        _lib.check(self._lib.nxsig_group_allreduce_f64(self.handle, v, len(values), 0 if op == "max" else 1))
        return [float(x) for x in v]

    def allgather(self, send_ptrs, counts_bytes, recv_ptrs):
        n = self.local_count
        s = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in send_ptrs])
        r = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in recv_ptrs])
        c = (C.c_int64 * self.world)(*[int(x) for x in counts_bytes])
        _lib.check(self._lib.nxsig_group_allgather(self.handle, s, c, r))

    def sync(self):
        for c in self.contexts:
            c.sync()

    def close(self):
        if self._h is not None:
            for c in self.contexts:
                c._h = None  # owned by the group
            self._lib.nxsig_group_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _stft_params(window, opts):
    from . import _resolve_stft_opts  # noqa: PLC0415

    return _resolve_stft_opts(window, opts)


def stft_sharded(group: Group, data, window, axis: str = "channels", gather: bool = False, **opts):
    """NxSignal.stft(data, window, **opts) sharded over `group` (window_padding must be "valid").

    data: host array [channels, L] / [L]  -> returns the assembled host spectrum c64[channels, M, K] / [M, K]
          (per-shard downloads, or with gather=True the RCCL all-gather followed by one download; on a RANKED group every
          process passes the whole tensor and gets its own part — gather=False — or the whole result — gather=True), or
          a list with one DeviceBuffer per LOCAL member holding that member's input shard (rows [c0, c1) or the sample
          span [s0, s1) of every row; see shard_channels / shard_frames) together with `length=` and `batch=` of the
          whole tensor -> returns the list of per-member output DeviceBuffers (shards, or full tensors with gather=True).
    """
    lib = _lib.load()
    w = np.ascontiguousarray(window, dtype=np.float32)
    length = opts.pop("length", None)
    batch = opts.pop("batch", None)
    p, N, hop, K = _stft_params(w, opts)
    ax = _AXES[axis]
    n = group.local_count
    if isinstance(data, (list, tuple)) and data and isinstance(data[0], DeviceBuffer):
        if length is None or batch is None:
            raise _lib.ArgumentError("device shards need length= and batch= of the whole tensor")
        M = int(_lib.check(lib.nxsig_num_frames(int(length), N, hop, _lib.PAD_VALID, 0, 0)))
        outs = []
        for i, r in enumerate(group.ranks):
            if gather:
                shape = (batch, M, K)
            elif ax == CHANNELS:
                c0, c1 = shard_channels(batch, group.world, r)
                shape = (c1 - c0, M, K)
            else:
                m0, m1, _, _ = shard_frames(M, N, hop, group.world, r)
                shape = (batch, m1 - m0, K)
            outs.append(group.contexts[i].empty(shape, np.complex64))
        # channel shards: rows of the full length; frame shards: every member's buffer is dense [batch][its span] and the spans
        # differ from member to member when the frame count does not divide evenly -> 0 = "dense per-member shards" (nxsig.h)
        stride = int(data[0].shape[-1]) if ax == CHANNELS else 0
        xs = (C.c_void_p * n)(*[C.c_void_p(d.ptr) for d in data])
        zs = (C.c_void_p * n)(*[C.c_void_p(o.ptr) for o in outs])
        _lib.check(lib.nxsig_stft_sharded_f32(group.handle, xs, int(length), int(batch), stride, w.ctypes.data_as(C.c_void_p),
                                              C.byref(p), ax, int(bool(gather)), zs, _lib.DEVICE))
        return outs
    x = np.ascontiguousarray(data, dtype=np.float32)
    squeeze = x.ndim == 1
    x2 = x.reshape(1, -1) if squeeze else x.reshape(-1, x.shape[-1])
    B, L = x2.shape
    M = int(_lib.check(lib.nxsig_num_frames(L, N, hop, _lib.PAD_VALID, 0, 0)))
    z = np.empty((B, M, K), np.complex64)
    xs = (C.c_void_p * n)(*([x2.ctypes.data_as(C.c_void_p)] + [C.c_void_p(0)] * (n - 1)))
    zs = (C.c_void_p * n)(*([z.ctypes.data_as(C.c_void_p)] + [C.c_void_p(0)] * (n - 1)))
    _lib.check(lib.nxsig_stft_sharded_f32(group.handle, xs, L, B, L, w.ctypes.data_as(C.c_void_p), C.byref(p), ax,
                                          int(bool(gather)), zs, _lib.HOST))
    return z[0] if squeeze else z.reshape(x.shape[:-1] + (M, K))


def istft_sharded(group: Group, data, window, axis: str = "channels", gather: bool = False, **opts):
    """NxSignal.istft(data, window, **opts) sharded over `group`; equals the unsharded call bit for bit.

    data: host array c64[channels, M, K] / [M, K] -> the assembled host signal c64[channels, out_len] / [out_len], or a list
          with one DeviceBuffer per LOCAL member holding that member's dense input shard (rows [c0, c1), or frames [f0, f1) of
          every row; see shard_channels / shard_istft) together with `num_frames=` and `batch=` of the whole tensor -> the
          list of per-member output DeviceBuffers (shards, or full tensors with gather=True).
    """
    lib = _lib.load()
    w = np.ascontiguousarray(window, dtype=np.float32)
    num_frames = opts.pop("num_frames", None)
    batch = opts.pop("batch", None)
    N = int(w.shape[0])
    o = {"fft_length": None, "overlap_length": None, "scaling": None, "sampling_rate": 1000}
    unknown = [k for k in opts if k not in o]
    if unknown:
        raise _lib.ArgumentError(f"unknown keys {unknown} in istft options, the allowed keys are: {list(o)}")
    o.update(opts)
    overlap = N // 2 if o["overlap_length"] is None else int(o["overlap_length"])
    if overlap >= N:
        raise _lib.ArgumentError(f"overlap_length must be a number less than the window size {N}, got: {N}")
    scal = {None: _lib.SCALE_NONE, "spectrum": _lib.SCALE_SPECTRUM, "psd": _lib.SCALE_PSD}
    if o["scaling"] not in scal:
        raise _lib.ArgumentError(f"invalid :scaling, expected one of :spectrum, :psd or nil, got: {o['scaling']!r}")
    hop = N - overlap
    p = _lib.StftParams(N, hop, N, 0, 0, 0, scal[o["scaling"]], 0, float(o["sampling_rate"] or 0.0))
    ax = _AXES[axis]
    n = group.local_count
    if isinstance(data, (list, tuple)) and data and isinstance(data[0], DeviceBuffer):
        if num_frames is None or batch is None:
            raise _lib.ArgumentError("device shards need num_frames= and batch= of the whole tensor")
        M = int(num_frames)
        out_len = int(_lib.check(lib.nxsig_ola_length(M, N, hop)))
        outs = []
        for i, r in enumerate(group.ranks):
            if gather:
                shape = (batch, out_len)
            elif ax == CHANNELS:
                c0, c1 = shard_channels(batch, group.world, r)
                shape = (c1 - c0, out_len)
            else:
                _, _, n0, n1 = shard_istft(M, N, hop, group.world, r)
                shape = (batch, n1 - n0)
            outs.append(group.contexts[i].empty(shape, np.complex64))
        zs = (C.c_void_p * n)(*[C.c_void_p(d.ptr) for d in data])
        ys = (C.c_void_p * n)(*[C.c_void_p(o_.ptr) for o_ in outs])
        _lib.check(lib.nxsig_istft_sharded_c64(group.handle, zs, M, int(batch), w.ctypes.data_as(C.c_void_p), C.byref(p), ax,
                                               int(bool(gather)), ys, _lib.DEVICE))
        return outs
    z = np.ascontiguousarray(data, dtype=np.complex64)
    if z.ndim < 2 or z.shape[-1] != N:
        raise _lib.ArgumentError("istft expects a tensor of shape {..., frames, frequencies} with fft_length == window length")
    squeeze = z.ndim == 2
    z3 = z.reshape((-1,) + z.shape[-2:])
    B, M, _ = z3.shape
    out_len = int(_lib.check(lib.nxsig_ola_length(M, N, hop)))
    y = np.empty((B, out_len), np.complex64)
    zs = (C.c_void_p * n)(*([z3.ctypes.data_as(C.c_void_p)] + [C.c_void_p(0)] * (n - 1)))
    ys = (C.c_void_p * n)(*([y.ctypes.data_as(C.c_void_p)] + [C.c_void_p(0)] * (n - 1)))
    _lib.check(lib.nxsig_istft_sharded_c64(group.handle, zs, M, B, w.ctypes.data_as(C.c_void_p), C.byref(p), ax, int(bool(gather)), ys,
                                           _lib.HOST))
    return y[0] if squeeze else y.reshape(z.shape[:-2] + (out_len,))


def fir_sharded(group: Group, data, taps, mode: str = "same", axis: str = "channels", gather: bool = False, length=None, batch=None):
    """Filters.fir (overlap-save FFT convolution, nxsig_fir_f32) sharded over `group`.

    data: host array [channels, L] / [L] (LOCAL groups) -> the assembled host result, or a list with one DeviceBuffer per LOCAL
          member holding that member's input shard (rows [c0, c1), or the sample span [s0, s1) of every row: shard_channels /
          shard_fir) with `length=` and `batch=` of the whole tensor -> the list of per-member result DeviceBuffers (shards, or the
          full tensor with gather=True).
    Sample shards have one exchange step: a row that holds an Inf / NaN ANYWHERE comes out NaN from end to end, like the one
    transform of Convolution.fftconvolve (lib/nx_signal/convolution.ex:276-284) leaves it — the members all-reduce one flag per row."""
    lib = _lib.load()
    h = np.ascontiguousarray(taps, dtype=np.float32)
    m = {"full": _lib.CONV_FULL, "same": _lib.CONV_SAME, "valid": _lib.CONV_VALID}.get(mode)
    if m is None:
        raise _lib.ArgumentError("expected mode to be one of [:full, :same, :valid]")
    n = group.local_count
    ax = _AXES[axis]
    if isinstance(data, (list, tuple)) and data and isinstance(data[0], DeviceBuffer):
        if length is None or batch is None:
            raise _lib.ArgumentError("device shards need length= and batch= of the whole tensor")
        n_out = int(_lib.check(lib.nxsig_conv_length(int(length), h.shape[0], m)))
        outs = []
        for i, r in enumerate(group.ranks):
            if gather:
                shape = (batch, n_out)
            elif ax == CHANNELS:
                c0, c1 = shard_channels(batch, group.world, r)
                shape = (c1 - c0, n_out)
            else:
                n0, n1, _, _ = shard_fir(int(length), int(h.shape[0]), group.world, r, mode)
                shape = (batch, n1 - n0)
            outs.append(group.contexts[i].empty(shape, np.float32))
        stride = int(data[0].shape[-1]) if ax == CHANNELS else 0   # 0 = dense per-member shards (nxsig.h)
        xs = (C.c_void_p * n)(*[C.c_void_p(d.ptr) for d in data])
        ys = (C.c_void_p * n)(*[C.c_void_p(o.ptr) for o in outs])
        _lib.check(lib.nxsig_fir_sharded_f32(group.handle, xs, int(length), int(batch), stride, h.ctypes.data_as(C.c_void_p),
                                             int(h.shape[0]), m, ax, int(bool(gather)), ys, _lib.DEVICE))
        return outs
    x = np.ascontiguousarray(data, dtype=np.float32)
    squeeze = x.ndim == 1
    x2 = x.reshape(1, -1) if squeeze else x.reshape(-1, x.shape[-1])
    B, L = x2.shape
    n_out = int(_lib.check(lib.nxsig_conv_length(L, h.shape[0], m)))
    y = np.empty((B, n_out), np.float32)
    xs = (C.c_void_p * n)(*([x2.ctypes.data_as(C.c_void_p)] + [C.c_void_p(0)] * (n - 1)))
    ys = (C.c_void_p * n)(*([y.ctypes.data_as(C.c_void_p)] + [C.c_void_p(0)] * (n - 1)))
    _lib.check(lib.nxsig_fir_sharded_f32(group.handle, xs, L, B, L, h.ctypes.data_as(C.c_void_p), int(h.shape[0]), m,
                                         ax, int(bool(gather)), ys, _lib.HOST))
    return y[0] if squeeze else y.reshape(x.shape[:-1] + (n_out,))


def mel_spectrogram_sharded(group: Group, data, window, axis: str = "channels", **opts):
    """`stft(data, window) |> stft_to_mel` (the fused log-mel of `nx_signal_amd.mel_spectrogram`) sharded over `group` — a sharded
    call with an exchange step (sample-sharded `fir_sharded` is the other): the clamp against `reduce_max(log_spec) - 8` (lib/nx_signal.ex:511) takes the maximum
    over the WHOLE tensor, so the members' running maxima are all-reduced (RCCL ncclAllReduce / ncclMax; through the host when
    the members of one process share a device) between the two passes.  window_padding must be "valid".

    data: host array [channels, L] / [L] (LOCAL groups: every member uploads its part, the result shards are downloaded into
          place -> f32[channels, M, mel_bins] / [M, mel_bins]), or a list with one DeviceBuffer per LOCAL member holding
          its input shard (rows [c0, c1) / the sample span [s0, s1) of every row) with `length=` and `batch=` of the whole
          tensor -> the list of per-member result DeviceBuffers (shards; they stay on their devices).
    Options: the stft options plus mel_bins (128), max_mel, mel_frequency_spacing."""
    from . import _mel_filters_for  # the host filterbank of the unsharded call (bit-identical to the reference's doctest)

    lib = _lib.load()
    w = np.ascontiguousarray(window, dtype=np.float32)
    length = opts.pop("length", None)
    batch = opts.pop("batch", None)
    mb = int(opts.pop("mel_bins", 128))
    fopts = {k: opts.pop(k) for k in ("max_mel", "mel_frequency_spacing") if k in opts and opts[k] is not None}
    p, N, hop, K = _stft_params(w, opts)
    filt = _mel_filters_for(K, mb, float(p.sampling_rate), fopts)
    ax = _AXES[axis]
    n = group.local_count

    def shard_shape(r, B, M):
        if ax == CHANNELS:
            c0, c1 = shard_channels(B, group.world, r)
            return (c1 - c0, M, mb)
        m0, m1, _, _ = shard_frames(M, N, hop, group.world, r)
        return (B, m1 - m0, mb)

    wp, fp = w.ctypes.data_as(C.c_void_p), filt.ctypes.data_as(C.c_void_p)
    mo = C.c_int64(0)
    if isinstance(data, (list, tuple)) and data and isinstance(data[0], DeviceBuffer):
        if length is None or batch is None:
            raise _lib.ArgumentError("device shards need length= and batch= of the whole tensor")
        M = int(_lib.check(lib.nxsig_num_frames(int(length), N, hop, _lib.PAD_VALID, 0, 0)))
        outs = [group.contexts[i].empty(shard_shape(r, int(batch), M), np.float32) for i, r in enumerate(group.ranks)]
        # channel shards: rows of the full length; frame shards: every member's buffer is dense [batch][its span] and the spans
        # differ from member to member when the frame count does not divide evenly -> 0 = "dense per-member shards" (nxsig.h)
        stride = int(data[0].shape[-1]) if ax == CHANNELS else 0
        xs = (C.c_void_p * n)(*[C.c_void_p(d.ptr) for d in data])
        ys = (C.c_void_p * n)(*[C.c_void_p(o.ptr) for o in outs])
        _lib.check(lib.nxsig_stft_mel_sharded_f32(group.handle, xs, int(length), int(batch), stride, wp, C.byref(p), mb, fp, ax, ys,
                                                  C.byref(mo), _lib.DEVICE))
        return outs
    x = np.ascontiguousarray(data, dtype=np.float32)
    squeeze = x.ndim == 1
    x2 = x.reshape(1, -1) if squeeze else x.reshape(-1, x.shape[-1])
    B, L = x2.shape
    M = int(_lib.check(lib.nxsig_num_frames(L, N, hop, _lib.PAD_VALID, 0, 0)))
    out = np.empty((B, M, mb), np.float32)
    xs = (C.c_void_p * n)(*([x2.ctypes.data_as(C.c_void_p)] + [C.c_void_p(0)] * (n - 1)))
    ys = (C.c_void_p * n)(*([out.ctypes.data_as(C.c_void_p)] + [C.c_void_p(0)] * (n - 1)))
    _lib.check(lib.nxsig_stft_mel_sharded_f32(group.handle, xs, L, B, L, wp, C.byref(p), mb, fp, ax, ys, C.byref(mo), _lib.HOST))
    return out[0] if squeeze else out.reshape(x.shape[:-1] + (M, mb))
