"""Multi-GPU sharding of the STFT / FIR path: one process per GPU, no data-path collective.

Frames of a stream are independent given their samples and channels are fully independent, so the path shards
with NO exchange step: rank r owns a contiguous block of channels, or a contiguous range of frames of one long
stream plus an (N - hop)-sample input halo that it reads redundantly.  The only collective is the OPTIONAL
final assembly (`gather=True`): an all-gather of the output shards through torch.distributed — backend "nccl"
is RCCL over xGMI on ROCm, "gloo" is used by the CPU tests.  Outputs stay sharded and device-resident by
default (assembling config 4 would move 51.6 GB into every GPU: 30x the compute time).

The helpers below are pure index arithmetic (unit-tested without a GPU).
"""
from __future__ import annotations

import numpy as np


def split_range(n: int, world: int, rank: int):
    """Contiguous near-equal split of range(n): the first n % world ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(int(n), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_channels(n_channels: int, world: int, rank: int):
    """Channel block [c0, c1) owned by `rank` (config 4/5: 64 channels -> 8 contiguous channels per GPU)."""
    return split_range(n_channels, world, rank)


def shard_frames(num_frames: int, frame_length: int, hop: int, world: int, rank: int):
    """Frame range [m0, m1) owned by `rank` of a :valid-framed stream and the input span [s0, s1) it needs:
    s0 = m0*hop, s1 = (m1-1)*hop + frame_length — neighbouring shards overlap by frame_length - hop samples (halo)."""
    m0, m1 = split_range(num_frames, world, rank)
    if m1 <= m0:
        return m0, m1, m0 * hop, m0 * hop
    return m0, m1, m0 * hop, (m1 - 1) * hop + frame_length


def shard_fir(length: int, num_taps: int, world: int, rank: int):
    """Output range [n0, n1) of a :same-mode FIR owned by `rank` and the input span [s0, s1) (clamped to the
    signal) it needs: (taps-1)//2 samples of look-ahead and taps-1-(taps-1)//2 of history."""
    n0, n1 = split_range(length, world, rank)
    ahead = (num_taps - 1) // 2
    behind = num_taps - 1 - ahead
    return n0, n1, max(0, n0 - behind), min(length, n1 + ahead)


def stft_sharded(data, window, rank: int, world: int, axis: str = "channels", gather: bool = False, group=None,
                 compute=None, **opts):
    """Computes this rank's shard of NxSignal.stft(data, window, **opts) (window_padding must be "valid").

    data: the FULL tensor [channels, L] (axis="channels") or [L] (axis="frames"); every rank slices its own part
          (in production each rank loads only its slice; the slicing rules are shard_channels / shard_frames).
    returns (z_local, (lo, hi)) or, with gather=True, the assembled full spectrum on every rank.
    compute: the per-shard stft callable — defaults to the HIP path (nx_signal_amd.stft); the CPU gloo tests pass
             a stand-in because no GPU exists there.
    """
    if opts.get("window_padding", "valid") != "valid":
        raise ValueError("sharded stft supports window_padding='valid' (padding belongs to the stream ends)")
    if compute is None:
        from . import stft as compute  # noqa: PLC0415
    w = np.asarray(window)
    N = int(w.shape[0])
    overlap = opts.get("overlap_length")
    hop = N - (N // 2 if overlap is None else int(overlap))
    if axis == "channels":
        c0, c1 = shard_channels(data.shape[0], world, rank)
        z, _, _ = compute(data[c0:c1], window, **opts)
        lo, hi = c0, c1
    elif axis == "frames":
        L = data.shape[-1]
        M = (L - N) // hop + 1
        m0, m1, s0, s1 = shard_frames(M, N, hop, world, rank)
        z, _, _ = compute(data[..., s0:s1], window, **opts)
        lo, hi = m0, m1
    else:
        raise ValueError("axis must be 'channels' or 'frames'")
    if not gather:
        return z, (lo, hi)
    return all_gather_shards(z, lo, hi, rank, world, group=group), (lo, hi)


def all_gather_shards(z_local, lo: int, hi: int, rank: int, world: int, group=None):
    """Final assembly: all-gather of unequal shards along axis 0 (RCCL when the tensors live on GPUs)."""
    import torch
    import torch.distributed as dist

    t = z_local if isinstance(z_local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(z_local))
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device), group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    if torch.is_complex(t):
        tr = torch.view_as_real(t)
    else:
        tr = t
    pad = torch.zeros((mx,) + tuple(tr.shape[1:]), dtype=tr.dtype, device=tr.device)
    pad[: tr.shape[0]] = tr
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    parts = [b[:n] for b, n in zip(bufs, sizes)]
    out = torch.cat(parts, dim=0)
    if torch.is_complex(t):
        out = torch.view_as_complex(out.contiguous())
    return out if isinstance(z_local, torch.Tensor) else out.numpy()
