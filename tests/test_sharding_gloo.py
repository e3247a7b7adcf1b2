"""Multi-process CPU tests (gloo, world_size 2) of the sharded STFT path: the shards partition the work exactly,
frame shards with their input halo reproduce the unsharded result bit for bit, and the optional all-gather
assembly returns the full spectrum on every rank.  The per-shard compute is the oracle here (no GPU in this
container); on GPUs the same code path calls the HIP stft."""
import os
import socket

import numpy as np
import pytest

from nx_signal_amd import sharding
from oracle import nx_oracle as O


def test_split_range_partitions_exactly():
    for n in (0, 1, 7, 64, 11247, 56247):
        for world in (1, 2, 3, 8):
            spans = [sharding.split_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_channels(64, 8, 3) == (24, 32)  # config 4: 8 contiguous channels per GPU


def test_frame_shards_cover_stream_with_halo():
    N, hop, L = 1024, 256, 2880000
    M = (L - N) // hop + 1
    for world in (2, 4, 8):
        prev_m1 = 0
        for r in range(world):
            m0, m1, s0, s1 = sharding.shard_frames(M, N, hop, world, r)
            assert m0 == prev_m1
            assert s0 == m0 * hop and s1 == (m1 - 1) * hop + N
            assert (s1 - s0 - N) // hop + 1 == m1 - m0  # the shard alone frames to exactly its frames
            if r > 0:
                assert prev_s1 - s0 == N - hop  # halo read redundantly, no communication
            prev_m1, prev_s1 = m1, s1
        assert prev_m1 == M and prev_s1 == L


def test_fir_shards():
    n0, n1, s0, s1 = sharding.shard_fir(1000, 257, 4, 1)
    assert (n0, n1) == (250, 500) and s0 == 250 - 128 and s1 == 500 + 128


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_stft(x, w, **opts):
    return O.stft(x, w, **opts)


def _worker(rank, world, port, axis, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = O.hann(64)
        opts = dict(overlap_length=48, fft_length=64, sampling_rate=8000)
        if axis == "channels":
            x = np.stack([O.synth_signal(4000, seed=10 + c) for c in range(5)])
        else:
            x = O.synth_signal(20000, seed=3)
        full, _, _ = O.stft(x, w, **opts)
        z_local, (lo, hi) = sharding.stft_sharded(x, w, rank, world, axis=axis, compute=_oracle_stft, **opts)
        ref_local = np.ascontiguousarray(full[lo:hi])
        z_local = np.ascontiguousarray(z_local)
        ok_local = np.array_equal(z_local.view(np.uint32), ref_local.view(np.uint32))
        z_all, _ = sharding.stft_sharded(x, w, rank, world, axis=axis, gather=True, compute=_oracle_stft, **opts)
        z_all = np.ascontiguousarray(z_all)
        ok_all = z_all.shape == full.shape and np.array_equal(z_all.view(np.uint32), full.view(np.uint32))
        q.put((rank, bool(ok_local), bool(ok_all), (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("axis", ["channels", "frames"])
def test_sharded_stft_world2_gloo(axis):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, axis, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), "shard differs from the matching slice of the unsharded result"
    assert all(r[2] for r in res), "all-gather assembly differs from the unsharded result"
    assert res[0][3][1] == res[1][3][0]
