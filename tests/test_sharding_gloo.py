"""Multi-process CPU tests (gloo, world_size 2) of the sharded STFT / FIR path: the shard plans (computed by the C
library: nxsig_shard_range / _frames / _fir) partition the work exactly, frame shards with their input halo reproduce
the unsharded result bit for bit, and an all-gather assembly of the unequal shards returns the full result on every
rank.  There is no GPU in this container, so the per-shard compute is the oracle and the collective is gloo (test
infrastructure); on GPUs the same plans drive the HIP kernels and RCCL inside libnxsig.so (tests/test_gpu_group.py).
The file rendezvous that carries the ncclUniqueId between processes is exercised here with two real processes."""
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
    # every mode / world: ranges tile the output, spans hold exactly the samples the outputs read, and the per-shard slices of
    # the full convolution of the span reproduce the unsharded filter
    rng = np.random.default_rng(5)
    x = rng.standard_normal(777).astype(np.float32)
    for taps in (1, 2, 31, 257):
        h = rng.standard_normal(taps).astype(np.float32)
        full = np.convolve(x.astype(np.float64), h.astype(np.float64))
        for mode in ("full", "same", "valid"):
            if mode == "valid" and taps > x.size:
                continue
            want = {"full": full, "same": full[(taps - 1) // 2:(taps - 1) // 2 + x.size],
                    "valid": full[taps - 1:x.size]}[mode]
            start = {"full": 0, "same": (taps - 1) // 2, "valid": taps - 1}[mode]
            for world in (1, 2, 3, 8):
                got, prev = [], 0
                for r in range(world):
                    n0, n1, s0, s1 = sharding.shard_fir(x.size, taps, world, r, mode=mode)
                    assert n0 == prev
                    prev = n1
                    if n1 > n0:
                        assert 0 <= s0 <= s1 <= x.size
                        sub = np.convolve(x[s0:s1].astype(np.float64), h.astype(np.float64))
                        k0 = n0 + start - s0
                        got.append(sub[k0:k0 + (n1 - n0)])
                assert prev == want.size
                assert np.allclose(np.concatenate(got), want, rtol=0, atol=1e-12), (taps, mode, world)


def test_istft_frame_shards_reproduce_the_overlap_add():
    """nxsig_shard_istft: kept sample ranges tile the output; the local overlap-add (oracle) of the member's frames, halo frames
    included, equals the global one on the kept samples exactly (same frames, same order)"""
    rng = np.random.default_rng(9)
    for N, hop, M in ((16, 4, 37), (16, 16, 9), (12, 5, 20), (8, 2, 3), (32, 8, 100)):
        z = (rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))).astype(np.complex64)
        w = np.hanning(N + 1)[:N].astype(np.float32) + np.float32(0.05)
        full = O.istft(z, w, overlap_length=N - hop, fft_length=N)
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                f0, f1, n0, n1 = sharding.shard_istft(M, N, hop, world, r)
                assert n0 == prev and 0 <= f0 <= f1 <= M
                prev = n1
                if n1 > n0:
                    local = O.istft(z[f0:f1], w, overlap_length=N - hop, fft_length=N)
                    k0 = n0 - f0 * hop
                    assert np.array_equal(local[k0:k0 + (n1 - n0)].view(np.uint32), full[n0:n1].view(np.uint32)), (N, hop, M, world, r)
            assert prev == full.shape[0]


def _rdzv_reader(path, q):
    import ctypes as C

    from nx_signal_amd import _lib

    buf = (C.c_ubyte * 128)()
    rc = _lib.load().nxsig_rendezvous_fetch(path.encode(), buf, 128, 20000, 600)
    q.put((rc, bytes(buf)))


def test_file_rendezvous_two_processes(tmp_path):
    """the 128-byte ncclUniqueId travels from rank 0 to the other ranks through a file: a reader that starts first waits
    for the writer, never sees a partial file, and a stale file (older than max_age) is ignored"""
    import ctypes as C
    import multiprocessing as mp
    import time

    from nx_signal_amd import _lib

    lib = _lib.load()
    path = str(tmp_path / "nxsig_rdzv_test")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rdzv_reader, args=(path, q))
    p.start()
    time.sleep(0.5)
    payload = bytes(range(128))
    assert lib.nxsig_rendezvous_publish(path.encode(), payload, 128) == 0
    rc, got = q.get(timeout=60)
    p.join(timeout=30)
    assert rc == 0 and got == payload
    # stale file: mtime two hours back -> fetch with max_age 600 s times out instead of returning old bytes
    old = time.time() - 7200
    os.utime(path, (old, old))
    buf = (C.c_ubyte * 128)()
    assert lib.nxsig_rendezvous_fetch(path.encode(), buf, 128, 200, 600) == _lib.ERR_INVALID_ARG
    assert "timed out" in _lib.last_error()
    assert lib.nxsig_rendezvous_fetch(path.encode(), buf, 128, 200, 0) == 0  # max_age 0 = any age


def test_group_needs_a_gpu():
    """no CPU fallback: creating a group without a GPU fails loudly"""
    import subprocess
    import sys

    code = ("import nx_signal_amd.sharding as s\n"
            "try:\n    s.Group.local(1)\n    print('CREATED')\n"
            "except Exception as e:\n    print(type(e).__name__, e)\n")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=120).stdout
    assert "NxSignalDeviceError" in out and "no CPU fallback" in out, out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_unequal(z_local, world):
    """all-gather of unequal shards along axis 0 over gloo (the CPU stand-in of nxsig_group_allgather)"""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(z_local).view(np.float32).reshape(z_local.shape[0], -1))
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([t.shape[0]], dtype=torch.int64))
    sizes = [int(s.item()) for s in sizes]
    pad = torch.zeros((max(sizes), t.shape[1]), dtype=t.dtype)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0).numpy()
    return out.view(np.complex64).reshape((out.shape[0],) + z_local.shape[1:])


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
        N, hop = 64, 16
        if axis == "channels":
            lo, hi = sharding.shard_channels(x.shape[0], world, rank)
            z_local, _, _ = O.stft(x[lo:hi], w, **opts)
        else:
            M = (x.shape[-1] - N) // hop + 1
            lo, hi, s0, s1 = sharding.shard_frames(M, N, hop, world, rank)
            z_local, _, _ = O.stft(x[s0:s1], w, **opts)
        ref_local = np.ascontiguousarray(full[lo:hi])
        z_local = np.ascontiguousarray(z_local)
        ok_local = np.array_equal(z_local.view(np.uint32), ref_local.view(np.uint32))
        z_all = np.ascontiguousarray(_gather_unequal(z_local, world))
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
