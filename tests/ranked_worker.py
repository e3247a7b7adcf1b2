"""Worker of tests/test_gpu_group_ranked.py: one RANKED member (one process per rank) of a 2- or 8-rank group whose ranks share the
single GPU of the test box.  RCCL refuses two ranks on one device of one host, so every rank claims its own host id
(NCCL_HOSTID) and the pair talks through RCCL's socket transport over loopback: the whole multi-process path — file
rendezvous of the ncclUniqueId, ncclCommInitRank, barrier, all-reduce, all-gather of EQUAL shards (ncclAllGather, in place)
and of UNEQUAL shards (one ncclBroadcast per rank), sharded stft on device shards with the assembly — runs for real."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import faulthandler

    faulthandler.dump_traceback_later(240, exit=True)  # a hang prints where every thread is and ends the process
    rank, world, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.dup2(2, 1)  # RCCL's banner goes to stdout: keep it away from the verdict line
    import nx_signal_amd as S
    from nx_signal_amd import sharding
    from oracle import nx_oracle as O

    g = sharding.Group.ranked(world=world, rank=rank, device=0, path=path, timeout_ms=150000)
    ctx = g.contexts[0]
    assert g.world == world and g.local_count == 1 and g.ranks == [rank] and g.has_rccl
    g.barrier()
    assert g.allreduce([float(rank + 1), 10.0 - rank], "max") == [float(world), 10.0]
    assert g.allreduce([float(rank + 1)], "sum") == [world * (world + 1) / 2.0]
    # unequal shards: one broadcast per rank
    parts = [(np.arange(1000 + 37 * r, dtype=np.uint32) * 2654435761 % 1000003 + r).astype(np.uint32) for r in range(world)]
    want = np.concatenate(parts)
    send = ctx.to_device(parts[rank])
    recv = ctx.empty((want.size,), np.uint32)
    g.allgather([send.ptr], [p.nbytes for p in parts], [recv.ptr])
    ctx.sync()
    assert np.array_equal(recv.numpy(), want), "unequal all-gather"
    # equal shards, in place
    eq = ctx.empty((world, 4096), np.uint32)
    mine = (np.arange(4096, dtype=np.uint32) + 100000 * (rank + 1)).astype(np.uint32)
    S._lib.check(S._lib.load().nxsig_upload(ctx.handle, eq.ptr + rank * 4096 * 4, mine.ctypes.data, mine.nbytes))
    g.allgather([eq.ptr + rank * 4096 * 4], [4096 * 4] * world, [eq.ptr])
    ctx.sync()
    assert np.array_equal(eq.numpy(), np.stack([np.arange(4096, dtype=np.uint32) + 100000 * (r + 1) for r in range(world)])), "in-place all-gather"
    # sharded stft on device shards: channels (3 + 2 of 5) and frame ranges of one stream, assembled on every rank
    B, L, N, hop = max(5, world + 3), 30000, 1024, 256   # every rank owns at least one channel (11 over 8 ranks: 2, 2, 2, 1, ...)
    x = np.stack([O.synth_signal(L, seed=100 + c) for c in range(B)])
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
    full, _, _ = S.stft(x, w, ctx=ctx, **opts)
    c0, c1 = sharding.shard_channels(B, world, rank)
    outs = sharding.stft_sharded(g, [ctx.to_device(x[c0:c1])], w, axis="channels", gather=True, length=L, batch=B, **opts)
    ctx.sync()
    assert np.array_equal(outs[0].numpy().view(np.uint32), full.view(np.uint32)), "channel shards + RCCL assembly"
    M = full.shape[1]
    m0, m1, s0, s1 = sharding.shard_frames(M, N, hop, world, rank)
    outs = sharding.stft_sharded(g, [ctx.to_device(x[0, s0:s1].reshape(1, -1))], w, axis="frames", gather=True, length=L, batch=1, **opts)
    ctx.sync()
    got = outs[0].numpy()[0]
    assert float(np.max(np.abs(got - full[0])) / np.max(np.abs(full[0]))) < 1e-6, "frame shards + RCCL assembly"
    lo, hi = m0 + (m0 & 1), m1 - ((m1 - m0 - (m0 & 1)) & 1)  # own frames whose frame PAIR (2j, 2j + 1) lies inside the shard
    if m0 % 2 == 0 and hi > lo:                                 # ... ride the same transform as in the unsharded launch: bit-identical
        assert np.array_equal(got[lo:hi].view(np.uint32), full[0][lo:hi].view(np.uint32))
    # HOST tensors on a ranked group (round 5): every process holds the whole tensor; gather=True assembles the whole result in every
    # process, gather=False fills only the own part (channels c0:c1)
    zh = sharding.stft_sharded(g, x, w, axis="channels", gather=True, **opts)
    assert np.array_equal(zh.view(np.uint32), full.view(np.uint32)), "host tensors, ranked group, assembled"
    zo = sharding.stft_sharded(g, x, w, axis="channels", gather=False, **opts)
    assert np.array_equal(zo[c0:c1].view(np.uint32), full[c0:c1].view(np.uint32)), "host tensors, ranked group, own part"
    yh = sharding.fir_sharded(g, x[:3], S.filters.firwin(257, [4000.0], sampling_rate=48000), mode="same", axis="samples", gather=True)
    yf3 = np.asarray(S.filters.fir(x[:3], S.filters.firwin(257, [4000.0], sampling_rate=48000), mode="same", ctx=ctx))
    assert float(np.max(np.abs(yh - yf3)) / np.max(np.abs(yf3))) < 1e-6, "host tensors, ranked group, sample shards of 3 rows assembled"
    # frame shards of a MULTI-ROW tensor, assembled on every rank: one ncclBroadcast per (rank, row)
    xs3 = np.ascontiguousarray(x[:3, s0:s1])
    outs = sharding.stft_sharded(g, [ctx.to_device(xs3)], w, axis="frames", gather=True, length=L, batch=3, **opts)
    ctx.sync()
    got3 = outs[0].numpy()
    assert got3.shape == full[:3].shape and float(np.max(np.abs(got3 - full[:3])) / np.max(np.abs(full[:3]))) < 1e-6, "multi-row frame shards + RCCL assembly"
    # the sharded log-mel: rank 0's channels are quiet, so its clamp floor must come from the OTHER rank's maximum (ncclAllReduce max)
    xq = x.copy()
    xq[: sharding.shard_channels(B, world, 0)[1]] = x[: sharding.shard_channels(B, world, 0)[1]] * np.float32(1e-3)
    mopts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000, mel_bins=80)
    mfull = S.mel_spectrogram(xq, w, ctx, **mopts)
    mine = sharding.mel_spectrogram_sharded(g, [ctx.to_device(xq[c0:c1])], w, axis="channels", length=L, batch=B, **mopts)[0].numpy()
    assert np.array_equal(mine.view(np.uint32), mfull[c0:c1].view(np.uint32)), "sharded log-mel: all-reduced maximum"
    if rank == 0:
        alone = S.mel_spectrogram(xq[c0:c1], w, ctx, **mopts)
        assert not np.array_equal(alone, mine), "the exchange step must matter for the quiet shard"
    # sample-sharded FIR: a NaN in the LAST rank's span must turn the row NaN on every rank (ncclAllReduce max of one flag per row),
    # the clean row stays what the unsharded filter gives
    h = S.filters.firwin(257, [4000.0], sampling_rate=48000)
    xf = x[:2].copy()
    xf[1, L - 9] = np.nan
    yfull = np.asarray(S.filters.fir(xf, h, mode="same", ctx=ctx))
    n0, n1, s0, s1 = sharding.shard_fir(L, 257, world, rank, "same")
    part = sharding.fir_sharded(g, [ctx.to_device(np.ascontiguousarray(xf[:, s0:s1]))], h, mode="same", axis="samples", length=L, batch=2)[0].numpy()
    assert part.shape == (2, n1 - n0)
    assert not np.isfinite(part[1]).any(), "sample shards: the non-finite row is NaN on every rank"
    assert float(np.max(np.abs(part[0] - yfull[0, n0:n1])) / np.max(np.abs(yfull[0]))) < 1e-6, "sample shards: the clean row"
    # the same call assembled: every rank ends up with both whole rows (row 1 NaN from end to end)
    whole = sharding.fir_sharded(g, [ctx.to_device(np.ascontiguousarray(xf[:, s0:s1]))], h, mode="same", axis="samples", gather=True, length=L, batch=2)[0].numpy()
    assert whole.shape == yfull.shape and not np.isfinite(whole[1]).any(), "assembled sample shards: the non-finite row"
    assert float(np.max(np.abs(whole[0] - yfull[0])) / np.max(np.abs(yfull[0]))) < 1e-6, "assembled sample shards: the clean row"
    # a rank whose own part FAILS must not leave its peers waiting in the assembly collective (round 6: one status word is all-reduced
    # before the gather): the last rank hands in a null shard; every rank gets an error back — its own, or "another rank ... failed" —
    # and the group keeps working afterwards
    import ctypes as C

    from nx_signal_amd import _lib
    if world > 1:
        lib = _lib.load()
        xs_ok = ctx.to_device(x[c0:c1])
        zs_buf = ctx.empty((B, full.shape[1], N), np.complex64)
        pst = _lib.StftParams(N, hop, N, _lib.PAD_VALID, 0, 0, _lib.SCALE_NONE, 0, 48000.0)
        xs = (C.c_void_p * 1)(C.c_void_p(0 if rank == world - 1 else xs_ok.ptr))
        zs = (C.c_void_p * 1)(C.c_void_p(zs_buf.ptr))
        rcf = lib.nxsig_stft_sharded_f32(g.handle, xs, L, B, L, w.ctypes.data_as(C.c_void_p), C.byref(pst), _lib.SHARD_CHANNELS, 1, zs, _lib.DEVICE)
        msg = _lib.last_error()
        assert rcf != 0, "a failed rank must fail the assembled call on every rank"
        assert ("null shard pointer" in msg) if rank == world - 1 else ("another rank" in msg), msg
        outs = sharding.stft_sharded(g, [xs_ok], w, axis="channels", gather=True, length=L, batch=B, **opts)   # the group is still usable
        ctx.sync()
        assert np.array_equal(outs[0].numpy().view(np.uint32), full.view(np.uint32)), "the call after the failed one"
    g.barrier()
    print(f"RANKED-OK rank {rank} of {world}", file=sys.stderr, flush=True)
    g.close()


if __name__ == "__main__":
    main()
