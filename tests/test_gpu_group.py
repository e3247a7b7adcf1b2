"""Multi-GPU groups through the C ABI (include/nxsig.h "multi-GPU groups") on ONE GPU:

  * a LOCAL group with two members on device 0 runs the HIP stft / fir under shard_channels / shard_frames and the result
    equals the unsharded HIP result — bit for bit wherever the kernels see the same operands (channel shards; frame shards
    that start on an even frame, because two adjacent frames share one complex transform), to 1e-6 otherwise;
  * the assembly (nxsig_group_allgather) of unequal shards returns the full tensor on every member (device-to-device copies
    when members share a device; RCCL when every member has its own GPU — exercised here with 1-member groups, which go
    through ncclCommInitAll / ncclCommInitRank, ncclAllReduce, ncclAllGather for real);
  * a RANKED group of world size 1 comes up without a rendezvous file.
No torch anywhere in this path."""
import os
import ctypes as C

import numpy as np
import pytest

from oracle import nx_oracle as O

import nx_signal_amd as S
from nx_signal_amd import _lib, sharding

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def pair():
    g = sharding.Group.local(2, devices=[0, 0])
    yield g
    g.close()


@pytest.fixture(scope="module")
def solo():
    g = sharding.Group.local(1)
    yield g
    g.close()


def test_group_shapes(pair, solo):
    assert pair.world == 2 and pair.local_count == 2 and pair.ranks == [0, 1] and not pair.has_rccl
    assert solo.world == 1 and solo.local_count == 1 and solo.has_rccl  # ncclCommInitAll on one device
    assert "gfx950" in pair.contexts[1].name()
    pair.barrier()
    solo.barrier()
    assert solo.allreduce([1.5, -2.0], "max") == [1.5, -2.0]


@pytest.mark.parametrize("gather", [False, True])
def test_stft_channel_shards_equal_unsharded_bit_for_bit(pair, gather):
    x = np.stack([O.synth_signal(30000, seed=100 + c) for c in range(5)])  # 5 channels over 2 members: 3 + 2
    w = S.windows.hann(1024)
    opts = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
    full, _, _ = S.stft(x, w, **opts)
    got = sharding.stft_sharded(pair, x, w, axis="channels", gather=gather, **opts)
    assert got.shape == full.shape and np.array_equal(bits(got), bits(full))
    zo, _, _ = O.stft(x[4], w, **opts)
    assert float(np.max(np.abs(got[4] - zo)) / np.max(np.abs(zo))) < 1e-5


@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("L,N,hop", [(1024 + 256 * 199, 1024, 256), (1024 + 256 * 200, 1024, 256), (50000, 400, 160), (2048 + 512 * 37, 2048, 512)])
def test_stft_frame_shards(pair, gather, L, N, hop):
    x = O.synth_signal(L, seed=7)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N if N != 400 else 512, sampling_rate=16000)
    full, _, _ = S.stft(x, w, **opts)
    got = sharding.stft_sharded(pair, x, w, axis="frames", gather=gather, **opts)
    assert got.shape == full.shape
    M = full.shape[0]
    m0, m1, s0, s1 = sharding.shard_frames(M, N, hop, 2, 1)
    assert s1 == (M - 1) * hop + N and s0 == m0 * hop
    if N == 1024 and m0 % 2 == 0:
        assert np.array_equal(bits(got), bits(full))  # same frame pairs ride the same transforms
    assert float(np.max(np.abs(got - full)) / np.max(np.abs(full))) < 1e-6
    zo, _, _ = O.stft(x, w, **opts)
    assert float(np.max(np.abs(got - zo)) / np.max(np.abs(zo))) < 1e-5


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("mem", ["host", "device"])
def test_assembled_frame_and_sample_shards_of_a_multi_row_tensor(world, mem):
    """`gather=True` on frame / sample shards was refused unless batch == 1.  The reference's vectorised axes have
    no such limit (lib/nx_signal.ex:358-363): batch = 3 rows, every member ends up with the WHOLE [3, M, K] spectrum / [3, L] signal —
    rank r's dense shard goes to its place row by row (one broadcast per (rank, row) with RCCL, one strided copy per pair without)."""
    g = sharding.Group.local(world, devices=[0] * world)
    try:
        B, N, hop = 3, 1024, 256
        L = N + hop * 203 + 17                     # 204 frames: 25.5 per member at world 8 — unequal spans
        x = np.stack([O.synth_signal(L, seed=300 + c) for c in range(B)])
        w = S.windows.hann(N)
        opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000)
        full, _, _ = S.stft(x, w, **opts)
        M = full.shape[1]
        h = S.filters.firwin(257, [4000.0], sampling_rate=48000)
        yfull = np.asarray(S.filters.fir(x, h, mode="same"))
        zi = O.stft(x[0], w, **opts)[0]
        zin = np.stack([zi, zi[::-1].copy(), (zi * np.complex64(0.5 + 0.25j)).astype(np.complex64)])   # non-Hermitian rows too
        ifull = np.asarray(S.istft(zin, w, **opts))
        if mem == "host":
            got = sharding.stft_sharded(g, x, w, axis="frames", gather=True, **opts)
            yg = sharding.fir_sharded(g, x, h, mode="same", axis="samples", gather=True)
            ig = sharding.istft_sharded(g, zin, w, axis="frames", gather=True, **opts)
            assert got.shape == full.shape and float(np.max(np.abs(got - full)) / np.max(np.abs(full))) < 1e-6
            assert yg.shape == yfull.shape and float(np.max(np.abs(yg - yfull)) / np.max(np.abs(yfull))) < 1e-6
            assert np.array_equal(bits(ig), bits(ifull))
        else:
            xs, hs, zs = [], [], []
            for i, r in enumerate(g.ranks):
                m0, m1, s0, s1 = sharding.shard_frames(M, N, hop, world, r)
                xs.append(g.contexts[i].to_device(np.ascontiguousarray(x[:, s0:s1])))
                n0, n1, t0, t1 = sharding.shard_fir(L, 257, world, r, "same")
                hs.append(g.contexts[i].to_device(np.ascontiguousarray(x[:, t0:t1])))
                f0, f1, _, _ = sharding.shard_istft(M, N, hop, world, r)
                zs.append(g.contexts[i].to_device(np.ascontiguousarray(zin[:, f0:f1])))
            outs = sharding.stft_sharded(g, xs, w, axis="frames", gather=True, length=L, batch=B, **opts)
            youts = sharding.fir_sharded(g, hs, h, mode="same", axis="samples", gather=True, length=L, batch=B)
            iouts = sharding.istft_sharded(g, zs, w, axis="frames", gather=True, num_frames=M, batch=B, **opts)
            for i in range(world):                 # EVERY member holds the whole tensor
                got = outs[i].numpy()
                assert got.shape == full.shape and float(np.max(np.abs(got - full)) / np.max(np.abs(full))) < 1e-6, i
                yg = youts[i].numpy()
                assert yg.shape == yfull.shape and float(np.max(np.abs(yg - yfull)) / np.max(np.abs(yfull))) < 1e-6, i
                assert np.array_equal(bits(iouts[i].numpy()), bits(ifull)), i
    finally:
        g.close()


def test_stft_device_shards_stay_on_their_device(pair):
    """DEVICE mode: every member is handed its input shard in HBM and keeps its output shard there (no host round trip)"""
    B, L, N, hop = 4, 20000, 512, 128
    x = np.stack([O.synth_signal(L, seed=200 + c) for c in range(B)])
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=8000)
    full, _, _ = S.stft(x, w, **opts)
    shards = []
    for i, r in enumerate(pair.ranks):
        c0, c1 = sharding.shard_channels(B, pair.world, r)
        shards.append(pair.contexts[i].to_device(x[c0:c1]))
    outs = sharding.stft_sharded(pair, shards, w, axis="channels", length=L, batch=B, **opts)
    pair.sync()
    got = np.concatenate([o.numpy() for o in outs], axis=0)
    assert np.array_equal(bits(got), bits(full))
    # assembly in place: every member ends up with the whole tensor
    outs = sharding.stft_sharded(pair, shards, w, axis="channels", gather=True, length=L, batch=B, **opts)
    pair.sync()
    for o in outs:
        assert o.shape == full.shape and np.array_equal(bits(o.numpy()), bits(full))
    # frame shards of one long stream, device-resident, assembled in place
    x1 = x[0]
    M = full.shape[1]
    fsh = []
    for i, r in enumerate(pair.ranks):
        m0, m1, s0, s1 = sharding.shard_frames(M, N, hop, pair.world, r)
        fsh.append(pair.contexts[i].to_device(x1[s0:s1].reshape(1, -1)))
    outs = sharding.stft_sharded(pair, fsh, w, axis="frames", gather=True, length=L, batch=1, **opts)
    pair.sync()
    for o in outs:
        assert float(np.max(np.abs(o.numpy()[0] - full[0])) / np.max(np.abs(full[0]))) < 1e-6


@pytest.mark.parametrize("world", [2, 8])
def test_device_frame_shards_with_several_rows_and_unequal_spans(world):
    """batch > 1, frames axis, frame count not divisible by the member count: the members' dense device shards have rows of
    DIFFERENT lengths (round 2 read the later members' rows 1.. at member 0's stride: ADVICE r02)"""
    grp = sharding.Group.local(world, devices=[0] * world)
    try:
        B, N, hop = 3, 512, 128
        M = 8 * 13 + 5                                   # 109 frames: 2 x 55 - 1, 8 x 14 - 3
        L = N + hop * (M - 1)
        x = np.stack([O.synth_signal(L, seed=300 + c) for c in range(B)])
        w = S.windows.hann(N)
        opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=8000)
        full, _, _ = S.stft(x, w, **opts)
        assert full.shape[1] == M
        shards, spans = [], []
        for i, r in enumerate(grp.ranks):
            m0, m1, s0, s1 = sharding.shard_frames(M, N, hop, grp.world, r)
            spans.append((m0, m1, s1 - s0))
            shards.append(grp.contexts[i].to_device(np.ascontiguousarray(x[:, s0:s1])))
        assert len({sp[2] for sp in spans}) > 1          # the spans really differ
        outs = sharding.stft_sharded(grp, shards, w, axis="frames", length=L, batch=B, **opts)
        grp.sync()
        for (m0, m1, _), o in zip(spans, outs):
            got = o.numpy()
            assert got.shape == (B, m1 - m0, N)
            assert float(np.max(np.abs(got - full[:, m0:m1])) / np.max(np.abs(full))) < 1e-6
        mel = sharding.mel_spectrogram_sharded(grp, shards, w, axis="frames", length=L, batch=B, mel_bins=40, **opts)
        grp.sync()
        ref = S.mel_spectrogram(x, w, mel_bins=40, **opts)
        for (m0, m1, _), o in zip(spans, mel):
            assert float(np.max(np.abs(o.numpy() - np.asarray(ref)[:, m0:m1]))) < 1e-4
    finally:
        grp.close()


def test_allgather_of_unequal_shards(pair, solo):
    rng = np.random.default_rng(3)
    parts = [rng.integers(0, 2 ** 31, size=n, dtype=np.int64).astype(np.uint32) for n in (1000, 37)]
    want = np.concatenate(parts)
    send = [pair.contexts[i].to_device(parts[i]) for i in range(2)]
    recv = [pair.contexts[i].empty((want.size,), np.uint32) for i in range(2)]
    pair.allgather([s.ptr for s in send], [p.nbytes for p in parts], [r.ptr for r in recv])
    pair.sync()
    for r in recv:
        assert np.array_equal(r.numpy(), want)
    # one member with a communicator: ncclAllGather of a single rank must reproduce the shard (out of place and in place)
    s1 = solo.contexts[0].to_device(parts[0])
    r1 = solo.contexts[0].empty((parts[0].size,), np.uint32)
    solo.allgather([s1.ptr], [parts[0].nbytes], [r1.ptr])
    solo.allgather([r1.ptr], [parts[0].nbytes], [r1.ptr])
    solo.sync()
    assert np.array_equal(r1.numpy(), parts[0])


@pytest.mark.parametrize("gather", [False, True])
def test_solo_group_runs_the_rccl_assembly(solo, gather):
    x = np.stack([O.synth_signal(9000, seed=300 + c) for c in range(3)])
    w = S.windows.hann(256)
    opts = dict(overlap_length=192, fft_length=256, sampling_rate=8000)
    full, _, _ = S.stft(x, w, **opts)
    got = sharding.stft_sharded(solo, x, w, axis="channels", gather=gather, **opts)
    assert np.array_equal(bits(got), bits(full))
    got = sharding.stft_sharded(solo, x[0], w, axis="frames", gather=gather, **opts)
    assert np.array_equal(bits(got), bits(full[0]))


@pytest.mark.parametrize("mode", ["same", "full", "valid"])
def test_fir_shards(pair, mode):
    x = np.stack([O.synth_signal(100000, seed=400 + c) for c in range(3)])
    h = S.filters.firwin(257, [4000.0], sampling_rate=48000)
    full = S.filters.fir(x, h, mode=mode)
    got = sharding.fir_sharded(pair, x, h, mode=mode, axis="channels")
    assert np.array_equal(bits(got), bits(full))
    for gather in (False, True):
        got1 = sharding.fir_sharded(pair, x[0], h, mode=mode, axis="samples", gather=gather)
        assert got1.shape == full[0].shape
        assert float(np.max(np.abs(got1 - full[0])) / np.max(np.abs(full[0]))) < 1e-6
    ref = np.convolve(x[0].astype(np.float64), h.astype(np.float64), mode=mode)
    assert float(np.max(np.abs(got1 - ref)) / np.max(np.abs(ref))) < 1e-5


@pytest.mark.parametrize("mode", ["same", "full"])
def test_fir_sample_shards_agree_on_non_finite_rows(pair, solo, mode):
    """Convolution.fftconvolve filters a row with ONE transform (convolution.ex:276-284): an Inf / NaN anywhere leaves no finite
    output in that row.  A sample shard sees only its own span, so the members exchange one flag per row (all-reduce, max) and
    every member poisons the rows any member flagged: the sharded result equals the unsharded one for such rows too — and the
    next, clean call is clean (the flags are consumed)."""
    L = 90000
    x = np.stack([O.synth_signal(L, seed=410 + c) for c in range(3)])
    xp = x.copy()
    xp[1, 1000] = np.nan          # first member's span
    xp[2, L - 7] = np.inf         # second member's span
    h = S.filters.firwin(257, [4000.0], sampling_rate=48000)
    full = S.filters.fir(xp, h, mode=mode)
    assert np.isfinite(full[0]).all() and not np.isfinite(full[1]).any() and not np.isfinite(full[2]).any()
    for g in (pair, solo):
        got = sharding.fir_sharded(g, xp, h, mode=mode, axis="samples")
        assert np.array_equal(np.isfinite(got), np.isfinite(full))
        assert float(np.max(np.abs(got[0] - full[0])) / np.max(np.abs(full[0]))) < 1e-6
        clean = sharding.fir_sharded(g, x, h, mode=mode, axis="samples")
        assert np.isfinite(clean).all()
        ref = S.filters.fir(x, h, mode=mode)
        assert float(np.max(np.abs(clean - ref)) / np.max(np.abs(ref))) < 1e-6
    got1 = sharding.fir_sharded(pair, xp[1], h, mode=mode, axis="samples", gather=True)   # assembled on the device
    assert not np.isfinite(got1).any()
    # device-resident sample shards (dense per-member spans): the same exchange, the shards stay on the device
    L2 = xp.shape[1]
    spans = [sharding.shard_fir(L2, 257, pair.world, r, mode) for r in pair.ranks]
    ins = [pair.contexts[i].to_device(np.ascontiguousarray(xp[:, s0:s1])) for i, (_, _, s0, s1) in enumerate(spans)]
    outs = sharding.fir_sharded(pair, ins, h, mode=mode, axis="samples", length=L2, batch=3)
    for (n0, n1, _, _), o in zip(spans, outs):
        part = o.numpy()
        assert part.shape == (3, n1 - n0)
        assert np.array_equal(np.isfinite(part), np.isfinite(full[:, n0:n1]))
        assert float(np.max(np.abs(part[0] - full[0, n0:n1])) / np.max(np.abs(full[0]))) < 1e-6
    for b_ in ins + outs:
        b_.free()


@pytest.mark.parametrize("taps", [5, 33, 257, 600, 1025, 2049, 5001])
@pytest.mark.parametrize("bad", [np.inf, -np.inf, np.nan])
def test_fir_sample_shards_non_finite_rows_for_every_filter_length(pair, taps, bad):
    """the exchange reads a member's "row poisoned" flag off the FIRST output of its slice, which relies on "an Inf / NaN
    sample leaves no finite output in its row" holding in EVERY kernel family a slice can take — the 32-point kernel (short filters),
    the 1024- / 2048-point overlap-save kernels, the edge kernels and the single-transform path beyond 4096 taps — and for Inf as well
    as NaN.  One bad sample in the second member's span, far from the seam: every member's slice of that row must come out
    non-finite from end to end, the clean rows must equal the unsharded filter, and the next clean call must be clean."""
    L = 60000
    x = np.stack([O.synth_signal(L, seed=430 + c) for c in range(2)])
    xp = x.copy()
    xp[1, L - 2000] = bad
    h = S.filters.firwin(taps, [4000.0], sampling_rate=48000)
    full = S.filters.fir(xp, h, mode="same")
    assert np.isfinite(full[0]).all() and not np.isfinite(full[1]).any()
    got = sharding.fir_sharded(pair, xp, h, mode="same", axis="samples")
    assert np.isfinite(got[0]).all() and not np.isfinite(got[1]).any(), (taps, bad)
    assert float(np.max(np.abs(got[0] - full[0])) / np.max(np.abs(full[0]))) < 1e-5
    clean = sharding.fir_sharded(pair, x, h, mode="same", axis="samples")
    assert np.isfinite(clean).all()


def test_fir_sample_shards_with_an_empty_part_leave_no_stale_flags():
    """with more members than outputs some parts are empty (out_len == 0).  Their flags hold the group maximum after the
    all-reduce and used to stay set in the context's scratch: the next unrelated FIR call on that context then turned rows NaN."""
    g = sharding.Group.local(8, devices=[0] * 8)
    try:
        h = np.array([0.25, 0.5, 0.25], np.float32)
        x = np.array([[1.0, np.nan, 3.0, 4.0, 5.0], [1.0, 2.0, 3.0, 4.0, 5.0]], np.float32)   # 5 outputs over 8 members: 3 empty parts
        got = sharding.fir_sharded(g, x, h, mode="same", axis="samples")
        assert not np.isfinite(got[0]).any() and np.isfinite(got[1]).all()
        spans = [sharding.shard_fir(5, 3, 8, r, "same") for r in g.ranks]
        assert any(n1 == n0 for n0, n1, _, _ in spans)
        ins = [g.contexts[i].to_device(np.ascontiguousarray(x[:, s0:s1])) if s1 > s0 else g.contexts[i].empty((2, 0), np.float32) for i, (_, _, s0, s1) in enumerate(spans)]
        outs = sharding.fir_sharded(g, ins, h, mode="same", axis="samples", length=5, batch=2)
        for (n0, n1, _, _), o in zip(spans, outs):
            if n1 > n0:
                part = o.numpy()
                assert not np.isfinite(part[0]).any() and np.isfinite(part[1]).all()
        # every member's context is clean afterwards: an ordinary FIR call on it returns finite rows
        ref = np.stack([O.synth_signal(4000, seed=440 + c) for c in range(2)])
        for c in g.contexts:
            y = S.filters.fir(c.to_device(ref), h, mode="same", ctx=c).numpy()
            assert np.isfinite(y).all()
    finally:
        g.close()


@pytest.mark.parametrize("N,hop,M,scaling", [
    (1024, 256, 61, None), (1024, 512, 40, "spectrum"), (1024, 1024, 9, None), (512, 128, 77, None), (2048, 512, 23, "psd"),
    (256, 64, 130, None), (400, 160, 51, None), (96, 24, 45, None), (1024, 256, 3, None),
])
def test_istft_shards_equal_unsharded_bit_for_bit(pair, solo, N, hop, M, scaling):
    """channels axis and frame ranges (halo FRAMES recomputed, partial overlap sums dropped): identical bits, host tensors and
    dense device shards, with and without the assembly; 3 members over 2... frames fewer than members included"""
    rng = np.random.default_rng(N + hop + M)
    z = (rng.standard_normal((3, M, N)) + 1j * rng.standard_normal((3, M, N))).astype(np.complex64)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=16000, scaling=scaling)
    full = S.istft(z, w, **opts)
    sopts = {k: v for k, v in opts.items() if k != "fft_length"}
    for gather in (False, True):
        got = sharding.istft_sharded(pair, z, w, axis="channels", gather=gather, **sopts)
        assert got.shape == full.shape and np.array_equal(bits(got), bits(full))
        got1 = sharding.istft_sharded(pair, z[1], w, axis="frames", gather=gather, **sopts)
        assert got1.shape == full[1].shape and np.array_equal(bits(got1), bits(full[1]))
        got1 = sharding.istft_sharded(solo, z[1], w, axis="frames", gather=gather, **sopts)
        assert np.array_equal(bits(got1), bits(full[1]))
    got = sharding.istft_sharded(pair, z, w, axis="frames", **sopts)  # several rows, per-shard downloads
    assert np.array_equal(bits(got), bits(full))
    # dense device shards stay on their members
    shards = []
    for i, r in enumerate(pair.ranks):
        f0, f1, n0, n1 = sharding.shard_istft(M, N, hop, pair.world, r)
        shards.append(pair.contexts[i].to_device(np.ascontiguousarray(z[:, f0:f1, :])))
    outs = sharding.istft_sharded(pair, shards, w, axis="frames", num_frames=M, batch=3, **sopts)
    pos = 0
    for i, r in enumerate(pair.ranks):
        f0, f1, n0, n1 = sharding.shard_istft(M, N, hop, pair.world, r)
        assert n0 == pos and outs[i].shape == (3, n1 - n0)
        assert np.array_equal(bits(outs[i].numpy()), bits(full[:, n0:n1]))
        pos = n1
    assert pos == full.shape[-1]


def test_istft_eight_frame_shards_of_a_long_stream():
    g = sharding.Group.local(8, devices=[0] * 8)
    try:
        x = O.synth_signal(1024 + 256 * 999, seed=77)
        w = S.windows.hann(1024)
        z, _, _ = S.stft(x, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
        full = S.istft(z, w, overlap_length=768, sampling_rate=48000)
        for gather in (False, True):
            got = sharding.istft_sharded(g, z, w, axis="frames", gather=gather, overlap_length=768, sampling_rate=48000)
            assert np.array_equal(bits(got), bits(full))
        spans = [sharding.shard_istft(1000, 1024, 256, 8, r) for r in range(8)]
        assert spans[0] == (0, 128, 0, 125 * 256) and spans[1][:2] == (120, 256) and spans[7][3] == 999 * 256 + 1024
    finally:
        g.close()


@pytest.mark.parametrize("axis", ["channels", "frames"])
@pytest.mark.parametrize("K,N,hop,mb", [(1024, 1024, 256, 128), (512, 400, 160, 80), (400, 400, 160, 80), (3000, 3000, 750, 40)])
def test_log_mel_shards_all_reduce_the_global_maximum(pair, solo, axis, K, N, hop, mb):
    """the sharded log-mel has an exchange step (the clamp needs reduce_max over the WHOLE tensor, lib/nx_signal.ex:511): two
    members whose shards have very different levels — only an all-reduced maximum clamps the quiet shard like the unsharded call
    does.  `pair`: members sharing the device reduce through the host; `solo`: the RCCL ncclAllReduce path (world of one)."""
    rng = np.random.default_rng(K + hop)
    L = N + hop * 333 + 5
    x = rng.standard_normal((4, L)).astype(np.float32)
    x[:2] *= np.float32(1e-3)            # channels shard: member 0 quiet, member 1 loud
    x[:, : L // 2] *= np.float32(1e-2)   # frames shard: the first half quiet
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K, sampling_rate=16000, mel_bins=mb)
    full = S.mel_spectrogram(x, w, **opts)
    lo = float(full.max()) - 2.0 - 1e-6  # (max - 8 + 4) / 4: the clamp floor of the whole tensor
    assert float(full.min()) >= lo and np.any(full[:2] <= lo + 1e-5)   # the quiet channels do hit the global floor
    for grp in (pair, solo):
        got = sharding.mel_spectrogram_sharded(grp, x, w, axis=axis, **opts)
        assert got.shape == full.shape and got.dtype == np.float32
        if axis == "channels":
            assert np.array_equal(bits(got), bits(full))                # same kernel per row, same maximum: identical
        else:
            assert float(np.max(np.abs(got - full))) < 2e-5             # frame pairing differs at the shard edge: fp32 rounding
    xd = [pair.contexts[i].to_device(x[2 * i: 2 * i + 2]) for i in range(2)]   # device-resident shards stay on their devices
    outs = sharding.mel_spectrogram_sharded(pair, xd, w, axis="channels", length=L, batch=4, **opts)
    assert np.array_equal(bits(np.concatenate([o.numpy() for o in outs])), bits(full))
    with pytest.raises(S.ArgumentError):
        sharding.mel_spectrogram_sharded(pair, x, w, window_padding="reflect", **opts)


def test_ranked_group_of_one():
    g = sharding.Group.ranked(world=1, rank=0, device=0, path="")
    try:
        assert g.world == 1 and g.has_rccl
        g.barrier()
        assert g.allreduce([3.0, 4.0], "max") == [3.0, 4.0]
        assert g.allreduce([3.0], "sum") == [3.0]
    finally:
        g.close()


def test_sharded_rejects_padding_modes(pair):
    x = O.synth_signal(5000, seed=1)
    with pytest.raises(_lib.ArgumentError, match="valid"):
        sharding.stft_sharded(pair, x, S.windows.hann(64), axis="frames", window_padding="reflect")


def test_config4_and_config5_partitioning_with_eight_members():
    """BASELINE configs 4 / 5 shard 64 channels over 8 GPUs (8 contiguous channels each, no halo, no exchange).  Eight members
    on the one GPU of this box run exactly that plan (N=2048 hop=512 Hann; 257-tap low-pass) on a shortened stream and must
    reproduce the unsharded HIP result bit for bit, with and without the assembly."""
    g = sharding.Group.local(8, devices=[0] * 8)
    try:
        assert [sharding.shard_channels(64, 8, r) for r in (0, 3, 7)] == [(0, 8), (24, 32), (56, 64)]
        x = np.stack([np.roll(O.synth_signal(40000, seed=900), 131 * c) * np.float32(1 + 0.01 * c) for c in range(64)])
        w = S.windows.hann(2048)
        opts = dict(overlap_length=2048 - 512, fft_length=2048, sampling_rate=48000)
        full, _, _ = S.stft(x, w, **opts)
        for gather in (False, True):
            got = sharding.stft_sharded(g, x, w, axis="channels", gather=gather, **opts)
            assert np.array_equal(bits(got), bits(full))
        h = S.filters.firwin(257, [4000.0], sampling_rate=48000)
        yfull = S.filters.fir(x, h, mode="same")
        for gather in (False, True):
            got = sharding.fir_sharded(g, x, h, mode="same", axis="channels", gather=gather)
            assert np.array_equal(bits(got), bits(yfull))
        # one long stream split into 8 frame ranges with a 1536-sample input halo each
        long = O.synth_signal(2048 + 512 * 799, seed=901)
        zl, _, _ = S.stft(long, w, **opts)
        got = sharding.stft_sharded(g, long, w, axis="frames", gather=True, **opts)
        assert got.shape == zl.shape and float(np.max(np.abs(got - zl)) / np.max(np.abs(zl))) < 1e-6
    finally:
        g.close()


def test_the_loaded_rccl_is_reported_and_new_enough(pair):
    """round 6: the library prefers ROCm's own librccl over whatever the bare soname resolves to, refuses one older than 2.18.0
    (group.cpp: kMinRccl) and reports path + version of what it loaded (nxsig_rccl_info; one stderr line at the first group)"""
    import ctypes as C

    from nx_signal_amd import _lib

    lib = _lib.load()
    v, buf = C.c_int32(0), C.create_string_buffer(512)
    assert lib.nxsig_rccl_info(C.byref(v), buf, 512) == 0, _lib.last_error()
    path = buf.value.decode()
    assert v.value >= 21800, v.value
    assert "librccl" in path and os.path.exists(path), path
    if os.path.exists("/opt/rocm/lib/librccl.so.1") and not os.environ.get("NXSIG_RCCL_LIB"):
        assert os.path.realpath(path) == os.path.realpath("/opt/rocm/lib/librccl.so.1"), path
