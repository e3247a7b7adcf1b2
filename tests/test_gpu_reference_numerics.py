"""The two numeric rules of the reference that round 2 left as documented deviations, now closed:

* **Non-finite samples.**  The reference transforms every frame on its own (`as_windowed` -> one `Nx.fft` row per frame,
  lib/nx_signal.ex:94-102; one `Nx.ifft` row per frame, :609), so an Inf / NaN reaches exactly the frames that contain it.
  The HIP kernels pack 2 ... 16 frames into one complex transform; a unit that holds a non-finite value leaves the packed
  route.  `fir` equals `Convolution.fftconvolve` (convolution.ex:276-284), which transforms the WHOLE row once: a
  non-finite sample leaves no finite output in its row.
* **`Nx.fft` / `Nx.ifft` eps clean-up** (SURVEY App. A rule 7): every component of a transform's result with |x| <= 1e-10
  becomes 0 before anything else (scaling, window product) touches it.

Every test compares with the oracle's pattern itself (not a widened one).
"""
import numpy as np
import pytest

from oracle import nx_oracle as O

import nx_signal_amd as S

pytestmark = pytest.mark.gpu


def nerr(got, ref):
    d = np.abs(np.asarray(got).astype(np.complex128) - np.asarray(ref).astype(np.complex128))
    return float(d.max()) / max(float(np.max(np.abs(ref))), 1e-30)


def poisoned(L, batch, seed, spots):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, L)).astype(np.float32)
    for r, i, v in spots:
        x[r, i] = v
    return x


# ------------------------------------------------------------------------------------------------ stft
@pytest.mark.parametrize("K,N,hop", [
    (1024, 1024, 256), (1024, 600, 200), (512, 512, 128), (512, 400, 160), (256, 256, 64), (256, 200, 80), (128, 128, 32),
    (128, 100, 50), (2048, 2048, 512), (2048, 1500, 500), (4096, 4096, 1024), (400, 400, 160), (400, 400, 100), (1000, 1000, 250),
    (300, 300, 75), (8192, 8192, 2048), (64, 64, 16), (3000, 3000, 750),
])
@pytest.mark.parametrize("pad", ["valid", "reflect"])
def test_stft_non_finite_samples_reach_only_their_own_frames(K, N, hop, pad):
    """streaming (interior) and bounds-checked (edge) units of every front-end: pair, quad J = 2 / 4 / 8, real-2x, the 20 x 20
    kernel, Bluestein on the wave core, four passes (8192) and the generic kernels"""
    L = max(20000, 6 * K)
    x = poisoned(L, 3, K + N + hop, [(0, 7001, np.inf), (1, 12345, np.nan), (1, 3, -np.inf), (2, L - 2, np.nan), (0, 9000, np.inf)])
    w = S.windows.hann(N)
    for scaling in (None, "spectrum"):
        opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling)
        z, _, _ = S.stft(x, w, **opts)
        zo, _, _ = O.stft(x, w, **opts)
        fin, fino = np.isfinite(z).all(axis=-1), np.isfinite(zo).all(axis=-1)
        assert np.array_equal(fin, fino), (K, N, hop, pad, np.argwhere(fin != fino)[:8])
        assert 0 < (~fin).sum() < fin.size
        assert nerr(z[fin], zo[fin]) < 1e-5
        # a frame that holds a non-finite sample has no finite bin at all, here as in the reference
        assert not np.isfinite(z[~fin].real).any() or not np.isfinite(zo[~fin].real).all()


@pytest.mark.parametrize("K,N,hop", [(1024, 1024, 256), (512, 512, 128), (256, 256, 64), (128, 128, 32), (2048, 2048, 512), (400, 400, 160),
                                     (1000, 1000, 250)])
@pytest.mark.parametrize("kind", ["magnitude", "power", "onesided"])
def test_spectrogram_sinks_non_finite_samples_reach_only_their_own_frames(K, N, hop, kind):
    """the fused magnitude / power / one-sided sinks share the front-ends: same rule"""
    x = poisoned(24000, 2, K + hop, [(0, 7001, np.inf), (1, 12345, np.nan)])
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K)
    zo, _, _ = O.stft(x, w, **opts)
    fino = np.isfinite(zo[..., : K // 2]).all(axis=-1)
    if kind == "onesided":
        got = S.stft_onesided(x, w, **opts)[0]
        ref = zo[..., : K // 2]
    else:
        got = S.spectrogram(x, w, kind=kind, **opts)[0]
        ref = np.abs(zo[..., : K // 2].astype(np.complex128))
        ref = ref if kind == "magnitude" else ref * ref
    got = np.asarray(got)
    fin = np.isfinite(got).all(axis=-1)
    assert np.array_equal(fin, fino)
    assert nerr(got[fin], ref[fin]) < 2e-5


def test_stft_solo_route_equals_the_paired_route_on_finite_frames():
    """the frames of a poisoned unit that do NOT hold the sample come out equal to what the same signal gives without it"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal(40000).astype(np.float32)
    w = S.windows.hann(512)
    opts = dict(overlap_length=384, fft_length=512)
    z0, _, _ = S.stft(x, w, **opts)
    xp = x.copy()
    xp[20000] = np.inf
    z1, _, _ = S.stft(xp, w, **opts)
    fin = np.isfinite(z1).all(axis=-1)
    assert (~fin).sum() == 4            # 512 / 128 frames contain sample 20000
    assert nerr(z1[fin], z0[fin]) < 5e-7   # solo and packed transforms differ by fp32 round-off only


# ------------------------------------------------------------------------------------------------ fir
@pytest.mark.parametrize("taps,mode", [(257, "same"), (101, "same"), (129, "full"), (513, "valid"), (1025, "same"), (33, "same"), (2049, "same")])
def test_fir_non_finite_sample_poisons_its_whole_row_like_one_transform(taps, mode):
    """Convolution.fftconvolve transforms the whole row at once (convolution.ex:276-284): a row that holds an Inf / NaN has no
    finite output; the other rows are untouched.  The next call on clean data is clean again (the flags are consumed)."""
    L = 60000
    x = poisoned(L, 4, taps, [(1, 31000, np.inf), (3, 5, np.nan)])
    h = S.filters.firwin(taps, [0.2])
    y = np.asarray(S.filters.fir(x, h, mode=mode))
    yo = np.stack([O.fftconvolve(x[r], h, mode=mode) for r in range(4)])
    fin, fino = np.isfinite(y), np.isfinite(yo)
    assert np.array_equal(fin, fino)
    assert not fin[1].any() and not fin[3].any() and fin[0].all() and fin[2].all()
    assert nerr(y[[0, 2]], yo[[0, 2]]) < 1e-5
    x2 = poisoned(L, 4, taps + 1, [])
    y2 = np.asarray(S.filters.fir(x2, h, mode=mode))
    assert np.isfinite(y2).all()
    assert nerr(y2, np.stack([O.fftconvolve(x2[r], h, mode=mode) for r in range(4)])) < 1e-5


# ------------------------------------------------------------------------------------------------ istft
@pytest.mark.parametrize("N,hop", [(1024, 256), (512, 128), (512, 256), (256, 64), (256, 128), (128, 32), (128, 64), (2048, 512), (400, 100),
                                   (4096, 1024), (64, 16)])
def test_istft_non_finite_bins_reach_only_their_own_frames(N, hop):
    rng = np.random.default_rng(N + hop)
    M = 61
    z = (rng.standard_normal((2, M, N)) + 1j * rng.standard_normal((2, M, N))).astype(np.complex64)
    z[0, 20, 5] = np.inf
    z[1, 33, N - 1] = complex(0.0, np.nan)
    z[1, 0, 0] = np.nan
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N)
    y = np.asarray(S.istft(z, w, **opts))
    yo = np.stack([O.istft(z[r], w, **opts) for r in range(2)])
    fin, fino = np.isfinite(y), np.isfinite(yo)
    assert np.array_equal(fin, fino), np.argwhere(fin != fino)[:8]
    assert nerr(y[fin], yo[fino]) < 1e-5



def test_istft_non_finite_report_list_is_consumed_by_the_pass_that_reads_it():
    """The frame-packing inverse kernels append their poisoned units to a device list; the fix-up pass that redoes them puts the
    count back to zero itself (no per-call memset).  A stale count would not change a value — the redo is idempotent — so the
    check is the time: with EVERY unit poisoned the pass recomputes every sample in double (milliseconds); the clean call after
    it must not pay that again, and a smaller clean problem must not be reached through stale entries."""
    import time
    N, hop, M = 512, 128, 3000
    rng = np.random.default_rng(5)
    z = (rng.standard_normal((2, M, N)) + 1j * rng.standard_normal((2, M, N))).astype(np.complex64)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N)
    ctx = S.default_context()
    zc = ctx.to_device(z)

    def timed(buf):
        ctx.sync()
        t0 = time.perf_counter()
        y = S.istft(buf, w, ctx=ctx, **opts)
        ctx.sync()
        return time.perf_counter() - t0, y

    timed(zc)                                   # tables, first-launch costs
    t_clean0, _ = timed(zc)
    zp = z.copy()
    zp[:, ::2, 7] = np.nan                      # one bin in every frame pair
    zpd = ctx.to_device(zp)
    t_poison, yp = timed(zpd)
    assert not np.isfinite(yp.numpy()[:, N:-N]).any()
    t_clean1, y1 = timed(zc)
    y1 = y1.numpy()
    assert np.isfinite(y1).all()
    assert nerr(y1[0], O.istft(z[0], w, **opts)) < 1e-5
    assert t_poison > 5 * t_clean0, (t_poison, t_clean0)          # the redo is what costs
    assert t_clean1 < 0.5 * t_poison, (t_clean1, t_poison)        # ... and it is not paid again
    zs = ctx.to_device(z[:1, :40])
    ys = S.istft(zs, w, ctx=ctx, **opts).numpy()
    assert np.isfinite(ys).all() and nerr(ys[0], O.istft(z[0, :40], w, **opts)) < 1e-5
    for b in (zc, zpd, zs):
        b.free()

# ------------------------------------------------------------------------------------------------ eps clean-up
def _cmp_cleaned(got, ref, what):
    """`ref` holds exact zeros where the reference cleaned a component.  fp32 arithmetic puts a value within round-off of the
    1e-10 threshold on either side of it, so a handful of components may differ in the decision; every other component the
    reference zeroed must be exactly zero here, and the non-zero ones must agree as usual."""
    g = np.asarray(got).view(np.float32).reshape(-1)
    r = np.asarray(ref).astype(np.complex64).view(np.float32).reshape(-1)
    zeroed = r == 0
    assert zeroed.any(), what
    near = np.abs(np.abs(g.astype(np.float64)) - 1e-10) < 2e-15    # |x| within fp32 round-off of the threshold
    wrong = (zeroed & (g != 0)) | (~zeroed & (g == 0))
    assert not (wrong & ~near).any(), (what, int((wrong & ~near).sum()), g[wrong & ~near][:5], r[wrong & ~near][:5])
    assert wrong.sum() <= max(4, zeroed.size // 2000), (what, int(wrong.sum()))
    keep = ~zeroed & (g != 0)
    assert keep.sum() > keep.size // 20, what
    assert np.max(np.abs(g[keep] - r[keep])) <= 1e-5 * max(float(np.max(np.abs(r))), 1e-30) + 1e-15, what


@pytest.mark.parametrize("K,hop", [(1024, 256), (512, 128), (256, 64), (128, 32), (2048, 512), (4096, 1024), (400, 160), (1000, 250), (8192, 2048),
                                   (64, 16), (300, 75)])
def test_stft_zeroes_what_nx_fft_zeroes(K, hop):
    """a signal whose spectrum sits around the 1e-10 threshold: the reference returns exact zeros for the components below it"""
    w = S.windows.hann(K)
    L = K * 12
    # spectrum components of a Hann-windowed N(0, a) signal have deviation a sqrt(3 K / 16): put it at 1.5e-10
    x = (O.synth_signal(L, seed=3) * np.float32(1.5e-10 / np.sqrt(3 * K / 16))).astype(np.float32)
    for scaling in (None, "spectrum"):
        opts = dict(overlap_length=K - hop, fft_length=K, sampling_rate=48000, scaling=scaling)
        z, _, _ = S.stft(x, w, **opts)
        zo, _, _ = O.stft(x, w, **opts)
        _cmp_cleaned(z, zo, (K, scaling))


def test_stft_exact_zero_input_and_silence():
    """digital silence: every component is cleaned (+0, never -0 or round-off)"""
    x = np.zeros(1024 * 8, np.float32)
    x[5000:] = O.synth_signal(1024 * 8 - 5000, seed=1)
    z, _, _ = S.stft(x, S.windows.hann(1024), overlap_length=768, fft_length=1024)
    zo, _, _ = O.stft(x, S.windows.hann(1024), overlap_length=768, fft_length=1024)
    silent = np.all(zo == 0, axis=-1)
    assert silent.sum() >= 10
    assert np.array_equal(z[silent].view(np.uint32), zo[silent].astype(np.complex64).view(np.uint32))   # bit pattern: +0.0
    assert nerr(z, zo) < 1e-5


@pytest.mark.parametrize("N,hop", [(1024, 256), (512, 128), (256, 64), (128, 32), (2048, 512), (4096, 1024), (400, 100), (64, 16), (1000, 250)])
def test_istft_zeroes_what_nx_ifft_zeroes(N, hop):
    """the clean-up sits between the inverse transform and the scale / window product: frames whose samples are ~1e-10 lose
    the components below the threshold BEFORE the overlap-add, as in the reference"""
    rng = np.random.default_rng(N)
    M = 40
    z = ((rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))) * (1.5e-10 * np.sqrt(N))).astype(np.complex64)
    w = S.windows.rectangular(N).astype(np.float32)
    # hop == N: no overlap, the output IS the cleaned inverse transform divided by 1 -> zeros are observable one by one
    y = np.asarray(S.istft(z, w, overlap_length=0, fft_length=N))
    yo = O.istft(z, w, overlap_length=0, fft_length=N)
    _cmp_cleaned(y, yo, ("istft", N))
    wh = S.windows.hann(N)
    y = np.asarray(S.istft(z, wh, overlap_length=N - hop, fft_length=N))
    yo = O.istft(z, wh, overlap_length=N - hop, fft_length=N)
    assert nerr(y, yo) < 2e-5


@pytest.mark.parametrize("n_in,K,inverse", [(1024, 1024, False), (2048, 2048, True), (4096, 4096, False), (512, 512, True), (1000, 1000, False),
                                            (16384, 16384, False), (65536, 65536, True), (5000, 5000, False), (33, 33, True)])
def test_fft_rows_zero_what_nx_fft_zeroes(n_in, K, inverse):
    rng = np.random.default_rng(K)
    amp = 2e-10 / np.sqrt(K) if not inverse else 2e-10 * np.sqrt(K)
    a = ((rng.standard_normal((3, n_in)) + 1j * rng.standard_normal((3, n_in))) * amp).astype(np.complex64)
    got = S.transforms.ifft_nd(a, axes=[-1]) if inverse else S.transforms.fft_nd(a, axes=[-1])
    ref = O.ifft(a) if inverse else O.fft(a)
    _cmp_cleaned(got, ref, (K, inverse))


@pytest.mark.parametrize("taps", [257, 101, 1025, 2049])
def test_fir_zeroes_what_the_inverse_transform_of_fftconvolve_zeroes(taps):
    """fftconvolve's result is an Nx.ifft output (convolution.ex:282-284): samples with |y| <= 1e-10 are exact zeros"""
    L = 30000
    x = (O.synth_signal(L, seed=9) * np.float32(4e-10)).astype(np.float32)
    h = S.filters.firwin(taps, [0.25])
    y = np.asarray(S.filters.fir(x, h, mode="same"))
    yo = O.fftconvolve(x, h, mode="same")
    # the reference also cleans the two forward spectra, which an overlap-save formulation cannot see; their effect on y is
    # below 1e-10 * sum|h| ~ 1e-10, i.e. it moves some samples across the threshold: allow the decision to differ there
    g, r = y.astype(np.float64), yo.astype(np.float64)
    assert (r == 0).any() and (g == 0).any()
    far = np.abs(np.abs(r) - 1e-10) > 3e-11          # samples the spectra's clean-up cannot move across the threshold
    assert np.array_equal((g == 0)[far & (np.abs(g) < 5e-11)], (r == 0)[far & (np.abs(g) < 5e-11)])
    assert np.max(np.abs(g - r)) < 2.5e-10


# ------------------------------------------------------------------------------------------------ log-mel
@pytest.mark.parametrize("K,N,hop,pad", [(1024, 1024, 256, "valid"), (512, 400, 160, "reflect"), (400, 400, 160, "valid"), (2048, 2048, 512, "valid"),
                                         (1000, 1000, 250, "valid"), (256, 256, 64, "valid")])
def test_log_mel_non_finite_sample_poisons_the_whole_tensor_like_reduce_max(K, N, hop, pad):
    """stft_to_mel clamps against Nx.reduce_max of the WHOLE tensor (lib/nx_signal.ex:511) and its dense Nx.dot multiplies a
    non-finite |z|^2 with every band's zero weights: one Inf / NaN sample leaves no finite value.  The fused sink (every front-end)
    and the two-step path do the same; clean data afterwards is clean again."""
    x = poisoned(30000, 2, K + hop, [(1, 12345, np.inf)])
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, sampling_rate=16000)
    zo, _, _ = O.stft(x, w, **opts)
    mo = O.stft_to_mel(zo.reshape(-1, K), 16000, K, mel_bins=40).reshape(zo.shape[:-1] + (40,))   # one tensor: one reduce_max
    assert not np.isfinite(mo).any()
    fused = np.asarray(S.mel_spectrogram(x, w, mel_bins=40, **opts))
    assert fused.shape == mo.shape and not np.isfinite(fused).any()
    z, _, _ = S.stft(x, w, **opts)
    two = np.asarray(S.stft_to_mel(z, 16000, fft_length=K, mel_bins=40))
    assert not np.isfinite(two).any()
    x[1, 12345] = 0.5
    fused = np.asarray(S.mel_spectrogram(x, w, mel_bins=40, **opts))
    zo, _, _ = O.stft(x, w, **opts)
    ref = O.stft_to_mel(zo.reshape(-1, K), 16000, K, mel_bins=40).reshape(fused.shape)
    assert np.isfinite(fused).all() and float(np.max(np.abs(fused - ref))) < 1e-4
