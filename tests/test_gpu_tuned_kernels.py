"""GPU parity tests aimed at the tuned wave kernels (kernels_wave.hip): every template variant and every seam
(odd frame counts, batch rows, chunk boundaries, run boundaries of the istft, stream/edge split of the FIR) against
the oracle, and against the generic kernels (NXSIG_DISABLE_WAVE=1 cannot be flipped inside one process, so the
generic path is exercised through shapes the tuned kernels decline)."""
import numpy as np
import pytest

from oracle import nx_oracle as O

import nx_signal_amd as S

pytestmark = pytest.mark.gpu


def nerr(got, ref):
    d = np.abs(np.asarray(got).astype(np.complex128) - np.asarray(ref).astype(np.complex128))
    return float(d.max()) / max(float(np.max(np.abs(ref))), 1e-30)


@pytest.mark.parametrize("K,N,hop,L,batch", [
    (1024, 1024, 256, 1024 + 256 * 6, 3),        # M = 7 (odd) per row, 3 rows: phantom second frame at every row end
    (1024, 1024, 256, 1024, 2),                  # M = 1: a lone frame per row
    (1024, 1024, 1, 1024 + 37, 1),               # hop 1
    (1024, 1024, 333, 1024 + 333 * 40 + 5, 2),   # hop not a multiple of anything, ragged tail dropped (:valid)
    (1024, 1000, 250, 20000, 2),                 # N < K: zero-padded frames (general path: reads must stop at N)
    (1024, 1500, 500, 20000, 1),                 # N > K: truncated frames
    (512, 512, 128, 512 + 128 * 8, 3),           # quad front-end: M = 9 (M % 4 == 1): three phantom frames per row
    (512, 512, 128, 512 + 128 * 9, 2),           # M % 4 == 2
    (512, 512, 128, 512 + 128 * 10, 2),          # M % 4 == 3
    (512, 512, 128, 512 + 128 * 11, 2),          # M % 4 == 0
    (512, 512, 160, 40000, 2),                   # speech-style hop
    (512, 400, 160, 16000, 1),                   # N = 400 < K = 512 (25 ms frames at 16 kHz): general path
    (512, 512, 128, 512, 1),                     # a lone frame
    (256, 256, 64, 256 + 64 * 16, 3),            # J = 4: M = 17 (M % 8 == 1), 3 rows
    (256, 256, 64, 256 + 64 * 21, 2),            # M % 8 == 6
    (256, 200, 80, 30000, 2),                    # N < K
    (256, 256, 64, 256, 1),
    (128, 128, 32, 128 + 32 * 32, 3),            # J = 8: M = 33 (M % 16 == 1)
    (128, 128, 32, 128 + 32 * 46, 2),            # M % 16 == 15
    (128, 128, 1, 128 + 100, 1),                 # hop 1
    (128, 100, 50, 9000, 2),
    (2048, 2048, 512, 2048 + 512 * 9, 3),        # real-2x front-end, odd M, 3 rows
    (2048, 2048, 511, 50000, 1),                 # odd hop: unaligned 4-byte loads
    (2048, 1024, 256, 30000, 2),                 # N < K on the real-2x front-end
    (4096, 4096, 1024, 4096 + 1024 * 9, 3),      # 2048-point core (32 points per lane), real-2x front-end
    (4096, 4096, 1023, 90000, 1),                # odd hop
    (4096, 2048, 512, 60000, 2),                 # N < K
    (4096, 4096, 4096, 4096 * 5, 2),             # no overlap
    (4096, 4096, 512, 4096, 1),                  # a lone frame
])
def test_stft_wave_variants(K, N, hop, L, batch):
    rng = np.random.default_rng(K + N + hop + L)
    x = rng.standard_normal((batch, L)).astype(np.float32)
    w = S.windows.hann(N)
    for scaling in (None, "psd"):
        opts = dict(overlap_length=N - hop, fft_length=K, sampling_rate=16000, scaling=scaling)
        z, _, _ = S.stft(x, w, **opts)
        zo, _, _ = O.stft(x, w, **opts)
        assert z.shape == zo.shape
        assert nerr(z, zo) < 1e-5, (K, N, hop, scaling, nerr(z, zo))


def test_stft_wave_chunk_seams_many_rows():
    """more rows x frames than workgroups: chunks straddle row boundaries"""
    rng = np.random.default_rng(77)
    x = rng.standard_normal((37, 1024 + 256 * 300 + 3)).astype(np.float32)
    w = S.windows.hamming(1024)
    z, _, _ = S.stft(x, w, overlap_length=768, fft_length=1024)
    zo, _, _ = O.stft(x, w, overlap_length=768, fft_length=1024)
    assert nerr(z, zo) < 1e-5


@pytest.mark.parametrize("pad", ["reflect", "same", [(100, 900)], [(-3, 50)]])
@pytest.mark.parametrize("K", [128, 256, 512, 1024, 2048, 4096])
def test_stft_wave_general_padding(pad, K):
    rng = np.random.default_rng(5 + K)
    if isinstance(pad, list) and K < 1024:
        pad = [(pad[0][0] // 8, pad[0][1] // 8)] if pad[0][0] > 0 else pad
    x = rng.standard_normal((2, 9000)).astype(np.float32)
    w = S.windows.blackman(K)
    opts = dict(overlap_length=K - K // 4, fft_length=K, window_padding=pad, scaling="spectrum", sampling_rate=8000)
    z, t, f = S.stft(x, w, **opts)
    zo, to, fo = O.stft(x, w, **opts)
    assert z.shape == zo.shape and nerr(z, zo) < 1e-5
    assert np.array_equal(t, to) and np.array_equal(f, fo)


@pytest.mark.parametrize("hop", [128, 256, 512, 1024])
@pytest.mark.parametrize("M", [7, 15, 64, 301])
def test_istft_wave_all_hops_and_run_seams(hop, M):
    """tuned istft (N = 1024): R = 8, 4, 2, 1; M spans 'fewer frames than 2R-1' (generic path) to several runs"""
    N = 1024
    rng = np.random.default_rng(hop + M)
    z = (rng.standard_normal((3, M, N)) + 1j * rng.standard_normal((3, M, N))).astype(np.complex64)
    w = S.windows.hann(N)
    for scaling in (None, "spectrum"):
        y = S.istft(z, w, overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=48000)
        yo = O.istft(z, w, overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=48000)
        assert y.shape == yo.shape
        assert nerr(y, yo) < 1e-5, (hop, M, scaling, nerr(y, yo))


@pytest.mark.parametrize("hop", [64, 128, 256, 512])
@pytest.mark.parametrize("M", [3, 16, 17, 64, 301])
def test_istft_wave_half_n512(hop, M):
    """N = 512: two frames per 1024-point inverse FFT (odd and even frame counts, every supported hop, run seams)"""
    N = 512
    rng = np.random.default_rng(hop * 7 + M)
    z = (rng.standard_normal((3, M, N)) + 1j * rng.standard_normal((3, M, N))).astype(np.complex64)
    w = S.windows.hann(N)
    for scaling in (None, "psd"):
        y = S.istft(z, w, overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=16000)
        yo = O.istft(z, w, overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=16000)
        assert y.shape == yo.shape
        assert nerr(y, yo) < 1e-5, (hop, M, scaling, nerr(y, yo))


@pytest.mark.parametrize("hop", [256, 512, 1024, 2048])
@pytest.mark.parametrize("M", [2, 9, 15, 64, 157])
def test_istft_wave_dbl_n2048(hop, M):
    """N = 2048: one frame per two 1024-point inverse FFTs (even / odd bins); every supported hop, run seams, and
    frame counts below 2R-1 that take the generic path"""
    N = 2048
    rng = np.random.default_rng(hop * 3 + M)
    z = (rng.standard_normal((2, M, N)) + 1j * rng.standard_normal((2, M, N))).astype(np.complex64)
    w = S.windows.hann(N)
    for scaling in (None, "spectrum"):
        y = S.istft(z, w, overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=48000)
        yo = O.istft(z, w, overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=48000)
        assert y.shape == yo.shape
        assert nerr(y, yo) < 1e-5, (hop, M, scaling, nerr(y, yo))


@pytest.mark.parametrize("N,hop", [(256, 32), (256, 64), (256, 128), (256, 256), (128, 16), (128, 32), (128, 64), (128, 128)])
@pytest.mark.parametrize("M", [3, 16, 17, 66, 301])
def test_istft_wave_quad_n256_n128(N, hop, M):
    """N = 256 / 128: 4 / 8 frames per 1024-point inverse FFT, overlap-add through the wave's LDS buffer; frame counts
    that are not multiples of the unit, run seams, every supported hop (M < 2R-1 takes the generic path)"""
    rng = np.random.default_rng(N + hop * 5 + M)
    z = (rng.standard_normal((3, M, N)) + 1j * rng.standard_normal((3, M, N))).astype(np.complex64)
    w = S.windows.hann(N)
    for scaling in (None, "spectrum"):
        y = S.istft(z, w, overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=16000)
        yo = O.istft(z, w, overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=16000)
        assert y.shape == yo.shape
        assert nerr(y, yo) < 1e-5, (N, hop, M, scaling, nerr(y, yo))


@pytest.mark.parametrize("hop", [160, 100, 200, 400, 50, 80, 2, 398, 133])
@pytest.mark.parametrize("M", [3, 16, 17, 66, 301])
def test_istft_r20_n400(hop, M):
    """N = 400 natively (20 x 20 inverse transform, three frames per wave iteration, gathered overlap-add): hops that do
    and do not divide the frame length, unit / run seams, frame counts below 2R-1 and an odd hop (generic path)"""
    N = 400
    if hop <= 4 and M > 66:
        pytest.skip("tiny hop: covered at smaller M")
    rng = np.random.default_rng(hop * 11 + M)
    z = (rng.standard_normal((2, M, N)) + 1j * rng.standard_normal((2, M, N))).astype(np.complex64)
    w = S.windows.hann(N)
    for scaling in (None, "spectrum"):
        y = S.istft(z, w, overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=16000)
        yo = O.istft(z, w, overlap_length=N - hop, fft_length=N, scaling=scaling, sampling_rate=16000)
        assert y.shape == yo.shape
        assert nerr(y, yo) < 1e-5, (hop, M, scaling, nerr(y, yo))


def test_istft_rectangular_window_no_edge_fix_needed():
    N, hop, M = 1024, 256, 40
    rng = np.random.default_rng(3)
    z = (rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))).astype(np.complex64)
    w = S.windows.rectangular(N, type="f32")
    y = S.istft(z, w, overlap_length=N - hop, fft_length=N)
    assert nerr(y, O.istft(z, w, overlap_length=N - hop, fft_length=N)) < 1e-5


@pytest.mark.parametrize("taps", [1, 2, 65, 129, 257, 385, 513, 300, 514, 641, 769, 1000, 1025, 1026])
@pytest.mark.parametrize("mode", ["full", "same", "valid"])
def test_fir_wave_stream_edge_split(taps, mode):
    """taps-1 multiples of 128 take the vectorised streaming kernel for interior block pairs and the bounds-checked
    kernel at the row ends; other tap counts take the bounds-checked kernel throughout; 514..1025 taps run on 2048-sample
    blocks (the 2048-point core), longer filters on the generic kernel"""
    rng = np.random.default_rng(taps)
    L = 20000 + (taps % 7)
    x = rng.standard_normal((3, L)).astype(np.float32)
    h = rng.standard_normal(taps).astype(np.float32) / max(1, taps) ** 0.5
    ctx = S.default_context()
    y = S.convolution.convolve(ctx.to_device(x), h, method="fft", mode=mode).numpy()
    n = {"full": L + taps - 1, "same": L, "valid": abs(L - taps) + 1}[mode]
    assert y.shape == (3, n)
    for r in range(3):
        full = O.direct_convolve_f64(x[r], h)
        start = (full.shape[0] - n) // 2 if mode != "full" else 0
        ref = full[start:start + n]
        assert nerr(y[r], ref) < 1e-5, (taps, mode, r, nerr(y[r], ref))


def test_fir_short_signal_and_odd_alignment():
    rng = np.random.default_rng(9)
    for L in (1, 5, 700, 1023, 1025, 3001):
        x = rng.standard_normal(L).astype(np.float32)
        h = S.filters.firwin(257, [0.2])
        y = S.filters.fir(x, h, mode="same")
        full = O.direct_convolve_f64(x, h)
        ref = full[128:128 + L]
        assert y.shape == (L,) and nerr(y, ref) < 1e-5, L


def test_device_resident_chain_matches_host_path():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 50000)).astype(np.float32)
    w = S.windows.hann(1024)
    opts = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
    ctx = S.default_context()
    zd, _, _ = S.stft(ctx.to_device(x), w, **opts)
    yd = S.istft(zd, w, **opts)
    zh, _, _ = S.stft(x, w, **opts)
    yh = S.istft(zh, w, **opts)
    assert np.array_equal(zd.numpy().view(np.uint32), zh.view(np.uint32))
    assert np.array_equal(yd.numpy().view(np.uint32), yh.view(np.uint32))
    fr = S.as_windowed(ctx.to_device(x), window_length=64, stride=16)
    assert np.array_equal(fr.numpy(), S.as_windowed(x, window_length=64, stride=16))
    ola = S.overlap_and_add(fr, overlap_length=48)
    assert np.array_equal(ola.numpy(), S.overlap_and_add(fr.numpy(), overlap_length=48))


def test_concurrent_contexts_and_threads():
    """dirty-NIF style usage: several OS threads, one context each, same GPU"""
    import threading

    x = O.synth_signal(100000, seed=4)
    w = S.windows.hann(1024)
    ref, _, _ = O.stft(x, w, overlap_length=768, fft_length=1024)
    errs = []

    def work():
        c = S.Context(0)
        for _ in range(5):
            z, _, _ = S.stft(x, w, ctx=c, overlap_length=768, fft_length=1024)
            errs.append(nerr(z, ref))
        c.close()

    ts = [threading.Thread(target=work) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert len(errs) == 20 and max(errs) < 1e-5


@pytest.mark.parametrize("K,N,hop,L,batch", [
    (400, 400, 160, 16000, 2),        # 25 ms / 10 ms speech framing at 16 kHz
    (400, 400, 160, 400 + 160 * 6, 3),  # odd frame count: phantom second frame of the last pair
    (100, 100, 25, 3000, 2),
    (500, 300, 100, 9000, 1),         # N < K: zero-padded frames
    (511, 511, 128, 8000, 2),         # odd length: odd-aligned output rows
    (512 - 1, 600, 200, 9000, 1),     # N > K: truncated frames
    (513, 513, 171, 9000, 2),         # first length on the 2048-point core
    (1000, 1000, 250, 30000, 2),
    (1023, 1023, 512, 20000, 1),
    (17, 17, 5, 400, 2),
    (400, 400, 160, 400, 1),          # a lone frame
])
def test_stft_bluestein_wave(K, N, hop, L, batch):
    """non-power-of-two fft_length <= 1024: chirp-z through the wave core, two frames per convolution"""
    rng = np.random.default_rng(K * 3 + N + hop)
    x = rng.standard_normal((batch, L)).astype(np.float32)
    w = S.windows.hann(N)
    for scaling in (None, "spectrum"):
        opts = dict(overlap_length=N - hop, fft_length=K, sampling_rate=16000, scaling=scaling)
        z, t, f = S.stft(x, w, **opts)
        zo, to, fo = O.stft(x, w, **opts)
        assert z.shape == zo.shape
        assert nerr(z, zo) < 1e-5, (K, N, hop, scaling, nerr(z, zo))
        assert np.array_equal(t, to, equal_nan=True) and np.array_equal(f, fo)  # M == 1: the reference's times are NaN


@pytest.mark.parametrize("pad", ["reflect", "same", [(37, 211)]])
@pytest.mark.parametrize("K", [400, 1000])
def test_stft_bluestein_wave_padding(pad, K):
    rng = np.random.default_rng(K)
    x = rng.standard_normal((2, 7000)).astype(np.float32)
    w = S.windows.hamming(K)
    opts = dict(overlap_length=K - K // 4, fft_length=K, window_padding=pad, scaling="psd", sampling_rate=8000)
    z, _, _ = S.stft(x, w, **opts)
    zo, _, _ = O.stft(x, w, **opts)
    assert z.shape == zo.shape and nerr(z, zo) < 1e-5


@pytest.mark.parametrize("K,N,hop", [(1024, 600, 200), (512, 400, 160), (2048, 1500, 500), (256, 200, 80), (128, 100, 40)])
def test_stft_interior_units_do_not_leak_samples_past_the_frame(K, N, hop):
    """frame_length < fft_length on the streaming kernels: an Inf that lies past a frame's last sample (but inside the
    fft_length samples the kernel loads) must not contaminate that frame, and frames that share one complex transform
    (2 at K=1024, 4 at 512, 8 at 256, 16 at 128) must not share non-finite values either: the finite / non-finite pattern
    is the oracle's, frame by frame (tests/test_gpu_reference_numerics.py covers every front-end and sink)."""
    rng = np.random.default_rng(K + N)
    x = rng.standard_normal((2, 20000)).astype(np.float32)
    x[0, 7001] = np.inf
    x[1, 12345] = np.nan
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K)
    z, _, _ = S.stft(x, w, **opts)
    zo, _, _ = O.stft(x, w, **opts)
    fin, fino = np.isfinite(z).all(axis=-1), np.isfinite(zo).all(axis=-1)
    assert np.array_equal(fin, fino)
    assert 0 < (~fin).sum() < fin.size // 4
    assert nerr(z[fin], zo[fin]) < 1e-5


@pytest.mark.parametrize("N,hop,pad,scaling,L", [
    (8192, 2048, "valid", None, 8192 * 5 + 77), (8192, 1024, "reflect", "spectrum", 8192 * 3 + 1), (8192, 8192, "same", None, 8192 * 4),
    (8192, 4096, [(100, 3000)], "psd", 8192 * 3), (6000, 1500, "valid", None, 50001), (6000, 2000, "reflect", None, 30000),
    (8192, 2047, "valid", None, 8192 * 3 + 5),
])
def test_stft_8192_on_four_passes_of_the_1024_core(N, hop, pad, scaling, L):
    """fft_length 8192 (kernels_wave_8k.hip): interior frames stream, edge frames / padding modes / short frames (N < 8192)
    take the bounds-checked instantiation; odd hops exercise the unaligned loads; batch rows and > 1 frame per wave"""
    x = np.stack([O.synth_signal(L, seed=70 + c) for c in range(3)])
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=8192, sampling_rate=48000, window_padding=pad, scaling=scaling)
    z, t, f = S.stft(x, w, **opts)
    for c in range(3):
        zo, to, fo = O.stft(x[c], w, **opts)
        d = float(np.max(np.abs(z[c] - zo)) / np.max(np.abs(zo)))
        assert z[c].shape == zo.shape and d < 1e-5, (N, hop, pad, c, d)
    assert np.array_equal(t, to) and np.array_equal(f, fo)
    zd, _, _ = S.stft(S.default_context().to_device(x[:, 1:]), w, **opts)  # odd (4-byte aligned only) base address
    zo, _, _ = O.stft(x[0, 1:], w, **opts)
    assert float(np.max(np.abs(zd.numpy()[0] - zo)) / np.max(np.abs(zo))) < 1e-5


@pytest.mark.parametrize("hop,scaling,M,batch", [(1024, None, 40, 3), (512, "spectrum", 23, 2), (2048, "psd", 9, 2), (4096, None, 5, 1), (1024, None, 7, 1),
                                                 (1024, None, 300, 8)])
def test_istft_4096_on_four_passes_of_the_1024_core(hop, scaling, M, batch):
    """istft N = 4096 (k_istft_wave_4k): non-Hermitian random spectra (the output is complex like the reference's), every hop
    the kernel takes, scaling modes, runs that start mid-row (halo frames), short inputs just above the 2R - 1 frame minimum"""
    N = 4096
    rng = np.random.default_rng(hop + M)
    z = (rng.standard_normal((batch, M, N)) + 1j * rng.standard_normal((batch, M, N))).astype(np.complex64)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=48000, scaling=scaling)
    y = S.istft(z, w, **opts)
    for b in range(min(batch, 2)):
        yo = O.istft(z[b], w, **opts)
        d = float(np.max(np.abs(y[b] - yo)) / np.max(np.abs(yo)))
        assert y[b].shape == yo.shape and d < 1e-5, (hop, scaling, M, b, d)
    if hop == N:
        return  # no overlap under a tapered window: the reference itself does not round-trip (normaliser guard, quirk B9)
    x = O.synth_signal(N + hop * (M - 1), seed=5)  # round trip of a real signal through stft (tuned fft_length-4096 kernel) and back
    zs, _, _ = S.stft(x, w, **opts)
    ys = S.istft(zs, w, **opts)
    core = slice(N, x.size - N) if x.size > 3 * N else slice(N // 2, N // 2 + 16)
    assert float(np.max(np.abs(ys.real[core] - x[core]))) < 1e-4 * max(1.0, float(np.max(np.abs(x))))


@pytest.mark.parametrize("taps,L,mode", [(5000, 60000, "same"), (20000, 100000, "full"), (48000, 48000 * 4, "valid"), (4097, 9000, "same"),
                                          (96001, 3_000_000, "same"),     # one 2^22-point transform per row (five-pass four-step)
                                          (70001, 900_000, "full")])      # 2^20 points: the two-pass tiled four-step
def test_fir_with_very_long_filters(taps, L, mode):
    """more than 4096 taps (reverb-length impulse responses): one big transform per row like the reference's fftconvolve"""
    from scipy import signal as ss
    rng = np.random.default_rng(taps)
    x = rng.standard_normal((2, L)).astype(np.float32)
    h = (rng.standard_normal(taps) * np.exp(-np.arange(taps) / (taps / 6.0)) / np.sqrt(taps / 12.0)).astype(np.float32)
    y = S.filters.fir(x, h, mode=mode)
    for r in range(2):
        ref = ss.fftconvolve(x[r].astype(np.float64), h.astype(np.float64), mode=mode)
        assert y[r].shape == ref.shape
        assert float(np.max(np.abs(y[r] - ref)) / np.max(np.abs(ref))) < 1e-5
    yd = S.filters.fir(S.default_context().to_device(x), h, mode=mode)
    assert np.array_equal(yd.numpy().view(np.uint32), y.view(np.uint32))


def test_freed_device_buffers_are_reused_not_reallocated():
    """nxsig_alloc / nxsig_free cache blocks per context: a caller that allocates its result per call (this mirror, the NIF's
    *_dev functions) must not pay a hipMalloc + hipFree of gigabytes every time (measured: 100 ms for 3.3 GB against 0.55 ms
    for the stft that fills it)"""
    import time
    ctx = S.Context(0)
    try:
        a = ctx.empty((3, 1 << 20), np.float32)
        p = a.ptr
        a.free()
        b = ctx.empty((3, 1 << 20), np.float32)       # same size: the parked block comes back
        assert b.ptr == p
        c = ctx.empty((3, (1 << 20) - 1000), np.float32)  # a similar size while b is live: a different block
        assert c.ptr != p
        b.free(); c.free()
        x = ctx.to_device(np.random.default_rng(0).standard_normal((8, 48000 * 60)).astype(np.float32))
        w = S.windows.hann(1024)
        z, _, _ = S.stft(x, w, ctx=ctx, overlap_length=768, sampling_rate=48000)   # first call allocates 737 MB
        ref = z.numpy()[3, 1000]
        del z
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            z, _, _ = S.stft(x, w, ctx=ctx, overlap_length=768, sampling_rate=48000)
            del z
        ctx.sync()
        per_call = (time.perf_counter() - t0) / 10
        assert per_call < 5e-3, per_call   # kernel ~0.15 ms; a fresh hipMalloc of 737 MB alone costs ~20 ms
        z, _, _ = S.stft(x, w, ctx=ctx, overlap_length=768, sampling_rate=48000)
        assert np.array_equal(z.numpy()[3, 1000].view(np.uint32), ref.view(np.uint32))
    finally:
        ctx.close()


@pytest.mark.parametrize("K,N,hop,pad,scaling", [
    (1024, 1024, 256, "valid", None), (1024, 1024, 512, "reflect", "spectrum"), (1024, 600, 200, "valid", None),
    (512, 400, 160, "reflect", None), (512, 512, 128, "valid", "psd"), (256, 256, 64, "same", None), (128, 128, 32, "valid", None),
    (2048, 2048, 512, "valid", None), (4096, 3000, 1000, "valid", "spectrum"),
    (400, 400, 160, "valid", None), (1000, 1000, 250, "valid", None), (64, 64, 16, "valid", None), (8192, 8192, 2048, "valid", None),
    (400, 400, 160, "reflect", "psd"), (400, 400, 100, "same", None), (400, 320, 160, "reflect", "spectrum"),   # 20 x 20 kernel's own sink
])
def test_stft_onesided_equals_the_first_half_of_stft_bit_for_bit(K, N, hop, pad, scaling):
    """the one-sided sink (tuned front-ends) and its two-step form (every other length): same bits as stft()[..., :K // 2]"""
    rng = np.random.default_rng(K + N + hop)
    x = rng.standard_normal((3, 30000 + 7)).astype(np.float32)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K, sampling_rate=16000, window_padding=pad, scaling=scaling)
    z, t, f = S.stft(x, w, **opts)
    h, th, fh = S.stft_onesided(x, w, **opts)
    assert h.shape == z.shape[:-1] + (K // 2,) and np.array_equal(th, t) and np.array_equal(fh, f[: K // 2])
    assert np.array_equal(h.view(np.uint32), np.ascontiguousarray(z[..., : K // 2]).view(np.uint32))
    hd, _, _ = S.stft_onesided(S.default_context().to_device(x), w, **opts)
    assert np.array_equal(hd.numpy().view(np.uint32), h.view(np.uint32))


# every fft length with a native A x B kernel (wave_rab.hpp: four lists, four translation units)
RAB_LENGTHS = [441, 882, 1764, 2205, 2400, 2880, 3840,   # round 6: radix 7 (10 / 20 / 40 / 50 ms at 44.1 kHz; 441 and 2205 = 35 x 63 are ODD: 8-byte drains) and the 50 / 60 / 80 ms frames of 48 kHz audio
               192, 288, 576, 1152, 1440, 1536, 1920, 32, 64, 100, 120, 160, 200, 240, 300, 360, 384, 320, 480, 640, 960, 500, 600, 720, 768, 800, 900, 1000, 1200, 1280, 1600]


@pytest.mark.parametrize("K", RAB_LENGTHS)
@pytest.mark.parametrize("shape", ["full", "short_frame", "long_frame", "reflect", "odd_hop", "ragged"])
def test_stft_composite_lengths_native_kernels(K, shape):
    """kernels_wave_rab*.hip (round 5): fft_length 320 = 16 x 20, 480 = 24 x 20, 640 = 32 x 20, 960 = 32 x 30 ... 1600 = 40 x 40 on two-pass
    wave kernels instead of Bluestein (<= 1024) or the generic kernels (above).  Against the oracle for: N == K at 75 % overlap, a shorter frame (zero-padded), a longer frame (truncated,
    lib/nx_signal.ex:102), :reflect padding (edge units through the bounds-checked staging), an odd hop (unaligned spans: 4-byte
    staging) and ragged frame counts (phantom frames of the last unit); every scaling; and the switch back to Bluestein agrees."""
    import nx_signal_amd as S
    from oracle import nx_oracle as O

    rng = np.random.default_rng(K)
    N, hop, pad, L = K, K // 4, "valid", 7 * K + 13
    if shape == "short_frame":
        N = K - K // 4
    elif shape == "long_frame":
        N = K + 64
    elif shape == "reflect":
        pad = "reflect"
    elif shape == "odd_hop":
        hop = K // 4 + 1
    elif shape == "ragged":
        L = 3 * K + hop * 5 + 1
    x = rng.standard_normal((3, L)).astype(np.float32)
    w = S.windows.hamming(N)
    for scaling in (None, "spectrum", "psd"):
        opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=16000)
        z, t, f = S.stft(x, w, **opts)
        zo, to, fo = O.stft(x, w, **opts)
        assert z.shape == zo.shape
        assert float(np.max(np.abs(z - zo)) / np.max(np.abs(zo))) < 1e-5, (K, shape, scaling)
        assert np.array_equal(t, to) and np.array_equal(f, fo)
    ctx = S.Context(0)
    native = not ctx.get_tuning("DISABLE_RAB")[0] and not ctx.get_tuning("DISABLE_WAVE")[0]   # (the switch matrix runs this test under them)
    zt = S.stft(ctx.to_device(x), w, ctx=ctx, **opts)[0].numpy()
    assert not native or ctx.last_dispatch().split("+")[0] == "stft.rab", ctx.last_dispatch()   # the native A x B kernel ran (dispatch record)
    ctx.set_tuning("NXSIG_DISABLE_RAB", 1)
    zb = S.stft(ctx.to_device(x), w, ctx=ctx, **opts)[0].numpy()
    assert "stft.rab" not in ctx.last_dispatch().split("+"), ctx.last_dispatch()               # ... and the switch really takes it away
    assert float(np.max(np.abs(zt - zb)) / np.max(np.abs(zb))) < 1e-5


def test_stft_composite_lengths_non_finite_samples_stay_in_their_frames():
    """a NaN / Inf sample reaches exactly the frames that contain it (the reference transforms frame by frame): the solo route of the
    A x B kernels, against the oracle's finite / non-finite pattern"""
    import nx_signal_amd as S
    from oracle import nx_oracle as O

    for K in RAB_LENGTHS:
        hop = K // 4
        x = np.random.default_rng(K + 1).standard_normal(12 * K).astype(np.float32)
        x[5 * K + 7] = np.nan
        x[9 * K + 1] = np.inf
        w = S.windows.hann(K)
        opts = dict(overlap_length=K - hop, fft_length=K, sampling_rate=16000)
        z = S.stft(x, w, **opts)[0]
        zo = O.stft(x, w, **opts)[0]
        bad = ~np.isfinite(zo).all(axis=1)
        assert bad.sum() >= 6 and np.array_equal(~np.isfinite(z).all(axis=1), bad), K
        assert float(np.max(np.abs(z[~bad] - zo[~bad])) / np.max(np.abs(zo[~bad]))) < 1e-5


@pytest.mark.parametrize("taps", [257, 386, 449, 513, 700, 769, 1000, 1025])
@pytest.mark.parametrize("mode", ["same", "full", "valid"])
def test_fir_real_block_kernel(taps, mode):
    """k_fir_r2k (round 5): one real 2048-sample block per 1024-point complex transform, W[k] = A[k] Z[k] + B[k] conj Z[-k].  Every tap
    count it serves by default (>= 386) and 257 under NXSIG_FIR_R2K=2, all modes, several rows, against the direct f64 convolution;
    unaligned rows (odd length / odd stride) fall back to the pair kernels and agree; a NaN poisons exactly its row."""
    import nx_signal_amd as S

    rng = np.random.default_rng(taps)
    L = 60000
    x = rng.standard_normal((3, L)).astype(np.float32)
    h = S.filters.firwin(taps if taps % 2 else taps + 1, [4000.0], sampling_rate=48000)[:taps].copy()
    ctx = S.Context(0)
    ctx.set_tuning("NXSIG_FIR_R2K", 2)
    y = S.filters.fir(ctx.to_device(x), h, mode=mode, ctx=ctx).numpy()
    ref = np.stack([np.convolve(r.astype(np.float64), h.astype(np.float64), mode=mode) for r in x])
    assert y.shape == ref.shape
    assert float(np.max(np.abs(y - ref)) / np.max(np.abs(ref))) < 1e-5, (taps, mode)
    ctx.set_tuning("NXSIG_FIR_R2K", 0)
    y0 = S.filters.fir(ctx.to_device(x), h, mode=mode, ctx=ctx).numpy()
    assert float(np.max(np.abs(y - y0)) / np.max(np.abs(ref))) < 2e-6
    ctx.set_tuning("NXSIG_FIR_R2K", 2)
    xo = np.ascontiguousarray(x[:, : L - 3])                      # 59 997 samples per row: rows not 16-byte aligned
    yo = S.filters.fir(ctx.to_device(xo), h, mode=mode, ctx=ctx).numpy()
    refo = np.stack([np.convolve(r.astype(np.float64), h.astype(np.float64), mode=mode) for r in xo])
    assert float(np.max(np.abs(yo - refo)) / np.max(np.abs(refo))) < 1e-5
    xn = x.copy()
    xn[1, 31000] = np.nan
    yn = S.filters.fir(ctx.to_device(xn), h, mode=mode, ctx=ctx).numpy()
    assert np.isfinite(yn[0]).all() and np.isfinite(yn[2]).all() and not np.isfinite(yn[1]).any()


@pytest.mark.parametrize("K", RAB_LENGTHS)
@pytest.mark.parametrize("hopsel", ["quarter", "half", "full", "eighth", "uneven"])
def test_istft_composite_lengths_native_kernels(K, hopsel):
    """k_istft_rab (round 5): the inverse of the A x B kernels — one frame per lane group, overlap-add through LDS with a carry strip,
    any hop (16-byte gathers for an even one, 8-byte for an odd one).  Non-Hermitian spectra, every scaling, several rows, frame counts that leave the last unit ragged; against the oracle
    (1e-5), bit-identical to itself across run alignments (two batch sizes change the run length) and against the generic path."""
    import nx_signal_amd as S
    from oracle import nx_oracle as O

    hop = {"quarter": K // 4, "half": K // 2, "full": K, "eighth": K // 8, "uneven": K // 4 + 6}[hopsel]
    rng = np.random.default_rng(K + hop)
    M = 57
    w = S.windows.hann(K)
    z = (rng.standard_normal((3, M, K)) + 1j * rng.standard_normal((3, M, K))).astype(np.complex64)
    for scaling in (None, "spectrum", "psd"):
        opts = dict(overlap_length=K - hop, fft_length=K, scaling=scaling, sampling_rate=16000)
        y = S.istft(z, w, **opts)
        yo = O.istft(z, w, **opts)
        assert y.shape == yo.shape and y.dtype == np.complex64
        assert float(np.max(np.abs(y - yo)) / np.max(np.abs(yo))) < 1e-5, (K, hop, scaling)
    ctx = S.Context(0)
    opts = dict(overlap_length=K - hop, fft_length=K, sampling_rate=16000)
    native = not ctx.get_tuning("DISABLE_RAB")[0] and not ctx.get_tuning("DISABLE_WAVE")[0]
    y3 = S.istft(ctx.to_device(z), w, ctx=ctx, **opts).numpy()
    y1 = S.istft(ctx.to_device(z[1:2]), w, ctx=ctx, **opts).numpy()
    assert np.array_equal(y3[1].view(np.uint32), y1[0].view(np.uint32))          # deterministic whatever the launch geometry
    ctx.set_tuning("NXSIG_DISABLE_RAB", 1)
    yg = S.istft(ctx.to_device(z), w, ctx=ctx, **opts).numpy()
    assert not native or not np.array_equal(yg.view(np.uint32), y3.view(np.uint32))   # two different kernels really ran
    assert float(np.max(np.abs(yg - y3)) / np.max(np.abs(y3))) < 1e-5
    # round trip of a real signal on interior samples
    x = rng.standard_normal(40 * hop + K).astype(np.float32)
    ctx.clear_tuning("DISABLE_RAB")
    zz = S.stft(x, w, **opts)[0]
    xr = S.istft(zz, w, **opts)
    if hop <= K // 2:   # a Hann window needs >= 50 % overlap for the normaliser to be well conditioned
        assert float(np.max(np.abs(xr.real[K: -K] - x[K: len(xr) - K]))) < 1e-4


@pytest.mark.parametrize("K", [1152, 1280, 1536, 1600, 1920, 2400, 2880, 3840])
@pytest.mark.parametrize("M,rows", [(57, 3), (9, 1), (700, 2)])
def test_istft_quarter_hop_overlap_add_in_registers(K, M, rows):
    """k_istft_rab_q (round 6): at hop = K / 4 the 50- / 60- / 64-point inverses add the overlapping frames in registers (the frames that
    share a sample sit in one lane).  Same sums in the same order as the LDS form (NXSIG_ISTFT_REGOLA=0): agreement to the last bit or two,
    bit-identical to itself across launch geometries; head and tail rows of the normaliser, ragged last unit, runs with halo (700 frames
    x 2 rows spread over the chip), non-finite bins included."""
    import nx_signal_amd as S
    from oracle import nx_oracle as O

    hop = K // 4
    rng = np.random.default_rng(K + M)
    w = S.windows.hann(K)
    z = (rng.standard_normal((rows, M, K)) + 1j * rng.standard_normal((rows, M, K))).astype(np.complex64)
    z[0, M // 2, 7] = np.nan
    opts = dict(overlap_length=K - hop, fft_length=K, scaling="spectrum", sampling_rate=16000)
    ctx = S.Context(0)
    native = not ctx.get_tuning("DISABLE_RAB")[0] and not ctx.get_tuning("DISABLE_WAVE")[0] and ctx.get_tuning("ISTFT_REGOLA") != (0, True)
    y = S.istft(ctx.to_device(z), w, ctx=ctx, **opts).numpy()
    assert ctx.last_dispatch().startswith("istft.rab.q") == native
    ctx.set_tuning("NXSIG_ISTFT_REGOLA", 0)
    y0 = S.istft(ctx.to_device(z), w, ctx=ctx, **opts).numpy()
    assert not ctx.last_dispatch().startswith("istft.rab.q")
    assert np.array_equal(np.isfinite(y), np.isfinite(y0))
    fin = np.isfinite(y0)
    # same sums in the same order; the two kernels compile the codelets on their own (fma contraction follows the surrounding code),
    # so a sample may differ in its last bit: 2400 / 2880 / 3840 / 1280 / 1536 happen to agree bit for bit, 1152 / 1600 / 1920 do not
    rms = float(np.median(np.abs(y0[fin])))          # (the first and last samples are huge: a tiny normaliser)
    excess = np.abs(y[fin] - y0[fin]) / (np.abs(y0[fin]) + rms)
    assert float(excess.max()) < 2e-6, (K, M, float(excess.max()), rms)
    y1 = S.istft(ctx.to_device(z[:1]), w, ctx=ctx, **{**opts}).numpy() if rows > 1 else None
    ctx.clear_tuning("ISTFT_REGOLA")
    if rows > 1:   # deterministic whatever the launch geometry (run lengths change with the row count)
        yq1 = S.istft(ctx.to_device(z[:1]), w, ctx=ctx, **opts).numpy()
        assert np.array_equal(yq1[0].view(np.uint32), y[0].view(np.uint32))
        assert np.array_equal(y1[0].view(np.uint32), y0[0].view(np.uint32))
    if M <= 60:
        yo = O.istft(z, w, **opts)
        assert np.array_equal(np.isfinite(y), np.isfinite(yo))
        ok = np.isfinite(yo)
        assert float(np.max(np.abs(y[ok] - yo[ok])) / np.max(np.abs(yo[ok]))) < 1e-5


def test_istft_composite_lengths_non_finite_bins_stay_in_their_frames():
    import nx_signal_amd as S
    from oracle import nx_oracle as O

    for K in RAB_LENGTHS:
        hop = K // 4   # (odd for 100 / 300 / 500 / 900: the 8-byte gathers)
        rng = np.random.default_rng(K + 3)
        z = (rng.standard_normal((2, 40, K)) + 1j * rng.standard_normal((2, 40, K))).astype(np.complex64)
        z[0, 17, 5] = np.nan
        z[1, 30, K - 1] = np.inf
        w = S.windows.hann(K)
        opts = dict(overlap_length=K - hop, fft_length=K, sampling_rate=16000)
        y = S.istft(z, w, **opts)
        yo = O.istft(z, w, **opts)
        assert np.array_equal(np.isfinite(y), np.isfinite(yo)), K
        ok = np.isfinite(yo)
        assert float(np.max(np.abs(y[ok] - yo[ok])) / np.max(np.abs(yo[ok]))) < 1e-5


@pytest.mark.parametrize("K,N,hop,pad", [(320, 320, 160, "reflect"), (480, 400, 160, "valid"), (640, 640, 160, "reflect"), (960, 960, 240, "valid"),
                                         (100, 100, 50, "valid"), (240, 200, 80, "reflect"), (500, 500, 125, "valid"), (768, 768, 192, "reflect"),
                                         (1000, 800, 250, "valid"), (1200, 1200, 300, "valid"), (1600, 1600, 400, "reflect")])
def test_composite_lengths_fused_sinks(K, N, hop, pad):
    """the log-mel, magnitude / power / dBFS and one-sided sinks of kernels_wave_rab.hip (round 5): bit-identical to the same sinks on the
    Bluestein kernels' INPUT (same oracle), i.e. against the oracle to the tolerances the other front-ends' sink tests use"""
    import nx_signal_amd as S
    from oracle import nx_oracle as O

    rng = np.random.default_rng(K + N)
    x = rng.standard_normal((2, 30 * hop + N)).astype(np.float32)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, sampling_rate=16000)
    zo = np.stack([O.stft(r, w, **opts)[0] for r in x])
    mel = S.mel_spectrogram(x, w, mel_bins=80, **opts)
    # reduce_max is over the WHOLE tensor (both rows): the oracle on the stacked spectrogram
    melo = O.stft_to_mel(zo.reshape(-1, K), 16000, K, 80).reshape(2, -1, 80)
    assert mel.shape == melo.shape and float(np.max(np.abs(mel - melo))) < 1e-4
    for kind in ("magnitude", "power"):
        g, _, _ = S.spectrogram(x, w, kind=kind, **opts)
        ref = np.abs(zo[..., : K // 2]) if kind == "magnitude" else np.abs(zo[..., : K // 2].astype(np.complex128)) ** 2
        assert g.shape == ref.shape and float(np.max(np.abs(g - ref)) / np.max(np.abs(ref))) < 1e-5, kind
    z1, _, _ = S.stft_onesided(S.default_context(0).to_device(x), w, **opts)
    zf, _, _ = S.stft(x, w, **opts)
    assert np.array_equal(z1.numpy().view(np.uint32), np.ascontiguousarray(zf[..., : K // 2]).view(np.uint32))   # the same bits as stft's first half


@pytest.mark.parametrize("K,hop", [(128, 40), (128, 100), (256, 80), (256, 37), (512, 160), (512, 200), (512, 441), (1024, 160), (1024, 300), (1024, 441)])
def test_istft_power_of_two_frames_any_hop(K, hop):
    """power-of-two frame lengths whose hop the N / hop in {1, 2, 4, 8} kernels do not take (e.g. 512-sample frames every 160 samples,
    lib/nx_signal.ex:609-637) run on the two-pass A x B inverses (kernels_wave_rab_p4.hip) instead of the generic path: against the
    oracle, every scaling, non-Hermitian spectra; deterministic across launch geometries; agrees with the generic path"""
    import nx_signal_amd as S
    from oracle import nx_oracle as O

    rng = np.random.default_rng(K * 7 + hop)
    M = 61
    w = S.windows.hann(K)
    z = (rng.standard_normal((3, M, K)) + 1j * rng.standard_normal((3, M, K))).astype(np.complex64)
    for scaling in (None, "spectrum", "psd"):
        opts = dict(overlap_length=K - hop, fft_length=K, scaling=scaling, sampling_rate=16000)
        y = S.istft(z, w, **opts)
        yo = O.istft(z, w, **opts)
        assert y.shape == yo.shape and y.dtype == np.complex64
        assert float(np.max(np.abs(y - yo)) / np.max(np.abs(yo))) < 1e-5, (K, hop, scaling)
    ctx = S.Context(0)
    opts = dict(overlap_length=K - hop, fft_length=K, sampling_rate=16000)
    native = not ctx.get_tuning("DISABLE_RAB")[0] and not ctx.get_tuning("DISABLE_WAVE")[0]
    y3 = S.istft(ctx.to_device(z), w, ctx=ctx, **opts).numpy()
    y1 = S.istft(ctx.to_device(z[1:2]), w, ctx=ctx, **opts).numpy()
    assert np.array_equal(y3[1].view(np.uint32), y1[0].view(np.uint32))
    ctx.set_tuning("NXSIG_DISABLE_RAB", 1)
    yg = S.istft(ctx.to_device(z), w, ctx=ctx, **opts).numpy()
    assert not native or not np.array_equal(yg.view(np.uint32), y3.view(np.uint32))   # two different kernels really ran
    assert float(np.max(np.abs(yg - y3)) / np.max(np.abs(y3))) < 1e-5
    zn = z.copy()
    zn[0, 20, 3] = np.nan
    yn = S.istft(zn, w, **opts)
    assert np.array_equal(np.isfinite(yn), np.isfinite(O.istft(zn, w, **opts)))


@pytest.mark.parametrize("taps", [1026, 1500, 2049, 4097, 5000, 9001])
@pytest.mark.parametrize("mode", ["same", "full", "valid"])
def test_fir_long_filters_partitioned(taps, mode):
    """filters of more than 1 025 taps (round 5): partitions of <= 1 025 taps through the tuned overlap-save kernels, summed with their
    delays (api.cpp launch_fir_partitioned) — against the direct f64 convolution (Convolution.convolve method: :fft,
    lib/nx_signal/convolution.ex:252-329, to fp32 rounding), several rows, every mode; a NaN poisons exactly its row; the path behind
    NXSIG_DISABLE_WAVE (8192-point workgroup kernel / one transform per row) agrees"""
    import nx_signal_amd as S

    rng = np.random.default_rng(taps)
    L = 40000
    x = rng.standard_normal((3, L)).astype(np.float32)
    h = (rng.standard_normal(taps) * np.hanning(taps)).astype(np.float32)
    ctx = S.Context(0)
    y = S.filters.fir(ctx.to_device(x), h, mode=mode, ctx=ctx).numpy()
    ref = np.stack([np.convolve(r.astype(np.float64), h.astype(np.float64), mode=mode) for r in x])
    assert y.shape == ref.shape
    assert float(np.max(np.abs(y - ref)) / np.max(np.abs(ref))) < 1e-5, (taps, mode)
    yh = S.filters.fir(x, h, mode=mode)                          # host tensors through the same path
    assert np.array_equal(yh.view(np.uint32), y.view(np.uint32))
    ctx.set_tuning("NXSIG_DISABLE_WAVE", 1)
    y0 = S.filters.fir(ctx.to_device(x), h, mode=mode, ctx=ctx).numpy()
    assert float(np.max(np.abs(y - y0)) / np.max(np.abs(ref))) < 1e-5
    ctx.clear_tuning("DISABLE_WAVE")
    xn = x.copy()
    xn[1, 17000] = np.inf
    yn = S.filters.fir(ctx.to_device(xn), h, mode=mode, ctx=ctx).numpy()
    assert np.isfinite(yn[0]).all() and np.isfinite(yn[2]).all() and not np.isfinite(yn[1]).any()
    xs = x[:, :700]                                              # rows shorter than the filter
    ys = S.filters.fir(xs, h, mode="full")
    refs = np.stack([np.convolve(r.astype(np.float64), h.astype(np.float64), mode="full") for r in xs])
    assert float(np.max(np.abs(ys - refs)) / np.max(np.abs(refs))) < 1e-5


@pytest.mark.parametrize("K,N,hop", [(512, 512, 128), (512, 400, 160), (256, 256, 64), (128, 128, 32), (512, 512, 200), (256, 256, 256)])
@pytest.mark.parametrize("L", [30001, 30002, 30003])
def test_stft_quad_front_ends_rows_that_do_not_start_on_16_bytes(K, N, hop, L):
    """rows of a length that is not a multiple of 4 (every row but the first starts off a 16-byte boundary) and a device tensor sliced at an
    odd offset: the staged quad input (round 5) starts its 16-byte loads up to 3 floats early and skips them when it reads the parked span.
    Spectrum, log-mel and magnitude sinks against the oracle; a NaN in the last sample of a row (right before the next row's first
    span) stays in its own frames."""
    import nx_signal_amd as S
    from oracle import nx_oracle as O

    rng = np.random.default_rng(K + hop + L)
    x = rng.standard_normal((5, L)).astype(np.float32)
    x[1, L - 1] = np.nan
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=K, sampling_rate=16000)
    ctx = S.Context(0)
    z = S.stft(ctx.to_device(x), w, ctx=ctx, **opts)[0].numpy()
    zo = O.stft(x, w, **opts)[0]
    ok = np.isfinite(zo)
    assert np.array_equal(np.isfinite(z), ok)
    assert float(np.max(np.abs(z[ok] - zo[ok])) / np.max(np.abs(zo[ok]))) < 1e-5
    x[1, L - 1] = 0.25
    zo = O.stft(x, w, **opts)[0]
    flat = ctx.to_device(np.concatenate([np.zeros(1, np.float32), x.reshape(-1)]))     # the same rows, one float into the allocation
    lib = S._lib.load()
    import ctypes as C
    M = zo.shape[1]
    zd = ctx.empty((5, M, K), np.complex64)
    p = S._lib.StftParams(N, hop, K, S._lib.PAD_VALID, 0, 0, S._lib.SCALE_NONE, 0, 16000.0)
    S._lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(flat.ptr + 4), L, 5, L, w.ctypes.data_as(C.c_void_p), C.byref(p), C.c_void_p(zd.ptr), None, S._lib.DEVICE))
    z2 = zd.numpy()
    assert float(np.max(np.abs(z2 - zo)) / np.max(np.abs(zo))) < 1e-5
    mel = S.mel_spectrogram(x, w, mel_bins=40, **opts)
    melo = O.stft_to_mel(zo.reshape(-1, K), 16000, K, 40).reshape(5, -1, 40)
    assert float(np.max(np.abs(mel - melo))) < 1e-4
    g, _, _ = S.spectrogram(x, w, kind="magnitude", **opts)
    assert float(np.max(np.abs(g - np.abs(zo[..., : K // 2]))) / np.max(np.abs(zo))) < 1e-5


@pytest.mark.parametrize("taps", [33, 101, 257, 513, 1025])
@pytest.mark.parametrize("mode", ["same", "full", "valid"])
def test_fir_rows_that_do_not_start_on_a_cache_line(taps, mode):
    """more than 32 rows of a length that is not a multiple of 32 samples: every row starts at a different offset inside a 128-byte line
    and gets its own grid phase (round 5, FirWaveArgs::row_mod: one common phase left the other rows' streaming stores on partial lines,
    0.26 instead of 0.50 of the roofline).  Against the direct f64 convolution; rows padded apart (batch_stride > length) through the C
    ABI; a NaN poisons exactly its row; NXSIG_FIR_PHASE=0 (no phase at all) agrees."""
    import ctypes as C
    import nx_signal_amd as S

    rng = np.random.default_rng(taps + len(mode))
    rows, L = 37, 9003
    x = rng.standard_normal((rows, L)).astype(np.float32)
    h = (rng.standard_normal(taps) / taps ** 0.5).astype(np.float32)
    ref = np.stack([np.convolve(r.astype(np.float64), h.astype(np.float64), mode=mode) for r in x])
    ctx = S.Context(0)
    y = S.filters.fir(ctx.to_device(x), h, mode=mode, ctx=ctx).numpy()
    assert y.shape == ref.shape and float(np.max(np.abs(y - ref)) / np.max(np.abs(ref))) < 1e-5, (taps, mode)
    ctx.set_tuning("NXSIG_FIR_PHASE", 0)
    y0 = S.filters.fir(ctx.to_device(x), h, mode=mode, ctx=ctx).numpy()
    ctx.clear_tuning("FIR_PHASE")
    assert float(np.max(np.abs(y - y0)) / np.max(np.abs(ref))) < 2e-6
    xn = x.copy()
    xn[20, 4000] = np.nan
    yn = S.filters.fir(ctx.to_device(xn), h, mode=mode, ctx=ctx).numpy()
    bad = ~np.isfinite(yn).all(axis=1)
    assert bad[20] and bad.sum() == 1 and not np.isfinite(yn[20]).any()
    # rows 9 007 floats apart
    stride = L + 4
    xs = np.zeros((rows, stride), np.float32)
    xs[:, :L] = x
    xd = ctx.to_device(xs)
    yd = ctx.empty(ref.shape, np.float32)
    lib = S._lib.load()
    cm = {"full": S._lib.CONV_FULL, "same": S._lib.CONV_SAME, "valid": S._lib.CONV_VALID}[mode]
    S._lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(xd.ptr), L, rows, stride, h.ctypes.data_as(C.c_void_p), taps, cm, C.c_void_p(yd.ptr), S._lib.DEVICE))
    assert float(np.max(np.abs(yd.numpy() - ref)) / np.max(np.abs(ref))) < 1e-5


@pytest.mark.parametrize("K", RAB_LENGTHS + [128, 256, 512, 1024])
def test_stft_of_complex_samples_on_the_two_pass_kernels(K):
    """k_stft_rab_c64 (round 5): c64 signals, one complex frame per transform, every composite length and the power-of-two lengths to 1024
    — against the oracle for N == K, a shorter frame, :reflect padding, every scaling, a ragged frame count; a NaN stays in its frames;
    NXSIG_DISABLE_RAB (two-step path / framed row kernels) agrees"""
    import nx_signal_amd as S
    from oracle import nx_oracle as O

    rng = np.random.default_rng(K + 5)
    hop = K // 4
    for N, pad, L in ((K, "valid", 9 * K + 7), (K - K // 4, "reflect", 5 * K + 3), (K, "reflect", 6 * K)):
        x = (rng.standard_normal((3, L)) + 1j * rng.standard_normal((3, L))).astype(np.complex64)
        w = S.windows.hamming(N)
        for scaling in (None, "spectrum", "psd"):
            opts = dict(overlap_length=N - hop, fft_length=K, window_padding=pad, scaling=scaling, sampling_rate=16000)
            z = S.stft(x, w, **opts)[0]
            zo = O.stft(x, w, **opts)[0]
            assert z.shape == zo.shape and float(np.max(np.abs(z - zo)) / np.max(np.abs(zo))) < 1e-5, (K, N, pad, scaling)
    ctx = S.Context(0)
    opts = dict(overlap_length=K - hop, fft_length=K, sampling_rate=16000)
    w = S.windows.hann(K)
    x = (rng.standard_normal((2, 8 * K)) + 1j * rng.standard_normal((2, 8 * K))).astype(np.complex64)
    x[1, 3 * K + 5] = np.nan
    z = S.stft(ctx.to_device(x), w, ctx=ctx, **opts)[0].numpy()
    zo = O.stft(x, w, **opts)[0]
    assert np.array_equal(np.isfinite(z).all(axis=-1), np.isfinite(zo).all(axis=-1))
    ctx.set_tuning("NXSIG_DISABLE_RAB", 1)
    zb = S.stft(ctx.to_device(x), w, ctx=ctx, **opts)[0].numpy()
    ok = np.isfinite(zo)
    assert float(np.max(np.abs(z[ok] - zb[ok])) / np.max(np.abs(zo[ok]))) < 1e-5


@pytest.mark.parametrize("taps", [257, 512, 513, 1000])
@pytest.mark.parametrize("off", [1, 2, 3])
def test_fir_of_a_tensor_sliced_at_an_odd_sample(taps, off):
    """x starts `off` floats into its allocation: the leading-zero-tap shift of the 2048-point blocks takes the offset into account, the
    1024-point blocks fall to the 4-byte kernel; results against the direct f64 convolution"""
    import ctypes as C
    import nx_signal_amd as S

    rng = np.random.default_rng(taps + off)
    rows, L = 3, 50000
    x = rng.standard_normal((rows, L)).astype(np.float32)
    h = (rng.standard_normal(taps) / taps ** 0.5).astype(np.float32)
    ref = np.stack([np.convolve(r.astype(np.float64), h.astype(np.float64), mode="same") for r in x])
    ctx = S.Context(0)
    flat = ctx.to_device(np.concatenate([np.zeros(off, np.float32), x.reshape(-1)]))
    yd = ctx.empty((rows, L), np.float32)
    lib = S._lib.load()
    S._lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(flat.ptr + 4 * off), L, rows, L, h.ctypes.data_as(C.c_void_p), taps, S._lib.CONV_SAME, C.c_void_p(yd.ptr), S._lib.DEVICE))
    assert float(np.max(np.abs(yd.numpy() - ref)) / np.max(np.abs(ref))) < 1e-5


def test_more_rows_than_one_launch_takes():
    """stft / istft / fir of more than 65 535 rows run as slabs of 65 504 rows (api.cpp launch_stft / launch_istft / launch_fir): every
    row equals the same row computed alone"""
    import nx_signal_amd as S

    rng = np.random.default_rng(77)
    rows, L, N, hop = 70000, 700, 256, 64
    x = rng.standard_normal((rows, L)).astype(np.float32)
    w = S.windows.hann(N)
    opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=16000)
    ctx = S.Context(0)
    xd = ctx.to_device(x)
    z = S.stft(xd, w, ctx=ctx, **opts)[0]
    zh = z.numpy()
    pick = [0, 1, 65503, 65504, 65505, 65535, 65536, rows - 1]
    zs = S.stft(ctx.to_device(np.ascontiguousarray(x[pick])), w, ctx=ctx, **opts)[0].numpy()
    assert np.array_equal(zh[pick].view(np.uint32), zs.view(np.uint32))
    y = S.istft(z, w, ctx=ctx, **opts).numpy()
    ys = S.istft(ctx.to_device(np.ascontiguousarray(zh[pick])), w, ctx=ctx, **opts).numpy()
    assert np.array_equal(y[pick].view(np.uint32), ys.view(np.uint32))
    h = S.filters.firwin(65, [0.2])
    f = S.filters.fir(xd, h, mode="same", ctx=ctx).numpy()
    fs = S.filters.fir(ctx.to_device(np.ascontiguousarray(x[pick])), h, mode="same", ctx=ctx).numpy()
    assert float(np.max(np.abs(f[pick] - fs))) < 1e-6
    # complex samples (nxsig_stft_c64, round 6: the same slabs; the limit used to be 65 535 rows): a native A x B length and a two-step one
    for Kc in (64, 48):
        xc = (x[:, :300] + 1j * x[:, 300:600]).astype(np.complex64)
        wc = S.windows.hann(Kc)
        oc = dict(overlap_length=Kc // 2, fft_length=Kc, sampling_rate=16000)
        zc = S.stft(ctx.to_device(xc), wc, ctx=ctx, **oc)[0].numpy()
        zcs = S.stft(ctx.to_device(np.ascontiguousarray(xc[pick])), wc, ctx=ctx, **oc)[0].numpy()
        assert np.array_equal(zc[pick].view(np.uint32), zcs.view(np.uint32)), Kc


def test_host_tensor_paths_return_the_same_bytes():
    """NXSIG_HOST calls move their tensors by chunked pageable copies with pre-faulting (default) or through pinned bounce slots with the
    DMA of one chunk beside the host copy of the previous one (NXSIG_HOST_PIPE=1, round 6; api.cpp: Staged): the same bytes either way,
    and the same as the device-resident call, for a transfer that spans several 32 MB chunks and ends inside one"""
    ctx = S.Context(0)
    rng = np.random.default_rng(77)
    x = rng.standard_normal((3, 1_234_567)).astype(np.float32)
    w = S.windows.hann(1024)
    opts = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
    z0 = S.stft(x, w, ctx=ctx, **opts)[0]
    zd = S.stft(ctx.to_device(x), w, ctx=ctx, **opts)[0].numpy()
    assert np.array_equal(z0.view(np.uint32), zd.view(np.uint32))
    for knob in (1, 3):
        ctx.set_tuning("HOST_PIPE", knob)
        z1 = S.stft(x, w, ctx=ctx, **opts)[0]
        y1 = S.filters.fir(x, np.ones(33, np.float32) / 33, mode="same", ctx=ctx)
        ctx.clear_tuning("HOST_PIPE")
        assert z1.nbytes > 96 << 20 and z1.nbytes % (32 << 20) != 0
        assert np.array_equal(z1.view(np.uint32), z0.view(np.uint32))
        assert np.array_equal(y1.view(np.uint32), S.filters.fir(x, np.ones(33, np.float32) / 33, mode="same", ctx=ctx).view(np.uint32))


def test_fir_delay_line_on_many_rows_stays_within_its_scratch():
    """kernels_wave_firlong.hip: a segment of the delay line holds at least 64 blocks of every row; with thousands of rows the rows go in
    groups so that the scratch stays within 1 GB (round 6).  1 200 rows x 2 049 taps against the direct f64 convolution of sampled rows;
    a NaN poisons exactly its row, also in the second group"""
    rng = np.random.default_rng(5)
    rows, L, taps = 1200, 9000, 2049
    x = rng.standard_normal((rows, L)).astype(np.float32)
    x[1100, 4000] = np.nan
    h = (rng.standard_normal(taps) * np.hanning(taps) / 30).astype(np.float32)
    ctx = S.Context(0)
    y = S.filters.fir(ctx.to_device(x), h, mode="same", ctx=ctx).numpy()
    native = not ctx.get_tuning("DISABLE_WAVE")[0] and ctx.get_tuning("FIR_DLINE") != (0, True)   # (the switch matrix runs this test under them)
    assert not native or ctx.last_dispatch().startswith("fir.dline"), ctx.last_dispatch()
    for r in (0, 1, 599, 1023, 1024, 1099, 1101, rows - 1):
        ref = np.convolve(x[r].astype(np.float64), h.astype(np.float64), mode="same")
        assert float(np.max(np.abs(y[r] - ref)) / np.max(np.abs(ref))) < 1e-5, r
    assert not np.isfinite(y[1100]).any() and np.isfinite(np.delete(y, 1100, axis=0)).all()
