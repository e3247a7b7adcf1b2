"""The C leg of the oracle (recursive radix-2 in double, the timed CPU baseline) agrees with the numpy
oracle (pinned on the reference's vectors): both are f64-internal and round to the same f32 values."""
import numpy as np

from oracle import bb_baseline, nx_oracle as O


def test_c_baseline_matches_numpy_oracle():
    for L, N, hop, K in [(48000, 1024, 256, 1024), (5000, 100, 25, 100), (4000, 64, 16, 128), (3000, 48, 12, 36), (20000, 2048, 512, 2048)]:
        x = O.synth_signal(L, seed=3)
        w = O.hann(N)
        z = bb_baseline.stft(x, w, hop, K, threads=2)
        zo, _, _ = O.stft(x, w, overlap_length=N - hop, fft_length=K)
        assert z.shape == zo.shape
        d = np.abs(z.astype(np.complex128) - zo.astype(np.complex128)).max() / np.abs(zo).max()
        assert d < 2e-7, (L, N, hop, K, d)
        same = np.mean(z.view(np.uint32) == zo.view(np.uint32))
        assert same > 0.98, same  # different f64 FFT factorizations round to the same f32 almost everywhere


def test_c_baseline_doctest_stft():
    z = bb_baseline.stft(np.arange(4, dtype=np.float32), np.ones(2, np.float32), 1, 2)
    assert z.tolist() == [[1, -1], [3, -1], [5, -1]]  # lib/nx_signal.ex:46-55
