"""Generates tests/golden/oracle_c1_fixture.npz: a regression fixture of the ORACLE itself (not of the reference, which
cannot run here: no BEAM) on BASELINE config 1 — 1 s mono 48 kHz, N=1024 hop=256 periodic Hann.  It pins the oracle
against silent drift (numpy / code changes); the reference-derived vectors live in reference_vectors.json.

    python tests/golden/make_oracle_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nx_oracle as O  # noqa: E402


def main():
    x = O.synth_signal(48000, seed=1234)
    w = O.hann(1024)
    z, t, f = O.stft(x, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    y = O.istft(z, w, overlap_length=768, fft_length=1024, sampling_rate=48000)
    frames = np.array([0, 1, 91, 183])
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_c1_fixture.npz")
    np.savez_compressed(out, x_head=x[:2048], frames=frames, z_frames=z[frames], z_colsum=z.sum(axis=0).astype(np.complex64),
                        z_abs_rowsum=np.abs(z).sum(axis=1).astype(np.float32), times=t, freqs=f,
                        y_head=y[:1536], y_tail=y[-1536:], shape_z=np.array(z.shape), shape_y=np.array(y.shape))
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
