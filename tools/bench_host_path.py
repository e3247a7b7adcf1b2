"""PCIe-inclusive rate of the host-buffer (NXSIG_HOST) STFT path: numpy in -> numpy out through the C ABI.
This is the number DESIGN.md quotes beside (never instead of) the device-resident benchmark value.
usage: python tools/bench_host_path.py [rows ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nx_signal_amd as S

def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [1, 8]
    N, hop, fs = 1024, 256, 48000
    w = S.windows.hann(N)
    for rows in rows_list:
        x = np.random.default_rng(0).standard_normal((rows, 60 * fs)).astype(np.float32)
        best = None
        for it in range(4):
            t0 = time.perf_counter()
            z, t, f = S.stft(x, w, overlap_length=N - hop, fft_length=N, sampling_rate=fs)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        frames = z.shape[0] * z.shape[1] if z.ndim == 3 else z.shape[0]
        print(json.dumps({"case": f"host-path stft N=1024 hop=256, {rows} x 60 s", "s": best, "frames_per_s": frames / best,
                          "output_GBps": z.nbytes / best / 1e9}))

if __name__ == "__main__":
    main()
