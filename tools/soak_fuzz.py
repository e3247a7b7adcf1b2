"""Soak run of the seeded fuzz families with fresh seeds (not part of the test suite): python tools/soak_fuzz.py [n] [seed0]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_fuzz as F  # noqa: E402
import test_gpu_nd as ND  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    fams = [F.test_fuzz_stft, F.test_fuzz_istft, F.test_fuzz_fir, F.test_fuzz_stft_long_rows_interior_edge_split, F.test_fuzz_fused_sinks, F.test_fuzz_istft_n400,
            F.test_fuzz_fir_any_taps_offsets_and_slices, F.test_fuzz_istft_filtered_and_direct_convolution,
            F.test_fuzz_stft_to_mel_bits, F.test_fuzz_long_rows_and_columns, ND.test_convolve_direct_register_window_kernel_fuzz,
            F.test_fuzz_non_finite_samples_follow_the_reference, F.test_fuzz_istft_non_finite_bins_and_packed_pair, F.test_fuzz_f64_tier]
    only = os.environ.get("SOAK_ONLY")
    if only:
        fams = [f for f in fams if only in f.__name__]
    bad = 0
    t0 = time.time()
    for i in range(n):
        for fn in fams:
            try:
                fn(seed0 + i)
            except Exception:  # noqa: BLE001
                bad += 1
                print(f"FAIL {fn.__name__} seed={seed0 + i}")
                traceback.print_exc(limit=2)
    print(f"soak: {n} seeds x {len(fams)} families, {bad} failures, {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
