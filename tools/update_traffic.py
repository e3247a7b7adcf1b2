"""Derives the HBM bytes per launch of the four bench workloads (headline + configs 3 / 4 / 5) from the rocprofv3 PMC summary of
bench.py itself (tools/profile_bench.sh -> profiles/<round>/bench_rocprofv3_summary.txt) and stores them in profiles/traffic.json,
which bench.py quotes as `roofline*.traffic` (labelled `traffic_source`: it is NOT measured in the bench run).
FETCH_SIZE reads 0.5000 of the bytes on gfx950, WRITE_SIZE 1.0 (profiles/<round>/pmc_calibration.txt); every kernel of a launch
(main + edge / fix-up passes) is added.   usage: python tools/update_traffic.py r03"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
text = open(os.path.join(ROOT, "profiles", rnd, "bench_rocprofv3_summary.txt")).read()
# (key, kernels of one launch as (name prefix, grid or None = the shape with the largest grid), algorithmic bytes)
CASES = {
    "stft_batch32": ([("k_stft_wave<1024, 0,", "max")], 32 * 11247 * 9216),
    "istft": ([("k_istft_wave<1024, 4,", "max"), ("nxsig::k_istft_edge_fix", "max")], 16 * 11247 * 10240),
    "stft2048": ([("k_stft_wave<1024, 1,", "max")], 8 * 56247 * 18432),
    "fir": ([("k_fir_wave<1024, true,", "max"), ("k_fir_wave<1024, false,", "max"), ("nxsig::k_fir_poison", "max")], 8 * 28800000 * 8),
}


def rows(counter):
    m = re.search(r"## " + counter + r".*?\n(.*?)(\n## |\Z)", text, re.S)
    out = []
    for line in (m.group(1) if m else "").splitlines():
        mm = re.match(r"\s*(.+?)\s+grid=(\d+) n=(\d+) median=([\d.]+) KiB", line)
        if mm:
            out.append((mm.group(1), int(mm.group(2)), int(mm.group(3)), float(mm.group(4))))
    return out


t = {"_about": f"HBM bytes per launch from the rocprofv3 PMC passes of bench.py itself (profiles/{rnd}/bench_rocprofv3_summary.txt, "
               "tools/profile_bench.sh): FETCH_SIZE x 2 (gfx950: the counter reads 0.5000 of the bytes moved, calibrated in "
               f"profiles/{rnd}/pmc_calibration.txt) + WRITE_SIZE (calibrated 1.0000 for sc1 nt buffer stores, 1.0024 for plain nt "
               "stores), KiB -> bytes, every kernel of a launch added.  Launches: headline 32 x 60 s (N=1024 hop=256); config 3 istft "
               "16 x 60 s; config 4 shard 8 ch x 600 s (N=2048); config 5 shard 8 ch x 600 s, 257 taps."}
for key, (kernels, algo) in CASES.items():
    total = 0.0
    for counter, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        rs = rows(counter)
        for prefix, _ in kernels:
            cand = [r for r in rs if r[0].startswith(prefix) and r[2] >= 10]
            if cand:
                total += mult * max(cand, key=lambda r: r[1])[3] * 1024.0
    t[key + "_bytes_per_launch"] = total
    t[key + "_algorithmic_bytes"] = algo
    t[key + "_ratio"] = total / algo
    print(key, total, algo, round(total / algo, 4))
json.dump(t, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
