"""Derives HBM bytes per launch of the secondary bench workloads (configs 3 / 4 / 5) from the rocprofv3 PMC summaries under
profiles/<round>/ and stores them in profiles/traffic.json next to the headline entry (tools only; bench.py reads the file).
FETCH_SIZE reads 0.5000 of the bytes on gfx950, WRITE_SIZE 1.0 (profiles/r02/pmc_calibration.txt); every kernel of the launch
(main + edge / fix-up) is added.  usage: python tools/update_traffic.py r02"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
CASES = {
    "istft": ("istft_rocprofv3_summary.txt", ("k_istft_wave", "k_istft_edge_fix"), 16 * 11247 * 10240),
    "stft2048": ("stft2048full_rocprofv3_summary.txt", ("k_stft_wave",), 8 * 56247 * 18432),
    "fir": ("fir_rocprofv3_summary.txt", ("k_fir_wave",), 8 * 28800000 * 8),
}


def section(text, title):
    m = re.search(r"## " + title + r".*?\n(.*?)(\n## |\Z)", text, re.S)
    return m.group(1) if m else ""


tpath = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tpath))
for key, (fname, kernels, algo) in CASES.items():
    text = open(os.path.join(ROOT, "profiles", rnd, fname)).read()
    total = 0.0
    for title, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        for line in section(text, title).splitlines():
            m = re.match(r"\s*(\S+)<.*n=(\d+) median=([\d.]+) KiB", line) or re.match(r"\s*(?:nxsig::)?(\S+)\(.*n=(\d+) median=([\d.]+) KiB", line)
            if m and any(m.group(1).endswith(k) or k in m.group(1) for k in kernels) and int(m.group(2)) >= 10:
                total += mult * float(m.group(3)) * 1024.0
    t[key + "_bytes_per_launch"] = total
    t[key + "_algorithmic_bytes"] = algo
    t[key + "_ratio"] = total / algo
    print(key, total, algo, round(total / algo, 4))
t["_about_secondary"] = ("istft / stft2048 / fir: the same derivation from profiles/%s/*_rocprofv3_summary.txt (tools/profile_secondary.sh, "
                         "tools/update_traffic.py); launches = config 3 (16 x 60 s), config 4 shard (8 ch x 600 s), config 5 shard (8 ch x 600 s, 257 taps)" % rnd)
json.dump(t, open(tpath, "w"), indent=1)
