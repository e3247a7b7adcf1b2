#!/usr/bin/env python
"""Average duration per (kernel, grid, LDS size) from a rocprofv3 --kernel-trace CSV directory.  usage: kernel_times.py <dir> [name-filter]"""
import collections
import csv
import glob
import sys

flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:70], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("LDS_Block_Size") or r.get("LDS_Block_Size_v"))
            d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v) / len(v) / 1e3:10.1f} us x {len(v):5d}  {k}")
