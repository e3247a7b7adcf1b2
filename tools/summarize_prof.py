"""Summarises rocprofv3 CSV output (kernel trace + PMC counters) into a small text report (tools only).
Dispatches are grouped by (kernel, grid size) so batched and single-stream launches are not mixed."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pattern):
    return sorted(glob.glob(os.path.join(root, sub, "**", pattern), recursive=True))


def short(k):
    return k.replace("void nxsig::", "").replace("(nxsig::", "(")[:64]


print(f"# rocprofv3 summary for {root}")
for f in find("trace", "*kernel_stats.csv"):
    print(f"\n## rocprofv3 --stats ({os.path.relpath(f, root)})")
    for r in list(csv.DictReader(open(f)))[:8]:
        print(f"  {short(r.get('Name','')):64s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} min_ns={r.get('MinNs')} max_ns={r.get('MaxNs')} pct={r.get('Percentage')}")
for f in find("trace", "*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    by = defaultdict(list)
    meta = {}
    for r in rows:
        key = (r["Kernel_Name"], r.get("Grid_Size_X"), r.get("Workgroup_Size_X"))
        by[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[key] = r
    print(f"\n## kernel durations by launch shape ({os.path.relpath(f, root)})")
    for key, v in sorted(by.items(), key=lambda kv: ('nxsig::' not in kv[0][0], -sum(kv[1])))[:24]:   # product kernels first, then the models / yardsticks
        v2 = sorted(v)
        r0 = meta[key]
        print(f"  {short(key[0]):64s} grid={key[1]} wg={key[2]} n={len(v)} mean_us={sum(v)/len(v)/1e3:.2f} median_us={v2[len(v2)//2]/1e3:.2f} min_us={v2[0]/1e3:.2f}"
              f" | VGPR={r0.get('VGPR_Count')} accVGPR={r0.get('Accum_VGPR_Count')} SGPR={r0.get('SGPR_Count')} LDS={r0.get('LDS_Block_Size')} scratch={r0.get('Scratch_Size')}")
        if 12 <= len(v) <= 80 and not any(t in key[0] for t in ("edge_fix", "nf_fix", "poison")):
            # bench.py's secondary blocks (configs 3 / 4 / 5): warm-up launches first, the TIMED window is the last 20 (istft) / 10 launches
            for nwin in (20, 10):
                if len(v) > nwin:
                    t = v[-nwin:]
                    print(f"      last {nwin} launches: mean_us={sum(t)/len(t)/1e3:.2f} median_us={sorted(t)[len(t)//2]/1e3:.2f} min_us={min(t)/1e3:.2f} max_us={max(t)/1e3:.2f}")
        if len(v) > 40:  # bench.py: pre-conditioning + warm-up launches come first, the TIMED window is the last --steps launches
            t = v[-20:]
            print(f"      last 20 launches (bench.py's timed window at --steps 20): mean_us={sum(t)/len(t)/1e3:.2f} min_us={min(t)/1e3:.2f} max_us={max(t)/1e3:.2f}"
                  f" | first 12 launches (us): {' '.join(str(round(x/1e3)) for x in v[:12])}")
for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(sub, "*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        by = defaultdict(list)
        for r in rows:
            if r.get("Counter_Name") == cname:
                by[(r["Kernel_Name"], r.get("Grid_Size"))].append(float(r["Counter_Value"]))
        print(f"\n## {cname} per dispatch, raw counter (KiB) by launch shape ({os.path.relpath(f, root)})")
        for key, v in sorted(by.items(), key=lambda kv: ('nxsig::' not in kv[0][0], -sum(kv[1])))[:24]:
            v2 = sorted(v)
            print(f"  {short(key[0]):64s} grid={key[1]} n={len(v)} median={v2[len(v2)//2]:.1f} KiB  min={v2[0]:.1f} max={v2[-1]:.1f}")
