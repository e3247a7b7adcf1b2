"""Summarises rocprofv3 CSV output (kernel stats + PMC counters) into a small text report (tools only)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pattern):
    return sorted(glob.glob(os.path.join(root, sub, "**", pattern), recursive=True))


print(f"# rocprofv3 summary for {root}")
for f in find("trace", "*kernel_stats.csv"):
    print(f"\n## kernel stats ({os.path.relpath(f, root)})")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        name = r.get("Name", "")[:70]
        print(f"  {name:70s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} min_ns={r.get('MinNs')} max_ns={r.get('MaxNs')} pct={r.get('Percentage')}")
for f in find("trace", "*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    by = defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"\n## kernel trace durations ({os.path.relpath(f, root)})")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:8]:
        v2 = sorted(v)
        print(f"  {k[:70]:70s} n={len(v)} mean_us={sum(v)/len(v)/1e3:.2f} median_us={v2[len(v2)//2]/1e3:.2f} min_us={v2[0]/1e3:.2f}")
        if rows:
            r0 = next(r for r in rows if r["Kernel_Name"] == k)
            print(f"      VGPR={r0.get('VGPR_Count')} SGPR={r0.get('SGPR_Count')} LDS={r0.get('LDS_Block_Size')} grid={r0.get('Grid_Size')} wg={r0.get('Workgroup_Size')}")
for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(sub, "*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        by = defaultdict(list)
        for r in rows:
            if r.get("Counter_Name") == cname:
                by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        print(f"\n## {cname} per dispatch ({os.path.relpath(f, root)}) [rocprofv3 unit: KiB... see MI355X_MICROARCH.md]")
        for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:6]:
            v2 = sorted(v)
            print(f"  {k[:70]:70s} n={len(v)} median={v2[len(v2)//2]:.1f} max={v2[-1]:.1f}")
