#!/bin/bash
# SQ/LDS counter passes for one bench_configs.py case (tools only).  usage: pmc_kernel.sh <tag> <case>
set -u
TAG=$1; CASE=$2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- python tools/bench_configs.py $CASE > $OUT/p1.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p2 -- python tools/bench_configs.py $CASE > $OUT/p2.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/p3 -- python tools/bench_configs.py $CASE > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "wave" in k or "generic" in k or "k_" in k:
            agg[(k[:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in agg.items():
    n = max(len(v) for v in cs.values())
    if n < 3: continue
    print(key)
    med = {c: sorted(v)[len(v)//2] for c, v in cs.items()}
    for c, v in sorted(med.items()): print(f"   {c:24s} {v:16.1f}")
    wc = med.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_VMEM","SQ_WAIT_INST_LDS"):
            if c in med: print(f"   {c}/WAVE_CYCLES = {med[c]/wc:.3f}")
PY
