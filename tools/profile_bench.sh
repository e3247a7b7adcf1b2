#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun). Output under gpurun_out/prof_<tag>/.
#   pass 1: --kernel-trace --stats              (per-kernel durations)
#   pass 2: --pmc FETCH_SIZE                    (HBM read side; gfx950: x2 for wide coalesced streams, see MI355X_MICROARCH.md)
#   pass 3: --pmc WRITE_SIZE                    (HBM write side)
# PMC passes never combine with sys/runtime traces (gpurun refuses that combination).
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 20 --warmup 5 --cpu-seconds 0 --no-verify"}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
# calibration of the two counters on known byte counts (2 GiB each) in this kernel's access patterns
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- ./tools/pmc_calib > $OUT/calib.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- ./tools/pmc_calib >> $OUT/calib.log 2>&1
python - <<PY >> $OUT/calib_summary.txt
import csv, glob
for sub, c in (("calib_fetch","FETCH_SIZE"),("calib_write","WRITE_SIZE")):
    for f in glob.glob("$OUT/"+sub+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and r["Kernel_Name"].startswith("k_"):
                kib = float(r["Counter_Value"]); print(f"{c:10s} {r['Kernel_Name'][:40]:40s} counter={kib:12.1f} KiB  moved=2097152 KiB  counter/moved={kib/2097152:.4f}")
PY
cat $OUT/calib_summary.txt
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
