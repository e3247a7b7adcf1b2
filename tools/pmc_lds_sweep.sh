#!/bin/bash
# LDS bank-conflict sweep over bench_configs.py cases (tools only): one --pmc pass per case, prints conflict cycles / LDS cycles
# and LDS wait / wave cycles per kernel; LDS cycles per LDS instruction (round 6: ~5 is healthy, 10+ means unaligned 8- / 16-byte accesses or
# conflicts — how the 4-byte-off wave buffers of the odd composite lengths were found).  usage: pmc_lds_sweep.sh <case> [<case> ...]
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for CASE in "$@"; do
  OUT=gpurun_out/lds_$(echo $CASE | tr ':' '_')
  rm -rf $OUT; mkdir -p $OUT
  timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/p -- python tools/bench_configs.py $CASE > $OUT/p.log 2>&1
  python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    n = max(len(v) for v in cs.values())
    if n < 3: continue
    med = {c: sorted(v)[len(v)//2] for c, v in cs.items()}
    idx = med.get("SQ_LDS_IDX_ACTIVE", 0.0); wc = med.get("SQ_WAVE_CYCLES", 0.0)
    if idx < 1e5: continue
    print(f"$CASE | {k:72s} | lds cycles/instr {idx/max(med.get('SQ_INSTS_LDS',1.0),1.0):5.1f} | conflict/lds {med.get('SQ_LDS_BANK_CONFLICT',0)/idx:5.2f} | lds_wait/wave {med.get('SQ_WAIT_INST_LDS',0)/wc:5.3f} | valu/wave {med.get('SQ_ACTIVE_INST_VALU',0)/wc:5.3f} | launches {n}")
PY
done
