#!/bin/bash
# rocprofv3 evidence for BASELINE configs 3 / 4 / 5 at their FULL per-GPU shard (run through gpurun):
#   istft N=1024 hop=256, 16 x 60 s  |  stft N=2048 hop=512, 8 ch x 600 s  |  fir 257 taps, 8 ch x 600 s
# per case: --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in their own passes (never with sys/runtime traces).
# Output under gpurun_out/prof2_<tag>/<case>/ plus one summary text per case (copy those into profiles/).
set -u
TAG=${1:-r02}
CASES=${2:-"istft stft2048full fir"}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for CASE in $CASES; do
  OUT=gpurun_out/prof2_$TAG/$CASE
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/bench_configs.py $CASE > $OUT/bench_trace.jsonl 2> $OUT/trace.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python tools/bench_configs.py $CASE > $OUT/bench_fetch.jsonl 2> $OUT/fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python tools/bench_configs.py $CASE > $OUT/bench_write.jsonl 2> $OUT/write.err
  python tools/bench_configs.py $CASE > $OUT/bench_unprofiled.jsonl 2> /dev/null
  { python tools/summarize_prof.py $OUT; echo; echo "## bench_configs.py $CASE (HIP events): under --kernel-trace / unprofiled"; cat $OUT/bench_trace.jsonl $OUT/bench_unprofiled.jsonl; } > $OUT/summary.txt 2>&1
  # keep the merge small: per-dispatch CSVs are large
  find $OUT -name "*kernel_trace.csv" -size +8M -delete
done
cat gpurun_out/prof2_$TAG/*/summary.txt
