#!/bin/bash
# Everything the round's profile evidence is made of, in ONE gpurun call (run on the GPU box):
#   1. tools/profile_bench.sh <tag>        rocprofv3 --kernel-trace --stats of bench.py at the driver's flags (headline + configs 3/4/5 in
#                                          the same process, same pre-conditioning), FETCH_SIZE / WRITE_SIZE in their own passes, calibration
#   2. tools/pmc_kernel.sh <tag>_<case>    SQ / LDS counters of the shipped fir / istft / stft kernels
#   3. rocm-smi clocks + package power     sampled while tools/loop_kernel.py keeps the fir / stft kernel running
# Output under gpurun_out/prof_<tag>/, gpurun_out/pmc_<tag>_*/, gpurun_out/power_<tag>.txt.   usage: tools/profile_round.sh r03
set -u
TAG=${1:-r03}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tools/profile_bench.sh $TAG > gpurun_out/profile_bench_$TAG.log 2>&1
for CASE in fir istft stft1024; do
  tools/pmc_kernel.sh ${TAG}_$CASE $CASE > gpurun_out/pmc_${TAG}_$CASE.txt 2>&1
done
{
  for K in fir stft; do
    python tools/loop_kernel.py $K 9 > gpurun_out/loop_$K.log 2>&1 &
    LP=$!
    sleep 4
    for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 1.2; done
    wait $LP
    echo "## $K: $(cat gpurun_out/loop_$K.log | tail -1)"
  done
} > gpurun_out/power_$TAG.txt 2>&1
tail -3 gpurun_out/profile_bench_$TAG.log
cat gpurun_out/power_$TAG.txt
