#!/bin/bash
# N RANKED processes on ONE GPU, launched the way the driver launches the multi-GPU bench:
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
# with --share-gpu (test mode: every rank uses device LOCAL_RANK % device_count and claims its own NCCL_HOSTID, because RCCL
# refuses two ranks on one device of one host; the ranks then talk through RCCL's socket transport over loopback).
# No scaling figure can come from this — it shakes out the rendezvous file, the watchdog and the unequal-shard broadcasts at the
# world size the 8-GPU node uses.   usage: tools/n_rank_one_gpu.sh [N=8] [out-dir=gpurun_out/n_rank]
cd "$(dirname "$0")/.."
N=${1:-8}
OUT=${2:-gpurun_out/n_rank}
mkdir -p "$OUT"
export NCCL_SOCKET_IFNAME=lo NCCL_IB_DISABLE=1 NCCL_DEBUG=WARN
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus "$N" --steps 10 --warmup 2 --streams 4 --cpu-seconds 0 --share-gpu ${NXSIG_BENCH_EXTRA:-} > "$OUT/bench_${N}_ranks.json" 2> "$OUT/bench_${N}_ranks.err"
echo "exit $?"
cat "$OUT/bench_${N}_ranks.json"
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^W0\|^\*\*\*" "$OUT/bench_${N}_ranks.err" | tail -8
