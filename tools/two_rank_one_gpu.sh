#!/bin/bash
# two RANKED processes on ONE GPU: RCCL refuses two ranks on one device of one host, so each rank claims its own host id
# (NCCL_HOSTID) and the pair talks through RCCL's socket transport over the loopback interface
cd "$(dirname "$0")/.."
export NCCL_SOCKET_IFNAME=lo NCCL_IB_DISABLE=1 NCCL_DEBUG=WARN MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_RANK=0
mkdir -p gpurun_out/two_rank
RANK=0 NCCL_HOSTID=nxsig-rank0 python bench.py --gpus 2 --steps 10 --warmup 2 --streams 8 --cpu-seconds 0 > gpurun_out/two_rank/rank0.json 2> gpurun_out/two_rank/rank0.err &
P0=$!
RANK=1 NCCL_HOSTID=nxsig-rank1 python bench.py --gpus 2 --steps 10 --warmup 2 --streams 8 --cpu-seconds 0 > gpurun_out/two_rank/rank1.json 2> gpurun_out/two_rank/rank1.err &
P1=$!
wait $P0; echo "rank0 exit $?"; wait $P1; echo "rank1 exit $?"
cat gpurun_out/two_rank/rank0.json
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/two_rank/rank0.err | tail -8
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/two_rank/rank1.err | tail -5
# the sharded log-mel: its clamp needs the maximum over both ranks' shards (ncclAllReduce / ncclMax between the two passes)
export MASTER_PORT=29534
RANK=0 NCCL_HOSTID=nxsig-rank0 python tools/two_rank_mel.py > gpurun_out/two_rank/mel_rank0.json 2> gpurun_out/two_rank/mel_rank0.err &
P0=$!
RANK=1 NCCL_HOSTID=nxsig-rank1 python tools/two_rank_mel.py > gpurun_out/two_rank/mel_rank1.json 2> gpurun_out/two_rank/mel_rank1.err &
P1=$!
wait $P0; echo "mel rank0 exit $?"; wait $P1; echo "mel rank1 exit $?"
cat gpurun_out/two_rank/mel_rank0.json gpurun_out/two_rank/mel_rank1.json
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/two_rank/mel_rank0.err | tail -5
