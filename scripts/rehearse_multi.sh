#!/bin/bash
# Rehearse bench.py's N>1 path on a 1-GPU box: two ranks share GPU 0.  RCCL refuses two ranks on
# one device, so the collective runs over gloo here; everything else (frame sharding, barriers,
# max-over-ranks timing, the label gather call) is the code the 8-GPU run uses.
mkdir -p gpurun_out
export LCCRF_BENCH_DEVICE=0 LCCRF_BENCH_BACKEND=${1:-gloo} HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/rehearse_multi.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/rehearse_multi.log
