#!/bin/bash
# Rehearse bench.py's N>1 path on a 1-GPU box: `python bench.py --gpus 2` starts its two ranks itself; both share GPU 0.
# RCCL refuses two ranks on one device, so the collective runs over gloo here; everything else (rank spawning, frame
# sharding, barriers, max-over-ranks timing, the per-step label gather on the kernel-written bit buffer) is the code an
# 8-GPU run uses.
mkdir -p gpurun_out
export LCCRF_BENCH_DEVICE=0 LCCRF_BENCH_BACKEND=${1:-gloo} HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --frames ${FRAMES:-2048} > gpurun_out/rehearse_multi.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/rehearse_multi.log | cut -c1-1500
