#!/bin/bash
# A/B of the fused engine's shape for full-size SLAM frames on one GPU box:  scripts/gpu_lean_ab.sh [workloads...]
# LCCRF_LEAN_SHAPE = 0 (one 1024-lane frame per CU), 1 (two frames per CU: 384 lanes, everything in registers), 2 (512 lanes, weights
# re-read per iteration), 3 (384 lanes, weights re-read); interleaved, two rounds.
export LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so   # (the switch exists in the instrumented library only)
for w in ${@:-c2}; do
for rep in 1 2; do
for sh in ${SHAPES:-0 2}; do
LCCRF_LEAN_SHAPE=$sh timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('shape $sh $w iters/s %.4g launch_ms %.4f | match %s maxdQ %s' % (d['value'], d['roofline']['launch_ms'], d['label_match_vs_cpu_reference'], d.get('max_abs_dQ_vs_cpu_reference')))"
done; done; done
