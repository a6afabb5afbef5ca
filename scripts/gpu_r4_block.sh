#!/bin/bash
# round 4: lanes per workgroup of the iteration kernels below 8 frames in flight (LCCRF_SMALL_F_BLOCK), library built as liblccrf_hip_blk.so
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_blk.so
for rep in 1 2; do
for F in 1 2 4; do
for B in 256 128 64; do
  LCCRF_SMALL_F_BLOCK=$B timeout 300 python bench.py --workload c5 --frames $F --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('block %3d F=$F us/iter/frame %.2f frac %.3f match %s dQ %s tiles %s' % ($B, 1e6/d['value'], r['frac'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference'], d['tiles_identical']))"
done; done; done
