#!/bin/bash
# round 4: two-hop neighbour table for the pair-fused blur passes of a single frame: parity + A/B
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_boundary.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for E in "LCCRF_NO_PAIR_FUSE=1" "LCCRF_NO_2HOP_TABLE=1" "X=1"; do
  env $E timeout 300 python bench.py --workload c5 --frames 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-24s F=1 us/iter %.2f frac %.3f  build_ms %.3f match %s dQ %s tiles %s' % ('[$E]', 1e6/d['value'], r['frac'], d['build_ms_per_batch'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference'], d['tiles_identical']))"
done; done
TAG=${TAG:-r4h} FRAMES="1" bash scripts/gpu_c5_small_f.sh 2>&1 | grep -E "^F=|k_blur|k_splat2|k_slice2|k_neighbors"
