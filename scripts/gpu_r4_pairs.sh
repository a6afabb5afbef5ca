#!/bin/bash
# round 4: pair-fused blur passes (one or two frames in flight) A/B + parity
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_boundary.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for F in ${FRAMES:-1 2 4}; do
for E in "LCCRF_NO_PAIR_FUSE=1" "LCCRF_PAIR_FUSE_MAX=4"; do
  env $E timeout 300 python bench.py --workload c5 --frames $F --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-24s F=$F us/iter/frame %.2f  us/iter %.2f frac %.3f  build_ms %.3f match %s dQ %s tiles %s' % ('[$E]', 1e6/d['value'], $F*1e6/d['value'], r['frac'], d['build_ms_per_batch'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference'], d['tiles_identical']))"
done; done; done
