#!/bin/bash
# round 4: the last blur pass inside the slice for 2 / 4 / 8 frames in flight (one pass per launch otherwise): A/B
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for F in 2 4 8; do
for E in "LCCRF_SLICE_BLUR_MAX=1" "LCCRF_SLICE_BLUR_MAX=8"; do
  env $E timeout 300 python bench.py --workload c5 --frames $F --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-26s F=$F us/iter/frame %.2f frac %.3f match %s dQ %s tiles %s' % ('[$E]', 1e6/d['value'], r['frac'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference'], d['tiles_identical']))"
done; done; done
