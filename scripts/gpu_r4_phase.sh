#!/bin/bash
# round 4: what a phase boundary costs (ubench/phasecost) + XCD-chunked grids below 8 frames, A/B on one box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG:-r4b}; mkdir -p $O
timeout 120 scripts/ubench/phasecost 2>&1 | tee $O/phasecost.txt
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "locality or c5 or xcd or caller_stream" 2>&1 | tail -3
FR="${FRAMES:-1 2 4}"
for rep in 1 2; do
for F in $FR; do
for E in "LCCRF_NO_XCD_CHUNK=1" "X=1"; do
  env $E timeout 300 python bench.py --workload c5 --frames $F --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-24s F=$F us/iter/frame %.2f  us/iter %.2f frac %.3f  build_ms %.3f match %s dQ %s' % ('[$E]', 1e6/d['value'], $F*1e6/d['value'], r['frac'], d['build_ms_per_batch'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"
done; done; done
