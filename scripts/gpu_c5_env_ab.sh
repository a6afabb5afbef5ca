#!/bin/bash
# C5 (8 frames) under a list of environment settings, interleaved twice:  scripts/gpu_c5_env_ab.sh "" "LCCRF_NO_PERM=1" ...
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for E in "$@"; do
  env $E timeout 300 python bench.py --workload c5 --frames ${FRAMES:-8} --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-40s iters/s %.5g  us/iter/frame %.2f  frac %.3f  build_ms %.3f match %s dQ %s' % ('[$E]', d['value'], 1e6/d['value'], r['frac'], d['build_ms_per_batch'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"
done; done
