#!/bin/bash
# A/B of the streaming build's compile-time switches on C5 (8 frames): prints build_ms per variant, twice
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
i=0
for FL in "$@"; do
  i=$((i+1))
  make -C lc-crf-slam_amd -j8 EXTRA="$FL" BUILD=build_ab$i LIB=liblccrf_hip_ab$i.so >/dev/null 2>&1 || { echo "build failed: $FL"; exit 1; }
done
for rep in 1 2; do
i=0
for FL in "$@"; do
  i=$((i+1))
  LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_ab$i.so timeout 300 python bench.py --workload c5 --frames 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-70s build_ms %.3f  us/iter/frame %.2f' % ('$FL', d['build_ms_per_batch'], 1e6/d['value']))"
done; done
