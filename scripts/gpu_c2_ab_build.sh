#!/bin/bash
# A/B of compile-time switches of the fused loop on the batched SLAM workloads: one library per flag set, runs interleaved twice.
#   scripts/gpu_c2_ab_build.sh "-DLCCRF_X=1" "-DLCCRF_CHAIN_TOP=4" ...      WORKLOADS="c2 c1" to add configs
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
i=0
for FL in "$@"; do
  i=$((i+1))
  make -C lc-crf-slam_amd -j8 EXTRA="$FL" BUILD=build_ab$i LIB=liblccrf_hip_ab$i.so >/dev/null 2>&1 || { echo "build failed: $FL"; exit 1; }
done
for rep in 1 2; do
for w in ${WORKLOADS:-c2}; do
i=0
for FL in "$@"; do
  i=$((i+1))
  LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_ab$i.so timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s %-4s iters/s %.4g launch_ms %.4f | one-launch %.4f ms | match %s dQ %s' % ('$FL', '$w', d['value'], d['roofline']['launch_ms'], d['end_to_end']['one_launch_ms_per_batch'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"
done; done; done
