#!/bin/bash
# small-frame workgroups (512 lanes, two frames per CU) on/off: C1 and N500 inference throughput, interleaved twice
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for w in ${WORKLOADS:-c1 n500}; do
for E in "LCCRF_X=1" "LCCRF_NO_SMALL_WG=1"; do
  env $E timeout 300 python bench.py --workload $w --frames ${FRAMES:-16384} --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s %-5s iters/s %.4g launch_ms %.4f | one-launch %.4f ms | match %s dQ %s' % ('[$E]', '$w', d['value'], d['roofline']['launch_ms'], d['end_to_end']['one_launch_ms_per_batch'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"
done; done; done
