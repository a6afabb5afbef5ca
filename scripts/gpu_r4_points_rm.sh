#!/bin/bash
# round 4: with the sorted build, points in (coarse) row-major order of their cells in the lattice basis vs the Z-order curve (LCCRF_POINTS_ZORDER=1)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for E in "LCCRF_POINTS_ZORDER=1" "X=1"; do
  env $E timeout 300 python bench.py --workload c5 --frames 8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-26s F=8 us/iter/frame %.2f frac %.3f  build_ms %.3f match %s dQ %s tiles %s' % ('[$E]', 1e6/d['value'], r['frac'], d['build_ms_per_batch'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference'], d['tiles_identical']))"
done; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4w/stats -o run -- python3 bench.py --workload c5 --frames 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-check > /dev/null 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r4w/stats/run_kernel_stats.csv')))
for r in rows[:8]:
    print("  %-58s calls %6s avg_us %9.2f" % (r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3))
PY
