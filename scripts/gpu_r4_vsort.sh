#!/bin/bash
# round 4: the sorted build of locality mode (entries sorted on the row-major code of their vertex; LCCRF_VERTEX_ORDER=1 / 0 forces it on / off):
# parity + A/B on C5 with 8 / 1 frames + kernel trace
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG:-r4v}; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "locality or c5 or large or single_frame or lazy" 2>&1 | tail -3
for rep in 1 2; do
for F in ${FRAMES:-8 1}; do
for E in "LCCRF_VERTEX_ORDER=0" "LCCRF_VERTEX_ORDER=1"; do
  env $E timeout 300 python bench.py --workload c5 --frames $F --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-24s F=$F us/iter/frame %.2f frac %.3f  build_ms %.3f match %s dQ %s tiles %s' % ('[$E]', 1e6/d['value'], r['frac'], d['build_ms_per_batch'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference'], d['tiles_identical']))"
done; done; done
LCCRF_VERTEX_ORDER=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --workload c5 --frames 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-check > $O/c5.json 2> $O/c5.err
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/stats/run_kernel_stats.csv')))
for r in rows[:26]:
    print("  %-58s calls %6s avg_us %9.2f total_ms %8.2f" % (r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
