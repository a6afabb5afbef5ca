#!/bin/bash
# C5 A/B on one box: locality mode off (LCCRF_NO_PERM=1) vs on, 8 frames in flight, + a kernel trace of the "on" run
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG:-c5ab}; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$1 iters/s %.5g  us/iter/frame %.2f  frac %.3f  build_ms %.3f match %s dQ %s' % (d['value'], 1e6/d['value'], r['frac'], d['build_ms_per_batch'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"; }
for rep in 1 2; do
LCCRF_NO_PERM=1 timeout 300 python bench.py --workload c5 --frames 8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>$O/off.err | line "perm OFF"
timeout 300 python bench.py --workload c5 --frames 8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>$O/on.err | line "perm ON "
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --workload c5 --frames 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-check > $O/c5.json 2> $O/c5.err
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/stats/run_kernel_stats.csv')))
for r in rows[:22]:
    print("%-58s calls %6s avg_us %9.2f total_ms %8.2f" % (r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
