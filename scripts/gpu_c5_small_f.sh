#!/bin/bash
# C5 with 1 / 2 / 4 frames in flight: a bench line and a kernel trace each (per-kernel averages of the iteration kernels)
#   TAG=... FRAMES="1 2 4" scripts/gpu_c5_small_f.sh [env...]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG:-c5f}; mkdir -p $O
for F in ${FRAMES:-1 2 4}; do
  env "$@" timeout 300 python bench.py --workload c5 --frames $F --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>$O/f$F.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('F=$F iters/s %.5g  us/iter/frame %.2f  us/iter %.2f frac %.3f  build_ms %.3f match %s dQ %s' % (d['value'], 1e6/d['value'], $F*1e6/d['value'], r['frac'], d['build_ms_per_batch'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f$F -o run -- python3 bench.py --workload c5 --frames $F --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-check > $O/c5_f$F.json 2> $O/c5_f$F.err
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/stats_f$F/run_kernel_stats.csv')))
for r in rows[:14]:
    print("  %-58s calls %6s avg_us %9.2f total_ms %8.2f" % (r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
