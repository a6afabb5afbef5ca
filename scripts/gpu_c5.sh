#!/bin/bash
# C5 (100k points, 6-D kernel) on the streaming engine: throughput vs frames in flight
for F in ${FRAMES:-1 2 4 8 16}; do
timeout 300 python bench.py --workload c5 --frames $F --steps 5 --warmup 2 --no-cpu-baseline --no-extras ${CHECK:---no-check} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('F=$F iters/s %.5g  us/iter/frame %.2f  frac %.3f  build_ms %.3f match %s' % (d['value'], 1e6/d['value'], r['frac'], d['build_ms_per_batch'], d['label_match_vs_cpu_reference']))"
done
