#!/bin/bash
# A/B of two PREBUILT libraries (liblccrf_hip.so = working tree, liblccrf_hip_ab.so = e.g. the previous commit,
# built locally with `make BUILD=build_ab LIB=liblccrf_hip_ab.so`), interleaved runs of the same bench lines.
for w in ${@:-c2}; do
for rep in 1 2 3; do
for lib in liblccrf_hip.so liblccrf_hip_ab.so; do
LCCRF_LIB=$PWD/lc-crf-slam_amd/$lib timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $w iters/s %.4g launch_ms %.4f | one-launch %.4f ms | match %s' % (d['value'], d['roofline']['launch_ms'], d['end_to_end']['one_launch_ms_per_batch'], d['label_match_vs_cpu_reference']))"
done; done; done
