#!/bin/bash
# A/B by COMPILE-TIME switch on one GPU box: one library per flag set built on the box ("" = the sources as they are), the same
# bench lines with each, interleaved twice.
#   scripts/gpu_ab_build.sh "" "-DLCCRF_FUSE_XP=0"                         C2
#   WORKLOAD="c2 c1" scripts/gpu_ab_build.sh "" "-DLCCRF_X=1"
#   WORKLOAD=c5 FRAMES=8 scripts/gpu_ab_build.sh "" "-DLCCRF_NT_NBR=0"      (prints build_ms too)
#   LATENCY=1 SIZES="500 2000" scripts/gpu_ab_build.sh "" "-DLCCRF_X=1"     single-frame latency from C++ (tools/latency_cpp.cpp) instead
#   PREBUILT="liblccrf_hip.so liblccrf_hip_ab.so" scripts/gpu_ab_build.sh    two libraries built beforehand (e.g. the previous commit)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
libs=()
if [ -n "$PREBUILT" ]; then for l in $PREBUILT; do libs+=("$PWD/lc-crf-slam_amd/$l"); done; set -- $PREBUILT
else
  i=0
  for FL in "$@"; do
    i=$((i+1)); mkdir -p /tmp/ab$i
    make -C lc-crf-slam_amd -j8 EXTRA="$FL" BUILD=build_ab$i LIB=/tmp/ab$i/liblccrf_hip.so >/dev/null 2>&1 || { echo "build failed: $FL"; exit 1; }
    libs+=("/tmp/ab$i/liblccrf_hip.so")
  done
fi
for rep in 1 2; do
  i=0
  for FL in "$@"; do
    lib=${libs[$i]}; i=$((i+1))
    if [ -n "$LATENCY" ]; then echo "== [$FL]"; bash scripts/gpu_latency_cpp.sh LD_LIBRARY_PATH=$(dirname $lib); continue; fi
    for w in ${WORKLOAD:-c2}; do
      extra=""; [ $w = c5 ] && extra="--frames ${FRAMES:-8} --steps 3 --warmup 1"
      LCCRF_LIB=$lib timeout 300 python bench.py --workload $w $extra --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e2e=d.get('end_to_end',{}).get('one_launch_ms_per_batch')
print('%-56s %-4s iters/s %.5g launch_ms %.4f build_ms %.3f | one-launch %s | match %s dQ %s' % ('[$FL]', '$w', d['value'], d['roofline']['launch_ms'], d['build_ms_per_batch'], ('%.4f' % e2e) if e2e else '-', d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"
    done
  done
done
