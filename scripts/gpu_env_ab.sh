#!/bin/bash
# A/B by ENVIRONMENT on one GPU box: bench lines under a list of settings, interleaved twice.  The switches live in the INSTRUMENTED
# library only (csrc/engine.h: ab_env), which this script selects; "" is the default behaviour of the same library.
#   scripts/gpu_env_ab.sh "" "LCCRF_SPLAT_REC=1"                        C5, 8 frames in flight
#   WORKLOAD=c5 FRAMES=1 scripts/gpu_env_ab.sh "" "LCCRF_NO_PAIR_FUSE=1"
#   WORKLOAD="c2 c3" scripts/gpu_env_ab.sh "LCCRF_LEAN_SHAPE=0" ""        the fused engine's shape for full-size SLAM frames
#   WORKLOAD="c1 n500" scripts/gpu_env_ab.sh "" "LCCRF_NO_SMALL_WG=1"     512-lane shapes for small frames
#   TRACE=1 ... adds a rocprofv3 kernel trace of the LAST setting (first workload) under gpurun_out/$TAG/stats
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so
[ -f "$LCCRF_LIB" ] || make -C lc-crf-slam_amd -j8 INSTRUMENT=1 >/dev/null || exit 1
args_of() { if [ "$1" = c5 ]; then echo "--workload c5 --frames ${FRAMES:-8} --steps 5 --warmup 2"; else echo "--workload $1 ${FRAMES:+--frames $FRAMES}"; fi; }
for rep in 1 2; do for w in ${WORKLOAD:-c5}; do for E in "$@"; do
  env $E timeout 300 python bench.py $(args_of $w) --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
e2e=d.get('end_to_end',{}).get('one_launch_ms_per_batch')
print('%-36s %-4s iters/s %.5g  launch_ms %.4f  frac %.3f  build_ms %.3f  one-launch %s | match %s dQ %s' % ('[$E]', '$w', d['value'], r['launch_ms'], r['frac'], d['build_ms_per_batch'], ('%.4f' % e2e) if e2e else '-', d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"
done; done; done
if [ -n "$TRACE" ]; then
  O=gpurun_out/${TAG:-envab}; mkdir -p $O; last="${@: -1}"; w=$(echo ${WORKLOAD:-c5} | cut -d' ' -f1)
  # (the program itself after `--`, never `env`: the profiler's preloaded library has initialised the GPU by then)
  for kv in $last; do export "$kv"; done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py $(args_of $w) --no-cpu-baseline --no-extras --no-check > $O/bench.json 2> $O/err.log
  python3 - <<PY
import csv, glob
for fn in glob.glob('$O/stats/**/run_kernel_stats.csv', recursive=True) + glob.glob('$O/stats/run_kernel_stats.csv'):
    for r in list(csv.DictReader(open(fn)))[:24]:
        print("  %-58s calls %6s avg_us %9.2f total_ms %8.2f" % (r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
    break
PY
fi
