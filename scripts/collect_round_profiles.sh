#!/bin/bash
# Collect the round's committed evidence on the GPU box:  scripts/collect_round_profiles.sh r4
#   <round>_fused_c2     default bench command (C2) -- kernel stats, FETCH/WRITE PMC passes, SQ counters; covers k_fused and k_frame
#   <round>_stream_c5    C5 with 8 frames in flight -- kernel stats + PMC passes
#   <round>_stream_c5_f1 C5 as BASELINE states it: ONE frame (XCD-chunked grids, two passes per launch) -- kernel stats + PMC passes
#   <round>_small_c1     C1 (1000 keypoints: the 512-lane shapes, two frames per CU) -- kernel stats + PMC passes
#   <round>_fused_c4     C4 (3000 keypoints, 3 points per lane: the lowest LDS fraction) -- kernel stats + PMC passes
#   <round>_stream_c5_hash_build  C5 x 8 with LCCRF_VERTEX_ORDER=0 (= LCCRF_OPT_VERTEX_ORDER 2: round 3's hash build + first-occurrence ids) -- kernel stats
R=${1:-r5}
bash scripts/profile.sh ${R}_fused_c2 > /dev/null 2>&1
bash scripts/pmc_sq.sh > gpurun_out/prof/${R}_fused_c2/sq_counters.txt 2>&1
bash scripts/profile.sh ${R}_stream_c5 --workload c5 --frames 8 > /dev/null 2>&1
bash scripts/profile.sh ${R}_stream_c5_f1 --workload c5 --frames 1 > /dev/null 2>&1
bash scripts/profile.sh ${R}_small_c1 --workload c1 > /dev/null 2>&1
bash scripts/profile.sh ${R}_fused_c4 --workload c4 --frames 8192 > /dev/null 2>&1
mkdir -p gpurun_out/prof/${R}_stream_c5_hash_build/stats
# (the switch lives in the instrumented library: csrc/engine.h ab_env)
(cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; export LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so LCCRF_VERTEX_ORDER=0; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${R}_stream_c5_hash_build/stats -o run -- \
    python3 bench.py --workload c5 --frames 8 --no-cpu-baseline --no-extras > gpurun_out/prof/${R}_stream_c5_hash_build/bench.json 2> gpurun_out/prof/${R}_stream_c5_hash_build/stats/err.log)
for t in ${R}_fused_c2 ${R}_stream_c5 ${R}_stream_c5_f1 ${R}_small_c1 ${R}_fused_c4; do
  python3 scripts/summarize_pmc.py gpurun_out/prof/$t > gpurun_out/prof/$t/pmc_summary.csv
done
(timeout 900 python bench.py) > gpurun_out/prof/${R}_bench_default.json 2> gpurun_out/prof/${R}_bench_default.err
# run-to-run spread of the headline: five back-to-back runs of the timed region alone
for i in 1 2 3 4 5; do timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $i: value %.5g iters/s  ms_per_step %.4f  launch_ms %.4f  frac %.4f  shape %sx%s  match %s dQ %s' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline'].get('lanes_per_frame'), d['roofline'].get('frames_per_cu'), d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"; done > gpurun_out/prof/${R}_fused_c2/bench_repeat.txt
# the one-frame-per-CU shape of the same kernel beside it (instrumented library: LCCRF_LEAN_SHAPE=0), SQ counters of both
LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so LCCRF_LEAN_SHAPE=0 bash scripts/pmc_sq.sh > gpurun_out/prof/${R}_fused_c2/sq_counters_one_frame_per_cu.txt 2>&1
cat gpurun_out/prof/${R}_fused_c2/bench_repeat.txt
for t in ${R}_fused_c2 ${R}_stream_c5 ${R}_stream_c5_f1 ${R}_small_c1 ${R}_fused_c4 ${R}_stream_c5_hash_build; do echo "== $t"; head -8 gpurun_out/prof/$t/stats/run_kernel_stats.csv | cut -c1-150; [ -f gpurun_out/prof/$t/pmc_summary.csv ] && cut -c1-200 gpurun_out/prof/$t/pmc_summary.csv | head -12; done
grep -v "^[EW]2026" gpurun_out/prof/${R}_fused_c2/sq_counters.txt
tail -c 400 gpurun_out/prof/${R}_bench_default.json
