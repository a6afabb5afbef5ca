#!/bin/bash
# Collect the round's committed evidence on the GPU box:  scripts/collect_round_profiles.sh r2
#   <round>_fused_c2 : default bench command (C2) -- kernel stats, FETCH/WRITE PMC passes, SQ counters; covers k_fused and k_frame
#   <round>_stream_c5: C5 with 8 frames in flight -- kernel stats + PMC passes
R=${1:-r2}
bash scripts/profile.sh ${R}_fused_c2 > /dev/null 2>&1
bash scripts/pmc_sq.sh > gpurun_out/prof/${R}_fused_c2/sq_counters.txt 2>&1
python3 scripts/summarize_pmc.py gpurun_out/prof/${R}_fused_c2 > gpurun_out/prof/${R}_fused_c2/pmc_summary.csv
bash scripts/profile.sh ${R}_stream_c5 --workload c5 --frames 8 > /dev/null 2>&1
python3 scripts/summarize_pmc.py gpurun_out/prof/${R}_stream_c5 > gpurun_out/prof/${R}_stream_c5/pmc_summary.csv
(timeout 500 python bench.py) > gpurun_out/prof/${R}_bench_default.json 2> gpurun_out/prof/${R}_bench_default.err
for t in ${R}_fused_c2 ${R}_stream_c5; do echo "== $t"; head -8 gpurun_out/prof/$t/stats/run_kernel_stats.csv | cut -c1-150; cat gpurun_out/prof/$t/pmc_summary.csv | cut -c1-200; done
grep -v "^[EW]2026" gpurun_out/prof/${R}_fused_c2/sq_counters.txt
tail -c 400 gpurun_out/prof/${R}_bench_default.json
