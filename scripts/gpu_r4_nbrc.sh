#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/nbrc
for F in 8 1; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nbrc -o q$F -- python3 bench.py --workload c5 --frames $F --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<PY
import csv,glob
for p in glob.glob("gpurun_out/nbrc/**/q${F}_kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(p)))
    for r in rows[:12]: print("%-60s calls %6s avg_us %9.2f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
