#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
for F in 8 4 1; do FRAMES=$F bash scripts/gpu_c5_env_ab.sh "" | sed "s/^/F=$F /"; done
mkdir -p gpurun_out/nbrc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nbrc -o r8 -- python3 bench.py --workload c5 --frames 8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for p in glob.glob("gpurun_out/nbrc/**/r8_kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(p)))
    for r in rows[:5]: print("%-60s calls %6s avg_us %9.2f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
    for r in rows:
        if "k_row_first" in r["Name"]: print("k_row_first", float(r["AverageNs"])/1e3)
PY
