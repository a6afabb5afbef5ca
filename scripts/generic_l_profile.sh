#!/bin/bash
# kernel trace of the generic (L-label) streaming kernels on large frames:  scripts/generic_l_profile.sh
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gl -o gl -- python3 scripts/generic_l_timing.py > gpurun_out/gl/out.txt 2>&1
tail -10 gpurun_out/gl/out.txt
python3 - <<'PY'
import csv,glob
for p in glob.glob("gpurun_out/gl/**/gl_kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(p)))
    for r in rows[:10]: print("%-64s calls %6s avg_us %9.2f pct %5s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
