#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known-byte access shapes (scripts/ubench/fetchcal.hip)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fetchcal; mkdir -p $O
timeout 300 ./scripts/ubench/fetchcal > $O/expect.csv 2> $O/err.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o run -- ./scripts/ubench/fetchcal > /dev/null 2>> $O/err.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o run -- ./scripts/ubench/fetchcal > /dev/null 2>> $O/err.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- ./scripts/ubench/fetchcal > /dev/null 2>> $O/err.log
cat $O/expect.csv; python3 - <<'PY'
import csv
for sub in ("fetch","write"):
    try:
        for r in csv.DictReader(open("gpurun_out/fetchcal/%s/run_counter_collection.csv" % sub)):
            print(sub, r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"])
    except Exception as e: print(sub, e)
PY
