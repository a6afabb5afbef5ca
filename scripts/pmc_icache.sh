#!/bin/bash
# Instruction-cache counters of the one-workgroup kernels (is the straight-line code of the frame kernels fetched from L2 every time?), on the GPU box.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof/icache
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/p1 -o run -- python3 bench.py --steps 2 --warmup 1 --lite --no-cpu-baseline --no-check --no-extras "$@" > /dev/null 2> $OUT/p1.err
python3 - <<'PY'
import csv, collections, glob
for fn in glob.glob("gpurun_out/prof/icache/p1/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        for kern in ("k_fused", "k_frame"):
            if kern in r["Kernel_Name"]:
                agg[(kern, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kern, k), v in sorted(agg.items()):
        print("%-9s %-24s mean/dispatch %.4g  (n=%d)" % (kern, k, sum(v)/len(v), len(v)))
PY
grep -v "^[EW]2026" $OUT/p1.err | tail -3
