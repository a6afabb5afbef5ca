#!/bin/bash
# SQ counters of the fused kernel (one pass of <= 8 SQ counters), run on the GPU box.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof/sq
rm -rf $OUT/p1 $OUT/p2; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/p1 -o run -- python3 bench.py --steps 2 --warmup 1 --lite --no-cpu-baseline --no-check --no-extras "$@" > /dev/null 2> $OUT/p1.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES \
  --kernel-trace --output-format csv -d $OUT/p2 -o run -- python3 bench.py --steps 2 --warmup 1 --lite --no-cpu-baseline --no-check --no-extras "$@" > /dev/null 2> $OUT/p2.err
python3 - <<'PY'
import csv, collections, glob
for p in ("p1","p2"):
    for fn in glob.glob("gpurun_out/prof/sq/%s/*counter_collection.csv" % p):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            for kern in ("k_fused", "k_frame", "k_blur2", "k_splat2", "k_slice2"):
                if kern in r["Kernel_Name"]:
                    agg[(kern, r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        # several instances of a template may match (the self-contained first inference, the once-per-build prepare launch, the
        # steady-state kernel): the instance with the most dispatches is reported
        best = {}
        for (kern, name, k), v in agg.items():
            if kern not in best or len(v) > best[kern][1]:
                best[kern] = (name, len(v))
        for (kern, name, k), v in sorted(agg.items()):
            if best[kern][0] == name:
                print("%-9s %-24s mean/dispatch %.4g  (n=%d)" % (kern, k, sum(v)/len(v), len(v)))
        for kern, (name, n) in sorted(best.items()):
            print("# %s = %s" % (kern, name[:150]))
PY
tail -3 $OUT/p1.err
