#!/bin/bash
# L2 (TCC) and memory-side (EA) counters of the inference kernel, run on the GPU box: hit rate, read requests by size, average
# read latency at the L2's memory side (RDREQ_LEVEL / RDREQ), DRAM-credit stalls.
#   scripts/pmc_tcc.sh [bench.py args...]        KERNELS="k_fused k_frame" (substrings)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof/tcc
rm -rf $OUT; mkdir -p $OUT
run() { timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -o run -- python3 bench.py --steps 2 --warmup 1 --lite --no-cpu-baseline --no-check --no-extras "${@:3}" > /dev/null 2> $OUT/$1.err; }
run p1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "$@"
run p2 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum" "$@"
run p3 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "$@"
python3 - <<'PY'
import csv, collections, glob, os
kerns = os.environ.get("KERNELS", "k_fused k_frame").split()
raw = collections.defaultdict(list)
for fn in glob.glob("gpurun_out/prof/tcc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        for k in kerns:
            if k in r["Kernel_Name"]:
                raw[(k, r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
# several instances of a template may match (the self-contained first inference, the once-per-build prepare launch, the steady-state
# kernel): the instance with the most dispatches is reported
best = {}
for (k, name, c), v in raw.items():
    if k not in best or len(v) > best[k][1]:
        best[k] = (name, len(v))
agg = {(k, c): v for (k, name, c), v in raw.items() if best[k][0] == name}
for k, (name, n) in sorted(best.items()):
    print("# %s = %s" % (k, name[:150]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
for (kern, c), v in sorted(m.items()):
    print("%-9s %-40s mean/dispatch %.5g  (n=%d)" % (kern, c, v, len(agg[(kern, c)])))
for kern in kerns:
    g = lambda c: m.get((kern, c))
    if g("TCC_REQ_sum"):
        print("%s: L2 hit rate %.3f" % (kern, g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1)))
    if g("TCC_EA0_RDREQ_sum"):
        rd, r32, bub = g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_32B_sum") or 0, g("TCC_BUBBLE_sum") or 0
        print("%s: EA read bytes (128 B x bubble + 64 B x rest + 32 B x 32B) %.3f GB; to DRAM address space %.3g requests of %.3g"
              % (kern, (bub * 128 + (rd - bub - r32) * 64 + r32 * 32) / 1e9, g("TCC_EA0_RDREQ_DRAM_sum") or 0, rd))
    if g("TCC_EA0_RDREQ_LEVEL_sum") and g("TCC_EA0_RDREQ_sum"):
        print("%s: average EA read latency %.0f L2 clocks; DRAM-credit stall cycles (sum over channels) %.4g" % (kern, g("TCC_EA0_RDREQ_LEVEL_sum") / g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum") or 0))
PY
tail -2 $OUT/p1.err
