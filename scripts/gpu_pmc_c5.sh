#!/bin/bash
# PMC passes over the C5 iteration kernels (8 frames): scripts/gpu_pmc_c5.sh <tag> [env...]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
i=0
for C in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
         "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o run -- python3 bench.py --workload c5 --frames 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-check > /dev/null 2> $O/p$i.err
done
python3 - <<PY
import csv, collections, glob
agg = collections.OrderedDict()
for fn in sorted(glob.glob("$O/p*/run_counter_collection.csv")):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if not any(x in k for x in ("k_blur2", "k_splat2", "k_slice2")): continue
        name = k.split("(")[0].split("::")[-1]
        agg.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in agg.items():
    print("%-14s %-44s n=%5d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
PY
