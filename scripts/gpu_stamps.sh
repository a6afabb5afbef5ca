#!/bin/bash
# phase stamps of the fused kernel under the timing-experiment knobs
for d in 0 1 2 3; do
echo "dbg=$d"; LCCRF_FUSED_DBG=$d LCCRF_FUSED_TIMING=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep "fused timing" | tail -1
done
