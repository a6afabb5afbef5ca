#!/bin/bash
# phase stamps of the fused kernel under the timing-experiment knobs (instrumented build only:
# the release library has no stamps and reads no LCCRF_FUSED_DBG)
make -C lc-crf-slam_amd -j4 INSTRUMENT=1 >/dev/null || exit 1
export LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so
for d in ${DBGS:-0 1 2 3}; do
echo "dbg=$d"; LCCRF_FUSED_DBG=$d LCCRF_FUSED_TIMING=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras 2>&1 | grep "fused timing" | tail -1
done
for k in 1 2; do
LCCRF_BUILD_TIMING=$k timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-extras 2>&1 | grep "build timing" | tail -1
done
LCCRF_FRAME_TIMING=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras 2>&1 | grep "frame timing" | tail -1
