#!/bin/bash
# phase stamps of k_fused for several lanes (instrumented build); DBG=8 adds the fine stamps
make -C lc-crf-slam_amd -j8 INSTRUMENT=1 EXTRA="$EXTRA" >/dev/null || exit 1
export LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so
for lane in ${LANES:-0 200 960}; do
echo "lane $lane"; LCCRF_FUSED_DBG=${DBG:-8} LCCRF_FUSED_TIMING_LANE=$lane LCCRF_FUSED_TIMING=${BLOCK:-1} timeout 200 python bench.py --workload ${WORKLOAD:-c2} --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras 2>&1 | grep "fused timing" | tail -1
done
