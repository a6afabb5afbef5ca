#!/bin/bash
# fine phase stamps of k_frame for several lanes (instrumented build)
make -C lc-crf-slam_amd -j8 INSTRUMENT=1 >/dev/null || exit 1
export LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so
for lane in ${LANES:-0 64 512 960}; do
echo "lane $lane"; LCCRF_FRAME_TIMING_LANE=$lane LCCRF_FRAME_TIMING=${BLOCK:-1} timeout 200 python bench.py --workload ${WORKLOAD:-c2} --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras 2>&1 | grep "frame timing" | tail -1
done
