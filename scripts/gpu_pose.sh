#!/bin/bash
# pose-optimisation kernel: parity tests, latency / throughput, and (instrumented build) the phase clocks of frame 0
python -m pytest tests/test_pose_optimization.py -m gpu -x -q 2>&1 | tail -3
python tests/perf/latency_pose.py 2>&1 | grep "F="
make -C lc-crf-slam_amd -j8 INSTRUMENT=1 >/dev/null || exit 1
LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so LCCRF_POSE_PROF=1 python tests/perf/latency_pose.py 2>&1 | grep "pose prof" | head -2
