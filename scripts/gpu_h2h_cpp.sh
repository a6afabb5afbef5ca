#!/bin/bash
# tools/host_pipeline.cpp by hand on the GPU box:  scripts/gpu_h2h_cpp.sh [taskset cpu list]   (builds, writes inputs, runs the three modes at B = 4096 and 256)
g++ -std=c++14 -O2 -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/host_pipeline.cpp -o /tmp/host_pipeline -L$PWD/lc-crf-slam_amd -l:liblccrf_hip.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/lc-crf-slam_amd -Wl,-rpath,/opt/rocm/lib || exit 1
python - <<'PY'
import importlib, numpy as np
wl = importlib.import_module("lc-crf-slam_amd.workloads")
pbs = [wl.slam_problem(2000, 1 + i) for i in range(64)]
with open("/tmp/frames.bin", "wb") as f:
    f.write(np.array([64, 2000, 5], np.int32).tobytes() + np.array([10.0, 30.0, 0.7], np.float32).tobytes())
    for pb in pbs:
        f.write(np.ascontiguousarray(pb["kernels"][0][0], np.float32).tobytes())
        f.write(np.ascontiguousarray(pb["kernels"][1][0], np.float32).tobytes())
        f.write(np.ascontiguousarray(pb["label"], np.int16).tobytes())
PY
PRE=""
[ -n "$1" ] && PRE="taskset -c $1"
for B in 4096 256; do for mode in serial pageable pinned; do
  nb=$((400000 / B)); h=3; [ $B -lt 1024 ] && h=6
  $PRE /tmp/host_pipeline /tmp/frames.bin $B $nb $h $mode
done; done
