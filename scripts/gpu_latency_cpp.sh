#!/bin/bash
# single-frame latency from a C++ caller (tools/latency_cpp.cpp), N = 500 / 1000 / 2000 / 3000
cd "$GRAFT_REPO_ROOT"
g++ -std=c++14 -O2 -Iinclude tools/latency_cpp.cpp -o /tmp/latency_cpp lc-crf-slam_amd/liblccrf_hip.so -Wl,-rpath,$PWD/lc-crf-slam_amd -Wl,-rpath,/opt/rocm/lib || exit 1
for N in ${SIZES:-500 1000 2000 3000}; do
python3 - "$N" <<'PY'
import importlib, sys, numpy as np
sys.path.insert(0, ".")
wl = importlib.import_module("lc-crf-slam_amd.workloads")
N = int(sys.argv[1])
with open("/tmp/in_%d.bin" % N, "wb") as f:
    f.write(np.int32(8).tobytes() + np.int32(N).tobytes())
    for s in range(1, 9):
        fr = wl.slam_frame(N, s)
        for a, dt in ((fr["obs"], np.float32), (fr["err"], np.float32), (fr["uv"], np.float32), (fr["init_label"], np.int16)):
            f.write(np.ascontiguousarray(a, dt).tobytes())
PY
env "$@" /tmp/latency_cpp /tmp/in_$N.bin 300
done
