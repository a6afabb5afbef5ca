#!/bin/bash
# Single-frame latency of the call site as a C++ caller sees it (no Python in the loop): builds
# tests/cpp/call_site_test.cpp against the in-tree libraries and times construct .. getMap .. destroy.
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
g++ -std=c++14 -O2 -Iinclude tests/cpp/call_site_test.cpp -o gpurun_out/call_site_test \
    lc-crf-slam_amd/liblccrf_hip.so oracle/liblccrf_oracle.so \
    -Wl,-rpath,$PWD/lc-crf-slam_amd -Wl,-rpath,$PWD/oracle -Wl,-rpath,/opt/rocm/lib
for N in 500 1000 2000 3000; do
python3 - "$N" <<'PY'
import importlib, sys, numpy as np
sys.path.insert(0, ".")
wl = importlib.import_module("lc-crf-slam_amd.workloads")
N = int(sys.argv[1]); fr = wl.slam_frame(N, 3)
with open("gpurun_out/in_%d.bin" % N, "wb") as f:
    f.write(np.int32(N).tobytes())
    for a in (fr["obs"], fr["err"], fr["uv"], fr["init_label"]):
        f.write(np.ascontiguousarray(a).tobytes())
PY
gpurun_out/call_site_test gpurun_out/in_$N.bin 200 | tail -1
done
