#!/bin/bash
# Where one frame's k_frame launch spends its time: shader-clock stamps of lane 0 for single frames through the object API
# (instrumented build; N from $SIZES).  Stamps: start | both lattice builds | normalisation pass (3 stamps: P, S, blur) +
# slice | per iteration: P barrier, S, blur, X | store.
cd "$GRAFT_REPO_ROOT"
make -C lc-crf-slam_amd -j8 INSTRUMENT=1 >/dev/null || exit 1
export LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so
for N in ${SIZES:-2000 500}; do
LCCRF_FRAME_TIMING=${TIMING_BLOCK:-1} python3 - "$N" 2>&1 <<'PY' | grep "frame timing" | tail -2
import importlib, sys, os
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
N = int(sys.argv[1])
for rep in range(6):
    pb = wl.slam_problem(N, 1 + rep % 2)
    c = pkg.DenseCRFHIP(N, 2); c.set_unary_from_label(pb["label"], pb["conf"])
    for f, w in pb["kernels"]: c.add_pairwise(f, w)
    c.inference(5, True); c.map(); c.close()
PY
done
