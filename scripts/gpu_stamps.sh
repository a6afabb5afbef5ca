#!/bin/bash
# Shader-clock phase stamps of the one-workgroup kernels (instrumented library only).
#   scripts/gpu_stamps.sh                       k_fused / k_fused_lean: C2, workgroup $BLOCK (default 1), lanes $LANES, DBG=8 adds the fine stamps
#   KERNEL=frame scripts/gpu_stamps.sh          k_frame (lccrf_batch_run)
#   KERNEL=build scripts/gpu_stamps.sh          k_build_small, kernel 1 and 2
#   KERNEL=object SIZES="2000 500" ...          one frame at a time through the object API (k_frame, two workgroups per frame)
#   any switch of the instrumented library may ride along in the environment (LCCRF_LEAN_SHAPE=0, LCCRF_FUSED_DBG via DBG=...)
cd "$GRAFT_REPO_ROOT" 2>/dev/null
make -C lc-crf-slam_amd -j8 INSTRUMENT=1 EXTRA="$EXTRA" >/dev/null || exit 1
export LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so
B="timeout 200 python bench.py --workload ${WORKLOAD:-c2} --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras"
case "${KERNEL:-fused}" in
fused) for lane in ${LANES:-0 200 960}; do echo "lane $lane"; LCCRF_FUSED_DBG=${DBG:-8} LCCRF_FUSED_TIMING_LANE=$lane LCCRF_FUSED_TIMING=${BLOCK:-1} $B 2>&1 | grep "fused timing" | tail -1; done ;;
frame) for lane in ${LANES:-0 64 512 960}; do echo "lane $lane"; LCCRF_FRAME_TIMING_LANE=$lane LCCRF_FRAME_TIMING=${BLOCK:-1} $B 2>&1 | grep "frame timing" | tail -1; done ;;
build) for k in 1 2; do LCCRF_BUILD_TIMING=$k $B 2>&1 | grep "build timing" | tail -1; done ;;
object) for N in ${SIZES:-2000 500}; do
LCCRF_FRAME_TIMING=${BLOCK:-1} python3 - "$N" 2>&1 <<'PY' | grep "frame timing" | tail -2
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
done ;;
esac
