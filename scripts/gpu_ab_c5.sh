#!/bin/bash
# A/B of a compile-time switch on C5 (frames in flight from $FRAMES):  scripts/gpu_ab_c5.sh "-DLCCRF_NT_NBR=0"
make -C lc-crf-slam_amd -j8 EXTRA="$1" BUILD=build_ab LIB=liblccrf_hip_ab.so >/dev/null || exit 1
for rep in 1 2; do for lib in liblccrf_hip.so liblccrf_hip_ab.so; do
echo -n "$lib "; LCCRF_LIB=$PWD/lc-crf-slam_amd/$lib FRAMES="${FRAMES:-8}" bash scripts/gpu_c5.sh
done; done
