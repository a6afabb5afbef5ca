#!/bin/bash
# single-frame latency (tools/latency_cpp.cpp) for builds of the library with different compile-time switches, interleaved twice
#   scripts/gpu_latency_ab_build.sh "-DLCCRF_X=1" "-DLCCRF_DUAL_SAME_XCD=1"        SIZES="500 2000"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
i=0
for FL in "$@"; do
  i=$((i+1)); mkdir -p /tmp/ab$i
  make -C lc-crf-slam_amd -j8 EXTRA="$FL" BUILD=build_ab$i LIB=/tmp/ab$i/liblccrf_hip.so >/dev/null 2>&1 || { echo "build failed: $FL"; exit 1; }
done
for rep in 1 2; do
i=0
for FL in "$@"; do
  i=$((i+1)); echo "== $FL"
  bash scripts/gpu_latency_cpp.sh LD_LIBRARY_PATH=/tmp/ab$i
done; done
