#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
FRAMES=8 bash scripts/gpu_c5_env_ab.sh "" "LCCRF_BLUR_UNROLL=2" "LCCRF_BLUR_UNROLL=4"
FRAMES=16 bash scripts/gpu_c5_env_ab.sh "" "LCCRF_BLUR_UNROLL=2" "LCCRF_BLUR_UNROLL=4"
