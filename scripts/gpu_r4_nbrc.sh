#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 400 python scripts/stress_locality.py 240 2>&1 | tail -3
FRAMES=2 bash scripts/gpu_c5_env_ab.sh "" "LCCRF_NO_COMPACT_NBR=1"
FRAMES=12 bash scripts/gpu_c5_env_ab.sh "" "LCCRF_BLUR_NT=0"
