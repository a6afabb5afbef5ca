#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
for F in 1 2 3 8; do FRAMES=$F bash scripts/gpu_c5_env_ab.sh "" "LCCRF_SPLAT_PASSES=1" | sed "s/^/F=$F /"; done
