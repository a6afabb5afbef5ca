#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 400 python scripts/stress_locality.py 200 2>&1 | tail -3
bash scripts/gpu_cycle.sh
