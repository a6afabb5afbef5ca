#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 python scripts/stress_object_api.py 2>&1 | tail -2
timeout 400 python scripts/stress_locality.py 150 2>&1 | tail -2
bash scripts/gpu_cycle.sh
