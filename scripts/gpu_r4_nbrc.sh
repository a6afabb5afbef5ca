#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python scripts/stress_locality.py 300 2>&1 | tail -3
