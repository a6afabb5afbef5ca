#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration (run on the GPU box via gpurun).
#   scripts/profile.sh <tag> [bench.py args...]
# Writes gpurun_out/prof/<tag>/{stats,fetch,write}/...  Counters are collected in their own
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never mixed with sys/hip traces).
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT/stats $OUT/fetch $OUT/write
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- \
    python3 bench.py "$@" --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/stats/err.log
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o run -- \
    python3 bench.py "$@" --steps 2 --warmup 1 --lite --no-cpu-baseline --no-check --no-extras > /dev/null 2> $OUT/fetch/err.log
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o run -- \
    python3 bench.py "$@" --steps 2 --warmup 1 --lite --no-cpu-baseline --no-check --no-extras > /dev/null 2> $OUT/write/err.log
find $OUT -type f | head -40
