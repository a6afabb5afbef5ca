#!/bin/bash
# round-3 baseline on one box: gpu tests, the default bench line, a C5 kernel trace
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -o run -- python3 bench.py --workload c5 --frames 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-check > $O/c5.json 2> $O/c5.err
head -30 $O/c5_stats/*/run_kernel_stats.csv 2>/dev/null || find $O/c5_stats | head
