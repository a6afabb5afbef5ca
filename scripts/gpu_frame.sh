#!/bin/bash
# quick loop for the one-launch-per-frame kernel on the GPU box: parity tests, phase stamps (instrumented build), bench line
OUT=gpurun_out/${1:-frame}; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_frame_engine.py -m gpu -x -q) > $OUT/frame.log 2>&1; tail -4 $OUT/frame.log
make -C lc-crf-slam_amd -j8 INSTRUMENT=1 >/dev/null || exit 1
for w in ${WORKLOADS:-c2}; do
LCCRF_LIB=$PWD/lc-crf-slam_amd/liblccrf_hip_instr.so LCCRF_FRAME_TIMING=1 timeout 200 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras 2>&1 | grep "frame timing" | tail -1
(timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extras) > $OUT/bench_$w.json 2> $OUT/bench_$w.err
python - $OUT/bench_$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s iters/s %.4g launch_ms %.4f frac %.3f | e2e one-launch %.4f ms (engine %s) two-kernel %.4f ms -> %.4g frames/s | match %s dQ %s" % (
    d["config"]["workload"][:2], d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["end_to_end"]["one_launch_ms_per_batch"],
    d["end_to_end"]["one_launch_engine"], d["end_to_end"]["two_kernel_ms_per_batch"], d["frames_per_s_end_to_end"],
    d["label_match_vs_cpu_reference"], d["max_abs_dQ_vs_cpu_reference"]))
PY
done
