#!/bin/bash
# quick loop check on the GPU box: fused parity tests, phase stamps, C2 bench line
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
# (phase stamps: scripts/gpu_stamps.sh, instrumented build)
for w in ${WORKLOADS:-c2}; do
timeout 200 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w iters/s %.4g ms/step %.4f launch_ms %.4f match %s dQ %s' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"
done
