timeout 800 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 300 python scripts/latency.py 2>&1 | grep hip
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('iters/s %.4g build_ms %.3f e2e frames/s %.4g match %s dQ %s' % (d['value'], d['build_ms_per_batch'], d['frames_per_s_end_to_end'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"
