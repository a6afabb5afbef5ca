#!/bin/bash
# the round-end cycle on one box: every -m gpu test, smoke(), the default bench line exactly as the driver runs it
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bench
[ -n "$SKIP_TESTS" ] || timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json gpurun_out/bench/bench_full.json > gpurun_out/bench/bench_default.out 2> gpurun_out/bench/bench_default.err; echo "bench rc=$?"
tail -n 1 gpurun_out/bench/bench_default.out > gpurun_out/bench/bench_line.json
python3 - <<PY
import json
last=open("gpurun_out/bench/bench_default.out").read().strip().splitlines()[-1]
print("compact line: %d bytes; stdout tail(8192) starts inside it: %s" % (len(last), len(last) > 8192))
print(last)
d=json.load(open("gpurun_out/bench/bench_full.json"))
r=d["roofline"]
print("C2 value %.4g ms/step %.3f frac %.3f chain_floor_ms %.3f lds_floor_ms %.3f launch_ms %.3f match %s frames_checked %s tiles %s" % (d["value"], d["ms_per_step"], r["frac"], r["chain_floor_ms"], r["lds_floor_ms"], r["launch_ms"], d["label_match_vs_cpu_reference"], d["frames_checked"], d["tiles_identical"]))
print("e2e frames/s %.4g" % d["frames_per_s_end_to_end"])
for k in ("c1","c3","c4","n500"):
    s=d[k]; print(k, "value %.4g frac %.3f e2e %.4g match %s checked %s tiles %s" % (s["value"], s["roofline"]["frac"], s["end_to_end"]["frames_per_s"], s["label_match_vs_cpu_reference"], s["frames_checked"], s["tiles_identical"]))
c=d["c5"]; print("c5 x8 value %.4g us/it/frame %.2f whole frac %.3f blur frac %.3f match %s tiles %s" % (c["value"], c["us_per_iteration_per_frame"], c["roofline_whole_iteration"]["frac"], c["roofline"]["frac"], c["label_match_vs_cpu_reference"], c["tiles_identical"]))
s=c["single_frame"]; print("c5 x1 value %.4g us/it %.2f frac events %.3f wall %.3f match %s" % (s["value"], s["us_per_iteration_per_frame"], s["roofline_whole_iteration"]["frac"], s["frac_wall"], s["label_match_vs_cpu_reference"]))
l=d["single_frame_latency_us"]; print("latency hip %.1f p90 %.1f ctypes %.1f cpu %.1f" % (l["hip"], l["hip_p90"], l["hip_ctypes"], l["cpu_reference"]))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
