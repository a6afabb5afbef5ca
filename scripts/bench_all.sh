#!/bin/bash
# All BASELINE.json configurations on one GPU (run via gpurun); one JSON line each.
mkdir -p gpurun_out/bench
for w in c1 c2 c3 c4; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 "$@" > gpurun_out/bench/$w.json 2> gpurun_out/bench/$w.err
done
timeout 600 python bench.py --workload c5 --steps 5 --warmup 1 "$@" > gpurun_out/bench/c5.json 2> gpurun_out/bench/c5.err
python - <<'PY'
import json
for w in ("c1","c2","c3","c4","c5"):
    try:
        d = json.load(open("gpurun_out/bench/%s.json" % w))
        cb = d.get("cpu_baseline", {})
        print(w, "value=%.4g iters/s" % d["value"], "ms/step=%.4g" % d["ms_per_step"], d["config"]["engine"],
              "roof=%.3f" % d["roofline"]["frac"], "match=%s dQ=%s" % (d["label_match_vs_cpu_reference"], d["max_abs_dQ_vs_cpu_reference"]),
              "build_ms=%.3g" % d["build_ms_per_batch"], "cpu=%.4g (%s)" % (cb.get("value", 0), cb.get("kind")))
    except Exception as e:
        print(w, "FAILED", e); print(open("gpurun_out/bench/%s.err" % w).read()[-1500:])
PY
