#!/bin/bash
# copy / kernel timeline of tools/host_pipeline.cpp (pinned mode, B = 4096): per-copy durations and rates
bash scripts/gpu_h2h_cpp.sh > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/h2htrace; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/h2htrace -o run -- /tmp/host_pipeline /tmp/frames.bin 4096 40 3 ${1:-pinned} > /tmp/h2h_run.txt 2>/tmp/h2h_err.txt
cat /tmp/h2h_run.txt; tail -3 /tmp/h2h_err.txt; find /tmp/h2htrace -name "*.csv" | head
python3 - <<'PY'
import csv, glob, collections
fn = glob.glob("/tmp/h2htrace/**/*memory_copy_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fn[0])))
print(rows[0].keys())
by = collections.defaultdict(list)
for r in rows:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    by[r["Direction"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), dur))
for k, v in by.items():
    v.sort()
    big = [x for x in v if x[2] > 2e-4]
    print(k, "copies", len(v), "long ones", len(big), "median dur ms", sorted(x[2] for x in big)[len(big)//2] * 1e3 if big else None)
    # overlap: total busy span vs sum
    if big:
        span = (big[-1][1] - big[0][0]) * 1e-9
        print("   span s", span, "sum of durations s", sum(x[2] for x in big))
kf = glob.glob("/tmp/h2htrace/**/*kernel_trace.csv", recursive=True)
ks = [r for r in csv.DictReader(open(kf[0])) if "k_frame" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in ks)
print("k_frame launches", len(ks), "median ms", d[len(d)//2], "max", d[-1])
PY
