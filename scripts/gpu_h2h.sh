#!/bin/bash
# tools/host_pipeline.cpp by hand on the GPU box (bench.py runs it for end_to_end.host_to_host):
#   scripts/gpu_h2h.sh                 the three modes at B = 4096 and 256
#   scripts/gpu_h2h.sh matrix          handles x staging threads
#   scripts/gpu_h2h.sh trace [mode]    rocprofv3 copy + kernel timeline of one mode at B = 4096: per-copy durations, link busy fraction
g++ -std=c++14 -O2 -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/host_pipeline.cpp -o /tmp/host_pipeline -L$PWD/lc-crf-slam_amd -l:liblccrf_hip.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/lc-crf-slam_amd -Wl,-rpath,/opt/rocm/lib || exit 1
python - <<'PY'
import importlib, numpy as np
wl = importlib.import_module("lc-crf-slam_amd.workloads")
pbs = [wl.slam_problem(2000, 1 + i) for i in range(64)]
with open("/tmp/frames.bin", "wb") as f:
    f.write(np.array([64, 2000, 5], np.int32).tobytes() + np.array([10.0, 30.0, 0.7], np.float32).tobytes())
    for pb in pbs:
        f.write(np.ascontiguousarray(pb["kernels"][0][0], np.float32).tobytes())
        f.write(np.ascontiguousarray(pb["kernels"][1][0], np.float32).tobytes())
        f.write(np.ascontiguousarray(pb["label"], np.int16).tobytes())
PY
case "${1:-modes}" in
modes) for B in 4096 256; do for mode in serial pageable pinned; do /tmp/host_pipeline /tmp/frames.bin $B $((400000 / B)) 4 $mode - 16; done; done ;;
matrix) for B in 4096 256; do for mode in pageable pinned; do for h in 3 4 6 8; do for th in 8 16 32; do
  [ $mode = pinned ] && [ $th != 8 ] && continue
  echo -n "B=$B $mode handles=$h threads=$th: "; /tmp/host_pipeline /tmp/frames.bin $B $((300000 / B)) $h $mode - $th | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g frames/s  %.1f GB/s' % (d['frames_per_s'], d['upload_GBs']))"
done; done; done; done ;;
trace) cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/h2htrace; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/h2htrace -o run -- /tmp/host_pipeline /tmp/frames.bin 4096 40 4 ${2:-pinned} - 16 2>/tmp/h2h_err.txt
  python3 - <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/h2htrace/**/*memory_copy_trace.csv", recursive=True)[0])))
by = collections.defaultdict(list)
for r in rows:
    by[r["Direction"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in by.items():
    v.sort(); big = [x for x in v if x[1] - x[0] > 200000]
    if big:
        d = sorted(x[1] - x[0] for x in big)
        print(k, "copies", len(v), "long", len(big), "median ms %.3f" % (d[len(d)//2] * 1e-6), "span s %.4f" % ((big[-1][1] - big[0][0]) * 1e-9), "sum of durations s %.4f" % (sum(d) * 1e-9))
ks = [r for r in csv.DictReader(open(glob.glob("/tmp/h2htrace/**/*kernel_trace.csv", recursive=True)[0])) if "k_frame" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in ks)
print("k_frame launches", len(ks), "median ms %.3f" % d[len(d)//2])
PY
  ;;
esac
