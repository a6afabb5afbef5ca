#!/bin/bash
# A/B by COMPILE-TIME switch with the memory-side counters beside the timing: per flag set ("" = the sources as they are) one library
# built on the box, the bench line (twice, interleaved), then FETCH_SIZE / WRITE_SIZE per launch of the inference kernel.
#   scripts/gpu_ab_fetch.sh "" "-DLCCRF_LEAN_NT=15"           WORKLOAD=c2 (default), KERNEL=k_fused (substring of the kernel name)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
libs=(); i=0
for FL in "$@"; do
  i=$((i+1)); mkdir -p /tmp/ab$i
  make -C lc-crf-slam_amd -j8 EXTRA="$FL" BUILD=build_ab$i LIB=/tmp/ab$i/liblccrf_hip.so >/dev/null 2>&1 || { echo "build failed: $FL"; exit 1; }
  libs+=("/tmp/ab$i/liblccrf_hip.so")
done
W=${WORKLOAD:-c2}
for rep in 1 2; do i=0; for FL in "$@"; do lib=${libs[$i]}; i=$((i+1))
  LCCRF_LIB=$lib timeout 300 python bench.py --workload $W --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s %-3s iters/s %.5g launch_ms %.4f | match %s dQ %s' % ('[$FL]', '$W', d['value'], d['roofline']['launch_ms'], d['label_match_vs_cpu_reference'], d['max_abs_dQ_vs_cpu_reference']))"
done; done
i=0
for FL in "$@"; do lib=${libs[$i]}; i=$((i+1)); export LCCRF_LIB=$lib
  for C in FETCH_SIZE WRITE_SIZE; do
    O=gpurun_out/abfetch/$i/$C; rm -rf $O; mkdir -p $O
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o run -- python3 bench.py --workload $W --steps 2 --warmup 1 --lite --no-cpu-baseline --no-check --no-extras > /dev/null 2> $O/err.log
    python3 - "$O" "$FL" "$C" "${KERNEL:-k_fused}" <<'PY'
import csv, glob, sys
o, fl, c, kern = sys.argv[1:5]
v = []
for fn in glob.glob(o + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == c and "ELi1EEEv" not in r["Kernel_Name"]:
            v.append(float(r["Counter_Value"]))
if v:
    m = sum(v) / len(v) * 1024.0
    print("[%s] %s per launch: %.3f GB as reported%s (n=%d)" % (fl, c, m / 1e9, (", x2 = %.3f GB" % (2 * m / 1e9)) if c == "FETCH_SIZE" else "", len(v)))
PY
  done
done
