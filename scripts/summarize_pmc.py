#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs to one row per (kernel, counter):
   scripts/summarize_pmc.py gpurun_out/prof/<tag> > profiles/<tag>/pmc_summary.csv
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  Per
MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 64 B per 128-B request of a
wide coalesced stream, i.e. half the bytes: the `bytes_corrected` column doubles it.
WRITE_SIZE needs no correction here (it reproduces the known output size exactly)."""
import collections
import csv
import os
import sys

root = sys.argv[1]
agg = collections.OrderedDict()
for sub in ("fetch", "write"):
    fn = os.path.join(root, sub, "run_counter_collection.csv")
    if not os.path.exists(fn):
        continue
    for r in csv.DictReader(open(fn)):
        if "lccrf" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"], r["Counter_Name"])
        agg.setdefault(key, []).append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_KiB_per_dispatch", "bytes_per_dispatch", "bytes_corrected"])
for (k, c), v in agg.items():
    mean = sum(v) / len(v)
    b = mean * 1024.0
    w.writerow([k, c, len(v), "%.1f" % mean, "%.0f" % b, "%.0f" % (2 * b if c == "FETCH_SIZE" else b)])
