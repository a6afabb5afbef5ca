#!/usr/bin/env python3
"""lccrf_batch_run + label download for small batches of 2000-keypoint frames (4 / 32 / 128 frames), median of 40:
the two-workgroup form of the frame kernel serves batches of up to 64 frames (LCCRF_NO_DUAL=1 for the A/B).  Run on the GPU box."""
import importlib, sys, os, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
import crf_cases as cc
for F in (4, 32, 128):
    pbs = [wl.slam_problem(2000, seed=700 + i) for i in range(F)]
    b = cc.batch_of(pkg, pbs) if hasattr(cc, "batch_of") else None
    if b is None:
        from test_frame_engine import _batch_of
        b = _batch_of(pbs)
    for _ in range(5): b.run(5, True); b.map()
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); b.run(5, True); b.map(); ts.append(time.perf_counter() - t0)
    print("F=%d  run+map median %.1f us" % (F, np.median(ts) * 1e6), flush=True)
    b.close()
