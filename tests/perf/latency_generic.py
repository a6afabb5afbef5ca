"""Object-API latency of configurations that run on the streaming engine (anything but L=2 / 2-D kernels),
next to the oracle's scalar C on one core."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as po
import crf_cases as cc
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
for N, dl, L in [(2000, [2, 2], 3), (2000, [3], 2), (2000, [5], 4), (20000, [5], 21), (100000, [6], 2)]:
    pb = wl.generic_problem(N, dl, L, seed=1, spread=3.0)
    def run(cls, reps):
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            c = cc.setup(cls, pb); c.inference_native(5, True); m = c.map(); c.close()
            t.append(time.perf_counter() - t0)
        return np.median(t) * 1e3, m
    th, mh = run(pkg.DenseCRFHIP, 12)
    to, mo = run(po.OracleCRF, 3)
    print("N=%6d d=%s L=%2d  hip %8.3f ms   oracle (1 core) %8.2f ms   labels equal: %s" % (N, dl, L, th, to, np.array_equal(mh, mo)))
