#!/usr/bin/env python3
"""Single-frame end-to-end latency through the object API (host buffers in, labels out):
what the call site src/Tracking.cc:1919-1930 would see per frame.  Run on the GPU box."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
pkg = importlib.import_module("lc-crf-slam_amd")
wl = importlib.import_module("lc-crf-slam_amd.workloads")
import pyoracle as po

def run(cls, fr, p, n_iter=5):
    c = cls(fr["N"], 2)
    c.set_unary_from_label(fr["init_label"], p["confidence"])
    c.add_pairwise(wl.appearance_features(fr), p["w1"])
    c.add_pairwise(wl.smooth_features(fr), p["w2"])
    c.inference_native(n_iter, True)
    m = c.map()
    c.close()
    return m

for N in (500, 1000, 2000, 3000):
    frames = [wl.slam_frame(N, s) for s in range(1, 9)]
    p = wl.TUM3
    for name, cls in (("hip", pkg.DenseCRFHIP), ("cpu-ref" if po.have_ref() else "cpu-port", po.RefCRF if po.have_ref() else po.OracleCRF)):
        for fr in frames[:3]: run(cls, fr, p)           # warm-up
        ts = []
        for rep in range(40):
            fr = frames[rep % len(frames)]
            t0 = time.perf_counter(); run(cls, fr, p); ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        print("N=%4d %-8s median %.0f us  min %.0f us  (create + unary + 2 kernels + 5 iterations + map, host to host)" % (N, name, np.median(ts), ts.min()))
