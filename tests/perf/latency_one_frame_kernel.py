import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
for N in (500, 1000, 2000):
    pb = wl.slam_problem(N, seed=5)
    b = pkg.BatchCRF(1, N, 2, [2, 2], [10.0, 30.0])
    b.set_inputs_host([N], [pb["kernels"][k][0][None] for k in range(2)], label=pb["label"][None], conf=0.7)
    for _ in range(5): b.run(5, True)
    b.synchronize()
    ts = []
    for _ in range(50):
        b.synchronize(); t0 = time.perf_counter(); b.run(5, True); b.synchronize(); ts.append(time.perf_counter() - t0)
    print("N=%d one-launch k_frame, launch+sync wall: median %.1f us" % (N, np.median(ts) * 1e6))
    b.close()
