#!/usr/bin/env python3
"""Mid-size frames on the streaming engine (launch-bound: 3000-6000 points, 6-D / 3-D kernels, 1 / 8 / 64 frames in flight): build and
10-iteration inference time per batch.  A/B by environment, e.g. `LCCRF_PAIR_FUSE_MAX=1 python scripts/mid_size_ab.py` (two blur passes
per launch for one frame only) against the default (also for small launches: <= 0.7 M vertices over all frames)."""
import importlib, sys, time, numpy as np
sys.path.insert(0, ".")
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
for N, d, F in ((5000, 6, 8), (3000, 6, 8), (6000, 3, 8), (5000, 6, 1), (2000, 6, 64)):
    pb = wl.bilateral_problem(N, 3) if d == 6 else wl.generic_problem(N, [d], 2, seed=4, spread=3.0)
    f = pb["kernels"][0][0]
    b = pkg.BatchCRF(F, N, 2, [d], [10.0])
    un = np.random.default_rng(1).uniform(0.1, 2, (F, N, 2)).astype(np.float32)
    b.set_inputs_host([N] * F, [np.repeat(f[None], F, 0)], unary=un)
    b.build(); b.synchronize(); b.build(); b.synchronize()
    bm = b.last_timing()["build_ms"]
    for _ in range(2): b.inference(10, True)
    b.synchronize(); t0 = time.perf_counter()
    for _ in range(5): b.inference(10, True)
    b.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("N %5d d %d F %2d  build %.3f ms  inference(10) %.3f ms  engine %d  Qsum %.6f" % (N, d, F, bm, dt * 1e3, b.engine(), float(b.probability().sum())))
    b.close()
