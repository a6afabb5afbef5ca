#!/usr/bin/env python3
"""Two labels, 100 000 points, COARSE kernels (a 2-D smoothness kernel: ~2000 vertices, rows of 150 entries on average): time per
mean-field iteration on the streaming engine (KernelDev::long_mode) and the build.    python scripts/coarse_l2_timing.py [N]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("lc-crf-slam_amd")
wl = importlib.import_module("lc-crf-slam_amd.workloads")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for dims in ([2, 5], [2], [3], [2, 2]):
    pb = wl.generic_problem(N, dims, 2, seed=3, spread=6.0)
    for F in (1, 4):
        b = pkg.BatchCRF(F, N, 2, dims, [float(w) for _, w in pb["kernels"]])
        b.set_inputs_host([N] * F, [np.repeat(f[None], F, 0) for f, _ in pb["kernels"]], unary=np.repeat(pb["unary"][None], F, 0))
        b.build(); b.synchronize(); b.build(); b.synchronize()
        b.inference(5, True); b.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            b.inference(5, True)
        b.synchronize()
        print("L=2 dims=%-6s F=%d V=%s  %.1f us per frame-iteration, build %.2f ms"
              % (dims, F, [int(b.lattice_sizes(k)[0]) for k in range(len(dims))], (time.perf_counter() - t0) / 15 / F * 1e6, b.last_timing()["build_ms"]))
        b.close()
