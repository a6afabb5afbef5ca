#!/usr/bin/env python3
"""Large frames with MORE than two labels (the generic kernels of the streaming engine: k_splat / k_blur / k_slice / k_softmax):
time per mean-field iteration and the CPU checker's, same bits.    python scripts/generic_l_timing.py [N]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as po            # noqa: E402  (the checker; this is a measurement script)
import crf_cases as cc           # noqa: E402

pkg = importlib.import_module("lc-crf-slam_amd")
wl = importlib.import_module("lc-crf-slam_amd.workloads")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for L, dims in ((2, [6]), (4, [6]), (8, [5]), (21, [5]), (21, [2, 5])):
    pb = wl.generic_problem(N, dims, L, seed=3, spread=6.0)
    for F in (1, 4):
        b = pkg.BatchCRF(F, N, L, dims, [float(w) for _, w in pb["kernels"]])
        b.set_inputs_host([N] * F, [np.repeat(f[None], F, 0) for f, _ in pb["kernels"]], unary=np.repeat(pb["unary"][None], F, 0))
        b.build(); b.synchronize()
        b.inference(5, True); b.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            b.inference(5, True)
        b.synchronize()
        dt = (time.perf_counter() - t0) / 15 / F
        line = "L=%2d dims=%-7s F=%d  V=%s  %.1f us per frame-iteration, build %.2f ms" % (L, dims, F, [int(b.lattice_sizes(k)[0]) for k in range(len(dims))], dt * 1e6, b.last_timing()["build_ms"])
        if F == 1:
            o = cc.setup(po.OracleCRF, pb)
            t0 = time.perf_counter()
            o.inference_native(5, True)
            cpu = (time.perf_counter() - t0) / 5
            line += "   CPU checker %.1f ms per iteration, same bits: %s" % (cpu * 1e3, cc.same_bits(b.probability()[0], o.probability()))
            o.close()
        print(line)
        b.close()
