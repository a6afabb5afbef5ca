#!/usr/bin/env python3
"""The reference's own demo (Thirdparty/DenseCRF/examples/example_cpu.cpp:79-103; README: "320x240, 21 classes, 10 iters: 225 ms"
end to end on the author's CPU): im1.ppm + anno1.ppm -> res1_cpu.ppm, 76 800 pixels, L = 21, a 2-D smoothness and a 5-D appearance
kernel, 10 mean-field iterations -- through the object API, host arrays in / labels out.  Checks the known answer (byte for byte)
and prints the host-to-host time of the CRF part and of the inference alone, beside the CPU checker's.
    python scripts/image_example.py [reps]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po            # noqa: E402  (the checker; this is a measurement script)

pkg = importlib.import_module("lc-crf-slam_amd")
z = np.load(os.path.join(ROOT, "tests", "golden", "example_im1.npz"))
im, res, lab, colors = z["im"], z["res"], z["label"], z["colors"]
H, W, _ = im.shape
wl = importlib.import_module("lc-crf-slam_amd.workloads")
f_smooth = wl.image_features(W, H, 3.0)
f_app = wl.image_features(W, H, 60.0, im, 20.0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
whole, inf = [], []
for _ in range(reps + 1):
    t0 = time.perf_counter()
    c = pkg.DenseCRFHIP(W * H, 21)
    c.set_unary_from_label(lab, 0.5)
    c.add_pairwise(f_smooth, 3.0)
    c.add_pairwise(f_app, 10.0)
    c.inference(10, True)
    m = c.map()
    whole.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    c.inference(10, True)
    m2 = c.map()
    inf.append(time.perf_counter() - t0)
    c.close()
col = colors[m]
out = np.stack([col & 255, (col >> 8) & 255, (col >> 16) & 255], -1).astype(np.uint8).reshape(H, W, 3)
o = po.OracleCRF(W * H, 21)
t0 = time.perf_counter()
o.set_unary_from_label(lab, 0.5)
o.add_pairwise(f_smooth, 3.0)
o.add_pairwise(f_app, 10.0)
o.inference_native(10, True)
cpu = time.perf_counter() - t0
print("known answer (res1_cpu.ppm) reproduced: %s, second inference identical: %s" % (np.array_equal(out, res), np.array_equal(m, m2)))
print("HIP object API: CRF part of the demo %.2f ms host to host, 10 iterations on the resident lattices %.2f ms (%.0f us per iteration); "
      "CPU checker (one core of this box) %.0f ms; reference README: 225 ms for the whole process on the author's machine"
      % (np.median(whole[1:]) * 1e3, np.median(inf[1:]) * 1e3, np.median(inf[1:]) * 1e5, cpu * 1e3))
