"""Adversarial inputs for the half-CU one-launch kernel (csrc/frame_lean.hip): feature clouds of every scale and offset -- all points in one
cell, two far clusters, ranges too wide for the id map, coordinates near the int16 wrap of the reference's keys, negative and huge values,
ragged sizes -- in batches large enough for the kernel (>= 256 frames).  lccrf_batch_run (grid build, this kernel's own vertex numbering,
flagged frames re-run) must equal lccrf_batch_build + lccrf_batch_inference (hash build, the reference's numbering) bit for bit."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("lc-crf-slam_amd")


def cloud(rng, n, kind):
    if kind == "slam":
        return rng.uniform(0, 1, (n, 2)) * rng.uniform(5, 60, 2) + rng.uniform(-30, 30, 2)
    if kind == "one_cell":
        return np.full((n, 2), rng.uniform(-5, 5)) + rng.uniform(0, 1e-3, (n, 2))
    if kind == "clusters":
        c = rng.uniform(-400, 400, (2, 2))
        return c[rng.integers(0, 2, n)] + rng.normal(0, 1.5, (n, 2))
    if kind == "wide":                                     # key range beyond the id map: the frame flags itself
        return rng.uniform(-3000, 3000, (n, 2))
    if kind == "wrap":                                     # near / beyond the int16 range of the reference's keys
        return rng.uniform(0, 40, (n, 2)) + rng.choice([9000.0, 10500.0, -10800.0, 12000.0])
    if kind == "line":
        t = rng.uniform(-200, 200, n)
        return np.stack([t, 0.5 * t + rng.normal(0, 0.01, n)], 1)
    if kind == "grid":                                     # points exactly on lattice boundaries (ties in the rank, quirk Q3)
        return rng.integers(-40, 40, (n, 2)).astype(np.float64) * 0.5
    raise ValueError(kind)


def run(seed, F=288, maxN=2048):
    rng = np.random.default_rng(seed)
    kinds = ["slam", "one_cell", "clusters", "wide", "wrap", "line", "grid"]
    top = int(rng.choice([700, 1024, 1500, 2048]))
    sizes = [int(rng.integers(1, top + 1)) for _ in range(F)]
    sizes[0] = top
    sizes[1] = 0
    feats = [np.zeros((F, maxN, 2), np.float32) for _ in range(2)]
    label = np.full((F, maxN), -1, np.int16)
    which = []
    for f in range(F):
        n = sizes[f]
        k0 = kinds[int(rng.integers(0, len(kinds)))] if rng.random() < 0.2 else "slam"
        k1 = kinds[int(rng.integers(0, len(kinds)))] if rng.random() < 0.2 else "slam"
        which.append((k0, k1))
        feats[0][f, :n] = cloud(rng, n, k0) * (rng.uniform(0.03, 0.2) if k0 == "slam" else 1.0)   # appearance: few vertices, long rows
        feats[1][f, :n] = cloud(rng, n, k1)
        label[f, :n] = rng.integers(-1, 2, n)
    b = pkg.BatchCRF(F, maxN, 2, [2, 2], [10.0, 30.0])
    b.set_inputs_host(sizes, feats, label=label, conf=0.7)
    b.run(5, True)
    q1, m1, fb, shape = b.probability().view(np.uint32).copy(), b.map().copy(), b.fallback_frames(), b.fused_shape()
    v1 = [b.lattice_sizes(k).copy() for k in range(2)]
    b.set_engine(1)
    b.build(); b.inference(5, True)
    q2, m2 = b.probability().view(np.uint32), b.map()
    v2 = [b.lattice_sizes(k) for k in range(2)]
    bad = []
    for f in range(F):
        n = sizes[f]
        if (q1[f, :n] != q2[f, :n]).any() or (m1[f, :n] != m2[f, :n]).any() or any(int(v1[k][f]) != int(v2[k][f]) for k in range(2)):
            bad.append((f, n, which[f], [int(v1[k][f]) for k in range(2)], [int(v2[k][f]) for k in range(2)]))
    b.close()
    print("seed %d: top %d, shape %s, %d flagged frames re-run, %d of %d frames differ %s" % (seed, top, shape, fb, len(bad), F, bad[:3]), flush=True)
    return len(bad)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    sys.exit(1 if sum(run(s) for s in range(1, n + 1)) else 0)
