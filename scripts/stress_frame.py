"""Race hunt for the one-launch-per-frame kernel: many copies of a few frames in one batch, many launches; every copy of
every launch must equal its original bit for bit, and the originals must equal the two-kernel path."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("lc-crf-slam_amd")
wl = importlib.import_module("lc-crf-slam_amd.workloads")


def run(N, F=4096, reps=40, distinct=4):
    base = [wl.slam_problem(N, seed=3100 + i) for i in range(distinct)]
    feats = [np.stack([base[f % distinct]["kernels"][k][0] for f in range(F)]) for k in range(2)]
    label = np.stack([base[f % distinct]["label"] for f in range(F)])
    b = pkg.BatchCRF(F, N, 2, [2, 2], [10.0, 30.0])
    b.set_inputs_host([N] * F, feats, label=label, conf=0.7)
    b.build()
    b.inference(5, True)
    qref = b.probability()[:distinct].view(np.uint32).copy()
    mref = b.map()[:distinct].copy()
    bad = 0
    for _ in range(reps):
        b.run(5, True)
        assert b.engine() == 3
        Q, M = b.probability().view(np.uint32), b.map()
        for d in range(distinct):
            bad += int((Q[d::distinct] != qref[d]).any(axis=(1, 2)).sum()) + int((M[d::distinct] != mref[d]).any(axis=1).sum())
    b.close()
    print("N=%d: %d bad frame-launches of %d" % (N, bad, reps * F))
    return bad


if __name__ == "__main__":
    total = sum(run(N) for N in (2000, 1000, 500, 3000, 2047))
    sys.exit(1 if total else 0)
