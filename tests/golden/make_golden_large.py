#!/usr/bin/env python3
"""Generate tests/golden/large.npz from the REFERENCE itself (oracle/_ref: the reference's own DenseCRF headers compiled in
place): frames above the 8192-point threshold at which the streaming engine switches to LOCALITY MODE (an internal Z-order
of the points, csrc/stream_engine.hip).  Results must not depend on that order -- these vectors pin that against the
reference directly, not through the restatement.

    python tests/golden/make_golden_large.py

Cases: a C5-shaped frame (9000 points, one 6-D kernel, labels, 3 iterations) and a generic one (8500 points, one 3-D
kernel, three labels, raw unaries, relax 0.9, 2 iterations).  Inputs and the reference's V, norm, Q and labels are stored.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402

wl = importlib.import_module("lc-crf-slam_amd.workloads")


def run(pb, n_iter, relax):
    c = po.RefCRF(pb["N"], pb["L"])
    if "unary" in pb:
        c.set_unary(pb["unary"])
    else:
        c.set_unary_from_label(pb["label"], pb["conf"])
    for f, w in pb["kernels"]:
        c.add_pairwise(f, w)
    kv = c.kernel(0)
    c.inference_native(n_iter, True, relax)
    out = dict(V=np.int32(kv["V"]), norm=kv["norm"].copy(), Q=c.probability().copy(), map=c.map().copy())
    c.close()
    return out


def main():
    assert po.have_ref(), "oracle/_ref is not built (needs /root/reference)"
    z = {}
    pb = wl.bilateral_problem(9000, 3)
    r = run(pb, 3, 1.0)
    z.update({"c5_features": pb["kernels"][0][0], "c5_w": np.float32(pb["kernels"][0][1]), "c5_label": pb["label"],
              "c5_conf": np.float32(pb["conf"]), "c5_iters": np.int32(3), "c5_relax": np.float32(1.0)})
    z.update({"c5_" + k: v for k, v in r.items()})
    pg = wl.generic_problem(8500, [3], 3, seed=17, spread=3.0)
    r = run(pg, 2, 0.9)
    z.update({"gen_features": pg["kernels"][0][0], "gen_w": np.float32(pg["kernels"][0][1]), "gen_unary": pg["unary"],
              "gen_iters": np.int32(2), "gen_relax": np.float32(0.9)})
    z.update({"gen_" + k: v for k, v in r.items()})
    path = os.path.join(HERE, "large.npz")
    np.savez_compressed(path, **z)
    print(path, os.path.getsize(path), "bytes; V =", int(z["c5_V"]), int(z["gen_V"]))


if __name__ == "__main__":
    main()
