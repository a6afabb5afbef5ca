"""Golden vectors for the two boundary entry points added in round 2, generated in THIS container from the
reference's own headers (oracle/_ref/liblccrf_ref.so, built by oracle/Makefile from /root/reference in place):

  apply_*   PairwisePotential::apply (densecrf_base.h:18; pairwise_cpu.h:53-57 == pairwise3d.h:73-78):
            out += w * norm * compute(in) for one kernel of a CRF
  filter_*  PermutohedralLatticeCPU::init + compute with an arbitrary value_size (permutohedral_cpu.h:241,634)

Run:  python tests/golden/make_golden_filter.py   ->  tests/golden/filter.npz  (data only)."""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as po  # noqa: E402
import crf_cases as cc  # noqa: E402

wl = importlib.import_module("lc-crf-slam_amd.workloads")


def main():
    po.build()
    assert po.have_ref(), "needs /root/reference (oracle/_ref/liblccrf_ref.so)"
    z, cases = {}, []
    rng = np.random.default_rng(2024)
    for name, pb in (("slam501", wl.slam_problem(501, seed=4)),
                     ("d5L3", wl.generic_problem(257, [5], 3, seed=5, lattice_ties=True)),
                     ("multiL4", wl.generic_problem(301, [2, 3], 4, seed=6))):
        r = cc.setup(po.RefCRF, pb)
        p = "apply_" + name + "_"
        z[p + "N"], z[p + "L"], z[p + "K"] = pb["N"], pb["L"], len(pb["kernels"])
        if "unary" in pb:
            z[p + "unary"] = pb["unary"]
        else:
            z[p + "label"], z[p + "conf"] = pb["label"], pb["conf"]
        for k, (f, w) in enumerate(pb["kernels"]):
            z[p + "feat%d" % k], z[p + "w%d" % k] = f, w
            x = rng.uniform(0, 1, (pb["N"], pb["L"])).astype(np.float32)
            out0 = rng.normal(0, 2, (pb["N"], pb["L"])).astype(np.float32)
            z[p + "in%d" % k], z[p + "out0_%d" % k], z[p + "out%d" % k] = x, out0, r.apply(k, out0, x)
        r.close()
        cases.append("apply_" + name)
    for name, N, d, vs in (("d2v1", 1003, 2, 1), ("d2v3", 700, 2, 3), ("d5v7", 300, 5, 7), ("d6v2", 2048, 6, 2), ("d3v21", 129, 3, 21)):
        f = rng.normal(0, 3.0, (N, d)).astype(np.float32)
        f[::9] = np.round(f[::9] * 2) / 2                     # lattice ties
        x = rng.normal(0, 1, (N, vs)).astype(np.float32)
        y, V = po.ref_lattice_filter(f, x)
        p = "filter_" + name + "_"
        z[p + "feat"], z[p + "in"], z[p + "out"], z[p + "V"] = f, x, y, V
        cases.append("filter_" + name)
    z["cases"] = np.array(cases)
    fn = os.path.join(HERE, "filter.npz")
    np.savez_compressed(fn, **z)
    print(fn, os.path.getsize(fn) // 1024, "KiB", cases)


if __name__ == "__main__":
    main()
