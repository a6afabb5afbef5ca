#!/usr/bin/env python3
"""Fuzz of the round-3 paths against the oracle (run on the GPU box): locality mode of the streaming engine (random sizes
above 8192, 1-6 feature dimensions, 2-3 labels, 1-2 kernels, uniform / clustered / duplicated features, labels or raw
unaries, ragged batches with empty frames, rebuilt twice) and the 512-lane shapes with per-frame fallback (random small
frames, a few sparse ones), and large single frames through the object API (inference() in locality mode, then a random
walk over the other entry points).  Every frame checked: lattice sizes, Q bit for bit, labels.
    python scripts/stress_locality.py [seconds]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle as po            # noqa: E402  (the checker; this is a test script)
import crf_cases as cc           # noqa: E402

pkg = importlib.import_module("lc-crf-slam_amd")
wl = importlib.import_module("lc-crf-slam_amd.workloads")


def features(rng, n, d, kind):
    spread = rng.uniform(1.5, 8.0)
    f = rng.normal(0.0, spread, (n, d)).astype(np.float32)
    if kind == 1 and n:                                   # clusters
        c = rng.normal(0.0, spread, (max(n // 500, 1), d)).astype(np.float32)
        f = (c[rng.integers(0, len(c), n)] + rng.normal(0, 0.2, (n, d))).astype(np.float32)
    if kind == 2 and n:                                   # a quarter of the points identical, some on lattice boundaries
        f[: n // 4] = f[0]
        q = rng.random(n) < 0.2
        f[q] = np.round(f[q] * 2) / 2
    return f


def check(b, pbs, sizes, n_iter, relax, tag):
    Q, M = b.probability(), b.map()
    K = len(pbs[0]["kernels"]) if pbs else 0
    Vs = [b.lattice_sizes(k) for k in range(K)]
    for f, pb in enumerate(pbs):
        n = sizes[f]
        if n == 0:
            continue
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(n_iter, True, relax)
        ok = cc.same_bits(Q[f, :n], o.probability()) and np.array_equal(M[f, :n], o.map())
        ok = ok and all(int(Vs[k][f]) == o.kernel(k)["V"] for k in range(K))
        o.close()
        if not ok:
            print("MISMATCH", tag, "frame", f, "n", n)
            return False
    return True


def large_case(rng):
    K, L = int(rng.integers(1, 3)), int(rng.integers(2, 4))
    dims = [int(rng.integers(1, 9)) for _ in range(K)]
    F = int(rng.integers(1, 10))
    # round 4: below 8 frames the XCD-chunked grids, one frame: two blur passes per launch (+ two-hop table, + the left-over pass in
    # the slice); sizes from 4200 points (streaming build, no permutation) to 20000 (locality mode); the vertex order option at random
    lo = 4200 if rng.random() < 0.3 else 8192
    maxN = int(rng.integers(lo, 20000 if lo == 8192 else 8192))
    sizes = [int(rng.integers(lo, maxN + 1)) if rng.random() > 0.15 else int(rng.integers(0, 5)) for _ in range(F)]
    sizes[0] = maxN
    vertex_order = int(rng.integers(0, 3))
    use_label = L == 2 and rng.random() < 0.5
    ws = [float(rng.uniform(1, 12)) for _ in range(K)]
    kind = int(rng.integers(0, 3))
    pbs = []
    feats = [np.zeros((F, maxN, d), np.float32) for d in dims]
    unary = np.zeros((F, maxN, L), np.float32)
    label = np.full((F, maxN), -1, np.int16)
    for f, n in enumerate(sizes):
        ks = [(features(rng, n, d, kind), np.float32(w)) for d, w in zip(dims, ws)]
        pb = dict(N=n, L=L, kernels=ks)
        if use_label:
            pb["label"] = rng.integers(-1, 2, n).astype(np.int16)
            pb["conf"] = np.float32(0.7)
            label[f, :n] = pb["label"]
        else:
            pb["unary"] = rng.uniform(0.05, 3.0, (n, L)).astype(np.float32)
            unary[f, :n] = pb["unary"]
        for k in range(K):
            feats[k][f, :n] = ks[k][0]
        pbs.append(pb)
    n_iter, relax = int(rng.integers(1, 4)), float(rng.choice([1.0, 0.8]))
    b = pkg.BatchCRF(F, maxN, L, dims, ws)
    b.set_option(pkg.OPT_VERTEX_ORDER, vertex_order)
    if use_label:
        b.set_inputs_host(sizes, feats, label=label, conf=0.7)
    else:
        b.set_inputs_host(sizes, feats, unary=unary)
    b.build()
    if rng.random() < 0.5:
        b.build()
    b.inference(n_iter, True, relax=relax)
    ok = check(b, pbs, sizes, n_iter, relax, "large K=%d L=%d dims=%s F=%d kind=%d label=%s maxN=%d vertex_order=%d" % (K, L, dims, F, kind, use_label, maxN, vertex_order))
    b.close()
    return ok


def object_case(rng):
    """One large frame through the OBJECT API: inference() in locality mode, then a random walk over the entry points that want the
    caller's point order (they re-build the lattices the plain way, once), new unaries, a recycled handle -- against the oracle."""
    K, L = int(rng.integers(1, 3)), int(rng.integers(2, 4))
    dims = [int(rng.integers(1, 9)) for _ in range(K)]
    n = int(rng.integers(8192, 16000))
    kind = int(rng.integers(0, 3))
    pb = dict(N=n, L=L, kernels=[(features(rng, n, d, kind), np.float32(rng.uniform(1, 12))) for d in dims])
    if L == 2 and rng.random() < 0.5:
        pb["label"], pb["conf"] = rng.integers(-1, 2, n).astype(np.int16), np.float32(0.7)
    else:
        pb["unary"] = rng.uniform(0.05, 3.0, (n, L)).astype(np.float32)
    h, o = cc.setup(pkg.DenseCRFHIP, pb), cc.setup(po.OracleCRF, pb)
    tag = "object K=%d L=%d dims=%s kind=%d n=%d" % (K, L, dims, kind, n)
    ok = True
    for step in range(int(rng.integers(2, 6))):
        op = int(rng.integers(0, 6))
        if op <= 1:
            it, relax = int(rng.integers(1, 4)), float(rng.choice([1.0, 0.8]))
            h.inference(it, True, relax); o.inference_native(it, True, relax)
            ok = ok and cc.same_bits(h.probability(), o.probability()) and np.array_equal(h.map(), o.map())
        elif op == 2:
            h.start_inference(); o.start_inference()
            h.step_inference(0.9); o.step_inference(0.9)
            ok = ok and cc.same_bits(h.probability(), o.probability())
        elif op == 3:
            x = rng.normal(0, 1, (n, L)).astype(np.float32)
            k = int(rng.integers(0, K))
            ok = ok and cc.same_bits(h.apply(k, np.zeros((n, L), np.float32), x), o.apply(k, np.zeros((n, L), np.float32), x))
        elif op == 4:
            k = int(rng.integers(0, K))
            kh, ko = h.kernel(k), o.kernel(k)
            ok = ok and kh["V"] == ko["V"] and np.array_equal(kh["offset"], ko["offset"]) and cc.same_bits(kh["norm"], ko["norm"])
        else:
            u = rng.uniform(0.05, 3.0, (n, L)).astype(np.float32)
            h.set_unary(u); o.set_unary(u)
        if not ok:
            print("MISMATCH", tag, "step", step, "op", op)
            break
    h.close(); o.close()
    return ok


def small_case(rng):
    F = int(rng.integers(256, 400))
    maxN = int(rng.integers(40, 1025))
    base = [wl.slam_problem(int(rng.integers(0, maxN + 1)), seed=int(rng.integers(1, 1 << 30))) for _ in range(6)]
    base[0] = wl.slam_problem(maxN, seed=int(rng.integers(1, 1 << 30)))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_hip_parity import _shaped_problem
    odd = _shaped_problem(wl, maxN, "sparse", seed=5) if maxN >= 300 else base[1]
    where = set(int(x) for x in rng.integers(0, F, 3))
    pbs = [odd if f in where else base[f % 6] for f in range(F)]
    sizes = [pb["N"] for pb in pbs]
    feats = [np.zeros((F, maxN, 2), np.float32) for _ in range(2)]
    label = np.full((F, maxN), -1, np.int16)
    for f, pb in enumerate(pbs):
        n = pb["N"]
        label[f, :n] = pb["label"]
        for k in range(2):
            feats[k][f, :n] = pb["kernels"][k][0]
    b = pkg.BatchCRF(F, maxN, 2, [2, 2], [10.0, 30.0])
    b.set_inputs_host(sizes, feats, label=label, conf=0.7)
    b.run(5, True)
    uniq = {}
    for f, pb in enumerate(pbs):
        uniq.setdefault(id(pb), f)
    idx = sorted(uniq.values())
    Q, M = b.probability(), b.map()
    ok = True
    for f in idx:
        pb, n = pbs[f], sizes[f]
        if n == 0:
            continue
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(5, True)
        for g in range(F):
            if pbs[g] is pb and not (cc.same_bits(Q[g, :n], o.probability()) and np.array_equal(M[g, :n], o.map())):
                print("MISMATCH small maxN=%d frame %d (copy of %d), fallback %d" % (maxN, g, f, b.fallback_frames()))
                ok = False
                break
        o.close()
        if not ok:
            break
    b.close()
    return ok


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(time.time()))
    t0, n_large, n_small, n_obj = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        if not large_case(rng):
            return 1
        n_large += 1
        if not small_case(rng):
            return 1
        n_small += 1
        if n_large % 4 == 0:
            if not object_case(rng):
                return 1
            n_obj += 1
    print("stress ok: %d large batches (locality mode), %d small batches (512-lane shapes + per-frame fallback), %d large object-API frames in %.0f s"
          % (n_large, n_small, n_obj, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
