"""scripts/sim_vertex_order.py -- the numpy restatement of the lattice behind round 4's vertex-order study -- must
describe the REAL lattice: its vertex count against the oracle's on a few shapes, and the identity the sorted build rests on (a
vertex's coordinates in the basis of the blur directions are integers; a blur neighbour is a unit step there)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.mark.parametrize("N,d", [(3000, 2), (2500, 3), (1501, 6)])
def test_simulator_lattice_is_the_oracles_lattice(po, wl, N, d):
    sim = importlib.import_module("sim_vertex_order")
    pb = wl.generic_problem(N, [d], 2, seed=5 + d, spread=3.0)
    f = pb["kernels"][0][0]
    keys, _ = sim.lattice(f)
    flat = keys.reshape(-1, d)
    uk, inv = np.unique(sim.pack(flat), return_inverse=True)
    import crf_cases as cc
    o = cc.setup(po.OracleCRF, pb)
    ko = o.kernel(0)
    # (the reference also hashes the phantom points of the last block of four; N % 4 == 0 here or the origin's simplex is added)
    phantom = 0
    if N % 4:
        z = np.zeros((1, d), np.float32)
        kz, _ = sim.lattice(z)
        phantom = len(set(sim.pack(kz.reshape(-1, d)).tolist()) - set(uk.tolist()))
    assert len(uk) + phantom == ko["V"]
    # same partition of the entries into vertices as the reference's offset_ (ids differ, the grouping must not)
    off = ko["offset"].reshape(-1)
    pair = np.unique(np.stack([inv, off], 1), axis=0)
    assert len(pair) == len(uk) == len(np.unique(off))
    # the lattice in the basis of its blur directions: integer coordinates, axis j < d = a unit step along coordinate j
    ukeys = np.zeros((len(uk), d), np.int32)
    ukeys[inv] = flat
    xd = -ukeys.sum(1)
    assert np.all((xd[:, None] - ukeys) % (d + 1) == 0)
    cgrid = (xd[:, None] - ukeys) // (d + 1)
    nbr = ko["nbr"]                                       # [d+1][V][2], reference ids
    id_of = np.empty(len(uk), np.int64)
    id_of[inv] = off                                      # simulator vertex -> reference id
    c_by_id = np.zeros((ko["V"], d), np.int64)
    c_by_id[id_of] = cgrid
    known = np.zeros(ko["V"], bool)
    known[id_of] = True
    for j in range(d + 1):
        n2 = nbr[j, :, 1]
        ok = (n2 >= 0) & known & known[np.maximum(n2, 0)]
        step = c_by_id[n2[ok]] - c_by_id[np.nonzero(ok)[0]]
        want = np.zeros(d, np.int64)
        if j < d:
            want[j] = 1
        else:
            want[:] = -1
        assert ok.sum() > 0 and np.all(step == want), j
    o.close()
