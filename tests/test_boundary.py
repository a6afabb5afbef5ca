"""The reference's plug-in points through the C-ABI (round 2): PairwisePotential::apply (densecrf_base.h:18,
pairwise3d.h:73-78), the bare lattice filter PermutohedralLatticeCPU::init + compute (permutohedral_cpu.h:241,634)
with an arbitrary value_size, and DenseCRF's protected virtuals (densecrf_base.h:34-36) -- against golden vectors
generated from the reference's own headers (tests/golden/filter.npz, make_golden_filter.py).

CPU part: the oracle reproduces those vectors bit for bit.  GPU part: so does the HIP path."""
import importlib
import os

import numpy as np
import pytest

import crf_cases as cc

pkg = importlib.import_module("lc-crf-slam_amd")
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "filter.npz"))
APPLY = [str(c)[6:] for c in Z["cases"] if str(c).startswith("apply_")]
FILTER = [str(c)[7:] for c in Z["cases"] if str(c).startswith("filter_")]


def _apply_case(name):
    p = "apply_" + name
    pb = cc.case_problem(Z, p)
    K = len(pb["kernels"])
    return pb, [(Z[p + "_in%d" % k], Z[p + "_out0_%d" % k], Z[p + "_out%d" % k]) for k in range(K)]


@pytest.mark.parametrize("name", APPLY)
def test_oracle_apply_matches_reference_fixture(po, name):
    pb, io = _apply_case(name)
    o = cc.setup(po.OracleCRF, pb)
    for k, (x, out0, exp) in enumerate(io):
        assert cc.same_bits(o.apply(k, out0, x), exp), (name, k)


@pytest.mark.parametrize("name", FILTER)
def test_oracle_filter_matches_reference_fixture(po, name):
    p = "filter_" + name + "_"
    y, V = po.oracle_lattice_filter(Z[p + "feat"], Z[p + "in"])
    assert V == int(Z[p + "V"]) and cc.same_bits(y, Z[p + "out"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", APPLY)
def test_hip_apply_matches_reference_fixture(name):
    pb, io = _apply_case(name)
    h = cc.setup(pkg.DenseCRFHIP, pb)
    for k, (x, out0, exp) in enumerate(io):
        assert cc.same_bits(h.apply(k, out0, x), exp), (name, k)
    h.inference(2, True)                                   # the CRF itself is undisturbed by apply()
    g = cc.setup(pkg.DenseCRFHIP, pb)
    g.inference(2, True)
    assert cc.same_bits(h.probability(), g.probability())


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILTER)
def test_hip_lattice_filter_matches_reference_fixture(name):
    p = "filter_" + name + "_"
    y, V = pkg.lattice_filter(Z[p + "feat"], Z[p + "in"])
    assert V == int(Z[p + "V"])
    assert cc.same_bits(y, Z[p + "out"]), np.abs(y - Z[p + "out"]).max()


@pytest.mark.gpu
def test_hip_filter_edge_cases(po):
    rng = np.random.default_rng(5)
    for N, d, vs in ((0, 2, 2), (1, 3, 1), (3, 2, 5), (4097, 4, 64)):
        f = rng.normal(0, 2, (N, d)).astype(np.float32)
        x = rng.normal(0, 1, (N, vs)).astype(np.float32)
        y, V = pkg.lattice_filter(f, x)
        yo, Vo = po.oracle_lattice_filter(f, x) if N else (x, 0)
        assert (V == Vo or N == 0) and cc.same_bits(y, yo), (N, d, vs)
    with pytest.raises(pkg.LccrfError):
        pkg.lattice_filter(np.zeros((4, 2), np.float32), np.zeros((4, 65), np.float32))   # value_size > LCCRF_MAX_LABELS


@pytest.mark.gpu
@pytest.mark.parametrize("L", [2, 5])
def test_hip_protected_virtuals_match_oracle(po, wl, L):
    """expAndNormalize / stepInit / buildMap on caller arrays (lccrf_exp_and_normalize, lccrf_step_init, lccrf_map_of)."""
    rng = np.random.default_rng(L)
    N = 777
    pb = wl.generic_problem(N, [2], L, seed=12)
    h, o = cc.setup(pkg.DenseCRFHIP, pb), cc.setup(po.OracleCRF, pb)
    assert cc.same_bits(h.step_init(), -pb["unary"])
    x = rng.normal(0, 6, (N, L)).astype(np.float32)
    x[::11] -= 40                                           # beyond fast_exp's cut-off
    old = rng.uniform(0, 1, (N, L)).astype(np.float32)
    lib = po.oracle_lib()
    for scale, relax in ((1.0, 1.0), (-1.0, 1.0), (1.0, 0.6)):
        exp = old.copy()
        lib.orc_exp_and_normalize(exp.ctypes.data_as(po._f32p), x.ctypes.data_as(po._f32p), N, L, scale, relax)
        got = h.exp_and_normalize(x, scale, relax, old=old)
        assert cc.same_bits(got, exp), (scale, relax)
    q = rng.uniform(0, 1, (N, L)).astype(np.float32)
    q[::5] = q[::5, :1]                                     # ties: the first maximum wins
    assert np.array_equal(h.map_of(q), q.argmax(-1).astype(np.int16))
