"""The unary builder (first "next" row): Tracking::ComputeMapPointErrAndObserv +
Tracking::RroughClassify for a whole frame.  PARITY UNPINNED against the reference (no fixture exists,
src/Tracking.cc is unbuildable here); the HIP path is checked against the oracle's restatement."""
import importlib

import numpy as np
import pytest

import crf_cases as cc

pkg = importlib.import_module("lc-crf-slam_amd")


def test_oracle_unary_build_known_answers(po, wl):
    """CPU: the whole-frame oracle equals its per-point pieces, drops empty points, uses the prior."""
    sc = wl.map_point_scene(300, 10, seed=3)
    obs, err, dep, lab = po.oracle_unary_build(sc["Xw"], sc["obs_ptr"], sc["obs_kf"], sc["obs_kp"],
                                               sc["kf_pose"], sc["kf_intr"], sc["kf_bounds"])
    n = np.diff(sc["obs_ptr"])
    assert np.array_equal(obs, n.astype(np.float32))
    assert np.all(lab[n == 0] == -1) and np.all((lab[n > 0] == 0) | (lab[n > 0] == 1))
    assert np.all(err[n == 0] == 0) and np.all(dep[n == 0] == 0)
    assert 0 < (lab == 0).sum() < (lab == 1).sum()                 # mostly static, some moving
    ref = po.oracle_rough_classify(obs[n > 0], err[n > 0], dep[n > 0])
    assert np.array_equal(ref, lab[n > 0])
    mp = np.full(300, 0.5)
    _, _, _, lab2 = po.oracle_unary_build(sc["Xw"], sc["obs_ptr"], sc["obs_kf"], sc["obs_kp"], sc["kf_pose"],
                                          sc["kf_intr"], sc["kf_bounds"], match_prob=mp)
    assert (lab2 == 0).sum() <= (lab == 0).sum()                   # a positive prior can only help


@pytest.mark.gpu
@pytest.mark.parametrize("n_points,n_kf,seed", [(0, 3, 1), (1, 1, 2), (257, 8, 3), (2000, 15, 4), (5000, 40, 5)])
def test_hip_unary_build_matches_oracle(po, wl, n_points, n_kf, seed):
    sc = wl.map_point_scene(n_points, n_kf, seed)
    args = (sc["Xw"], sc["obs_ptr"], sc["obs_kf"], sc["obs_kp"], sc["kf_pose"], sc["kf_intr"], sc["kf_bounds"])
    for mp in (None, np.random.default_rng(seed).uniform(0, 1, n_points)):
        o = po.oracle_unary_build(*args, match_prob=mp)
        h = pkg.unary_build(*args, match_prob=mp)
        for name, a, b in zip(("observs", "error", "depth"), o, h):
            assert cc.same_bits(a, b), name                        # bit-identical statistics
        assert np.array_equal(o[3], h[3])                          # identical rough labels


@pytest.mark.gpu
def test_hip_unary_build_feeds_the_crf(po, wl):
    """End to end as in Tracking::DynamicDetectionWithCRF: unary builder -> drop empty points -> CRF."""
    sc = wl.map_point_scene(1500, 12, seed=8)
    obs, err, dep, lab = pkg.unary_build(sc["Xw"], sc["obs_ptr"], sc["obs_kf"], sc["obs_kp"], sc["kf_pose"],
                                         sc["kf_intr"], sc["kf_bounds"])
    keep = lab >= 0                                                # Tracking.cc:1858
    xy = np.random.default_rng(1).uniform([0, 0], [640, 480], (int(keep.sum()), 2)).astype(np.float32)
    p = wl.TUM3
    res = []
    for cls in (po.OracleCRF, pkg.DenseCRFHIP):
        c = cls(int(keep.sum()), 2)
        c.set_unary_from_label(lab[keep], p["confidence"])
        f = np.stack([obs[keep] / np.float32(p["stdev_beta"]), err[keep] / np.float32(p["stdev_alpha"])], 1)
        c.add_pairwise(f.astype(np.float32), p["w1"])
        c.add_pairwise((xy / np.float32(p["point2d_stdev"])).astype(np.float32), p["w2"])
        c.inference_native(5, True)
        res.append((c.probability(), c.map()))
    assert cc.same_bits(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_unary_build_argument_checks():
    lib = pkg.lib()
    assert lib.lccrf_unary_build(0, -1, None, None, None, None, 0, None, None, None, None, None, None, None,
                                 None, None) == -1
