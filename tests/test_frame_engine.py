"""The one-launch-per-frame kernel (csrc/frame_engine.hip, lccrf_batch_run and the object API's inference()):
lattice build + normalisation + mean-field inference of a frame in ONE launch, against
  (1) the committed golden vectors generated from the reference itself (tests/golden/slam.npz),
  (2) the oracle on fresh seeded frames of every size class (every N % 4, 1 .. 4 points per lane),
  (3) adversarial lattices (one cell, long rows, rows on the chain's unit boundaries, lattices too large
      for the kernel -> its fallback), ragged batches, raw unaries, relax != 1, a single kernel,
  (4) the two-kernel path (LCCRF_NO_FRAME) bit for bit.
Bar: labels identical, Q bit-identical, V equal to the reference's M_.
"""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import crf_cases as cc
from test_hip_parity import _shaped_problem

pkg = importlib.import_module("lc-crf-slam_amd")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch_of(pbs, maxN=None, use_unary=False):
    F = len(pbs)
    maxN = maxN or max(max(pb["N"] for pb in pbs), 1)
    K = len(pbs[0]["kernels"])
    feats = [np.zeros((F, maxN, 2), np.float32) for _ in range(K)]
    label = np.full((F, maxN), -1, np.int16)
    unary = np.zeros((F, maxN, 2), np.float32)
    for f, pb in enumerate(pbs):
        n = pb["N"]
        if "label" in pb:
            label[f, :n] = pb["label"]
        if "unary" in pb:
            unary[f, :n] = pb["unary"]
        for k in range(K):
            feats[k][f, :n] = pb["kernels"][k][0]
    b = pkg.BatchCRF(F, maxN, 2, [2] * K, [float(pbs[0]["kernels"][k][1]) for k in range(K)])
    if use_unary:
        b.set_inputs_host([pb["N"] for pb in pbs], feats, unary=unary)
    else:
        b.set_inputs_host([pb["N"] for pb in pbs], feats, label=label, conf=pbs[0].get("conf", 0.7))
    return b


def _check_vs_oracle(po, pbs, b, n_iter, relax=1.0, expect_engine=3):
    b.run(n_iter, True, relax=relax)
    Q, M = b.probability(), b.map()
    assert b.engine() == expect_engine
    Vs = [b.lattice_sizes(k) for k in range(len(pbs[0]["kernels"]))]
    for f, pb in enumerate(pbs):
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(n_iter, True, relax)
        n = pb["N"]
        for k in range(len(pb["kernels"])):
            assert Vs[k][f] == (o.kernel(k)["V"] if n else 0), (f, k)
        assert cc.same_bits(Q[f, :n], o.probability()), (f, n)
        assert np.array_equal(M[f, :n], o.map()), (f, n)
        o.close()


def _slam_cases():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "slam.npz"))
    return [str(c) for c in z["cases"]]


@pytest.mark.parametrize("case", _slam_cases())
def test_frame_kernel_on_reference_fixtures(golden, case):
    """Golden vectors generated from the reference's own headers: Q after t iterations and the labels, for every
    recorded t, each as one launch."""
    z = golden["slam"]
    pb, exp = cc.case_problem(z, case), cc.case_expected(z, case)
    b = _batch_of([pb, pb])
    for t in sorted(exp["Q"]):
        b.run(t, True, relax=exp["relax"])
        Q, M = b.probability(), b.map()
        assert b.engine() == 3
        for f in range(2):
            assert cc.same_bits(Q[f, :pb["N"]], exp["Q"][t]), (case, t)
            assert np.array_equal(M[f, :pb["N"]], exp["map"][t]), (case, t)
        for k, V in enumerate(exp["V"]):
            assert list(b.lattice_sizes(k)) == [V, V]
    b.close()


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 63, 64, 65, 511, 1000, 1001, 1024, 1025, 2000, 2002, 2047, 2999, 3000, 3073, 4096])
def test_frame_kernel_matches_oracle_sizes(po, wl, N):
    pbs = [wl.slam_problem(N, seed=700 + N), wl.slam_problem(max(N - 1, 1), seed=701 + N)]
    b = _batch_of(pbs)
    _check_vs_oracle(po, pbs, b, 5)
    b.close()


def test_frame_kernel_ragged_batch_and_idempotence(po, wl):
    sizes = [2000, 0, 1, 777, 1999, 5, 1024, 2000, 0, 333]
    pbs = [wl.slam_problem(n, seed=40 + i) for i, n in enumerate(sizes)]
    b = _batch_of(pbs, maxN=2000)
    _check_vs_oracle(po, pbs, b, 5)
    Q, M = b.probability(), b.map()
    b.run(5, True)                                         # running the same batch again changes nothing
    assert cc.same_bits(b.probability(), Q) and np.array_equal(b.map(), M)
    b.close()


@pytest.mark.parametrize("shape,N", [("one_cell", 2000), ("one_cell", 4096), ("one_cell", 70), ("two_clusters", 2047),
                                     ("two_clusters", 2600), ("rows_of_8", 2000), ("rows_of_8", 1030)])
def test_frame_kernel_on_adversarial_lattices(po, wl, shape, N):
    pb = _shaped_problem(wl, N, shape, seed=5)
    b = _batch_of([pb])
    _check_vs_oracle(po, [pb], b, 4)
    b.close()
    pb1 = dict(pb, kernels=pb["kernels"][:1])              # and as the only kernel (K = 1), damped
    b = _batch_of([pb1])
    _check_vs_oracle(po, [pb1], b, 3, relax=0.8)
    b.close()


def test_frame_kernel_falls_back_when_a_lattice_does_not_fit(po, wl):
    """Every point in its own cell: V = 3N vertices per kernel cannot live in one workgroup's LDS.  The launch flags
    that FRAME; the library re-runs exactly the flagged frames on the build + inference kernels and scatters their
    results back: same answers, the batch stays a one-launch batch (engine 3)."""
    pbs = [_shaped_problem(wl, 1200, "sparse", seed=5), wl.slam_problem(1200, seed=9)]
    b = _batch_of(pbs)
    b.run(4, True)
    Q, M = b.probability(), b.map()
    assert b.engine() == 3 and b.fallback_frames() == 1
    Vs = [b.lattice_sizes(k) for k in range(2)]
    for f, pb in enumerate(pbs):
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(4, True)
        assert cc.same_bits(Q[f, :pb["N"]], o.probability()) and np.array_equal(M[f, :pb["N"]], o.map())
        assert [Vs[k][f] for k in range(2)] == [o.kernel(k)["V"] for k in range(2)]
    b.close()
    b = _batch_of(pbs[:1])                                 # every frame flagged: the whole batch takes the two-kernel path in place
    b.run(4, True)
    assert b.engine() in (1, 2) and b.fallback_frames() == 1
    o = cc.setup(po.OracleCRF, pbs[0])
    o.inference_native(4, True)
    assert cc.same_bits(b.probability()[0, :1200], o.probability()) and np.array_equal(b.map()[0, :1200], o.map())
    b.close()


def test_one_outlier_frame_costs_one_frame(po, wl):
    """1 frame that does not fit among 1023 that do.  Only that frame is re-run; labels, Q and the
    packed label bits of EVERY frame are right; raw unaries take the same route."""
    import time
    F, N = 1024, 1200
    base = [wl.slam_problem(N, seed=3100 + i) for i in range(4)]
    odd = _shaped_problem(wl, N, "sparse", seed=5)
    where = 517
    pbs = [odd if f == where else base[f % 4] for f in range(F)]
    b, ball = _batch_of(pbs), _batch_of([base[f % 4] for f in range(F)])
    for x in (b, ball):
        x.run(5, True); x.synchronize(); x.run(5, True); x.synchronize()      # warm: allocations, the fallback sub-engine
    ts = {}
    for name, x in (("odd", b), ("all", ball), ("odd2", b), ("all2", ball)):
        t0 = time.perf_counter()
        for _ in range(5):
            x.run(5, True)
            x.synchronize()
        ts[name] = (time.perf_counter() - t0) / 5
    assert b.engine() == 3 and b.fallback_frames() == 1 and ball.fallback_frames() == 0
    Q, M = b.probability(), b.map()
    refs = []
    for pb in base + [odd]:
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(5, True)
        refs.append((o.probability().copy(), o.map().copy()))
        o.close()
    for f in range(F):
        q, m = refs[4] if f == where else refs[f % 4]
        assert cc.same_bits(Q[f, :N], q) and np.array_equal(M[f, :N], m), f
    import torch
    ptr, words = b.device_label_bits()

    class _View:
        __cuda_array_interface__ = dict(shape=(F, words), typestr="<i8", data=(int(ptr), False), version=2)
    bits = torch.as_tensor(_View(), device="cuda:0").cpu().numpy().view(np.uint64)
    unpacked = ((bits[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)).reshape(F, -1)[:, :N]
    assert np.array_equal(unpacked.astype(np.int16), M[:, :N])
    # the outlier costs its OWN re-run -- one synchronisation + the ~90 small launches of the streaming build and
    # inference of one 3600-vertex frame, a constant ~0.35 ms whatever the batch size -- not a second pass over the batch
    # (round 2 re-ran all 1024 frames on the two-kernel path: build + inference of the whole batch again)
    slow, fast = min(ts["odd"], ts["odd2"]), min(ts["all"], ts["all2"])
    print("one outlier in %d frames: %.3f ms vs %.3f ms all-normal" % (F, slow * 1e3, fast * 1e3))
    assert slow - fast < 0.8e-3, (slow, fast)
    b.close(); ball.close()


def test_small_frames_share_a_cu_and_give_the_same_bits(po, wl):
    """Frames of up to 1024 points run as 512-lane workgroups in half the CU's LDS when a batch has at least 256 frames (two
    frames per CU: fused_loop.h kNTSmall) -- in the one-launch kernel and in the inference kernel on built lattices.  Ragged
    sizes around the 512-lane boundary, an empty frame, one sparse frame that does not fit the half-LDS plan (flagged and
    re-run alone); every frame against the oracle, and against the same frames in a batch too small for that shape."""
    sizes = [1024, 0, 1, 5, 511, 512, 513, 333, 64, 1000, 1023, 129]
    base = [wl.slam_problem(n, seed=2300 + i) for i, n in enumerate(sizes)]
    odd = _shaped_problem(wl, 900, "sparse", seed=5)
    F = 300
    pbs = [odd if f == 77 else base[f % len(base)] for f in range(F)]
    b = _batch_of(pbs, maxN=1024)
    b.run(5, True)
    q1, m1 = b.probability(), b.map()
    assert b.engine() == 3 and b.fallback_frames() == 1
    v1 = [b.lattice_sizes(k) for k in range(2)]
    b.build(); b.inference(5, True)
    q2, m2 = b.probability(), b.map()
    assert b.engine() in (1, 2)                            # (the sparse frame's lattices decide the engine for the batch)
    b.close()
    refs = {}
    for f in list(range(len(base))) + [77]:
        pb = pbs[f]
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(5, True)
        refs[f] = (o.probability().copy(), o.map().copy(), [o.kernel(k)["V"] if pb["N"] else 0 for k in range(2)])
        o.close()
    for f in range(F):
        q, m, V = refs[77] if f == 77 else refs[f % len(base)]
        n = pbs[f]["N"]
        assert cc.same_bits(q1[f, :n], q) and np.array_equal(m1[f, :n], m), f
        assert cc.same_bits(q2[f, :n], q) and np.array_equal(m2[f, :n], m), f
        assert [int(v1[k][f]) for k in range(2)] == V, f
    # without the outlier the built lattices fit the half-LDS plan: the inference kernel takes the 512-lane shape too
    b = _batch_of([base[f % len(base)] for f in range(F)], maxN=1024)
    b.build(); b.inference(5, True)
    assert b.engine() == 2
    q3 = b.probability()
    for f in range(F):
        n = base[f % len(base)]["N"]
        assert cc.same_bits(q3[f, :n], refs[f % len(base)][0]), f
    b.close()
    small = _batch_of(base, maxN=1024)                      # 12 frames: the 1024-lane shape
    small.run(5, True)
    for f in range(len(base)):
        assert cc.same_bits(small.probability()[f, :sizes[f]], refs[f][0])
    small.close()


def test_inference_after_run_uses_the_sized_engine(po, wl):
    """build -> inference -> run -> inference.  The second inference must run on the engine learn_sizes chose
    (the fused kernel here), not silently on the streaming engine because a one-launch run came in between."""
    pbs = [wl.slam_problem(1500, seed=77), wl.slam_problem(900, seed=78)]
    b = _batch_of(pbs)
    b.build(); b.inference(5, True)
    assert b.engine() == 2
    q0 = b.probability()
    b.run(5, True)
    assert b.engine() == 3 and cc.same_bits(b.probability(), q0)
    b.inference(5, True)
    assert b.engine() == 2 and cc.same_bits(b.probability(), q0)
    t = b.last_timing()["inference_ms"]
    b.set_engine(1); b.inference(5, True)
    assert b.engine() == 1 and cc.same_bits(b.probability(), q0)
    assert b.last_timing()["inference_ms"] > 2 * t          # (the streaming engine's ~9 launches per iteration show)
    b.close()


def test_frame_kernel_with_raw_unaries_and_unknown_labels(po, wl):
    rng = np.random.default_rng(11)
    pb = wl.slam_problem(1500, seed=21)
    pu = dict(pb)
    del pu["label"]
    pu["unary"] = rng.uniform(0.05, 3.0, (1500, 2)).astype(np.float32)
    pu["unary"][::50] = np.float32([0.0, 45.0])             # beyond fast_exp's cut-off
    b = _batch_of([pu], use_unary=True)
    _check_vs_oracle(po, [pu], b, 5)
    b.close()
    pl = dict(pb, label=pb["label"].copy())
    pl["label"][::7] = -1                                   # unknown labels: uniform energies (densecrf3d.h:119)
    b = _batch_of([pl])
    _check_vs_oracle(po, [pl], b, 5)
    b.close()


def test_frame_kernel_equals_two_kernel_path_bitwise(wl):
    """lccrf_batch_run (one launch) vs lccrf_batch_build + lccrf_batch_inference (build_small + fused)."""
    sizes = [2000, 1999, 1000, 2000, 0, 7, 3000, 2500]
    pbs = [wl.slam_problem(n, seed=60 + i) for i, n in enumerate(sizes)]
    b = _batch_of(pbs, maxN=3000)
    b.run(5, True)
    q1, m1 = b.probability(), b.map()
    assert b.engine() == 3
    v1 = [b.lattice_sizes(k) for k in range(2)]
    b.build()
    b.inference(5, True)
    q2, m2 = b.probability(), b.map()
    assert b.engine() == 2
    for f, n in enumerate(sizes):
        assert cc.same_bits(q1[f, :n], q2[f, :n]) and np.array_equal(m1[f, :n], m2[f, :n]), f
    for k in range(2):
        assert np.array_equal(v1[k], b.lattice_sizes(k))
    b.close()


def test_object_api_inference_is_one_launch_and_probes_still_work(po, wl):
    """DenseCRFHIP.inference() right after add_pairwise runs the frame kernel (nothing was built); the parity
    probes afterwards build the lattices on demand and agree with the oracle's numbering."""
    pb = wl.slam_problem(2000, seed=88)
    o, h = cc.setup(po.OracleCRF, pb), cc.setup(pkg.DenseCRFHIP, pb)
    o.inference_native(5, True)
    h.inference(5, True)
    assert cc.same_bits(h.probability(), o.probability()) and np.array_equal(h.map(), o.map())
    assert cc.same_bits(h.unary(), o.unary())
    for k in range(2):
        ko, kh = o.kernel(k), h.kernel(k)
        assert ko["V"] == kh["V"]
        for name in ("offset", "bary", "nbr", "norm"):
            assert cc.same_bits(ko[name], kh[name]), (k, name)
    h.inference(3, True)                                   # lattices are in HBM now: iterate on those
    o.inference_native(3, True)
    assert cc.same_bits(h.probability(), o.probability())
    h.step_inference()
    o.step_inference()
    assert cc.same_bits(h.probability(), o.probability())


def test_object_api_labels_are_taken_as_they_land(po, wl):
    """getMap() right behind inference(n, true) reads the labels out of pinned memory as the kernel's last stores arrive
    (no stream synchronisation); the handle goes back to the cache without one either.  A tracker's sequence: frames
    of changing size, one handle after the other, map first, probabilities only sometimes -- every frame against the oracle."""
    rng = np.random.default_rng(11)
    for i in range(40):
        N = int(rng.integers(1, 2600))
        pb = wl.slam_problem(N, seed=300 + i)
        o, h = cc.setup(po.OracleCRF, pb), cc.setup(pkg.DenseCRFHIP, pb)
        o.inference_native(5, True)
        h.inference(5, True)
        assert np.array_equal(h.map(), o.map()), (i, N)
        if i % 3 == 0:
            assert cc.same_bits(h.probability(), o.probability()), (i, N)
            assert np.array_equal(h.map(), o.map())            # (a second getMap: nothing in flight, the ordinary path)
        h.close()


def test_object_api_frame_that_does_not_fit_is_seen_through_the_label_array(po, wl):
    """A frame whose lattices do not fit the one-launch kernel writes -2 where its first label would go: getMap() leaves the
    early path and the two-kernel path re-runs the frame, same answers; the next frame on the cached handle is a normal one."""
    for pb in (_shaped_problem(wl, 1200, "sparse", seed=5), wl.slam_problem(1200, seed=9), _shaped_problem(wl, 900, "sparse", seed=6)):
        o, h = cc.setup(po.OracleCRF, pb), cc.setup(pkg.DenseCRFHIP, pb)
        o.inference_native(4, True)
        h.inference(4, True)
        assert np.array_equal(h.map(), o.map())
        assert cc.same_bits(h.probability(), o.probability())
        h.close()


def test_done_word_switch_gives_the_same_bits(wl):
    """LCCRF_NO_DONE_WORD (wait on the stream instead of the pinned words) in a child process."""
    import subprocess, sys, os, tempfile
    code = (
        "import importlib, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "pkg = importlib.import_module('lc-crf-slam_amd'); wl = importlib.import_module('lc-crf-slam_amd.workloads')\n"
        "out = []\n"
        "for i, N in enumerate((7, 500, 2000, 3001)):\n"
        "    pb = wl.slam_problem(N, seed=40 + i)\n"
        "    h = pkg.DenseCRFHIP(N, 2); h.set_unary_from_label(pb['label'], pb['conf'])\n"
        "    for f, w in pb['kernels']: h.add_pairwise(f, w)\n"
        "    h.inference(5, True); out.append(h.map().astype(np.float32)); out.append(h.probability().ravel()); h.close()\n"
        "np.save(sys.argv[1], np.concatenate(out))\n" % ROOT)
    res = []
    with tempfile.TemporaryDirectory() as td:
        for j, env in enumerate(({}, {"LCCRF_NO_DONE_WORD": "1"})):
            fn = os.path.join(td, "o%d.npy" % j)
            subprocess.run([sys.executable, "-c", code, fn], check=True, env=cc.switch_env(env), timeout=300)
            res.append(np.load(fn))
    assert cc.same_bits(res[0], res[1])


def test_no_frame_switch_gives_the_same_bits(wl):
    """LCCRF_NO_FRAME (two-kernel path for the object API) in a child process."""
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
out = []
for N in (5, 700, 2000, 3000):
    pb = wl.slam_problem(N, seed=123)
    h = pkg.DenseCRFHIP(pb["N"], pb["L"]); h.set_unary_from_label(pb["label"], pb["conf"])
    for f, w in pb["kernels"]: h.add_pairwise(f, w)
    h.inference(5, True); out.append(h.probability()); out.append(h.map().astype(np.float32)); h.close()
np.save(sys.argv[1], np.concatenate([o.ravel() for o in out]))
""" % ROOT
    res = []
    for env in ({}, {"LCCRF_NO_FRAME": "1"}):
        path = os.path.join(ROOT, "gpurun_out", "noframe_%d.npy" % len(res))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=cc.switch_env(env), timeout=600)
        res.append(np.load(path))
    assert cc.same_bits(res[0], res[1])


def test_two_workgroup_form_gives_the_same_bits(wl):
    """Single frames through the object API run the frame kernel as TWO workgroups (one per lattice build, hand-off through
    device memory: frame_engine.hip DUAL).  LCCRF_NO_DUAL=1 in a child process is the one-workgroup form: same bits for every
    points-per-lane shape, a one-kernel CRF (not split), a sparse frame (the helper's lattice does not fit beside the main
    workgroup's: fallback) and alternating sizes (the hand-off area is reused by every frame); batches of 3 and 64 frames
    (two workgroups per frame as well) and of 65 (one each)."""
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
from test_hip_parity import _shaped_problem
out = []
for rep in range(3):
    for N in (5, 700, 2000, 1024, 3000, 4096, 2047, 1):
        pb = wl.slam_problem(N, seed=321 + rep)
        h = pkg.DenseCRFHIP(pb["N"], pb["L"]); h.set_unary_from_label(pb["label"], pb["conf"])
        for f, w in pb["kernels"]: h.add_pairwise(f, w)
        h.inference(5, True); out.append(h.probability()); out.append(h.map().astype(np.float32)); h.close()
for pb in (_shaped_problem(wl, 1200, "sparse", seed=5), _shaped_problem(wl, 2000, "one_cell", seed=5)):
    h = pkg.DenseCRFHIP(pb["N"], pb["L"]); h.set_unary_from_label(pb["label"], pb["conf"])
    for f, w in pb["kernels"]: h.add_pairwise(f, w)
    h.inference(4, True); out.append(h.probability()); h.close()
    h = pkg.DenseCRFHIP(pb["N"], pb["L"]); h.set_unary_from_label(pb["label"], pb["conf"])
    h.add_pairwise(*pb["kernels"][1]); h.inference(3, True); out.append(h.probability()); h.close()
# small batches take the same form (up to 64 frames); 65 frames do not
from test_frame_engine import _batch_of
for F in (3, 64, 65):
    pbs = [wl.slam_problem(int(n), seed=900 + i) for i, n in enumerate(np.random.default_rng(F).integers(1, 2300, F))]
    if F == 3: pbs[1] = _shaped_problem(wl, 1200, "sparse", seed=7)      # one frame of the batch falls back
    b = _batch_of(pbs); b.run(5, True); out.append(b.probability()); out.append(b.map().astype(np.float32)); b.close()
np.save(sys.argv[1], np.concatenate([o.ravel() for o in out]))
""" % (ROOT, os.path.join(ROOT, "tests"))
    res = []
    for env in ({}, {"LCCRF_NO_DUAL": "1"}):
        path = os.path.join(ROOT, "gpurun_out", "dual_%d.npy" % len(res))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=cc.switch_env(env), timeout=600)
        res.append(np.load(path))
    assert cc.same_bits(res[0], res[1])


def test_a_helper_that_never_comes_cannot_hang_the_launch(po, wl):
    """LCCRF_DUAL_DROP_HELPER (child process, INSTRUMENTED library -- the release library carries no fault-injection hook): the helper
    workgroup of the two-workgroup form leaves at once.  The main workgroup's poll is bounded (~0.1 s), the frame flags itself and is
    re-run on the two-kernel path: same labels, same Q, no hang."""
    instr = os.path.join(ROOT, "lc-crf-slam_amd", "liblccrf_hip_instr.so")
    if not os.path.exists(instr):
        subprocess.run(["make", "-C", os.path.join(ROOT, "lc-crf-slam_amd"), "-j4", "INSTRUMENT=1"], check=True, stdout=subprocess.DEVNULL)
    code = r"""
import importlib, sys, time, numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
pb = wl.slam_problem(1500, seed=77)
out = []
t0 = time.perf_counter()
for rep in range(2):
    h = pkg.DenseCRFHIP(pb["N"], pb["L"]); h.set_unary_from_label(pb["label"], pb["conf"])
    for f, w in pb["kernels"]: h.add_pairwise(f, w)
    h.inference(5, True); out.append(h.map().astype(np.float32)); out.append(h.probability().ravel()); h.close()
np.save(sys.argv[1], np.concatenate(out + [np.float32([time.perf_counter() - t0])]))
""" % ROOT
    path = os.path.join(ROOT, "gpurun_out", "drop_helper.npy")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, LCCRF_DUAL_DROP_HELPER="1", LCCRF_LIB=instr), timeout=120)
    res = np.load(path)
    pb = wl.slam_problem(1500, seed=77)
    o = cc.setup(po.OracleCRF, pb)
    o.inference_native(5, True)
    want = np.concatenate([o.map().astype(np.float32), o.probability().ravel()])
    assert cc.same_bits(res[:want.size], want) and cc.same_bits(res[want.size:2 * want.size], want)
    assert res[-1] < 20.0                                   # two bounded waits + two re-runs + the child's start-up
    assert res[-1] > 0.05                                   # ... and the hook did fire (the helper was really dropped)
    # the release library neither reads the switch nor contains its name
    rel = open(os.path.join(ROOT, "lc-crf-slam_amd", "liblccrf_hip.so"), "rb").read()
    assert b"LCCRF_DUAL_DROP_HELPER" not in rel


def test_single_workgroup_option_gives_the_same_bits(po, wl):
    """lccrf_set_option(h, LCCRF_OPT_SINGLE_WORKGROUP, 1) / lccrf_batch_set_option / lccrf_set_default_option: a tracker on a shared
    GPU opts out of the two-workgroup form through the API (no environment variable); results are the oracle's either way, and a
    handle taken from the cache starts from the defaults again."""
    pb = wl.slam_problem(1800, seed=91)
    o = cc.setup(po.OracleCRF, pb)
    o.inference_native(5, True)
    for mode in ("default", "handle", "process"):
        if mode == "process":
            pkg.set_default_option(pkg.OPT_SINGLE_WORKGROUP, 1)
        try:
            h = pkg.DenseCRFHIP(pb["N"], pb["L"])
            if mode == "handle":
                h.set_option(pkg.OPT_SINGLE_WORKGROUP, 1)
            h.set_unary_from_label(pb["label"], pb["conf"])
            for f, w in pb["kernels"]:
                h.add_pairwise(f, w)
            h.inference(5, True)
            assert np.array_equal(h.map(), o.map()) and cc.same_bits(h.probability(), o.probability()), mode
            h.close()
        finally:
            pkg.set_default_option(pkg.OPT_SINGLE_WORKGROUP, 0)
    with pytest.raises(pkg.LccrfError):
        h = pkg.DenseCRFHIP(4, 2)
        try:
            h.set_option(99, 1)
        finally:
            h.close()
    # batches of up to 64 two-kernel frames take the two-workgroup form too
    F = 6
    pbs = [wl.slam_problem(1500 + 7 * i, seed=92 + i) for i in range(F)]
    res = []
    for single in (0, 1):
        b = pkg.BatchCRF(F, 1600, 2, [2, 2], [10.0, 30.0])
        b.set_option(pkg.OPT_SINGLE_WORKGROUP, single)
        feats = [np.zeros((F, 1600, 2), np.float32) for _ in range(2)]
        label = np.zeros((F, 1600), np.int16)
        for i, p in enumerate(pbs):
            for k in range(2):
                feats[k][i, :p["N"]] = p["kernels"][k][0]
            label[i, :p["N"]] = p["label"]
        b.set_inputs_host([p["N"] for p in pbs], feats, label=label, conf=0.7)
        b.run(5, True)
        res.append((b.probability(), b.map()))
        assert b.engine() == 3
        b.close()
    assert cc.same_bits(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    for i, p in enumerate(pbs):
        oo = cc.setup(po.OracleCRF, p)
        oo.inference_native(5, True)
        assert cc.same_bits(res[1][0][i, :p["N"]], oo.probability()) and np.array_equal(res[1][1][i, :p["N"]], oo.map()), i
        oo.close()


def test_frame_kernel_many_copies_are_identical(wl):
    """Race hunt: 1536 frames in flight (6 per CU), three distinct frames tiled, three runs -- every copy must equal its
    original bit for bit, every time (workgroups of different frames share nothing but the kernel's code)."""
    F, N = 1536, 2000
    base = [wl.slam_problem(N, seed=1200 + i) for i in range(3)]
    feats = [np.stack([base[f % 3]["kernels"][k][0] for f in range(F)]) for k in range(2)]
    label = np.stack([base[f % 3]["label"] for f in range(F)])
    b = pkg.BatchCRF(F, N, 2, [2, 2], [10.0, 30.0])
    b.set_inputs_host([N] * F, feats, label=label, conf=0.7)
    ref = None
    for _ in range(3):
        b.run(5, True)
        Q, M = b.probability(), b.map()
        assert b.engine() == 3
        for f in range(3, F):
            assert cc.same_bits(Q[f], Q[f % 3]) and np.array_equal(M[f], M[f % 3]), f
        if ref is None:
            ref = (Q[:3].copy(), M[:3].copy())
        else:
            assert cc.same_bits(Q[:3], ref[0]) and np.array_equal(M[:3], ref[1])
    b.close()


@pytest.mark.parametrize("N", [2000, 4096])
def test_two_label_softmax_dense_sweep(po, wl, N):
    """The loop's two-label softmax (one polynomial exp, range reduction by compares + ldexp, two divisions sharing a
    refined reciprocal -- csrc/device_math.h: softmax2, fast_exp_nonpos) against the restatement of expAndNormalize
    (densecrf3d.h:51-98) on ~10^6 energy pairs: every fast_exp threshold (0.69 * 2^k) and the cut-off at 20 to within
    +-8 ulp, denormal and zero differences, log-uniform differences up to 25, both orders, random common offsets.
    A pairwise kernel of weight 0 makes next = -unary exactly, so Q after any number of iterations is that softmax --
    through the one-launch kernel and through the inference kernel on built lattices."""
    rng = np.random.default_rng(N)
    F = 256
    total = F * N
    special = []
    for t in [0.69 * 2 ** k for k in range(0, 6)] + [20.0, 1.0, 2.0, 0.5]:
        c = np.float32(t)
        ulps = np.arange(-8, 9)
        special.append((c.view(np.int32) + ulps).astype(np.int32).view(np.float32))
        special.append((np.float32(np.float64(t)).view(np.int32) + ulps + 1).astype(np.int32).view(np.float32))
    special.append(np.float32([0.0, 1e-45, 1e-40, 1e-38, 1.1754944e-38, 1e-30, 1e-20, 19.999999, 20.000002, 21.0, 25.0, 80.0]))
    special = np.concatenate(special).astype(np.float32)
    d = np.exp(rng.uniform(np.log(1e-8), np.log(25.0), total)).astype(np.float32)
    d[:special.size] = special
    d[special.size:2 * special.size] = special
    base = rng.uniform(0.0, 3.0, total).astype(np.float32)
    base[:special.size] = 0.0                               # the differences themselves, exactly
    flip = rng.integers(0, 2, total).astype(bool)
    flip[:special.size] = False
    flip[special.size:2 * special.size] = True
    un = np.empty((total, 2), np.float32)
    un[:, 0] = np.where(flip, base + d, base)
    un[:, 1] = np.where(flip, base, base + d)
    expect = np.zeros_like(un)
    po.oracle_lib().orc_exp_and_normalize(expect.ctypes.data_as(po._f32p), un.ctypes.data_as(po._f32p), total, 2, -1.0, 1.0)
    feats = rng.uniform(0.0, 30.0, (F, N, 2)).astype(np.float32)
    b = pkg.BatchCRF(F, N, 2, [2], [0.0])
    b.set_inputs_host([N] * F, [feats], unary=un.reshape(F, N, 2))
    b.run(2, True)
    assert b.engine() == 3
    assert cc.same_bits(b.probability().reshape(total, 2), expect)
    assert np.array_equal(b.map().reshape(total), (expect[:, 1] > expect[:, 0]).astype(np.int16))
    b.build()
    b.inference(3, True)
    assert b.engine() == 2
    assert cc.same_bits(b.probability().reshape(total, 2), expect)
    b.close()


@pytest.mark.parametrize("shape", ["", "2", "2-noprep"])
def test_full_size_frames_share_a_cu_and_give_the_same_bits(po, wl, shape):
    """"2-noprep" = the self-contained kernel (no prepared launch records, round 6: LCCRF_NO_LEAN_PREP) in the instrumented twin;
    shape "" = the RELEASE library (liblccrf_hip.so, environment untouched: the default shape is the lean one), "2" = the same
    shape selected by switch in the instrumented twin (the sweep has to hit the shipped object too).
    Round 5: frames of 1025 .. ~2300 points run TWO per CU as well when a batch has at least 256 frames -- the lean plan of
    csrc/fused_lean.h (one shared product buffer, the large lattice's neighbour table read from HBM/L2, chain rows placed by a scan;
    384 lanes x 6 points or 512 lanes x 4 points with the weights re-read per iteration).  Ragged sizes around every
    points-per-lane boundary of both shapes, an empty and a tiny frame among them; every frame against the oracle, bit for bit."""
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import crf_cases as cc, pyoracle as po
from test_frame_engine import _batch_of
wl = importlib.import_module("lc-crf-slam_amd.workloads")
shape = %r
noprep = shape == "2-noprep"
lanes = 512 if shape in ("", "2", "2-noprep") else 384
top = 2048 if lanes == 512 else 2304
sizes = [2000, 1025, 1536, 1537, 1999, top, 1100, 1920, 1921, top - 1, 0, 700, 1152, 1153, 3, 2001]
base = [wl.slam_problem(n, seed=5100 + i) for i, n in enumerate(sizes)]
F = 272
pbs = [base[f %% len(base)] for f in range(F)]
refs = []
for n_iter, relax in ((5, 1.0), (3, 0.5)):
    b = _batch_of(pbs, maxN=top)
    b.build(); b.inference(n_iter, True, relax=relax)
    Q, M = b.probability(), b.map()
    assert b.engine() == 2 and b.fused_shape() == (lanes, 2), (b.engine(), b.fused_shape())
    # the second inference on the same lattices writes the prepared launch records and runs from them, the third reuses them (round 6)
    for again in range(2):
        b.inference(n_iter, True, relax=relax)
        assert cc.same_bits(b.probability(), Q) and np.array_equal(b.map(), M), ("prepared", n_iter, again)
    assert b.last_prepare()[1] == (0 if noprep else 1)
    b.close()
    for i, pb in enumerate(base):
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(n_iter, True, relax)
        for f in range(i, F, len(base)):
            n = pb["N"]
            assert cc.same_bits(Q[f, :n], o.probability()), (n_iter, f, n)
            assert np.array_equal(M[f, :n], o.map()), (n_iter, f, n)
        o.close()
# frames of 513 .. 1024 points take the plan too (two points per lane, everything in registers)
small = [wl.slam_problem(n, seed=5300 + i) for i, n in enumerate([1024, 513, 700, 1000, 0, 640, 1023, 514])]
pbs2 = [small[f %% len(small)] for f in range(F)]
b = _batch_of(pbs2, maxN=1024)
b.build(); b.inference(5, True)
Q, M = b.probability(), b.map()
assert b.engine() == 2 and b.fused_shape() == (512, 2), (b.engine(), b.fused_shape())
for again in range(2):
    b.inference(5, True)
    assert cc.same_bits(b.probability(), Q) and np.array_equal(b.map(), M), ("prepared small", again)
b.close()
for i, pb in enumerate(small):
    o = cc.setup(po.OracleCRF, pb)
    o.inference_native(5, True)
    for f in range(i, F, len(small)):
        assert cc.same_bits(Q[f, :pb["N"]], o.probability()) and np.array_equal(M[f, :pb["N"]], o.map()), ("small", f)
    o.close()
# one kernel only (the chain kernel alone, then the short-row kernel alone), and two short-row kernels
for pick in ((0,), (1,), (1, 1)):
    pb1 = []
    for pb in base:
        q = dict(pb); q["kernels"] = [pb["kernels"][k] for k in pick]; pb1.append(q)
    pbs1 = [pb1[f %% len(pb1)] for f in range(F)]
    b = _batch_of(pbs1, maxN=top)
    b.build(); b.inference(4, True)
    Q = b.probability()
    for again in range(2):
        b.inference(4, True)
        assert cc.same_bits(b.probability(), Q), ("prepared", pick, again)
    # (two large lattices do not fit half a CU's LDS: that batch keeps the 1024-lane shape)
    assert b.engine() == 2 and b.fused_shape() == ((lanes, 2) if len(pick) == 1 else (1024, 1)), (pick, b.engine(), b.fused_shape())
    b.close()
    for i, pb in enumerate(pb1):
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(4, True)
        assert cc.same_bits(Q[i, :pb["N"]], o.probability()) and cc.same_bits(Q[i + 256, :pb["N"]], o.probability()), (pick, i)
        o.close()
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), shape)
    env = dict(os.environ)
    if shape == "2-noprep":
        env = cc.switch_env(LCCRF_LEAN_SHAPE="2", LCCRF_NO_LEAN_PREP="1")
    elif shape:
        env = cc.switch_env(LCCRF_LEAN_SHAPE=shape)
    if not shape:
        if "LCCRF_LIB" in env:
            pytest.skip("the suite is running on another library (LCCRF_LIB): this case is the RELEASE library's")
        assert "LCCRF_LEAN_SHAPE" not in env
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


def test_prepared_launch_records_follow_the_lattices_and_not_the_unaries(po, wl):
    """Round 6 (fused_lean.h: LeanPrepPlan): the two-frames-per-CU inference kernel starts from per-frame blocks the FIRST inference
    behind a build writes (ranking / placement of the chain rows, vertex addresses, product slots, the LDS tables) -- the SECOND one, so
    that one inference per lattice pays nothing.  They must be rewritten when -- and only when -- a lattice changes."""
    F, maxN = 264, 2048
    def batch(seed0):
        sizes = [2000, 1300, 1999, 1100, 2048, 1537, 3, 0]
        return [wl.slam_problem(sizes[f % 8], seed=seed0 + f % 8) for f in range(F)]

    def check(b, pbs, n_iter, tag, conf=None):
        Q, M = b.probability(), b.map()
        for f in list(range(8)) + [F - 8 + i for i in range(8)]:
            pb = dict(pbs[f])
            if conf is not None:
                pb["conf"] = conf
            o = cc.setup(po.OracleCRF, pb)
            o.inference_native(n_iter, True)
            assert cc.same_bits(Q[f, :pb["N"]], o.probability()), (tag, f)
            assert np.array_equal(M[f, :pb["N"]], o.map()), (tag, f)
            o.close()

    pbs = batch(9100)
    b = _batch_of(pbs, maxN=maxN)
    assert b.last_prepare()[1] == 0
    b.build()
    b.inference(5, True)                                     # the first inference behind a build: the self-contained kernel
    assert b.engine() == 2 and b.fused_shape() == (512, 2)
    assert b.last_prepare() == (0.0, 0)
    check(b, pbs, 5, "first")
    b.inference(3, True)                                     # the second one writes the blocks and runs from them
    ms, runs = b.last_prepare()
    assert runs == 1 and ms > 0
    check(b, pbs, 3, "second")
    b.inference(5, True)                                     # ... and the third reuses them
    assert b.last_prepare()[1] == 1
    check(b, pbs, 5, "third")
    # other unaries (another confidence): new inputs through the host path invalidate the lattices -- build again; the blocks follow
    feats = [np.stack([np.pad(pb["kernels"][k][0], ((0, maxN - pb["N"]), (0, 0))) for pb in pbs]) for k in range(2)]
    label = np.stack([np.pad(pb["label"], (0, maxN - pb["N"]), constant_values=-1) for pb in pbs]).astype(np.int16)
    b.set_inputs_host([pb["N"] for pb in pbs], feats, label=label, conf=0.9)
    b.build()
    b.inference(5, True)
    assert b.last_prepare()[1] == 1                          # (first inference on the new lattices)
    check(b, pbs, 5, "conf 0.9, first", conf=np.float32(0.9))
    b.inference(5, True)
    assert b.last_prepare()[1] == 2
    check(b, pbs, 5, "conf 0.9, second", conf=np.float32(0.9))
    # other frames: the lattices change, the blocks are rewritten
    pbs2 = batch(9200)
    feats2 = [np.stack([np.pad(pb["kernels"][k][0], ((0, maxN - pb["N"]), (0, 0))) for pb in pbs2]) for k in range(2)]
    label2 = np.stack([np.pad(pb["label"], (0, maxN - pb["N"]), constant_values=-1) for pb in pbs2]).astype(np.int16)
    b.set_inputs_host([pb["N"] for pb in pbs2], feats2, label=label2, conf=pbs2[0]["conf"])
    b.build()
    for rep in range(3):
        b.inference(5, True)
        assert b.last_prepare()[1] == (2 if rep == 0 else 3)
        check(b, pbs2, 5, "other frames, inference %d" % rep)
    b.close()


@pytest.mark.parametrize("case", ["c4", "n500", "sparse", "k1_chain", "k1_short", "short_rows", "few_frames"])
def test_repeated_inference_runs_from_prepared_records_in_every_fused_shape(po, wl, case):
    """Round 6: k_fused (1024 lanes, one frame per CU: 3000 keypoints; 512 lanes: frames of up to 512 points) takes the prepared launch
    records as well.  Inference three times on the same lattices -- self-contained, prepare + run, run -- gives the same bits, and
    they are the oracle's; a handful of frames (< 64) never prepares."""
    F = 72
    if case == "c4":
        sizes, top, shape = [3000, 2500, 2049, 2999, 3, 0, 2700, 3072], 3072, (1024, 1)
        base = [wl.slam_problem(n, seed=9400 + i) for i, n in enumerate(sizes)]
    elif case == "n500":
        sizes, top, shape, F = [500, 512, 1, 0, 333, 257, 400, 64], 512, (512, 2), 264
        base = [wl.slam_problem(n, seed=9500 + i) for i, n in enumerate(sizes)]
    elif case == "sparse":                                   # large lattices: the tables take two 16-byte pieces per lane
        top, shape = 850, (1024, 1)                          # (V = 3 N for the first kernel: ~30 KB of neighbour words at 820 points)
        base = [_shaped_problem(wl, n, "sparse", seed=20 + i) for i, n in enumerate([700, 760, 820, 640])] + [wl.slam_problem(850, seed=9600)]
    elif case in ("k1_chain", "k1_short"):
        top, shape = 3000, (1024, 1)
        base = []
        for i, n in enumerate([3000, 2600, 2100]):
            q = dict(wl.slam_problem(n, seed=9700 + i)); q["kernels"] = [q["kernels"][0 if case == "k1_chain" else 1]]; base.append(q)
    elif case == "short_rows":                               # no chain kernel: a frame whose appearance features spread out
        top, shape = 2500, (1024, 1)
        base = []
        for i, n in enumerate([2500, 2100, 2300]):
            q = dict(wl.slam_problem(n, seed=9800 + i)); q["kernels"] = [q["kernels"][1], q["kernels"][1]]; base.append(q)
    else:
        sizes, top, shape, F = [3000, 2500, 2800], 3000, (1024, 1), 8
        base = [wl.slam_problem(n, seed=9900 + i) for i, n in enumerate(sizes)]
    pbs = [base[f % len(base)] for f in range(F)]
    b = _batch_of(pbs, maxN=top)
    b.build()
    b.inference(5, True)
    assert b.engine() == 2 and b.fused_shape() == shape, (case, b.engine(), b.fused_shape())
    Q, M = b.probability().copy(), b.map().copy()
    for again in range(2):
        b.inference(5, True)
        assert cc.same_bits(b.probability(), Q) and np.array_equal(b.map(), M), (case, again)
    assert b.last_prepare()[1] == (0 if case == "few_frames" else 1), case
    b.close()
    for i, pb in enumerate(base):
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(5, True)
        for f in (i, i + len(base) * ((F - 1 - i) // len(base))):
            assert cc.same_bits(Q[f, :pb["N"]], o.probability()), (case, f)
            assert np.array_equal(M[f, :pb["N"]], o.map()), (case, f)
        o.close()


def test_event_timing_option_only_switches_the_events_off(po, wl):
    """LCCRF_OPT_EVENT_TIMING = 0: no HIP events around build / inference / run (a packet less between two launches), timings read 0,
    results unchanged; 1 brings them back."""
    pbs = [wl.slam_problem(700, seed=9950 + i) for i in range(4)]
    b = _batch_of(pbs)
    b.build(); b.inference(5, True)
    Q = b.probability().copy()
    assert b.last_timing()["inference_ms"] > 0 and b.last_timing()["build_ms"] > 0
    b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 0)
    b.build(); b.inference(5, True)
    assert b.last_timing() == dict(inference_ms=0.0, build_ms=0.0)
    assert cc.same_bits(b.probability(), Q)
    b.run(5, True)
    assert b.last_timing()["inference_ms"] == 0.0 and cc.same_bits(b.probability(), Q)
    b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 1)
    b.run(5, True)
    assert b.last_timing()["inference_ms"] > 0 and cc.same_bits(b.probability(), Q)
    b.close()


def test_two_handles_on_their_own_streams_give_each_its_results(po, wl):
    """Round 6: the frames in flight held by two handles whose launches alternate on the handles' OWN streams (`lccrf_batch_get_stream`;
    what `bench.py`'s `two_handles` record and a replay loop do): nothing is ordered between the handles, every handle must still hold
    its own frames' results -- against the oracle, after many interleaved launches, through inference and through the one-launch path."""
    F, maxN = 264, 2048
    sets = [[wl.slam_problem(n, seed=9960 + 10 * h + i) for i, n in enumerate([2000, 1300, 1999, 1100, 2048, 1537, 3, 0])] for h in range(2)]
    bs = [_batch_of([sets[h][f % 8] for f in range(F)], maxN=maxN) for h in range(2)]
    streams = [b.own_stream() for b in bs]
    assert all(streams) and streams[0] != streams[1]
    for b in bs:
        b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 0)
        b.build()
    for rep in range(6):
        for b in bs:
            b.inference(5, True)
    for h, b in enumerate(bs):
        b.synchronize()
        assert b.engine() == 2 and b.fused_shape() == (512, 2) and b.last_prepare()[1] == 1
        Q, M = b.probability(), b.map()
        for i, pb in enumerate(sets[h]):
            o = cc.setup(po.OracleCRF, pb)
            o.inference_native(5, True)
            for f in (i, F - 8 + i):
                assert cc.same_bits(Q[f, :pb["N"]], o.probability()) and np.array_equal(M[f, :pb["N"]], o.map()), (h, f)
            o.close()
    for rep in range(3):
        for b in bs:
            b.run(5, True)
    for h, b in enumerate(bs):
        Q = b.probability()
        o = cc.setup(po.OracleCRF, sets[h][0])
        o.inference_native(5, True)
        assert b.engine() == 3 and cc.same_bits(Q[0, :2000], o.probability()) and cc.same_bits(Q[F - 8, :2000], o.probability()), h
        o.close()
        b.close()


def test_one_launch_record_area_grows_with_the_frames_bound(po, wl):
    """The half-CU one-launch kernel's record area (96 KB per frame) is sized for the frames BOUND, not for the handle's capacity
    (ADVICE r5): a handle for 1024 frames runs 264, then 520 (a larger area), then 264 again -- each batch against the oracle."""
    base = [wl.slam_problem(n, seed=9980 + i) for i, n in enumerate([2000, 1300, 1999, 1100, 2048, 1537, 3, 0])]
    refs = []
    for pb in base:
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(5, True)
        refs.append((o.probability().copy(), o.map().copy()))
        o.close()
    maxN = 2048
    b = pkg.BatchCRF(1024, maxN, 2, [2, 2], [float(base[0]["kernels"][k][1]) for k in range(2)])
    for F in (264, 520, 264):
        feats = [np.zeros((F, maxN, 2), np.float32) for _ in range(2)]
        label = np.full((F, maxN), -1, np.int16)
        for f in range(F):
            pb = base[f % 8]
            label[f, :pb["N"]] = pb["label"]
            for k in range(2):
                feats[k][f, :pb["N"]] = pb["kernels"][k][0]
        b.set_inputs_host([base[f % 8]["N"] for f in range(F)], feats, label=label, conf=base[0]["conf"])
        b.run(5, True)
        Q, M = b.probability(), b.map()
        assert b.engine() == 3 and b.fused_shape() == (512, 2), (F, b.engine(), b.fused_shape())
        for f in list(range(8)) + list(range(F - 8, F)):
            q, m = refs[f % 8]
            n = base[f % 8]["N"]
            assert cc.same_bits(Q[f, :n], q) and np.array_equal(M[f, :n], m), (F, f)
    b.close()


def test_full_size_frames_share_a_cu_in_the_one_launch_kernel(po, wl):
    """Round 5: lccrf_batch_run on batches of >= 256 full-size two-kernel frames (1025 .. 2048 points) runs the WHOLE frame -- both
    lattice builds, normalisation, inference -- in 512-lane workgroups on half a CU (csrc/frame_lean.hip: LDS scratch laid out by
    lifetime, the per-point records through HBM/L2, fused_lean.h's loop).  Ragged sizes around the points-per-lane boundaries, an
    empty and a tiny frame, one sparse frame that does not fit (flagged, re-run alone), labels and raw unaries, relax != 1;
    every frame against the oracle bit for bit, V against the reference's M_; then build + inference on the same handle."""
    sizes = [2000, 1025, 1536, 1537, 1999, 2048, 1100, 1920, 1921, 2047, 0, 700, 1152, 3, 2001, 1300]
    base = [wl.slam_problem(n, seed=6100 + i) for i, n in enumerate(sizes)]
    odd = _shaped_problem(wl, 1800, "sparse", seed=7)
    F = 272
    refs = {}
    rng = np.random.default_rng(12)
    for n_iter, relax, use_unary in ((5, 1.0, False), (3, 0.5, True)):
        if use_unary:                                      # raw energies instead of labels, some beyond fast_exp's cut-off
            def raw(pb):
                pu = {k: v for k, v in pb.items() if k != "label"}
                pu["unary"] = rng.uniform(0.05, 3.0, (pb["N"], 2)).astype(np.float32)
                pu["unary"][::50] = np.float32([0.0, 45.0])
                return pu
            base, odd = [raw(pb) for pb in base], raw(odd)
        pbs = [odd if f == 99 else base[f % len(base)] for f in range(F)]
        b = _batch_of(pbs, maxN=2048, use_unary=use_unary)
        b.run(n_iter, True, relax=relax)
        Q, M = b.probability(), b.map()
        assert b.engine() == 3 and b.fused_shape() == (512, 2) and b.fallback_frames() == 1, (b.engine(), b.fused_shape(), b.fallback_frames())
        Vs = [b.lattice_sizes(k) for k in range(2)]
        for i, pb in enumerate(base + [odd]):
            o = cc.setup(po.OracleCRF, pb)
            o.inference_native(n_iter, True, relax)
            n = pb["N"]
            for f in ([99] if i == len(base) else [f for f in range(i, F, len(base)) if f != 99]):
                assert cc.same_bits(Q[f, :n], o.probability()), (n_iter, f, n)
                assert np.array_equal(M[f, :n], o.map()), (n_iter, f, n)
                assert [int(Vs[k][f]) for k in range(2)] == [o.kernel(k)["V"] if n else 0 for k in range(2)], (f, n)
            if (n_iter, i) == (5, 0):
                refs["q0"] = o.probability().copy()
            o.close()
        if n_iter == 5:
            # the kernel kept its records in the batch's lattice arrays: a build + inference on the same handle starts over
            b.build(); b.inference(5, True)
            assert b.engine() in (1, 2) and cc.same_bits(b.probability()[0, :sizes[0]], refs["q0"])
            b.run(5, True)                                 # ... and lccrf_batch_inference after ANOTHER run builds again by itself
            assert b.engine() == 3 and cc.same_bits(b.probability(), Q)
            b.inference(5, True)
            assert b.engine() in (1, 2) and cc.same_bits(b.probability()[0, :sizes[0]], refs["q0"])
        b.close()
    # a batch whose frames mostly do NOT fit: the engine stops asking for the shape
    pbs = [odd if f % 2 else base[0] for f in range(F)]
    b = _batch_of(pbs, maxN=2048, use_unary=True)
    b.run(2, True)
    q = b.probability()
    assert b.fallback_frames() == F // 2 and b.fused_shape() == (512, 2)
    b.run(2, True)
    assert b.fused_shape() == (1024, 1) and cc.same_bits(b.probability(), q)
    b.close()


def test_frame_lean_switch_gives_the_same_bits(wl):
    """LCCRF_NO_FRAME_LEAN (instrumented library): the same batch through one 1024-lane workgroup per frame -- the same bits."""
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import crf_cases as cc
from test_frame_engine import _batch_of
wl = importlib.import_module("lc-crf-slam_amd.workloads")
pbs = [wl.slam_problem(n, seed=6200 + i) for i, n in enumerate([2000, 1500, 1900, 1026] * 64)]
b = _batch_of(pbs, maxN=2000)
b.run(5, True)
np.save(sys.argv[1], np.concatenate([b.probability().ravel().view(np.uint32), np.asarray(b.fused_shape(), np.uint32)]))
""" % (ROOT, os.path.join(ROOT, "tests"))
    outs = []
    for env_extra in ({}, {"LCCRF_NO_FRAME_LEAN": "1"}):
        path = os.path.join("/tmp", "lean_switch_%d_%d.npy" % (os.getpid(), len(outs)))
        r = subprocess.run([sys.executable, "-c", code, path], env=cc.switch_env(**env_extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(np.load(path))
        os.remove(path)
    assert tuple(outs[0][-2:]) == (512, 2) and tuple(outs[1][-2:]) == (1024, 1)
    assert np.array_equal(outs[0][:-2], outs[1][:-2])


def _host_bits(b):
    b.download_async(pkg.BatchCRF.DOWNLOAD_LABEL_BITS)
    return b.wait_download()["bits"]


def test_half_cu_frame_kernel_corner_calls(po, wl):
    """frame_lean.hip at the edges of the call: zero iterations (Q = softmax(-unary), densecrf_base.h:78-80), no MAP, every points-per-lane
    shape (1 .. 4), a handle reused with SMALLER frames (the label bits beyond a frame's points must read 0, and nothing of the previous
    batch may leak through the kernel's record area), labels with unknown entries -- against the oracle, bit for bit."""
    F = 256
    for top in (400, 900, 1400, 2048):                     # 1, 2, 3, 4 points per lane
        base = [wl.slam_problem(max(top - 37 * i, 1), seed=6400 + i) for i in range(5)]
        for pb in base:
            pb["label"] = pb["label"].copy()
            pb["label"][::9] = -1
        pbs = [base[f % 5] for f in range(F)]
        b = _batch_of(pbs, maxN=2048)
        refs = []
        for pb in pbs[:5]:
            o = cc.setup(po.OracleCRF, pb)
            o.inference_native(0, True)
            q0 = o.probability().copy()
            o.inference_native(4, True)
            refs.append((q0, o.probability().copy(), o.map().copy()))
            o.close()
        b.run(0, True)
        assert b.engine() == 3 and b.fused_shape() == (512, 2)
        Q = b.probability()
        for f in range(F):
            assert cc.same_bits(Q[f, :pbs[f]["N"]], refs[f % 5][0]), (top, f)
        b.run(4, False)                                    # no MAP asked for: Q all the same
        Q = b.probability()
        for f in range(F):
            assert cc.same_bits(Q[f, :pbs[f]["N"]], refs[f % 5][1]), (top, f)
        b.run(4, True)
        M, bits = b.map(), _host_bits(b)
        for f in range(0, F, 17):
            n = pbs[f]["N"]
            assert np.array_equal(M[f, :n], refs[f % 5][2])
            want = np.zeros(bits.shape[1] * 64, np.uint8)
            want[:n] = refs[f % 5][2] == 1
            assert np.array_equal(np.unpackbits(bits[f].view(np.uint8), bitorder="little"), want), (top, f)
        b.close()
    # one handle, a batch of large frames, then a batch of small ones
    big = [wl.slam_problem(2000, seed=6500 + f % 3) for f in range(F)]
    small = [wl.slam_problem(300 + f % 3, seed=6600 + f % 3) for f in range(F)]
    b = _batch_of(big, maxN=2048)
    b.run(3, True)
    feats = [np.zeros((F, 2048, 2), np.float32) for _ in range(2)]
    label = np.full((F, 2048), -1, np.int16)
    for f, pb in enumerate(small):
        label[f, :pb["N"]] = pb["label"]
        for k in range(2):
            feats[k][f, :pb["N"]] = pb["kernels"][k][0]
    b.set_inputs_host([pb["N"] for pb in small], feats, label=label, conf=small[0].get("conf", 0.7))
    b.run(3, True)
    Q, bits = b.probability(), _host_bits(b)
    for i in range(3):
        o = cc.setup(po.OracleCRF, small[i])
        o.inference_native(3, True)
        n = small[i]["N"]
        for f in range(i, F, 3):
            assert cc.same_bits(Q[f, :n], o.probability()), f
            want = np.zeros(bits.shape[1] * 64, np.uint8)
            want[:n] = o.map() == 1
            assert np.array_equal(np.unpackbits(bits[f].view(np.uint8), bitorder="little"), want), f
        o.close()
    b.close()
