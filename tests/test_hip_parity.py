"""Parity tests proper: the HIP path, called through the C-ABI, against
  (1) the committed golden vectors generated from the reference itself,
  (2) the oracle (oracle/lccrf_oracle.c, proven bit-identical to the reference) on fresh
      seeded inputs, including the lattice internals,
  (3) the reference's own known-answer image (res1_cpu.ppm).

Bar (BASELINE.json north_star): labels identical; mean-field Q within 1e-5.  The path is
built to do better -- ordered splat, no FMA, same rounding -- so these tests demand
BIT-IDENTICAL Q and normalisation, and identical lattice numbering.
"""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import crf_cases as cc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("lc-crf-slam_amd")
pytestmark = pytest.mark.gpu

Q_TOL = 1e-5   # the stated tolerance; used only where a test says so


def _cases(name):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    return [str(c) for c in z["cases"]]


@pytest.mark.parametrize("case", _cases("slam"))
def test_hip_slam_fixture(golden, case):
    z = golden["slam"]
    c = cc.setup(pkg.DenseCRFHIP, cc.case_problem(z, case))
    cc.check_against_expected(c, cc.case_expected(z, case))


@pytest.mark.parametrize("case", _cases("generic"))
def test_hip_generic_fixture(golden, case):
    z = golden["generic"]
    c = cc.setup(pkg.DenseCRFHIP, cc.case_problem(z, case))
    cc.check_against_expected(c, cc.case_expected(z, case))


def test_hip_bilateral_fixture(golden):
    z = golden["bilateral"]
    c = cc.setup(pkg.DenseCRFHIP, cc.case_problem(z, "c5"))
    cc.check_against_expected(c, cc.case_expected(z, "c5"))


def test_hip_reproduces_reference_known_answer_image(po, golden):
    z = golden["example_im1"]
    im, res, lab, colors = z["im"], z["res"], z["label"], z["colors"]
    H, W, _ = im.shape
    c = pkg.DenseCRFHIP(W * H, 21)
    c.set_unary_from_label(lab, 0.5)
    c.add_pairwise(po.oracle_image_features(W, H, 3.0), 3.0)
    c.add_pairwise(po.oracle_image_features(W, H, 60.0, im, 20.0), 10.0)
    c.inference(10, True)
    col = colors[c.map()]
    out = np.stack([col & 255, (col >> 8) & 255, (col >> 16) & 255], -1).astype(np.uint8)
    assert np.array_equal(out.reshape(H, W, 3), res)


@pytest.mark.parametrize("N", [0, 1, 2, 3, 4, 5, 63, 64, 65, 511, 1000, 2000, 3000])
def test_hip_matches_oracle_slam_sizes(po, wl, N):
    pb = wl.slam_problem(N, seed=21)
    o, h = cc.setup(po.OracleCRF, pb), cc.setup(pkg.DenseCRFHIP, pb)
    assert cc.same_bits(o.unary(), h.unary())
    for k in range(2):
        ko, kh = o.kernel(k), h.kernel(k)
        assert ko["V"] == kh["V"]
        for name in ("offset", "bary", "nbr", "norm"):
            assert cc.same_bits(ko[name], kh[name]), (k, name)
    assert cc.same_bits(o.run_trace(5), h.run_trace(5))
    o.build_map(), h.build_map()
    assert np.array_equal(o.map(), h.map())


def _shaped_problem(wl, N, shape, seed):
    """SLAM-shaped frames whose lattices stress one code path of the fused engine each."""
    pb = wl.slam_problem(N, seed=seed)
    rng = np.random.default_rng(seed)
    f0, f1 = pb["kernels"][0][0].copy(), pb["kernels"][1][0].copy()
    if shape == "one_cell":            # every point in one lattice cell: 3 vertices, rows of N products
        f0[:] = f0[0]
        f1[:] = f1[0]
    elif shape == "two_clusters":      # two very long rows per kernel plus stragglers
        half = N // 2
        f0[:half], f0[half:] = f0[0], f0[-1] + np.float32(7.5)
        f0[::97] += rng.normal(0, 3, f0[::97].shape).astype(np.float32)
    elif shape == "rows_of_8":         # row lengths around the 8-product units of chain_rows
        cells = max(N // 8, 1)
        f0 = (np.stack([np.arange(N) % cells, np.arange(N) % cells], 1) * np.float32(4.0)).astype(np.float32)
    elif shape == "sparse":            # every point its own cell: V = 3N, far beyond the chain's vertex limit
        f0 = (np.stack([np.arange(N), (np.arange(N) * 7) % 1013], 1) * np.float32(9.0)).astype(np.float32)
    pb["kernels"] = [(f0, pb["kernels"][0][1]), (f1, pb["kernels"][1][1])]
    return pb


@pytest.mark.parametrize("shape,N", [("one_cell", 2000), ("one_cell", 4096), ("one_cell", 70), ("two_clusters", 2047),
                                     ("two_clusters", 2600), ("rows_of_8", 2000), ("rows_of_8", 1030),
                                     ("sparse", 1200)])
def test_fused_engine_on_adversarial_lattices(po, wl, shape, N):
    """Very long rows (chain path, also with the shared product buffer and 3-4 points per lane), row
    lengths on the chain's unit boundaries, and lattices too large for the chain path: the engine
    the library picks must still reproduce the oracle bit for bit."""
    pb = _shaped_problem(wl, N, shape, seed=5)
    o, h = cc.setup(po.OracleCRF, pb), cc.setup(pkg.DenseCRFHIP, pb)
    o.inference_native(4, True)
    h.inference(4, True)
    assert cc.same_bits(o.probability(), h.probability())
    assert np.array_equal(o.map(), h.map())
    pb["kernels"] = pb["kernels"][:1]                      # and as the only kernel (K = 1)
    o, h = cc.setup(po.OracleCRF, pb), cc.setup(pkg.DenseCRFHIP, pb)
    o.inference_native(3, True, 0.8)
    h.inference(3, True, 0.8)
    assert cc.same_bits(o.probability(), h.probability())
    assert np.array_equal(o.map(), h.map())


@pytest.mark.parametrize("d,L", [(1, 2), (2, 3), (3, 2), (4, 4), (5, 21), (6, 2), (7, 2), (8, 5)])
def test_hip_matches_oracle_generic(po, wl, d, L):
    pb = wl.generic_problem(403, [d], L, seed=31, lattice_ties=True)
    o, h = cc.setup(po.OracleCRF, pb), cc.setup(pkg.DenseCRFHIP, pb)
    ko, kh = o.kernel(0), h.kernel(0)
    assert ko["V"] == kh["V"]
    for name in ("offset", "bary", "nbr", "norm"):
        assert cc.same_bits(ko[name], kh[name]), name
    assert cc.same_bits(o.run_trace(3, relax=0.75), h.run_trace(3, relax=0.75))


def test_hip_slam_factories_match_reference_division(po, wl):
    """lccrf_add_appearance_kernel / lccrf_add_smooth_kernel divide like pairwise3d.h:41-66."""
    fr = wl.slam_frame(1500, seed=5)
    p = wl.TUM3
    o = po.OracleCRF(1500, 2)
    o.set_unary_from_label(fr["init_label"], p["confidence"])
    o.add_pairwise(wl.appearance_features(fr), p["w1"])
    o.add_pairwise(wl.smooth_features(fr), p["w2"])
    h = pkg.DenseCRFHIP(1500, 2)
    h.set_unary_from_label(fr["init_label"], p["confidence"])
    h.add_appearance_kernel(p["w1"], fr["obs"], fr["err"], p["stdev_beta"], p["stdev_alpha"])
    h.add_smooth_kernel(p["w2"], fr["uv"], p["point2d_stdev"])
    o.inference_native(5, True)
    h.inference(5, True)
    assert cc.same_bits(o.probability(), h.probability())
    assert np.array_equal(o.map(), h.map())
    assert 0 < (h.map() == 0).sum() < 1500          # the workload is not degenerate


def test_hip_no_pairwise_and_unknown_labels(po):
    lab = np.array([-1, 0, 1, 1, -1, 0, 1], np.int16)
    o, h = po.OracleCRF(7, 2), pkg.DenseCRFHIP(7, 2)
    for c in (o, h):
        c.set_unary_from_label(lab, [0.7, 0.9])
    assert cc.same_bits(o.unary(), h.unary())
    assert cc.same_bits(o.run_trace(2), h.run_trace(2))


def test_hip_extreme_unaries_hit_the_fast_exp_cutoff(po):
    """Energy gaps > 20 make fast_exp return exactly 0 (densecrf3d.h:58)."""
    rng = np.random.default_rng(0)
    N = 300
    unary = rng.uniform(0, 60, (N, 3)).astype(np.float32)
    f = rng.normal(0, 2, (N, 2)).astype(np.float32)
    o, h = po.OracleCRF(N, 3), pkg.DenseCRFHIP(N, 3)
    for c in (o, h):
        c.set_unary(unary)
        c.add_pairwise(f, 4.0)
    to, th = o.run_trace(4), h.run_trace(4)
    assert (to == 0).any()
    assert cc.same_bits(to, th)


def test_hip_batch_matches_oracle_ragged(po, wl):
    """Frames in flight: ragged sizes (incl. an empty frame) in one batch."""
    sizes = [2000, 0, 1, 777, 1999, 5, 1024, 2000]
    maxN = 2000
    pbs = [wl.slam_problem(n, seed=40 + i) for i, n in enumerate(sizes)]
    F = len(sizes)
    feats = [np.zeros((F, maxN, 2), np.float32) for _ in range(2)]
    label = np.full((F, maxN), -1, np.int16)
    for f, pb in enumerate(pbs):
        n = pb["N"]
        label[f, :n] = pb["label"]
        for k in range(2):
            feats[k][f, :n] = pb["kernels"][k][0]
    b = pkg.BatchCRF(F, maxN, 2, [2, 2], [pbs[0]["kernels"][0][1], pbs[0]["kernels"][1][1]])
    b.set_inputs_host(sizes, feats, label=label, conf=pbs[0]["conf"])
    b.build()
    b.inference(5, True)
    Q, M = b.probability(), b.map()
    V0, V1 = b.lattice_sizes(0), b.lattice_sizes(1)
    for f, pb in enumerate(pbs):
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(5, True)
        n = pb["N"]
        assert (V0[f], V1[f]) == (o.kernel(0)["V"], o.kernel(1)["V"])
        assert cc.same_bits(Q[f, :n], o.probability()), f
        assert np.array_equal(M[f, :n], o.map()), f
    # idempotence: running the same batch again changes nothing
    b.inference(5, True)
    assert cc.same_bits(b.probability(), Q) and np.array_equal(b.map(), M)


def test_hip_full_size_properties(wl):
    """BASELINE config sizes (too slow for the oracle in bulk): size-independent properties."""
    F, N = 64, 3000
    pbs = [wl.slam_problem(N, seed=100 + (i % 4)) for i in range(F)]   # 4 distinct frames, repeated
    feats = [np.stack([pb["kernels"][k][0] for pb in pbs]) for k in range(2)]
    label = np.stack([pb["label"] for pb in pbs])
    b = pkg.BatchCRF(F, N, 2, [2, 2], [10.0, 30.0])
    b.set_inputs_host([N] * F, feats, label=label, conf=0.7)
    b.build()
    b.inference(5, True)
    Q, M = b.probability(), b.map()
    assert np.isfinite(Q).all()
    np.testing.assert_allclose(Q.sum(-1), 1.0, atol=2e-6)              # rows are distributions
    assert np.array_equal(M, Q.argmax(-1).astype(np.int16))            # first max wins (L=2: ties -> 0)
    for i in range(4, F):                                              # copies agree bit for bit
        assert cc.same_bits(Q[i], Q[i % 4]) and np.array_equal(M[i], M[i % 4])
    # a single-frame object run equals its slot in the batch
    c = cc.setup(pkg.DenseCRFHIP, pbs[1])
    c.inference(5, True)
    assert cc.same_bits(c.probability(), Q[1]) and np.array_equal(c.map(), M[1])


@pytest.mark.parametrize("N,n_iter,relax", [(2000, 5, 1.0), (3000, 5, 1.0), (1000, 10, 1.0), (257, 3, 0.5),
                                             (1, 2, 1.0), (3400, 2, 1.0), (4096, 2, 1.0)])
def test_fused_and_streaming_engines_agree_bitwise(wl, N, n_iter, relax):
    """Engine 2 (one workgroup per frame, LDS) vs engine 1 (streaming kernels)."""
    F = 6
    sizes = [N, max(N - 1, 0), N // 2, N, 0, min(N, 7)]
    pbs = [wl.slam_problem(n, seed=60 + i) for i, n in enumerate(sizes)]
    feats = [np.zeros((F, N, 2), np.float32) for _ in range(2)]
    label = np.full((F, N), -1, np.int16)
    for f, pb in enumerate(pbs):
        n = pb["N"]
        label[f, :n] = pb["label"]
        for k in range(2):
            feats[k][f, :n] = pb["kernels"][k][0]
    res = {}
    for eng in (1, 2):
        b = pkg.BatchCRF(F, N, 2, [2, 2], [10.0, 30.0])
        b.set_engine(eng)
        b.set_inputs_host(sizes, feats, label=label, conf=0.7)
        b.build()
        b.inference(n_iter, True, relax=relax)
        assert b.engine() == eng
        res[eng] = (b.probability(), b.map())
        b.close()
    for f, n in enumerate(sizes):
        assert cc.same_bits(res[1][0][f, :n], res[2][0][f, :n]), f
        assert np.array_equal(res[1][1][f, :n], res[2][1][f, :n]), f


def test_fused_engine_single_kernel(po, wl):
    pb = wl.slam_problem(1500, seed=77)
    pb["kernels"] = pb["kernels"][1:]                      # smoothness kernel only (K = 1)
    o = cc.setup(po.OracleCRF, pb)
    o.inference_native(4, True)
    h = cc.setup(pkg.DenseCRFHIP, pb)
    h.inference(4, True)
    assert cc.same_bits(o.probability(), h.probability()) and np.array_equal(o.map(), h.map())


def test_oversize_frames_fall_back_to_streaming_engine(wl):
    """More than 4 x 1024 keypoints do not fit the fused engine's one workgroup per frame:
    automatic engine choice must pick the streaming engine, and forcing the fused one must fail loudly."""
    N = 4100
    pb = wl.slam_problem(N, seed=3)
    feats = [pb["kernels"][k][0][None] for k in range(2)]
    b = pkg.BatchCRF(1, N, 2, [2, 2], [10.0, 30.0])
    b.set_inputs_host([N], feats, label=pb["label"][None], conf=0.7)
    b.build()
    b.inference(2, True)
    assert b.engine() == 1
    b.set_engine(2)
    with pytest.raises(pkg.LccrfError):
        b.inference(2, True)


def test_handle_cache_reuses_resources_across_frame_sizes(po, wl):
    """The reference builds one DenseCRF3D per frame; destroyed handles are recycled.  A recycled
    handle must behave like a fresh one for a different N (larger-capacity buffers, stale data)."""
    pkg.lib().lccrf_trim_cache()
    for N, seed in ((2000, 1), (500, 2), (1999, 3), (3, 4), (2000, 5)):
        pb = wl.slam_problem(N, seed=seed)
        o, h = cc.setup(po.OracleCRF, pb), cc.setup(pkg.DenseCRFHIP, pb)
        o.inference_native(5, True)
        h.inference(5, True)
        assert cc.same_bits(o.probability(), h.probability()), N
        assert np.array_equal(o.map(), h.map()), N
        for k in range(2):
            assert o.kernel(k)["V"] == h.kernel(k)["V"]
        h.close()
    assert pkg.lib().lccrf_trim_cache() >= 1


@pytest.mark.parametrize("N", [5, 700, 2000, 3000])
def test_fused_build_equals_streaming_build(po, wl, N):
    """build_small.hip (one launch) vs the 19-launch streaming build: identical lattices.  The streaming build of a SLAM-size
    frame is a cross-check switch (LCCRF_NO_FUSED_BUILD) of the instrumented library: it runs in a child process."""
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import crf_cases as cc
pkg = importlib.import_module("lc-crf-slam_amd")
wl = importlib.import_module("lc-crf-slam_amd.workloads")
pb = wl.slam_problem(int(sys.argv[2]), seed=123)
h = cc.setup(pkg.DenseCRFHIP, pb)
ks = [h.kernel(k) for k in range(2)]
h.inference(3, True)
out = {"Q": h.probability()}
for k in range(2):
    out["V%%d" %% k] = np.int64(ks[k]["V"])
    for name in ("offset", "bary", "nbr", "norm"):
        out["%%s%%d" %% (name, k)] = ks[k][name]
h.close()
np.savez(sys.argv[1], **out)
""" % (ROOT, os.path.join(ROOT, "tests"))
    res = {}
    for mode, env in (("fused", {}), ("streaming", {"LCCRF_NO_FUSED_BUILD": "1"})):
        path = os.path.join(ROOT, "gpurun_out", "build_%s_%d.npz" % (mode, N))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.run([sys.executable, "-c", code, path, str(N)], check=True, env=cc.switch_env(env), timeout=300)
        res[mode] = np.load(path)
    for k in range(2):
        assert int(res["fused"]["V%d" % k]) == int(res["streaming"]["V%d" % k])
        for name in ("offset", "bary", "nbr", "norm"):
            assert cc.same_bits(res["fused"]["%s%d" % (name, k)], res["streaming"]["%s%d" % (name, k)]), (k, name)
    assert cc.same_bits(res["fused"]["Q"], res["streaming"]["Q"])
    pb = wl.slam_problem(N, seed=123)
    o = cc.setup(po.OracleCRF, pb)
    for k in range(2):
        ko = o.kernel(k)
        for name in ("offset", "bary", "nbr", "norm"):
            assert cc.same_bits(ko[name], res["fused"]["%s%d" % (name, k)]), (k, name)


def test_step_api_after_deferred_build(po, wl):
    """add_pairwise only stages features; the first step / probe triggers one joint build."""
    pb = wl.slam_problem(900, seed=77)
    o, h = cc.setup(po.OracleCRF, pb), cc.setup(pkg.DenseCRFHIP, pb)
    h.start_inference()
    o.start_inference()
    assert cc.same_bits(o.probability(), h.probability())        # no lattice needed yet
    h.step_inference()
    o.step_inference()
    assert cc.same_bits(o.probability(), h.probability())


@pytest.mark.gpu
def test_object_api_from_several_threads(po, wl):
    """Handles are independent (own stream, own buffers; the parked-handle cache and the error string
    are the only shared state): four threads running frames of different sizes concurrently must each
    get the oracle's results.  ctypes releases the GIL during the calls, so the threads really overlap."""
    import threading
    sizes = [700, 1300, 2000, 2600]
    pbs = [wl.slam_problem(n, seed=80 + i) for i, n in enumerate(sizes)]
    want = []
    for pb in pbs:
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(5, True)
        want.append((o.probability().copy(), o.map().copy()))
        o.close()
    errors = []

    def worker(t):
        try:
            for rep in range(40):
                j = (t + rep) % len(pbs)
                h = cc.setup(pkg.DenseCRFHIP, pbs[j])
                h.inference(5, True)
                if not (np.array_equal(h.map(), want[j][1]) and cc.same_bits(h.probability(), want[j][0])):
                    errors.append((t, rep, j))
                h.close()
        except Exception as e:                                 # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


@pytest.mark.gpu
def test_batch_mixing_long_row_and_short_row_frames(po, wl):
    """One batch, one launch: the chain decision is made once for the batch (longest row over all
    frames), so frames with short rows ride the chain path of their long-row neighbours and vice versa."""
    shapes = [("one_cell", 1800), (None, 2000), ("rows_of_8", 1999), ("two_clusters", 2048), (None, 7), ("one_cell", 64)]
    pbs = [wl.slam_problem(n, seed=90 + i) if sh is None else _shaped_problem(wl, n, sh, seed=90 + i)
           for i, (sh, n) in enumerate(shapes)]
    F, maxn = len(pbs), max(pb["N"] for pb in pbs)
    feats = [np.zeros((F, maxn, 2), np.float32) for _ in range(2)]
    label = np.full((F, maxn), -1, np.int16)
    for f, pb in enumerate(pbs):
        n = pb["N"]
        label[f, :n] = pb["label"]
        for k in range(2):
            feats[k][f, :n] = pb["kernels"][k][0]
    for eng in (0, 1):
        b = pkg.BatchCRF(F, maxn, 2, [2, 2], [10.0, 30.0])
        b.set_engine(eng)
        b.set_inputs_host([pb["N"] for pb in pbs], feats, label=label, conf=0.7)
        b.build()
        b.inference(5, True)
        Q, M = b.probability(), b.map()
        if eng == 0:
            assert b.engine() == 2                         # everything fits one workgroup per frame
        b.close()
        for f, pb in enumerate(pbs):
            o = cc.setup(po.OracleCRF, pb)
            o.inference_native(5, True)
            n = pb["N"]
            assert cc.same_bits(Q[f, :n], o.probability()), (eng, f)
            assert np.array_equal(M[f, :n], o.map()), (eng, f)
            o.close()


def test_batch_on_a_caller_stream_right_after_create(wl):
    """build/inference on a foreign stream must be ordered behind the engine's own stream
    (allocation memsets, the unary kernel of bind_inputs_device).  Fresh handles every round so that
    the zeroing of several hundred MB of lattice arrays is still in flight when the build is queued."""
    import torch
    F, N = 96, 2000
    pbs = [wl.slam_problem(N, seed=300 + (i % 3)) for i in range(F)]
    feats = [np.stack([pb["kernels"][k][0] for pb in pbs]) for k in range(2)]
    label = np.stack([pb["label"] for pb in pbs])
    dev = torch.device("cuda", 0)
    d_feats = [torch.from_numpy(f).to(dev) for f in feats]
    d_label = torch.from_numpy(label).to(dev)
    d_np = torch.full((F,), N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ref = None
    for use_foreign in (False, True, True, True):
        st = torch.cuda.Stream(device=dev) if use_foreign else None
        b = pkg.BatchCRF(F, N, 2, [2, 2], [10.0, 30.0])
        b.bind_inputs_device(F, d_np.data_ptr(), [t.data_ptr() for t in d_feats], d_label=d_label.data_ptr(), conf=0.7)
        b.build(stream=st.cuda_stream if st else None)
        b.inference(5, True, stream=st.cuda_stream if st else None)
        Q, M = b.probability(), b.map()
        b.close()
        if ref is None:
            ref = (Q, M)
        else:
            assert cc.same_bits(Q, ref[0]) and np.array_equal(M, ref[1])


def test_lazy_allocations_are_zeroed_on_the_callers_stream(po, wl):
    """buffers allocated lazily DURING a batch call -- the two-workgroup hand-off area of lccrf_batch_run for
    up to 64 frames, the point sort's scratch of locality mode (>= 8192 points) -- must be zeroed on the stream the call's kernels
    run on, not on the engine's own stream (a memset that lands late wipes hand-off records or the permutation).  Fresh handles,
    caller streams, several rounds so that the first-use path runs each time; results against the oracle."""
    import torch
    dev = torch.device("cuda", 0)
    # (1) two-workgroup form: 6 SLAM frames through lccrf_batch_run on a foreign stream, fresh handle every round
    F, N = 6, 1900
    pbs = [wl.slam_problem(N - 13 * i, seed=410 + i) for i in range(F)]
    want = []
    for pb in pbs:
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(5, True)
        want.append((o.probability().copy(), o.map().copy()))
        o.close()
    feats = [np.zeros((F, N, 2), np.float32) for _ in range(2)]
    label = np.full((F, N), -1, np.int16)
    for f, pb in enumerate(pbs):
        label[f, :pb["N"]] = pb["label"]
        for k in range(2):
            feats[k][f, :pb["N"]] = pb["kernels"][k][0]
    d_feats = [torch.from_numpy(x).to(dev) for x in feats]
    d_label = torch.from_numpy(label).to(dev)
    d_np = torch.tensor([pb["N"] for pb in pbs], dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for rnd in range(4):
        st = torch.cuda.Stream(device=dev)
        b = pkg.BatchCRF(F, N, 2, [2, 2], [10.0, 30.0])
        b.bind_inputs_device(F, d_np.data_ptr(), [t.data_ptr() for t in d_feats], d_label=d_label.data_ptr(), conf=0.7)
        b.run(5, True, stream=st.cuda_stream)
        Q, M = b.probability(), b.map()
        assert b.engine() == 3
        b.close()
        for f, pb in enumerate(pbs):
            assert cc.same_bits(Q[f, :pb["N"]], want[f][0]) and np.array_equal(M[f, :pb["N"]], want[f][1]), (rnd, f)
    # (2) locality mode: build + inference of two 9000-point frames on a foreign stream, fresh handle every round
    N2 = 9000
    pb2 = wl.bilateral_problem(N2, seed=415)
    o = cc.setup(po.OracleCRF, pb2)
    o.inference_native(3, True)
    f2 = torch.from_numpy(np.repeat(pb2["kernels"][0][0][None], 2, 0)).to(dev)
    l2 = torch.from_numpy(np.repeat(pb2["label"][None], 2, 0)).to(dev)
    n2 = torch.full((2,), N2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for rnd in range(4):
        st = torch.cuda.Stream(device=dev)
        b = pkg.BatchCRF(2, N2, 2, [6], [float(pb2["kernels"][0][1])])
        b.bind_inputs_device(2, n2.data_ptr(), [f2.data_ptr()], d_label=l2.data_ptr(), conf=pb2["conf"])
        b.build(stream=st.cuda_stream)
        b.inference(3, True, stream=st.cuda_stream)
        Q, M = b.probability(), b.map()
        b.close()
        for g in range(2):
            assert cc.same_bits(Q[g], o.probability()) and np.array_equal(M[g], o.map()), (rnd, g)
    o.close()


def test_bound_n_points_out_of_range_is_reported_not_followed(wl):
    """a device-bound n_points[f] > max_points must not make the kernels run past the frame stride."""
    import torch
    F, N = 4, 500
    pbs = [wl.slam_problem(N, seed=310 + i) for i in range(F)]
    feats = [np.stack([pb["kernels"][k][0] for pb in pbs]) for k in range(2)]
    label = np.stack([pb["label"] for pb in pbs])
    dev = torch.device("cuda", 0)
    d_feats = [torch.from_numpy(f).to(dev) for f in feats]
    d_label = torch.from_numpy(label).to(dev)
    d_np = torch.tensor([N, 10 * N, -3, N], dtype=torch.int32, device=dev)
    b = pkg.BatchCRF(F, N, 2, [2, 2], [10.0, 30.0])
    b.bind_inputs_device(F, d_np.data_ptr(), [t.data_ptr() for t in d_feats], d_label=d_label.data_ptr(), conf=0.7)
    b.build()
    with pytest.raises(pkg.LccrfError) as ei:
        b.inference(5, True)
    assert ei.value.code == -6
    b.close()


def test_hip_c5_full_size_matches_oracle(po, wl):
    """BASELINE config 5 at FULL size: 100 000 points, one 6-D kernel (V ~ 5.9e5: hash capacity
    2^21, 700k-entry scans, the multi-launch iteration), 20 iterations -- one frame through the object API and one
    through the batch API against the oracle: lattice size, vertex ids, barycentrics, neighbour table, norm, Q, labels."""
    pb = wl.bilateral_problem(100000, 1)
    o = cc.setup(po.OracleCRF, pb)
    o.inference_native(20, True)
    ko = o.kernel(0)
    h = cc.setup(pkg.DenseCRFHIP, pb)
    h.inference(20, True)
    kh = h.kernel(0)
    assert kh["V"] == ko["V"] and kh["V"] > 500000
    assert np.array_equal(kh["offset"], ko["offset"]) and np.array_equal(kh["nbr"], ko["nbr"])
    assert cc.same_bits(kh["bary"], ko["bary"]) and cc.same_bits(kh["norm"], ko["norm"])
    assert cc.same_bits(h.probability(), o.probability())
    assert np.array_equal(h.map(), o.map())
    h.close()
    b = pkg.BatchCRF(1, pb["N"], 2, [6], [float(pb["kernels"][0][1])])
    b.set_inputs_host([pb["N"]], [pb["kernels"][0][0][None]], label=pb["label"][None], conf=pb["conf"])
    b.build()
    b.inference(20, True)
    assert cc.same_bits(b.probability()[0], o.probability()) and np.array_equal(b.map()[0], o.map())
    assert int(b.lattice_sizes(0)[0]) == ko["V"]
    b.close()


@pytest.mark.parametrize("N,d,L", [(6000, 2, 3), (5000, 3, 2), (700, 5, 4)])
def test_streaming_build_with_very_long_rows(po, wl, N, d, L):
    """clustered / identical features put thousands of entries on one vertex; the streaming build orders such
    rows with a per-row sort instead of the quadratic rank (k_csr_sort_long).  Lattice and results vs the oracle."""
    rng = np.random.default_rng(7)
    pb = wl.generic_problem(N, [d], L, seed=3)
    f = pb["kernels"][0][0].copy()
    f[: N // 2] = f[0]                                     # half the points in ONE cell: d+1 rows of N/2 entries
    f[N // 2: N // 2 + N // 4] = f[-1] + rng.normal(0, 0.01, (N // 4, d)).astype(np.float32)   # a tight cluster
    pb["kernels"] = [(f, pb["kernels"][0][1])]
    o, h = cc.setup(po.OracleCRF, pb), cc.setup(pkg.DenseCRFHIP, pb)
    ko, kh = o.kernel(0), h.kernel(0)
    assert ko["V"] == kh["V"]
    for name in ("offset", "bary", "nbr", "norm"):
        assert cc.same_bits(ko[name], kh[name]), name
    o.inference_native(3, True)
    h.inference(3, True)
    assert cc.same_bits(o.probability(), h.probability()) and np.array_equal(o.map(), h.map())


def test_streaming_engine_xcd_aware_grid_with_many_frames(po, wl):
    """With >= 8 frames in flight the streaming iteration kernels use the XCD-aware 1-D grid (one XCD per frame): 11 ragged
    frames (a partial last group of eight, an empty frame) on engine 1 against the oracle."""
    sizes = [700, 0, 1, 333, 699, 5, 512, 700, 64, 257, 700]
    maxN = 700
    pbs = [wl.slam_problem(n, seed=140 + i) for i, n in enumerate(sizes)]
    F = len(sizes)
    feats = [np.zeros((F, maxN, 2), np.float32) for _ in range(2)]
    label = np.full((F, maxN), -1, np.int16)
    for f, pb in enumerate(pbs):
        n = pb["N"]
        label[f, :n] = pb["label"]
        for k in range(2):
            feats[k][f, :n] = pb["kernels"][k][0]
    b = pkg.BatchCRF(F, maxN, 2, [2, 2], [10.0, 30.0])
    b.set_engine(1)
    b.set_inputs_host(sizes, feats, label=label, conf=0.7)
    b.build()
    b.inference(4, True)
    assert b.engine() == 1
    Q, M = b.probability(), b.map()
    for f, pb in enumerate(pbs):
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(4, True)
        assert cc.same_bits(Q[f, :pb["N"]], o.probability()) and np.array_equal(M[f, :pb["N"]], o.map()), f


@pytest.mark.gpu
@pytest.mark.parametrize("F", [1, 2, 3, 4, 5, 6, 7, 9, 12, 13])
def test_streaming_engine_xcd_chunked_grid_below_eight_frames(po, wl, F):
    """Any number of frames in flight (round 4): the launch's blocks -- frame after frame -- are cut into eight contiguous parts,
    one per XCD, so with fewer than 8 frames an XCD holds a contiguous chunk of a frame's vertex / row / point range and with 3, 6,
    12, 13 frames a part spans a frame boundary.  Ragged frames, an empty one, on engine 1 against the oracle -- and the plain grid
    (LCCRF_NO_XCD_CHUNK is read once per process, so the A/B itself is `FRAMES=1 WORKLOAD=c5 scripts/gpu_env_ab.sh LCCRF_NO_XCD_CHUNK=1 ""`;
    here the results must simply be the oracle's)."""
    sizes = [700, 333, 0, 699, 5, 512, 257, 700, 1, 650, 0, 300, 699][:F]
    maxN = 700
    pbs = [wl.slam_problem(n, seed=640 + i) for i, n in enumerate(sizes)]
    feats = [np.zeros((F, maxN, 2), np.float32) for _ in range(2)]
    label = np.full((F, maxN), -1, np.int16)
    for f, pb in enumerate(pbs):
        n = pb["N"]
        label[f, :n] = pb["label"]
        for k in range(2):
            feats[k][f, :n] = pb["kernels"][k][0]
    b = pkg.BatchCRF(F, maxN, 2, [2, 2], [10.0, 30.0])
    b.set_engine(1)
    b.set_inputs_host(sizes, feats, label=label, conf=0.7)
    b.build()
    b.inference(4, True)
    assert b.engine() == 1
    Q, M = b.probability(), b.map()
    b.close()
    for f, pb in enumerate(pbs):
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(4, True)
        assert cc.same_bits(Q[f, :pb["N"]], o.probability()) and np.array_equal(M[f, :pb["N"]], o.map()), f
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("d_list,N", [([6], 9000), ([1], 8200), ([2, 5], 8500), ([3], 5000), ([4, 2], 4500), ([8], 8192)])
def test_single_frame_blur_passes_go_two_per_launch(po, wl, d_list, N):
    """ONE frame in flight on the streaming engine (BASELINE config 5 as written; round 4): the blur passes go two per launch
    (k_blur2x2t on the two-hop table the streaming build leaves for single-frame batches, k_blur2x2 without a table), and with an
    odd number of passes the one left over is done inside the slice (k_slice2<D1, true>).  Every d + 1 from 2 to 9 (pairs only,
    pairs + left-over), two kernels of different d, locality mode on (>= 8192 points) and off: lattice sizes, Q and labels against
    the oracle, bit for bit, over two builds and a second inference."""
    pb = wl.generic_problem(N, d_list, 2, seed=700 + N % 97 + len(d_list), spread=3.0)
    ws = [float(w) for _, w in pb["kernels"]]
    b = pkg.BatchCRF(1, N, 2, d_list, ws)
    b.set_inputs_host([N], [f[None] for f, _ in pb["kernels"]], unary=pb["unary"][None])
    b.build(); b.build()
    b.inference(3, True, relax=0.9)
    b.inference(3, True, relax=0.9)
    assert b.engine() == 1
    Q, M = b.probability()[0], b.map()[0]
    o = cc.setup(po.OracleCRF, pb)
    o.inference_native(3, True, 0.9)
    for k in range(len(d_list)):
        assert int(b.lattice_sizes(k)[0]) == o.kernel(k)["V"], k
    assert cc.same_bits(Q, o.probability()) and np.array_equal(M, o.map())
    # the same frame as one of two frames in flight takes the one-pass-per-launch kernels: same bits
    b2 = pkg.BatchCRF(2, N, 2, d_list, ws)
    b2.set_inputs_host([N, N], [np.repeat(f[None], 2, 0) for f, _ in pb["kernels"]], unary=np.repeat(pb["unary"][None], 2, 0))
    b2.build()
    b2.inference(3, True, relax=0.9)
    Q2 = b2.probability()
    assert cc.same_bits(Q2[0], Q) and cc.same_bits(Q2[1], Q)
    b.close(); b2.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("d_list,L", [([6], 2), ([3, 2], 3), ([1, 4], 2)])
def test_large_ragged_batch_matches_oracle(po, wl, d_list, L):
    """A ragged batch of large frames (up to 9000 points: hash tables, scans and CSR far beyond the one-workgroup engines)
    with mixed feature dimensions and label counts on the streaming engine: lattice sizes, Q and labels against the oracle,
    bit for bit.  (Written for an experiment that numbered vertices along a Z-order curve -- notes/r2_experiments.md --
    and kept: results must not depend on how a build numbers its vertices.)"""
    sizes = [9000, 8192, 0, 4097, 9000]
    F, maxN = len(sizes), 9000
    pbs = [wl.generic_problem(n, d_list, L, seed=300 + i, spread=3.0) for i, n in enumerate(sizes)]
    feats = [np.zeros((F, maxN, d), np.float32) for d in d_list]
    unary = np.zeros((F, maxN, L), np.float32)
    for f, pb in enumerate(pbs):
        n = pb["N"]
        unary[f, :n] = pb["unary"]
        for k in range(len(d_list)):
            feats[k][f, :n] = pb["kernels"][k][0]
    ws = [float(pbs[0]["kernels"][k][1]) for k in range(len(d_list))]
    b = pkg.BatchCRF(F, maxN, L, d_list, ws)
    b.set_option(pkg.OPT_VERTEX_ORDER, 1 if L == 2 else 2)   # (mixed dimensions, ragged and empty frames through the sorted build / the hash build)
    b.set_inputs_host(sizes, feats, unary=unary)
    b.build()
    b.inference(3, True, relax=0.9)
    Q, M = b.probability(), b.map()
    for f, pb in enumerate(pbs):
        n = pb["N"]
        if n == 0:
            continue
        pbw = dict(pb, kernels=[(pb["kernels"][k][0], np.float32(ws[k])) for k in range(len(d_list))])
        o = cc.setup(po.OracleCRF, pbw)
        o.inference_native(3, True, 0.9)
        for k in range(len(d_list)):
            assert int(b.lattice_sizes(k)[f]) == o.kernel(k)["V"], (f, k)
        assert cc.same_bits(Q[f, :n], o.probability()), f
        assert np.array_equal(M[f, :n], o.map()), f
        o.close()
    # frames of this size are built in LOCALITY MODE (an internal Z-order of the points, csrc/stream_engine.hip:
    # launch_sort_points): what the caller sees stays in the caller's point order -- norm included
    for k in range(len(d_list)):
        nm = b.norm(k)
        for f in (0, 3):
            pbw = dict(pbs[f], kernels=[(pbs[f]["kernels"][j][0], np.float32(ws[j])) for j in range(len(d_list))])
            o = cc.setup(po.OracleCRF, pbw)
            assert cc.same_bits(nm[f, :sizes[f]], o.kernel(k)["norm"]), (k, f)
            o.close()
    # new inputs into the same handle (frames swapped, one frame shrunk), rebuilt twice: the order is decided afresh
    order = [4, 3, 2, 1, 0]
    sizes2 = [sizes[i] for i in order]
    sizes2[0] = 5000
    unary2 = unary[order].copy()
    feats2 = [f[order].copy() for f in feats]
    b.set_inputs_host(sizes2, feats2, unary=unary2)
    b.build(); b.build()
    b.inference(2, True)
    Q, M = b.probability(), b.map()
    for f in (0, 1, 4):
        n = sizes2[f]
        pb = dict(N=n, L=L, unary=unary2[f, :n], kernels=[(feats2[k][f, :n], np.float32(ws[k])) for k in range(len(d_list))])
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(2, True)
        assert cc.same_bits(Q[f, :n], o.probability()) and np.array_equal(M[f, :n], o.map()), f
        o.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("vertex_order", [0, 1, 2])
@pytest.mark.parametrize("name", ["c5", "gen"])
def test_locality_mode_on_reference_vectors(golden, name, vertex_order):
    """Locality mode against vectors generated by the reference build itself (tests/golden/large.npz): a frame alone,
    and eight copies (XCD-aware grids), must reproduce the reference's V, norm, Q and labels bit for bit although the
    points are processed in an internal order -- and the lattice built by sorting the entries on the row-major code of their vertex
    (round 4: the sorted build, no hash table; LCCRF_OPT_VERTEX_ORDER 0 = automatic: the eight copies, 1 = on: the single frame
    too, 2 = off: the hash build for both; results must not depend on how vertices are found or numbered)."""
    from test_oracle_golden import _large_case
    z = golden["large"]
    pb, n_iter, relax = _large_case(z, name)
    N, L = pb["N"], pb["L"]
    f, w = pb["kernels"][0]
    for F in (1, 8):
        b = pkg.BatchCRF(F, N, L, [f.shape[1]], [float(w)])
        b.set_option(pkg.OPT_VERTEX_ORDER, vertex_order)
        feats = [np.repeat(f[None], F, 0)]
        if "unary" in pb:
            b.set_inputs_host([N] * F, feats, unary=np.repeat(pb["unary"][None], F, 0))
        else:
            b.set_inputs_host([N] * F, feats, label=np.repeat(pb["label"][None], F, 0), conf=pb["conf"])
        b.build()
        b.inference(n_iter, True, relax=relax)
        Q, M, V, nm = b.probability(), b.map(), b.lattice_sizes(0), b.norm(0)
        assert b.engine() == 1
        for g in range(F):
            assert int(V[g]) == int(z[name + "_V"]) and cc.same_bits(nm[g], z[name + "_norm"]), (F, g)
            assert cc.same_bits(Q[g], z[name + "_Q"]) and np.array_equal(M[g], z[name + "_map"]), (F, g)
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("d,spread", [(8, 60.0), (6, 1200.0)])
def test_sorted_build_falls_back_to_the_hash_when_codes_overflow(po, wl, d, spread):
    """The sorted build of locality mode codes a vertex by its row-major position in the bounding box of the frame's lattice (62
    bits).  Features spread over hundreds of cells in each of 8 dimensions (or thousands in 6) overflow that: the plan kernel
    raises a flag in pinned memory and the engine rebuilds those lattices with the hash table -- same results, no error."""
    N, F = 8300, 8
    pb = wl.generic_problem(N, [d], 2, seed=77 + d, spread=spread)
    w = float(pb["kernels"][0][1])
    o = cc.setup(po.OracleCRF, pb)
    o.inference_native(2, True)
    b = pkg.BatchCRF(F, N, 2, [d], [w])
    b.set_inputs_host([N] * F, [np.repeat(pb["kernels"][0][0][None], F, 0)], unary=np.repeat(pb["unary"][None], F, 0))
    b.build()
    b.inference(2, True)
    Q, M, V = b.probability(), b.map(), b.lattice_sizes(0)
    for f in range(F):
        assert int(V[f]) == o.kernel(0)["V"] and cc.same_bits(Q[f], o.probability()) and np.array_equal(M[f], o.map()), f
    b.build()                                              # (the engine remembers: straight to the hash build)
    b.inference(2, True)
    assert cc.same_bits(b.probability(), Q)
    b.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("L,dims,F", [(2, [2], 1), (2, [5, 2], 3), (5, [3], 1), (21, [2, 5], 2), (1 + 2, [2], 8)])
def test_streaming_engine_long_rows_go_to_a_workgroup_each(po, wl, L, dims, F):
    """A coarse kernel over many points, and a third of the points IDENTICAL: rows of tens of entries everywhere and a few of thousands
    (the reference's image demo has them: uniformly coloured regions).  The adds of a row go one by one in point order (quirk Q6);
    the loads do not: rows beyond 512 entries are listed by the build and summed by a workgroup each (k_splat_long: products of a
    tile in LDS, one lane per label adds its column), the others by k_splat / k_splat4 with their loads up front, or -- two labels
    -- by k_splat2l / k_splat2v; the sorted build orders a giant vertex's bucket by a bitmap over the original ids.  Ragged frames,
    locality mode on and off (8300 / 6000 points): the oracle's bits."""
    for N in (8300, 6000):
        pb = wl.generic_problem(N, dims, L, seed=77 + L, spread=2.0)
        for k, (f, w) in enumerate(pb["kernels"]):
            f[: N // 3] = f[5]                               # one giant vertex (per remainder class) + its neighbourhood
            f[N // 3: N // 2] = np.round(f[N // 3: N // 2])  # ... and many shared cells
        sizes = [N - 211 * i for i in range(F)]
        b = pkg.BatchCRF(F, N, L, dims, [float(w) for _, w in pb["kernels"]])
        b.set_engine(1)
        b.set_option(pkg.OPT_VERTEX_ORDER, 2 if (L + F) % 2 else 0)      # (the hash build lists its long rows too)
        b.set_inputs_host(sizes, [np.repeat(f[None], F, 0) for f, _ in pb["kernels"]], unary=np.repeat(pb["unary"][None], F, 0))
        b.build(); b.inference(3, True, relax=0.9)
        Q, M = b.probability(), b.map()
        for i in sorted(set([0, F - 1])):
            n = sizes[i]
            q = dict(pb, N=n, unary=pb["unary"][:n], kernels=[(f[:n], w) for f, w in pb["kernels"]])
            o = cc.setup(po.OracleCRF, q)
            o.inference_native(3, True, 0.9)
            assert all(int(b.lattice_sizes(k)[i]) == o.kernel(k)["V"] for k in range(len(dims))), (N, i)
            assert cc.same_bits(Q[i, :n], o.probability()) and np.array_equal(M[i, :n], o.map()), (N, i)
            o.close()
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [4096, 4097, 6000])
@pytest.mark.parametrize("labels", [False, True])
def test_object_api_inputs_beyond_the_frame_kernel_are_uploaded(po, wl, N, labels):
    """Up to 4096 points a handle's kernels read the point count, the labels and the features straight from pinned host memory (the
    one-launch kernel's range); beyond, they are uploaded once (a dozen kernels of thousands of wavefronts would each cross PCIe).
    Either side of the line, with labels and with raw unaries, twice on a recycled handle with other inputs: the oracle's bits."""
    for seed in (5, 6):
        pb = wl.generic_problem(N, [2, 3], 2, seed=seed, spread=2.0)
        if labels:
            pb = dict(pb, label=np.random.default_rng(seed).integers(-1, 2, N).astype(np.int16), conf=np.float32([0.8, 0.6]))
            del pb["unary"]
        h, o = cc.setup(pkg.DenseCRFHIP, pb), cc.setup(po.OracleCRF, pb)
        h.inference(3, True); o.inference_native(3, True)
        assert cc.same_bits(h.probability(), o.probability()) and np.array_equal(h.map(), o.map()), (N, seed)
        assert cc.same_bits(h.unary(), o.unary())
        h.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("d_list,labels", [([6], False), ([3, 2], True), ([5], True)])
def test_object_api_large_frame_runs_inference_in_locality_mode(po, wl, d_list, labels):
    """BASELINE config 5 through the reference's OWN interface: a DenseCRF of >= 8192 points.  `inference()` on such a handle runs in
    locality mode (internal point order, sorted build, blur passes in the splat) -- and every other entry point of the object API
    still answers in the caller's point order: stepInference continues from the result, PairwisePotential::apply, the norm, the
    lattice probe (the reference's own vertex numbering) and the unary probe re-build the lattices the plain way, once; new unaries
    and a second inference; a recycled handle starts afresh.  Everything against the oracle, bit for bit."""
    N = 9100
    pb = wl.generic_problem(N, d_list, 2, seed=913 + len(d_list), spread=2.8)
    if labels:
        rng = np.random.default_rng(5)
        pb = dict(pb, label=rng.integers(-1, 2, N).astype(np.int16), conf=np.float32([0.7, 0.6]))
        del pb["unary"]
    for order in range(3):
        h = cc.setup(pkg.DenseCRFHIP, pb)
        o = cc.setup(po.OracleCRF, pb)
        if order == 1:                                     # a probe first: the handle is built the plain way and stays so
            assert np.array_equal(h.kernel(0)["offset"], o.kernel(0)["offset"])
        h.inference(3, True, 0.9)
        o.inference_native(3, True, 0.9)
        assert cc.same_bits(h.probability(), o.probability()) and np.array_equal(h.map(), o.map()), order
        if order == 2:                                     # new unaries on the same lattices, then again
            u = np.random.default_rng(8).uniform(0.1, 2.0, (N, 2)).astype(np.float32)
            h.set_unary(u); o.set_unary(u)
            h.inference(2, True); o.inference_native(2, True)
            assert cc.same_bits(h.probability(), o.probability()), order
        h.step_inference(0.8)                              # continues from inference()'s Q -- on plain lattices from here on
        o.step_inference(0.8)
        assert cc.same_bits(h.probability(), o.probability()), order
        x = np.random.default_rng(3).normal(0, 1, (N, 2)).astype(np.float32)
        assert cc.same_bits(h.apply(0, np.zeros((N, 2), np.float32), x), o.apply(0, np.zeros((N, 2), np.float32), x))
        kh, ko = h.kernel(0), o.kernel(0)
        assert kh["V"] == ko["V"] and np.array_equal(kh["offset"], ko["offset"]) and cc.same_bits(kh["norm"], ko["norm"])
        assert cc.same_bits(h.unary(), o.unary())
        h.inference(2, True); o.inference_native(2, True)  # (banned from locality mode now: same bits all the same)
        assert cc.same_bits(h.probability(), o.probability()) and np.array_equal(h.map(), o.map())
        h.close(); o.close()                               # parked: the next round reuses the handle


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"LCCRF_SPLAT_PASSES": "1"}, {"LCCRF_SPLAT_PASSES": "2"}, {"LCCRF_NO_SPLAT_BLUR": "1"}, {"LCCRF_NO_XCD_CHUNK": "1"},
                                 {"LCCRF_NO_COMPACT_NBR": "1"}, {"LCCRF_NO_PAIR_FUSE": "1"}, {"LCCRF_SPLAT_REC": "1"},
                                 {"LCCRF_SPLAT_REC": "1", "LCCRF_SPLAT_512X2": "1"}])
def test_streaming_engine_switches_do_not_change_a_bit(po, wl, env):
    """The sorted build lets the splat take the first blur passes along (axis 0: adjacent ids; axes 1 and 2: ids within the halo of an
    LDS window), reads a compact neighbour table with 3-5 frames in flight, pairs the passes of a single frame ...: every one of
    these is a choice of HOW, read once per process from the environment for A/B runs.  A child process per switch: one, three
    and eight frames of a 6-D and of a (3-D, 2-D) problem in locality mode -- the same bits as this process's default path and
    as the oracle."""
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
def run():
    out = []
    for dims, N in (([6], 9000), ([3, 2], 8400)):
        pb = wl.generic_problem(N, dims, 2, seed=41 + len(dims), spread=2.5)
        for F in (1, 3, 8):
            b = pkg.BatchCRF(F, N, 2, dims, [float(w) for _, w in pb["kernels"]])
            b.set_inputs_host([N - 37 * f for f in range(F)], [np.repeat(f[None], F, 0) for f, _ in pb["kernels"]], unary=np.repeat(pb["unary"][None], F, 0))
            b.build(); b.inference(3, True)
            out += [b.probability().ravel(), b.map().astype(np.float32).ravel()]
            b.close()
    return np.concatenate(out)
if __name__ == "__main__":
    np.save(sys.argv[1], run())
""" % ROOT
    path = os.path.join(ROOT, "gpurun_out", "switch_%s.npy" % "_".join(env))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    subprocess.run([sys.executable, "-c", code, path], check=True, env=cc.switch_env(env), timeout=600)
    got = np.load(path)
    ns = {}
    exec(compile(code.replace('if __name__ == "__main__":', "if False:"), "<switch>", "exec"), ns)
    want = ns["run"]()
    assert cc.same_bits(got, want), env
    # ... and the default path against the oracle on the first problem's single frame
    pb = wl.generic_problem(9000, [6], 2, seed=42, spread=2.5)
    o = cc.setup(po.OracleCRF, pb)
    o.inference_native(3, True)
    assert cc.same_bits(want[:9000 * 2].reshape(9000, 2), o.probability())
    o.close()


@pytest.mark.gpu
def test_sorted_build_tables_are_checked_and_abandoned_when_they_do_not_hold(po, wl):
    """Two short cuts of the sorted build rest on the ids following the row-major codes, and the build CHECKS both instead of trusting
    the argument.  (1) With 3-5 large frames in flight the blur passes read a COMPACT neighbour table: 16-bit offsets from a base per
    64 vertices -- a block's neighbours span about a block, but nothing guarantees it: every offset is checked, a pinned flag is
    raised and the engine then reads the 32-bit table.  LCCRF_NBRC_SPAN (INSTRUMENTED library only, child process) lowers the width
    the check allows to 3 bits, which real lattices exceed at once.  (2) The first blur pass rides in the splat because a vertex's
    axis-0 neighbours are the adjacent ids; the neighbour search verifies that for every vertex, and LCCRF_FAST0_BREAK (same library)
    raises its flag: the pass gets its own launch back.  Same bits as the oracle through both fallbacks and without them."""
    instr = os.path.join(ROOT, "lc-crf-slam_amd", "liblccrf_hip_instr.so")
    if not os.path.exists(instr):
        subprocess.run(["make", "-C", os.path.join(ROOT, "lc-crf-slam_amd"), "-j4", "INSTRUMENT=1"], check=True, stdout=subprocess.DEVNULL)
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
N, F, d = 9000, 3, 5
pb = wl.generic_problem(N, [d], 2, seed=31)
b = pkg.BatchCRF(F, N, 2, [d], [float(pb["kernels"][0][1])])
b.set_inputs_host([N, N - 700, N], [np.repeat(pb["kernels"][0][0][None], F, 0)], unary=np.repeat(pb["unary"][None], F, 0))
b.build(); b.inference(3, True)
np.save(sys.argv[1], np.concatenate([b.probability().ravel(), b.map().astype(np.float32).ravel()]))
""" % ROOT
    N, F, d = 9000, 3, 5
    pb = wl.generic_problem(N, [d], 2, seed=31)
    want = []
    for n in (N, N - 700):
        q = dict(pb, N=n, unary=pb["unary"][:n], kernels=[(pb["kernels"][0][0][:n], pb["kernels"][0][1])])
        o = cc.setup(po.OracleCRF, q)
        o.inference_native(3, True)
        want.append((o.probability().copy(), o.map().copy()))
        o.close()
    for env, said in (({"LCCRF_NBRC_SPAN": "8"}, b"compact neighbour table abandoned"), ({"LCCRF_FAST0_BREAK": "1"}, b"first blur pass keeps its own launch"),
                      ({}, None)):
        path = os.path.join(ROOT, "gpurun_out", "nbrc_span.npy")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        r = subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, LCCRF_LIB=instr, **env), timeout=300,
                           stderr=subprocess.PIPE)
        assert (said in r.stderr) if said else (b"lccrf:" not in r.stderr), r.stderr[-400:]
        res = np.load(path)
        Q, M = res[:F * N * 2].reshape(F, N, 2), res[F * N * 2:].reshape(F, N)
        for f, n in enumerate((N, N - 700, N)):
            wq, wm = want[0] if n == N else want[1]
            assert cc.same_bits(Q[f, :n], wq) and np.array_equal(M[f, :n], wm.astype(np.float32)), (env, f)
    rel = open(os.path.join(ROOT, "lc-crf-slam_amd", "liblccrf_hip.so"), "rb").read()
    assert b"LCCRF_NBRC_SPAN" not in rel and b"LCCRF_FAST0_BREAK" not in rel


@pytest.mark.gpu
@pytest.mark.parametrize("d", [3, 4])
def test_sorted_build_leaves_wrapped_keys_to_the_hash(wl, d):
    """(d = 3: d + 1 divides 65536, so a key wrapped by 65536 still has integral grid coordinates and only the range check of
    k_points can see it -- ADVICE r4; d = 4: the integrality check of k_ecode sees it too.)  Features thousands of lattice cells wide make the int16 keys wrap (the reference's `short` arithmetic wraps the same way); such
    keys are no lattice points any more and a row-major code cannot tell them apart, but 3 dimensions of them still fit 62 bits.  The
    sorted build notices (a key whose grid coordinates are not integers) and the engine rebuilds with the hash table, which compares
    the keys themselves: automatic == the hash build, bit for bit.  (No oracle here: out-of-range float -> short conversions are
    outside what the reference's own tests pin.)"""
    N, F = 8300, 2
    pb = wl.generic_problem(N, [d], 2, seed=99, spread=6000.0)
    f = np.clip(pb["kernels"][0][0], -20000, 20000).astype(np.float32)
    res = []
    for vo in (0, 2):
        b = pkg.BatchCRF(F, N, 2, [d], [3.0])
        b.set_option(pkg.OPT_VERTEX_ORDER, vo)
        b.set_inputs_host([N] * F, [np.repeat(f[None], F, 0)], unary=np.repeat(pb["unary"][None], F, 0))
        b.build()
        b.inference(2, True)
        assert b.locality_mode() == (True, False)          # the points keep their internal order, the vertices came from the hash: the fallback fired
        res.append((b.probability(), b.map(), b.lattice_sizes(0)))
        b.close()
    assert cc.same_bits(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    # ... and frames of ordinary width keep the sorted build
    pb = wl.bilateral_problem(N, seed=5)
    b = pkg.BatchCRF(1, N, 2, [6], [10.0])
    b.set_inputs_host([N], [pb["kernels"][0][0][None]], label=pb["label"][None], conf=0.7)
    b.build()
    assert b.locality_mode() == (True, True)
    b.close()


@pytest.mark.gpu
def test_locality_mode_with_labels_and_device_inputs(po, wl):
    """Locality mode end to end the way bench.py drives C5: device-bound inputs, unaries from labels (derived in the
    internal order), 8 frames (XCD-aware grids), two builds, against the oracle -- and one adversarial frame whose
    points all share a cell (one bucket of the sort holds everything)."""
    import torch
    N, F = 12000, 8
    dev = torch.device("cuda", 0)
    pbs = [wl.bilateral_problem(N, seed=31 + i) for i in range(3)]
    odd = dict(pbs[0], kernels=[(np.tile(pbs[0]["kernels"][0][0][:1], (N, 1)).copy(), pbs[0]["kernels"][0][1])])
    frames = [pbs[i % 3] for i in range(F - 1)] + [odd]
    f = torch.from_numpy(np.stack([pb["kernels"][0][0] for pb in frames])).to(dev)
    lab = torch.from_numpy(np.stack([pb["label"] for pb in frames])).to(dev)
    npt = torch.tensor([N, N - 1, N - 2, N - 3, N, 9001, 8192, N], dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    b = pkg.BatchCRF(F, N, 2, [6], [10.0])
    b.bind_inputs_device(F, npt.data_ptr(), [f.data_ptr()], d_label=lab.data_ptr(), conf=0.7)
    b.build(); b.build()
    for rep in range(2):
        b.inference(4, True)
    Q, M, Vs = b.probability(), b.map(), b.lattice_sizes(0)
    # the same batch built by the hash table instead of the sort, and back (option set between two builds of one handle): same bits
    for vo in (2, 1, 0):
        b.set_option(pkg.OPT_VERTEX_ORDER, vo)
        b.build()
        b.inference(4, True)
        assert cc.same_bits(b.probability(), Q) and np.array_equal(b.map(), M) and np.array_equal(b.lattice_sizes(0), Vs), vo
    with pytest.raises(pkg.LccrfError):
        b.set_option(pkg.OPT_VERTEX_ORDER, 3)
    b.run(4, True)                                         # lccrf_batch_run on frames beyond the one-launch kernel: rebuild + infer
    assert b.engine() == 1 and cc.same_bits(b.probability(), Q) and np.array_equal(b.map(), M)
    for i, pb in enumerate(frames):
        n = int(npt[i])
        o = cc.setup(po.OracleCRF, dict(pb, N=n, label=pb["label"][:n], kernels=[(pb["kernels"][0][0][:n], np.float32(10.0))]))
        o.inference_native(4, True)
        assert int(Vs[i]) == o.kernel(0)["V"], i
        assert cc.same_bits(Q[i, :n], o.probability()) and np.array_equal(M[i, :n], o.map()), i
        o.close()
    b.close()


@pytest.mark.gpu
def test_locality_threshold_below_the_one_workgroup_build_keeps_the_callers_order(po, wl):
    """LCCRF_PERM_MIN (instrumented library) lowers the size from which locality mode permutes the points; frames the
    one-workgroup build still accepts (N = 2000 with two 2-D kernels) must then NOT be built by it -- that build knows nothing of the
    internal point order, and the engine would iterate permuted unaries over lattices in the caller's order.  build() + inference()
    with the threshold at 1500, under LCCRF_NO_FRAME as well: the oracle's bits."""
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
pb = wl.slam_problem(2000, seed=77)
F = 3
b = pkg.BatchCRF(F, 2000, 2, [2, 2], [float(w) for _, w in pb["kernels"]])
lab = np.repeat(pb["label"][None], F, 0)
b.set_inputs_host([2000, 1800, 2000], [np.repeat(f[None], F, 0) for f, _ in pb["kernels"]], label=lab, conf=pb["conf"])
b.build(); b.inference(5, True)
q1 = b.probability().copy()
mode = b.locality_mode()[0]
b.run(5, True)
np.save(sys.argv[1], np.concatenate([q1.ravel(), b.probability().ravel(), np.float32([mode])]))
""" % ROOT
    pb = wl.slam_problem(2000, seed=77)
    o = cc.setup(po.OracleCRF, pb)
    o.inference_native(5, True)
    want = o.probability().copy()
    o.close()
    for extra in ({}, {"LCCRF_NO_FRAME": "1"}):
        path = os.path.join("/tmp", "perm_min_%d.npy" % os.getpid())
        subprocess.run([sys.executable, "-c", code, path], check=True, env=cc.switch_env(LCCRF_PERM_MIN="1500", **extra), timeout=600)
        got = np.load(path)
        os.remove(path)
        q = got[:-1].reshape(2, 3, 2000, 2)
        assert got[-1] == 1.0, extra                        # (the switch took: the lattices were built in the internal point order)
        for which in range(2):                             # build + inference, then lccrf_batch_run
            assert cc.same_bits(q[which, 0], want) and cc.same_bits(q[which, 2], want), (extra, which)
