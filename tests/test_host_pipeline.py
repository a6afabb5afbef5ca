"""The asynchronous host path of the batch API (round 5): lccrf_batch_set_inputs_host_async / _download_async / _wait_download.

The reference pays its per-frame cost host to host (src/Tracking.cc:1919-1930: arrays on the host in, labels on the host out), so a
caller with many frames in flight has to move its inputs under the kernels.  What must hold:
  * the same bits as the synchronous path (and the oracle), with the caller's buffers overwritten the moment the call returns,
    on the batch's own stream and on a caller's stream;
  * with LCCRF_HOST_PINNED the DMA reads the caller's (pinned) memory itself;
  * three handles used round-robin (batch i+1 uploads under batch i's kernels) give every batch ITS results;
  * a frame the one-launch kernel cannot take is re-run at wait_download and the host copies are refreshed;
  * the label bits are the int16 labels.
"""
import importlib

import numpy as np
import pytest

import crf_cases as cc
from test_hip_parity import _shaped_problem

pkg = importlib.import_module("lc-crf-slam_amd")
pytestmark = pytest.mark.gpu


def _arrays(pbs, maxN):
    F = len(pbs)
    feats = [np.zeros((F, maxN, 2), np.float32) for _ in range(2)]
    label = np.full((F, maxN), -1, np.int16)
    for f, pb in enumerate(pbs):
        n = pb["N"]
        label[f, :n] = pb["label"]
        for k in range(2):
            feats[k][f, :n] = pb["kernels"][k][0]
    return np.array([pb["N"] for pb in pbs], np.int32), feats, label


def _new_batch(pbs, maxN, F=None):
    return pkg.BatchCRF(F or len(pbs), maxN, 2, [2, 2], [float(pbs[0]["kernels"][k][1]) for k in range(2)])


def _sync_reference(pbs, maxN, n_iter):
    npts, feats, label = _arrays(pbs, maxN)
    b = _new_batch(pbs, maxN)
    b.set_inputs_host(npts, feats, label=label, conf=pbs[0]["conf"])
    b.run(n_iter, True)
    q, m = b.probability(), b.map()
    b.close()
    return q, m


def _bits_of(m, npts):
    F, maxN = m.shape
    words = (maxN + 63) // 64
    out = np.zeros((F, words), np.uint64)
    for f in range(F):
        lab = np.zeros(words * 64, np.uint64)
        lab[:npts[f]] = (m[f, :npts[f]] == 1)
        out[f] = (lab.reshape(words, 64) << np.arange(64, dtype=np.uint64)).sum(1, dtype=np.uint64)
    return out


@pytest.mark.parametrize("caller_stream", [False, True])
def test_async_inputs_give_the_same_bits_and_free_the_callers_buffers_at_once(po, wl, caller_stream):
    import torch
    sizes = [2000, 0, 1, 700, 1999, 1024, 333, 1500] * 4
    pbs = [wl.slam_problem(n, seed=8100 + i) for i, n in enumerate(sizes)]
    maxN = 2000
    q_ref, m_ref = _sync_reference(pbs, maxN, 5)
    npts, feats, label = _arrays(pbs, maxN)
    b = _new_batch(pbs, maxN)
    stream = torch.cuda.Stream() if caller_stream else None
    for rep in range(3):                                   # the handle is reused: staging and device copies are overwritten
        npts, feats, label = _arrays(pbs, maxN)
        b.set_inputs_host_async(npts, feats, label=label, conf=pbs[0]["conf"])
        for a in feats:                                    # the caller's buffers are free the moment the call returns
            a.fill(np.nan)
        label.fill(7)
        npts_keep = npts.copy()
        npts.fill(-5)
        b.run(5, True, stream=stream.cuda_stream if stream else None)
        b.download_async(pkg.BatchCRF.DOWNLOAD_LABEL_BITS | pkg.BatchCRF.DOWNLOAD_MAP | pkg.BatchCRF.DOWNLOAD_PROBABILITY)
        out = b.wait_download()
        for f, n in enumerate(sizes):
            assert cc.same_bits(out["prob"][f, :n], q_ref[f, :n]), (rep, f)
            assert np.array_equal(out["map"][f, :n], m_ref[f, :n]), (rep, f)
        assert np.array_equal(out["bits"], _bits_of(m_ref, npts_keep)), rep
    # against the oracle too (three frames)
    for f in (0, 3, 5):
        o = cc.setup(po.OracleCRF, pbs[f])
        o.inference_native(5, True)
        assert cc.same_bits(out["prob"][f, :sizes[f]], o.probability())
        o.close()
    b.close()


def test_pinned_caller_memory_is_read_in_place(wl):
    import torch
    sizes = [1800, 1200, 2000, 64] * 8
    pbs = [wl.slam_problem(n, seed=8200 + i) for i, n in enumerate(sizes)]
    maxN = 2000
    q_ref, m_ref = _sync_reference(pbs, maxN, 4)
    npts, feats, label = _arrays(pbs, maxN)
    t_feats = [torch.from_numpy(a).pin_memory() for a in feats]
    t_label = torch.from_numpy(label).pin_memory()
    b = _new_batch(pbs, maxN)
    b.set_inputs_host_async(npts, [t.numpy() for t in t_feats], label=t_label.numpy(), conf=pbs[0]["conf"], pinned=True)
    b.run(4, True)
    b.download_async(pkg.BatchCRF.DOWNLOAD_MAP | pkg.BatchCRF.DOWNLOAD_PROBABILITY)
    out = b.wait_download(copy=False)
    for f, n in enumerate(sizes):
        assert cc.same_bits(out["prob"][f, :n], q_ref[f, :n]) and np.array_equal(out["map"][f, :n], m_ref[f, :n]), f
    b.close()


def test_three_handles_round_robin_keep_every_batch_apart(wl):
    """Batch i+1 is staged and uploaded while batch i's kernel runs and batch i-1's labels travel back: every batch must
    come back with its own results."""
    maxN, F, n_batches = 1000, 96, 9
    batches = [[wl.slam_problem(300 + (37 * (i * F + f)) % 700, seed=9000 + i * F + f) for f in range(F)] for i in range(4)]
    refs = [_sync_reference(pbs, maxN, 5) for pbs in batches]
    handles = [_new_batch(batches[0], maxN, F) for _ in range(3)]
    got = {}
    for i in range(n_batches + 3):
        h = handles[i % 3]
        if i >= 3:
            got[i - 3] = h.wait_download()
        if i < n_batches:
            npts, feats, label = _arrays(batches[i % 4], maxN)
            h.set_inputs_host_async(npts, feats, label=label, conf=batches[0][0]["conf"])
            h.run(5, True)
            h.download_async(pkg.BatchCRF.DOWNLOAD_LABEL_BITS | pkg.BatchCRF.DOWNLOAD_PROBABILITY)
    for i in range(n_batches):
        q_ref, m_ref = refs[i % 4]
        npts = np.array([pb["N"] for pb in batches[i % 4]], np.int32)
        for f in range(F):
            assert cc.same_bits(got[i]["prob"][f, :npts[f]], q_ref[f, :npts[f]]), (i, f)
        assert np.array_equal(got[i]["bits"], _bits_of(m_ref, npts)), i
    for h in handles:
        h.close()


def test_a_frame_that_falls_back_is_settled_by_wait_download(po, wl):
    pbs = [wl.slam_problem(900, seed=8300 + i) for i in range(40)]
    pbs[17] = _shaped_problem(wl, 900, "sparse", seed=5)    # lattices too large for the one-launch kernel: re-run at the wait
    maxN = 900
    npts, feats, label = _arrays(pbs, maxN)
    b = _new_batch(pbs, maxN)
    b.set_inputs_host_async(npts, feats, label=label, conf=pbs[0]["conf"])
    b.run(5, True)
    b.download_async(pkg.BatchCRF.DOWNLOAD_LABEL_BITS | pkg.BatchCRF.DOWNLOAD_MAP | pkg.BatchCRF.DOWNLOAD_PROBABILITY)
    out = b.wait_download()
    assert b.fallback_frames() == 1
    for f in (16, 17, 18):
        o = cc.setup(po.OracleCRF, pbs[f])
        o.inference_native(5, True)
        assert cc.same_bits(out["prob"][f, :900], o.probability()), f
        assert np.array_equal(out["map"][f, :900], o.map()), f
        o.close()
    assert np.array_equal(out["bits"], _bits_of(out["map"], npts))
    b.close()


@pytest.mark.parametrize("settler", ["fallback_frames", "synchronize", "engine", "next_inputs"])
def test_a_fallback_settled_before_wait_download_still_refreshes_the_host_copies(po, wl, settler):
    """any call between download_async and wait_download that settles the flagged frames (and clears the pending state)
    used to leave the pinned copies at their pre-re-run bits.  The re-run itself now queues the copies again."""
    pbs = [wl.slam_problem(900, seed=8300 + i) for i in range(40)]
    pbs[17] = _shaped_problem(wl, 900, "sparse", seed=5)
    maxN = 900
    npts, feats, label = _arrays(pbs, maxN)
    b = _new_batch(pbs, maxN)
    b.set_inputs_host_async(npts, feats, label=label, conf=pbs[0]["conf"])
    b.run(5, True)
    b.download_async(pkg.BatchCRF.DOWNLOAD_LABEL_BITS | pkg.BatchCRF.DOWNLOAD_MAP | pkg.BatchCRF.DOWNLOAD_PROBABILITY)
    if settler == "fallback_frames":
        assert b.fallback_frames() == 1
    elif settler == "synchronize":
        b.synchronize()
    elif settler == "engine":
        assert b.engine() == 3
    else:                                                   # the next batch's inputs arrive before the caller collects this one
        pbs2 = [wl.slam_problem(900, seed=8400 + i) for i in range(40)]
        n2, f2, l2 = _arrays(pbs2, maxN)
        b.set_inputs_host_async(n2, f2, label=l2, conf=pbs[0]["conf"])
    out = b.wait_download()
    for f in (0, 16, 17, 18, 39):
        o = cc.setup(po.OracleCRF, pbs[f])
        o.inference_native(5, True)
        assert cc.same_bits(out["prob"][f, :900], o.probability()), (settler, f)
        assert np.array_equal(out["map"][f, :900], o.map()), (settler, f)
        o.close()
    assert np.array_equal(out["bits"], _bits_of(out["map"], npts))
    b.close()


def test_pipeline_of_256_frame_batches_runs_two_frames_per_cu_and_settles_its_fallbacks(po, wl):
    """From 256 frames per batch the one-launch kernel is csrc/frame_lean.hip (two frames per CU, records in the handle's own area): two
    handles round-robin, batches of ragged full-size frames with one frame that the half-CU plan cannot take in every batch -- every
    batch comes back with its own bits, the flagged frame's included, and the handles report the shape."""
    maxN, F = 2048, 256
    base = [wl.slam_problem(1500 + 61 * i, seed=8600 + i) for i in range(8)]
    odd = _shaped_problem(wl, 1700, "sparse", seed=9)
    batches = [[odd if f == 40 + i else base[(f + i) % 8] for f in range(F)] for i in range(3)]
    refs = {}
    for pb in base + [odd]:
        o = cc.setup(po.OracleCRF, pb)
        o.inference_native(5, True)
        refs[id(pb)] = (o.probability().copy(), o.map().copy())
        o.close()
    handles = [_new_batch(batches[0], maxN, F) for _ in range(2)]
    got = {}
    for i in range(3 + 2):
        h = handles[i % 2]
        if i >= 2:
            got[i - 2] = (h.wait_download(), h.fallback_frames(), h.fused_shape())
        if i < 3:
            npts, feats, label = _arrays(batches[i], maxN)
            h.set_inputs_host_async(npts, feats, label=label, conf=base[0]["conf"])
            h.run(5, True)
            h.download_async(pkg.BatchCRF.DOWNLOAD_LABEL_BITS | pkg.BatchCRF.DOWNLOAD_PROBABILITY)
    for i in range(3):
        out, fb, shape = got[i]
        assert fb == 1 and shape == (512, 2), (i, fb, shape)
        npts = np.array([pb["N"] for pb in batches[i]], np.int32)
        m = np.zeros((F, maxN), np.int16)
        for f, pb in enumerate(batches[i]):
            q, mm = refs[id(pb)]
            assert cc.same_bits(out["prob"][f, :pb["N"]], q), (i, f)
            m[f, :pb["N"]] = mm
        assert np.array_equal(out["bits"], _bits_of(m, npts)), i
    for h in handles:
        h.close()


def test_async_path_refuses_what_it_cannot_do(wl):
    pbs = [wl.slam_problem(100, seed=1)]
    b = _new_batch(pbs, 100)
    with pytest.raises(pkg.LccrfError):
        b.wait_download()                                  # nothing queued
    with pytest.raises(pkg.LccrfError):
        b.download_async(0)
    with pytest.raises(pkg.LccrfError):
        b.download_async(64)
    b.close()
