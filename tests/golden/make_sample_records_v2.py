#!/usr/bin/env python3
"""Writes tests/golden/sample_sections_v2.lccrfrec: a version-2 capture file (include/lccrf_record.h) whose frames carry
all three optional sections -- unary builder, BfMatch, PoseOptimization -- next to the CRF call site's arrays.

THE OUTPUTS IN THIS FILE ARE SYNTHETIC: they are produced by this repository's own CPU restatements (oracle/) on
synthetic scenes (lc-crf-slam_amd/workloads.py), because src/Tracking.cc and src/Optimizer.cc cannot be built in this
image (OpenCV, Eigen, g2o absent).  The file's header says so (origin = LCCRF_REC_ORIGIN_SYNTHETIC) and tools/replay.py
repeats it.  It documents the format and exercises writer, reader and replay; it pins nothing.  The file that pins rows
a2/a3/f1/f3/f4 is the same format written by the dump code of INTEGRATION.md section 4 inside the reference.

    python tests/golden/make_sample_records_v2.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po                                     # noqa: E402

rec = importlib.import_module("lc-crf-slam_amd.records")
wl = importlib.import_module("lc-crf-slam_amd.workloads")


def frame(n_cand, n_kf, seed, frame_id, with_prob):
    rng = np.random.default_rng([seed, 2])
    p = dict(wl.TUM3)
    sc = wl.map_point_scene(n_cand, n_kf, seed)
    n_key = n_cand + 60                                   # keypoints of the frame; the candidates are those with a map point
    fid = np.sort(rng.choice(n_key, n_cand, replace=False)).astype(np.int32)
    mp = rng.uniform(0.0, 0.4, n_cand) * (rng.random(n_cand) < 0.5) if with_prob else None
    params = po.default_params()
    obs, err, dep, lab = po.oracle_unary_build(sc["Xw"], sc["obs_ptr"], sc["obs_kf"], sc["obs_kp"], sc["kf_pose"], sc["kf_intr"],
                                               sc["kf_bounds"], match_prob=mp, params=params)
    unary = dict(Xw=sc["Xw"], fid=fid, obs_ptr=sc["obs_ptr"], obs_kf=sc["obs_kf"], obs_kp=sc["obs_kp"], kf_pose=sc["kf_pose"],
                 kf_intr=sc["kf_intr"], kf_bounds=sc["kf_bounds"], match_prob=mp, observs=obs, error=err, depth=dep, rough_label=lab)
    kept = obs != 0                                       # Tracking.cc:1857
    n = int(kept.sum())
    uv_all = (rng.random((n_key, 2)) * np.array([wl.IMG_W, wl.IMG_H])).astype(np.float32)
    fr = dict(frame_id=frame_id, n_iterations=5, params=p, vobservs=obs[kept], verrors=err[kept], vdepths=dep[kept],
              coord2d=uv_all[fid[kept]], init_label=lab[kept], match_prob=(mp[kept] if mp is not None else None))
    # the CRF itself: reference results from the reference's own headers when oracle/_ref is built, else the restatement
    cls = po.RefCRF if po.have_ref() else po.OracleCRF
    app = np.stack([fr["vobservs"] / np.float32(p["stdev_beta"]), fr["verrors"] / np.float32(p["stdev_alpha"])], 1).astype(np.float32)
    smooth = (fr["coord2d"] / np.float32(p["point2d_stdev"])).astype(np.float32)
    c = cls(n, 2)
    c.set_unary_from_label(fr["init_label"], np.float32(p["confidence"]))
    c.add_pairwise(app, np.float32(p["w1"]))
    c.add_pairwise(smooth, np.float32(p["w2"]))
    c.inference_native(5, True)
    fr["ref_label"], fr["ref_prob"] = c.map().copy(), c.probability().copy()
    c.close()
    # BfMatch: the frame's descriptors against an older frame's -- some true matches (a few flipped bits), the rest noise
    dq = rng.integers(0, 256, (n_key, 32), dtype=np.uint8)
    dt = rng.integers(0, 256, (n_key - 17, 32), dtype=np.uint8)
    hit = rng.choice(n_key - 17, 90, replace=False)
    dt[hit] = dq[rng.choice(n_key, 90, replace=False)] ^ (1 << rng.integers(0, 8, (90, 32))).astype(np.uint8) * (rng.random((90, 32)) < 0.1)
    asso, _ = po.oracle_bf_match(dq, dt, 0.6)
    bfm = dict(ratio=0.6, desc_query=dq, desc_train=dt, asso=asso)
    # PoseOptimization on the frame's keypoints: valid = has a map point and was not labelled moving by the CRF
    ps = wl.pose_scene(n_key, seed + 11)
    crf_index = np.full(n_key, -1, np.int32)
    crf_index[fid[kept]] = np.arange(n, dtype=np.int32)
    valid = np.zeros(n_key, np.uint8)
    valid[fid] = 1                                        # every candidate has a map point ...
    moving = fid[kept][fr["ref_label"] == 0]
    valid[moving] = 0                                     # ... until the CRF nulls the moving ones (Tracking.cc:1945-1955)
    To, outl, ninl, _ = po.oracle_pose_optimization(ps["Xw"], ps["kp"], ps["u_right"], ps["inv_sigma2"], valid, ps["K4"], ps["bf"], ps["T_init"])
    pose = dict(n_inliers=ninl, K4=ps["K4"], bf=ps["bf"], Xw=ps["Xw"], kp=ps["kp"], u_right=ps["u_right"], inv_sigma2=ps["inv_sigma2"],
                valid=valid, outlier=outl, Tcw_in=ps["T_init"], Tcw_out=To, crf_index=crf_index)
    fr["sections"] = dict(unary=unary, bfmatch=bfm, pose=pose)
    return fr


def main():
    frames = [frame(260, 9, 41, 100, True), frame(181, 6, 42, 101, False), frame(97, 12, 43, 102, True)]
    path = os.path.join(HERE, "sample_sections_v2.lccrfrec")
    rec.write_records(path, frames, origin=rec.ORIGIN_SYNTHETIC)
    print(path, os.path.getsize(path), "bytes,", len(frames), "frames")


if __name__ == "__main__":
    main()
