#!/usr/bin/env python3
"""Replay capture records (include/lccrf_record.h) through the MI355X path and check the labels.

    python tools/replay.py frames.lccrfrec [--batch 256] [--device 0] [--engine 0]

Every frame is run exactly as the call site does (Tracking.cc:1919-1930): unary from the recorded
initial labels and confidence, appearance kernel (vobservs / stdev_beta, verrors / stdev_alpha),
smoothness kernel (coord2d / point2d_stdev), n_iterations mean-field iterations, MAP.  Frames are
processed `--batch` at a time (frames in flight).  When a record carries the reference's own
results they must be reproduced exactly: labels identical, probabilities bit-identical.

Version-2 records may carry sections for the tracker steps around the CRF; each is replayed through its entry point
of include/lccrf.h and compared with the recorded outputs:
  unary    lccrf_unary_build        observation counts and rough labels EXACT; mean error / depth within 1e-6 relative
                                    (the HIP kernel equals this repository's restatement bit for bit; a real capture shows
                                    whether the restatement equals OpenCV's cv::Mat arithmetic -- max ulp is reported);
                                    the frame's CRF inputs must be the kept candidates, bit for bit
  bfmatch  lccrf_bf_match           asso EXACT (integers)
  pose     lccrf_pose_optimization  outlier flags and inlier count EXACT, pose within 1e-5 absolute per element (the 6x6
                                    normal equations are summed in a different order than g2o does; max |d| is reported)
Prints one JSON line; exit status 1 on any mismatch (a file whose origin is "synthetic" -- outputs made by this
repository's own restatements -- is replayed the same way but pins nothing: the JSON says so).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def features(fr):
    p = fr["params"]
    n = len(fr["init_label"])
    app = np.empty((n, 2), np.float32)
    app[:, 0] = fr["vobservs"] / np.float32(p["stdev_beta"])          # pairwise3d.h:41-44
    app[:, 1] = fr["verrors"] / np.float32(p["stdev_alpha"])
    smooth = (fr["coord2d"] / np.float32(p["point2d_stdev"])).astype(np.float32)   # pairwise3d.h:64-66
    return app, smooth


UNARY_REL_TOL = 1e-6
POSE_ABS_TOL = 1e-5


def _ulps(a, b):
    """max distance in float32 ulps between two arrays (0 for bit-identical)."""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    if a.size == 0:
        return 0
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return int(np.abs(ia - ib).max())


def check_sections(pkg, fr, device, out):
    """Replay the version-2 sections of one frame; accumulates into out["sections"]."""
    so = out["sections"]
    secs = fr.get("sections") or {}
    u = secs.get("unary")
    if u is not None:
        r = so.setdefault("unary", dict(frames=0, candidates=0, observs_mismatches=0, label_mismatches=0, error_max_ulp=0,
                                        depth_max_ulp=0, error_max_rel=0.0, depth_max_rel=0.0, crf_input_mismatches=0))
        params = pkg.CrfParams(*[float(fr["params"][k]) for k, _ in pkg.CrfParams._fields_])
        obs, err, dep, lab = pkg.unary_build(u["Xw"], u["obs_ptr"], u["obs_kf"], u["obs_kp"], u["kf_pose"], u["kf_intr"], u["kf_bounds"],
                                             match_prob=u["match_prob"], params=params, device=device)
        r["frames"] += 1
        r["candidates"] += len(u["fid"])
        r["observs_mismatches"] += int((obs != u["observs"]).sum())
        r["label_mismatches"] += int((lab != u["rough_label"]).sum())
        kept = u["observs"] != 0
        for name, mine, ref in (("error", err, u["error"]), ("depth", dep, u["depth"])):
            r[name + "_max_ulp"] = max(r[name + "_max_ulp"], _ulps(mine[kept], ref[kept]))
            if kept.any():
                rel = np.abs(mine[kept].astype(np.float64) - ref[kept]) / np.maximum(np.abs(ref[kept].astype(np.float64)), 1e-30)
                r[name + "_max_rel"] = max(r[name + "_max_rel"], float(rel.max()))
        # the frame's CRF arrays are the kept candidates in order (Tracking.cc:1857-1868)
        bad = 0
        if int(kept.sum()) != len(fr["init_label"]):
            bad = 1
        else:
            bad += int((u["observs"][kept].view(np.int32) != fr["vobservs"].view(np.int32)).sum())
            bad += int((u["error"][kept].view(np.int32) != fr["verrors"].view(np.int32)).sum())
            bad += int((u["depth"][kept].view(np.int32) != fr["vdepths"].view(np.int32)).sum())
            bad += int((u["rough_label"][kept] != fr["init_label"]).sum())
        r["crf_input_mismatches"] += bad
    m = secs.get("bfmatch")
    if m is not None:
        r = so.setdefault("bfmatch", dict(frames=0, queries=0, asso_mismatches=0))
        asso, _ = pkg.bf_match(m["desc_query"], m["desc_train"], ratio=m["ratio"], device=device)
        r["frames"] += 1
        r["queries"] += len(m["asso"])
        r["asso_mismatches"] += int((asso != m["asso"]).sum())
    q = secs.get("pose")
    if q is not None:
        r = so.setdefault("pose", dict(frames=0, outlier_mismatches=0, inlier_count_mismatches=0, max_abs_dT=0.0))
        T, outl, ninl = pkg.pose_optimization(q["Xw"], q["kp"], q["u_right"], q["inv_sigma2"], q["K4"], q["bf"], q["Tcw_in"],
                                              valid=q["valid"], device=device)
        r["frames"] += 1
        v = q["valid"] != 0
        r["outlier_mismatches"] += int((outl[v] != q["outlier"][v]).sum())
        r["inlier_count_mismatches"] += int(ninl != q["n_inliers"])
        r["max_abs_dT"] = max(r["max_abs_dT"], float(np.abs(T.reshape(4, 4) - q["Tcw_out"]).max()))


def sections_ok(so):
    u, m, q = so.get("unary"), so.get("bfmatch"), so.get("pose")
    ok = True
    if u:
        ok &= not (u["observs_mismatches"] or u["label_mismatches"] or u["crf_input_mismatches"])
        ok &= u["error_max_rel"] <= UNARY_REL_TOL and u["depth_max_rel"] <= UNARY_REL_TOL
    if m:
        ok &= not m["asso_mismatches"]
    if q:
        ok &= not (q["outlier_mismatches"] or q["inlier_count_mismatches"]) and q["max_abs_dT"] <= POSE_ABS_TOL
    return bool(ok)


def replay(path, batch=256, device=0, engine=0):
    pkg = importlib.import_module("lc-crf-slam_amd")
    rec = importlib.import_module("lc-crf-slam_amd.records")
    frames = list(rec.read_records(path))
    origin = rec.file_origin(path)
    out = dict(file=os.path.basename(path), frames=len(frames), points=0, checked_frames=0, label_mismatches=0,
               prob_mismatches=0, max_abs_dQ=0.0, dynamic_points=0, sections={},
               origin="synthetic (outputs from this repository's restatements: pins nothing)" if origin == rec.ORIGIN_SYNTHETIC
               else "reference")
    for fr in frames:
        check_sections(pkg, fr, device, out)
    out["sections_ok"] = sections_ok(out["sections"])
    t_gpu = 0.0
    i = 0
    while i < len(frames):
        p0 = frames[i]["params"]
        group = [frames[i]]
        while i + len(group) < len(frames) and len(group) < batch:
            nxt = frames[i + len(group)]
            same = all(nxt["params"][k] == p0[k] for k in ("w1", "w2", "confidence")) and \
                nxt["n_iterations"] == frames[i]["n_iterations"]
            if not same:
                break
            group.append(nxt)
        i += len(group)
        F = len(group)
        maxn = max(1, max(len(g["init_label"]) for g in group))
        sizes = [len(g["init_label"]) for g in group]
        feats = [np.zeros((F, maxn, 2), np.float32) for _ in range(2)]
        label = np.full((F, maxn), -1, np.int16)
        for f, g in enumerate(group):
            a, s = features(g)
            feats[0][f, :sizes[f]], feats[1][f, :sizes[f]] = a, s
            label[f, :sizes[f]] = g["init_label"]
        t0 = time.perf_counter()
        b = pkg.BatchCRF(F, maxn, 2, [2, 2], [float(p0["w1"]), float(p0["w2"])], device=device)
        b.set_engine(engine)
        b.set_inputs_host(sizes, feats, label=label, conf=float(p0["confidence"]))
        if engine == 0:
            b.run(int(group[0]["n_iterations"]), True)          # one launch per frame (lattice build + inference)
        else:
            b.build()
            b.inference(int(group[0]["n_iterations"]), True)
        M, Q = b.map(), b.probability()
        t_gpu += time.perf_counter() - t0
        b.close()
        for f, g in enumerate(group):
            n = sizes[f]
            out["points"] += n
            out["dynamic_points"] += int((M[f, :n] == 0).sum())
            if g["ref_label"] is not None:
                out["checked_frames"] += 1
                out["label_mismatches"] += int((M[f, :n] != g["ref_label"]).sum())
            if g["ref_prob"] is not None:
                d = Q[f, :n].view(np.uint32) != g["ref_prob"].view(np.uint32)
                out["prob_mismatches"] += int(d.any(-1).sum())
                if n:
                    out["max_abs_dQ"] = max(out["max_abs_dQ"], float(np.abs(Q[f, :n] - g["ref_prob"]).max()))
    out["frames_per_s_host_to_host"] = len(frames) / t_gpu if t_gpu > 0 else None
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("records")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--engine", type=int, default=0, help="0 auto, 1 streaming, 2 fused")
    a = ap.parse_args()
    out = replay(a.records, a.batch, a.device, a.engine)
    print(json.dumps(out))
    return 1 if (out["label_mismatches"] or out["prob_mismatches"] or not out["sections_ok"]) else 0


if __name__ == "__main__":
    sys.exit(main())
