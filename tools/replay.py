#!/usr/bin/env python3
"""Replay capture records (include/lccrf_record.h) through the MI355X path and check the labels.

    python tools/replay.py frames.lccrfrec [--batch 256] [--device 0] [--engine 0]

Every frame is run exactly as the call site does (Tracking.cc:1919-1930): unary from the recorded
initial labels and confidence, appearance kernel (vobservs / stdev_beta, verrors / stdev_alpha),
smoothness kernel (coord2d / point2d_stdev), n_iterations mean-field iterations, MAP.  Frames are
processed `--batch` at a time (frames in flight).  When a record carries the reference's own
results they must be reproduced exactly: labels identical, probabilities bit-identical.
Prints one JSON line; exit status 1 on any mismatch.
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


def replay(path, batch=256, device=0, engine=0):
    pkg = importlib.import_module("lc-crf-slam_amd")
    rec = importlib.import_module("lc-crf-slam_amd.records")
    frames = list(rec.read_records(path))
    out = dict(file=os.path.basename(path), frames=len(frames), points=0, checked_frames=0, label_mismatches=0,
               prob_mismatches=0, max_abs_dQ=0.0, dynamic_points=0)
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
    return 1 if (out["label_mismatches"] or out["prob_mismatches"]) else 0


if __name__ == "__main__":
    sys.exit(main())
