#!/usr/bin/env python3
"""Generates tests/golden/sample_frames.lccrfrec: record-format frames (include/lccrf_record.h) whose
ref_label / ref_prob were computed by THE REFERENCE'S OWN HEADERS compiled in place (oracle/_ref),
i.e. what an instrumented LC-CRF-SLAM would have written at Tracking.cc:1930 for these inputs.
Run in the build container (needs /root/reference):  python tests/golden/make_sample_records.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402

wl = importlib.import_module("lc-crf-slam_amd.workloads")
rec = importlib.import_module("lc-crf-slam_amd.records")
assert po.have_ref(), "build oracle/_ref first (make -C oracle)"

frames = []
for fid, (n, seed) in enumerate([(1000, 41), (2000, 42), (1999, 43), (0, 44), (5, 45), (3000, 46), (2000, 47), (731, 48)]):
    fr = rec.synthetic_frame(wl, n, seed, frame_id=100 + fid)
    p = fr["params"]
    c = po.RefCRF(n, 2)
    c.set_unary_from_label(fr["init_label"], np.float32(p["confidence"]))
    app = np.stack([fr["vobservs"] / np.float32(p["stdev_beta"]), fr["verrors"] / np.float32(p["stdev_alpha"])], 1)
    c.add_pairwise(app.astype(np.float32).reshape(n, 2), np.float32(p["w1"]))
    c.add_pairwise((fr["coord2d"] / np.float32(p["point2d_stdev"])).astype(np.float32).reshape(n, 2), np.float32(p["w2"]))
    c.inference_native(5, True)
    fr["ref_label"] = c.map().copy()
    fr["ref_prob"] = c.probability().copy()
    if fid % 2:
        fr["match_prob"] = np.random.default_rng(seed).uniform(0, 1, n)
    c.close()
    frames.append(fr)
path = os.path.join(HERE, "sample_frames.lccrfrec")
print(rec.write_records(path, frames), "frames ->", path, os.path.getsize(path), "bytes")
