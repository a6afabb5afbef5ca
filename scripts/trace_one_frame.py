import importlib, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
N = 2000
frames = [wl.slam_frame(N, s) for s in range(1, 5)]
p = wl.TUM3
for rep in range(30):
    fr = frames[rep % 4]
    fa, fs = wl.appearance_features(fr), wl.smooth_features(fr)
    c = pkg.DenseCRFHIP(N, 2)
    c.set_unary_from_label(fr["init_label"], p["confidence"])
    c.add_pairwise(fa, p["w1"]); c.add_pairwise(fs, p["w2"])
    c.inference(5, True)
    m = c.map()
    c.close()
