"""Probe (GPU box): how much does the ORDER of a frame's points (which lane owns which point) matter to the C2 inference kernel?
The same frames with their points as generated (uniformly random order), sorted along a Z-order curve of the image position, and sorted
by smoothness-lattice cell.  Different inputs give different (equally valid) results; only the launch time is compared."""
import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
dev = torch.device("cuda", 0)
def morton(x, y):
    def spread(v):
        v = v.astype(np.uint32) & 0xffff
        v = (v | (v << 8)) & 0x00ff00ff; v = (v | (v << 4)) & 0x0f0f0f0f; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    return spread(x) | (spread(y) << 1)
def run(order, F=8192, name="c2"):
    pbs, idx, feats, label, dims, weights = bench.make_batch(wl, name, F, 0, 64)
    feats = [f.copy() for f in feats]; label = label.copy()
    if order != "random":
        for f in range(F):
            uv = feats[1][f]                               # smoothness features = (u, v) / 18
            if order == "zorder":
                key = morton((uv[:, 0] * 4).astype(np.int64), (uv[:, 1] * 4).astype(np.int64))
            else:
                key = np.floor(uv[:, 1]).astype(np.int64) * 4096 + np.floor(uv[:, 0]).astype(np.int64)
            p = np.argsort(key, kind="stable")
            for k in range(2): feats[k][f] = feats[k][f][p]
            label[f] = label[f][p]
    d_feats = [torch.from_numpy(f).to(dev) for f in feats]; d_label = torch.from_numpy(label).to(dev)
    N = feats[0].shape[1]
    d_np = torch.full((F,), N, dtype=torch.int32, device=dev)
    b = pkg.BatchCRF(F, N, 2, dims, weights, device=0)
    b.bind_inputs_device(F, d_np.data_ptr(), [t.data_ptr() for t in d_feats], d_label=d_label.data_ptr(), conf=pbs[0]["conf"])
    b.build(); b.synchronize()
    ms = []
    for _ in range(6):
        b.inference(5, True); b.inference(5, True)
        ms.append(b.last_timing()["inference_ms"])
    b.run(5, True); b.run(5, True); r = b.last_timing()["inference_ms"]
    print("%-4s %-8s inference launch %.4f ms  one-launch %.4f ms  (shape %s)" % (name, order, float(np.median(ms[1:])), r, b.fused_shape()))
    b.close()
for name in ("c2", "c4"):
    for order in ("random", "zorder", "cell", "random"):
        run(order, 8192 if name == "c2" else 4096, name)
