import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
dev = torch.device("cuda", 0)
def make(F, rank=0):
    pbs, idx, feats, label, dims, weights = bench.make_batch(wl, "c2", F, rank, 64)
    d_feats = [torch.from_numpy(f).to(dev) for f in feats]; d_label = torch.from_numpy(label).to(dev)
    d_np = torch.full((F,), 2000, dtype=torch.int32, device=dev)
    b = pkg.BatchCRF(F, 2000, 2, dims, weights, device=0)
    b.bind_inputs_device(F, d_np.data_ptr(), [t.data_ptr() for t in d_feats], d_label=d_label.data_ptr(), conf=pbs[0]["conf"])
    b.build(); b.synchronize()
    b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 0)
    return b, (d_feats, d_label, d_np)
def run(handles, streams, steps=20, warm=5):
    def step():
        for h, s in zip(handles, streams):
            h.inference(5, True, stream=s.cuda_stream)
    for _ in range(warm): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps
for nh in (1, 2, 4):
    F = 16384 // nh
    hs = [make(F) for _ in range(nh)]
    ss = [torch.cuda.Stream() for _ in range(nh)]
    for rep in range(2):
        dt = run([h[0] for h in hs], ss)
        print("handles %d x %d frames: %.4f ms per step, %.4g iters/s" % (nh, F, dt * 1e3, 16384 * 5 / dt))
    for h in hs: h[0].close()
    del hs; torch.cuda.empty_cache()
