import importlib, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
pkg = importlib.import_module("lc-crf-slam_amd"); wl = importlib.import_module("lc-crf-slam_amd.workloads")
dev = torch.device("cuda", 0)
N, F = 2000, 16384
pbs = [wl.slam_problem(N, 1 + i) for i in range(64)]
idx = [i % 64 for i in range(F)]
feats = [np.stack([pbs[i]["kernels"][k][0] for i in idx]) for k in range(2)]
label = np.stack([pbs[i]["label"] for i in idx])
d_feats = [torch.from_numpy(f).to(dev) for f in feats]; d_label = torch.from_numpy(label).to(dev)
d_np = torch.full((F,), N, dtype=torch.int32, device=dev); torch.cuda.synchronize()
b = pkg.BatchCRF(F, N, 2, [2, 2], [10.0, 30.0])
b.bind_inputs_device(F, d_np.data_ptr(), [t.data_ptr() for t in d_feats], d_label=d_label.data_ptr(), conf=0.7)
b.build(); b.synchronize()
for name, st in (("torch stream", torch.cuda.current_stream(dev).cuda_stream), ("own stream", None), ("torch stream", torch.cuda.current_stream(dev).cuda_stream), ("own stream", None)):
    for _ in range(3): b.inference(5, True, stream=st)
    torch.cuda.synchronize()
    for steps in (20, 100):
        t0 = time.perf_counter()
        for _ in range(steps): b.inference(5, True, stream=st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        print("%-12s steps %3d ms_per_step %.4f  (kernel by events %.4f)" % (name, steps, dt, b.last_timing()["inference_ms"]))
