#!/usr/bin/env python3
"""bench.py -- CRF mean-field iterations/s on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--workload c2|c1|c3|c4|c5]

One "step" = one pass of the hot path over one batch: DenseCRF::inference(n_iter, with_map)
(densecrf_base.h:65-73: startInference + n_iter x stepInference + buildMap) for F independent
frames resident in HBM, through the C-ABI (lccrf_batch_inference).  Lattice construction +
normalisation (the PottsPotential3D ctor) is done once per batch before the timed region and
reported separately (build_ms, frames_per_s_end_to_end), as SURVEY.md section 8(d) defines
the metric.  Frames are independent, so N GPUs shard frames (weak scaling, F per rank); the
only collective is the final label gather (RCCL all_gather), inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def pmc_traffic(name, engine, F):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/<tag>/pmc_summary.csv: FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE).
    Counters cannot be collected inside this process, so this is the figure of the committed
    profile of the SAME command line; None when no matching profile exists."""
    import csv
    tag = {("c2", 2, 4096): "r1_fused_c2"}.get((name, engine, F))
    fn = os.path.join(ROOT, "profiles", tag or "", "pmc_summary.csv")
    if not tag or not os.path.exists(fn):
        return None
    tot = 0.0
    for r in csv.DictReader(open(fn)):
        if "k_fused" in r["kernel"]:
            tot += float(r["bytes_corrected"])
    return tot or None

WORKLOADS = {
    # name: (N, n_iter, obs_cap, description)
    "c1": (1000, 5, None, "C1: 1000 keypoints, 5 iters, two 2-D kernels (TUM3.yaml), L=2"),
    "c2": (2000, 5, None, "C2: 2000 keypoints, 5 iters, two 2-D kernels (TUM3.yaml), L=2"),
    "c3": (2000, 10, 10, "C3: 2000 keypoints, 10 iters, <=10 observations/point, L=2"),
    "c4": (3000, 5, None, "C4: 3000 keypoints, 5 iters, frames in flight, L=2"),
    "c5": (100000, 20, None, "C5: 100k points, one 6-D bilateral kernel, 20 iters, L=2"),
}


def algorithmic_bytes_per_iter(N, L, dims, Vs):
    """each array counted once per pass, gathers assumed cached."""
    b = 16.0 * N * L
    for d, V in zip(dims, Vs):
        b += 12.0 * N * L + 16.0 * N * (d + 1) + 4.0 * N + 8.0 * V * L + (d + 1) * (8.0 * V * L + 8.0 * V)
    return b


class CudaView:
    """Zero-copy torch view of a device buffer owned by the C library."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False),
                                             version=2)


def make_batch(wl, name, F, rank, distinct):
    """F frames for this rank: `distinct` different synthetic frames, tiled."""
    N, n_iter, cap, _ = WORKLOADS[name]
    pbs = []
    for i in range(distinct):
        seed = 1000 * rank + i + 1
        pbs.append(wl.bilateral_problem(N, seed) if name == "c5" else wl.slam_problem(N, seed, obs_cap=cap))
    K = len(pbs[0]["kernels"])
    idx = [i % distinct for i in range(F)]
    feats = [np.stack([pbs[i]["kernels"][k][0] for i in idx]) for k in range(K)]
    label = np.stack([pbs[i]["label"] for i in idx])
    dims = [pbs[0]["kernels"][k][0].shape[1] for k in range(K)]
    weights = [float(pbs[0]["kernels"][k][1]) for k in range(K)]
    return pbs, idx, feats, label, dims, weights


def cpu_baseline(pbs, n_iter, budget_s=12.0):
    """The reference CPU path on this box's host cores (1 core: the reference is
    single-threaded).  oracle/_ref (the reference's own headers, prebuilt) if present,
    else the oracle port.  Bounded sample, rank 0 only.  The checker, never the product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    kind, cls = ("reference", po.RefCRF) if po.have_ref() else ("port", po.OracleCRF)
    t_inf = t_all = 0.0
    frames = 0
    t_begin = time.perf_counter()
    while time.perf_counter() - t_begin < budget_s:
        pb = pbs[frames % len(pbs)]
        t0 = time.perf_counter()
        c = cls(pb["N"], pb["L"])
        c.set_unary_from_label(pb["label"], pb["conf"])
        for f, w in pb["kernels"]:
            c.add_pairwise(f, w)
        t1 = time.perf_counter()
        c.inference_native(n_iter, True)
        t2 = time.perf_counter()
        c.close()
        t_inf += t2 - t1
        t_all += t2 - t0
        frames += 1
    return dict(value=frames * n_iter / t_inf, unit="iters/s", cores=1, kind=kind,
                sample="%d frames x %d iters of the same workload, inference only, %.1f s of CPU work; "
                       "end-to-end incl. lattice build: %.1f frames/s" % (frames, n_iter, t_all, frames / t_all),
                frames_per_s_end_to_end=frames / t_all)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=0, help="frames in flight per GPU (0 = workload default)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--engine", type=int, default=0, help="0 auto, 1 streaming, 2 fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal knobs for a 1-GPU box (scripts/rehearse_multi.sh): every rank on one device and a
    # backend that tolerates that.  The driver's runs never set them.
    if "LCCRF_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["LCCRF_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("LCCRF_BENCH_BACKEND", "nccl"),   # "nccl" is RCCL on ROCm
                                rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = importlib.import_module("lc-crf-slam_amd")
    wl = importlib.import_module("lc-crf-slam_amd.workloads")

    name = args.workload
    N, n_iter, _, desc = WORKLOADS[name]
    F = args.frames or (1 if name == "c5" else 4096)   # frames in flight per GPU: 16 full waves of workgroups on 256 CUs, ~2.8 GB
    distinct = 1 if name == "c5" else min(F, 16)
    pbs, idx, feats, label, dims, weights = make_batch(wl, name, F, rank, distinct)
    L = 2

    # inputs resident in HBM before the timed region (torch = allocator plumbing)
    d_feats = [torch.from_numpy(f).to(dev) for f in feats]
    d_label = torch.from_numpy(label).to(dev)
    d_np = torch.full((F,), N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    b = pkg.BatchCRF(F, N, L, dims, weights, device=local_rank)
    b.set_engine(args.engine)
    b.bind_inputs_device(F, d_np.data_ptr(), [t.data_ptr() for t in d_feats], d_label=d_label.data_ptr(),
                         conf=pbs[0]["conf"])
    b.build()                                        # first build: also allocates and zeroes the lattice arrays
    b.synchronize()
    b.build()                                        # steady state (what a replay loop pays per batch)
    b.synchronize()
    build_ms = b.last_timing()["build_ms"]
    engine = b.engine()
    Vs = [b.lattice_sizes(k).astype(np.float64).mean() for k in range(len(dims))]

    sh = importlib.import_module("lc-crf-slam_amd.sharding")
    d_map_ptr, _ = b.device_buffers()
    map_view = torch.as_tensor(CudaView(d_map_ptr, (F, N), "<i2"), device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        b.inference(n_iter, True)
    b.synchronize()
    if world > 1:                                   # untimed: RCCL sets its rings up on first use
        sh.gather_labels(map_view, d_np, n_labels=L)

    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b.inference(n_iter, True)
    b.synchronize()
    if world > 1:                                   # the one collective: final label gather (RCCL)
        gathered, _ = sh.gather_labels(map_view, d_np, n_labels=L)
    barrier()
    t1 = time.perf_counter()
    dt = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())

    # HIP-event duration of the inference launch(es) on the library's own stream
    for _ in range(5):
        b.inference(n_iter, True)
        kernel_ms.append(b.last_timing()["inference_ms"])
    inf_ms = float(np.median(kernel_ms))

    # parity gate on the timed configuration: labels vs the CPU reference path
    label_match = None
    max_dq = None
    if rank == 0 and not args.no_check:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        M, Q = b.map(), b.probability()
        n_chk = min(distinct, 4 if name != "c5" else 1)
        same = tot = 0
        max_dq = 0.0
        for i in range(n_chk):
            pb = pbs[i]
            o = po.OracleCRF(pb["N"], pb["L"])
            o.set_unary_from_label(pb["label"], pb["conf"])
            for f, w in pb["kernels"]:
                o.add_pairwise(f, w)
            o.inference_native(n_iter, True)
            fidx = idx.index(i)
            same += int((M[fidx] == o.map()).sum())
            tot += pb["N"]
            max_dq = max(max_dq, float(np.abs(Q[fidx] - o.probability()).max()))
            o.close()
        label_match = same / tot

    if rank == 0:
        total_iters = float(F) * n_iter * args.steps * world
        value = total_iters / dt
        bytes_iter = algorithmic_bytes_per_iter(N, L, dims, Vs)
        bytes_launch = bytes_iter * n_iter * F
        achieved = bytes_launch / (inf_ms * 1e-3) / 1e9
        out = {
            "metric": "CRF mean-field iters/sec",
            "value": value,
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc, "frames_in_flight_per_gpu": F, "n_points": N, "n_iters": n_iter,
                       "n_labels": L, "kernel_dims": dims, "mean_lattice_vertices": Vs,
                       "engine": {1: "streaming", 2: "fused"}.get(engine, str(engine)),
                       "sharding": "frames over ranks, no data-path collective; final RCCL label all_gather"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(name, engine, F),
                         "kernel": "inference (start + %d mean-field iterations + map), HIP events" % n_iter,
                         "algorithmic_bytes_per_launch": bytes_launch, "launch_ms": inf_ms,
                         "note": "SLAM-size working sets are LDS/register-resident: bound by the CU's LDS pipe and latency, not by HBM"
                                 if name != "c5" else "lattice values exceed LDS; L2/MALL-resident"},
            "build_ms_per_batch": build_ms,
            "frames_per_s_end_to_end": F * world / ((build_ms + inf_ms) * 1e-3),
            "single_frame_latency_note": "see DESIGN.md; this line is batched throughput",
            "label_match_vs_cpu_reference": label_match,
            "max_abs_dQ_vs_cpu_reference": max_dq,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pbs, n_iter)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
