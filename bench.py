#!/usr/bin/env python3
"""bench.py -- CRF mean-field iterations/s on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--workload c2|c1|c3|c4|c5]

One "step" = one pass of the hot path over one batch: DenseCRF::inference(n_iter, with_map)
(densecrf_base.h:65-73: startInference + n_iter x stepInference + buildMap) for F independent
frames resident in HBM, through the C-ABI (lccrf_batch_inference).  Lattice construction +
normalisation (the PottsPotential3D ctor) is done once per batch before the timed region and
reported separately (build_ms, frames_per_s_end_to_end), as SURVEY.md section 8(d) defines the
metric.  Frames are independent, so N GPUs shard frames (weak scaling, F per rank); the only
collective is the label gather -- ONE RCCL all_gather of the bit-packed labels per step, inside
the timed region.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child
processes under torch.distributed.run, started before this process touches torch or the GPU).

Prints ONE JSON line on rank 0.  Besides the contract's keys it carries
  roofline          the dominant kernel against the roof that actually binds it (LDS for the
                    one-workgroup-per-frame engine, HBM for the streaming engine), frac <= 1
  c5                the HBM-roofline configuration (100k points, 6-D kernel) timed in the same run
  single_frame_latency_us   the plug-in surface as the tracker uses it: one frame, host to host
  cpu_baseline      the reference's own CPU path on this box's cores (1 pinned core, and all cores)
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8 TB/s spec; LDS bytes per clock per CU by instruction; 256 CUs at 2.4 GHz
HBM_PEAK_GBS = 8000.0
N_CU, CLK_HZ = 256, 2.4e9
LDS_RATE = {"read_b64": 256.0, "read_b128": 256.0, "read_b32": 128.0, "read_u16": 64.0,
            "write_b32": 64.0, "write_b64": 85.0}

DEFAULT_FRAMES = 16384         # sized for 288 GB of HBM: throughput still grows with frames in flight (4096: -5 %, 8192: -2 %, 32768: +1 %)

WORKLOADS = {
    # name: (N, n_iter, obs_cap, description)
    "c1": (1000, 5, None, "C1: 1000 keypoints, 5 iters, two 2-D kernels (TUM3.yaml), L=2"),
    "c2": (2000, 5, None, "C2: 2000 keypoints, 5 iters, two 2-D kernels (TUM3.yaml), L=2"),
    "c3": (2000, 10, 10, "C3: 2000 keypoints, 10 iters, <=10 observations/point, L=2"),
    "c4": (3000, 5, None, "C4: 3000 keypoints, 5 iters, frames in flight, L=2"),
    "c5": (100000, 20, None, "C5: 100k points, one 6-D bilateral kernel, 20 iters, L=2"),
    # not a BASELINE.json config: the size live SLAM mostly sees (N = matched, observed map points, "typically hundreds")
    "n500": (500, 5, None, "N500: 500 keypoints (live-SLAM-typical), 5 iters, two 2-D kernels (TUM3.yaml), L=2"),
}
SUB_FRAMES = {"c1": 16384, "c3": 8192, "c4": 8192, "n500": 32768}    # frames in flight of the sub-records of the default line


def algorithmic_bytes_per_iter(N, L, dims, Vs):
    """each array counted once per pass, gathers assumed cached."""
    b = 16.0 * N * L
    for d, V in zip(dims, Vs):
        b += 12.0 * N * L + 16.0 * N * (d + 1) + 4.0 * N + 8.0 * V * L + (d + 1) * (8.0 * V * L + 8.0 * V)
    return b


def fused_lds_model(N, dims, Vs, chain0, lean=False):
    """LDS bytes one mean-field iteration of one frame moves in the fused engine, by instruction class
    (DESIGN.md section 4.5 states the same table).  L = 2, every kernel 2-D:
      P     every point stores 3 products x 2 labels per kernel (chain kernel: 6 ds_write_b32, others 3 ds_write_b64)
      S     every product is read once (chain: ds_read_b128, short rows: ds_read_b64); per vertex two u16 row
            pointers, one u16 slot index and one 8-byte value store
      blur  3 passes x per vertex: neighbour pair (b32), centre + two neighbours (3 x b64), one b64 store
      X     every point gathers 3 float2 values per kernel (ds_read_b64)
    lean (round 5, two full-size frames per CU: csrc/fused_lean.h): every kernel's neighbour pairs but the chain kernel's come from
    HBM / L2 through buffer loads, not from LDS -- they are not counted.
    Returns (total bytes, minimum LDS-pipe clocks at the per-instruction peak rates)."""
    by = {k: 0.0 for k in LDS_RATE}
    for k, (d, V) in enumerate(zip(dims, Vs)):
        E = 3.0 * N
        chain = chain0 and k == 0
        by["write_b32" if chain else "write_b64"] += 8.0 * E          # P
        by["read_b128" if chain else "read_b64"] += 8.0 * E           # S: products
        by["read_u16"] += 6.0 * V                                      # S: row[v], row[v+1], perm[v]
        by["write_b64"] += 8.0 * V                                     # S: value store
        if not lean or chain:
            by["read_b32"] += 3 * 4.0 * V                              # blur: neighbour pairs
        by["read_b64"] += 3 * 24.0 * V                                 # blur: centre + 2 neighbours
        by["write_b64"] += 3 * 8.0 * V                                 # blur: store
        by["read_b64"] += 8.0 * E                                      # X: gathers
    total = sum(by.values())
    clocks = sum(v / LDS_RATE[k] for k, v in by.items())
    return total, clocks, by


# ---------------------------------------------------------------------------------------------
# Output discipline: the driver parses the LAST stdout line and captures a bounded tail, so that line is compact
# (< 4 KB, no note strings).  Everything else -- the sub-workload records, the latency records, the notes -- is printed BEFORE it,
# one `{"record": name, ...}` line each, and the complete nested record goes to a file (`--full-json`, default bench_full.json).
# ---------------------------------------------------------------------------------------------
COMPACT_LIMIT = 4096


def _r(x, sig=6):
    """Numbers to `sig` significant digits (the line is a report, not a checkpoint), everything else unchanged."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(full):
    """The one line the driver parses, built from the full record: the contract's keys, `roofline` and `cpu_baseline` as the task
    statement defines them, the parity gate, and one number per secondary configuration.  No free text but the workload name."""
    g = lambda *p: _r(_pick(full, *p))
    roof = full.get("roofline") or {}
    line = {k: _r(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                         "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    line["config"] = {k: _r(cfg.get(k)) for k in ("workload", "frames_in_flight_per_gpu", "handles", "distinct_frames", "n_points", "n_iters",
                                                  "n_labels", "kernel_dims", "mean_lattice_vertices", "engine") if k in cfg}
    line["roofline"] = {k: _r(roof.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "launch_ms", "traffic",
                                                     "hbm_counter_frac", "frames_per_launch", "lanes_per_frame", "frames_per_cu",
                                                     "algorithmic_hbm_bytes_per_launch", "algorithmic_bytes_per_launch") if k in roof}
    if _pick(roof, "valu_issue", "frac") is not None:
        line["roofline"]["valu_issue_frac"] = g("roofline", "valu_issue", "frac")
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind", "cpu_model", "nproc", "frames_per_s_end_to_end")}
        line["cpu_baseline"]["sample"] = "%s frames x %s iters, one pinned core, %.0f s" % (
            cb.get("sample_frames", "?"), cfg.get("n_iters", "?"), cb.get("sample_seconds", 0.0))
        if cb.get("all_cores"):
            line["cpu_baseline"]["all_cores_value"] = g("cpu_baseline", "all_cores", "value")
    for k in ("label_match_vs_cpu_reference", "max_abs_dQ_vs_cpu_reference", "frames_checked", "tiles_identical",
              "build_ms_per_batch", "prepare_ms_per_batch", "frames_per_s_end_to_end", "scaling_measured", "label_gather_ok", "device", "lib_sha16"):
        if k in full:
            line[k] = _r(full[k])
    if full.get("end_to_end"):
        line["end_to_end"] = {"one_launch_ms_per_batch": g("end_to_end", "one_launch_ms_per_batch"),
                              "fallback_frames": g("end_to_end", "fallback_frames"),
                              "hbm_bytes_per_frame": g("end_to_end", "one_launch_hbm_bytes_per_frame")}
        h2h = _pick(full, "end_to_end", "host_to_host")
        if h2h:
            line["host_to_host_frames_per_s"] = _r(h2h.get("best_frames_per_s"))
            line["host_to_host_frac_of_link"] = _r(h2h.get("best_frac_of_link"))
    if "two_handles" in full:
        line["two_handles"] = {"value": g("two_handles", "value"), "ms_per_step": g("two_handles", "ms_per_step"),
                               "frames_per_s_end_to_end": g("two_handles", "frames_per_s_end_to_end"),
                               "label_match": g("two_handles", "label_match_vs_cpu_reference")}
    for sub in ("c1", "c3", "c4", "n500"):                       # the other SLAM configurations: value, LDS fraction, end to end
        if sub in full:
            ms = _pick(full, sub, "end_to_end", "one_launch_ms_per_batch")
            fr = _pick(full, sub, "frames_in_flight")
            line[sub] = {"value": g(sub, "value"), "frac": g(sub, "roofline", "frac"), "cpu_value": g(sub, "cpu_baseline", "value"),
                         "frames_per_cu": g(sub, "roofline", "frames_per_cu"),
                         "frames_per_s_end_to_end": _r(fr / (ms * 1e-3)) if (ms and fr) else None,
                         "label_match": g(sub, "label_match_vs_cpu_reference"), "max_abs_dQ": g(sub, "max_abs_dQ_vs_cpu_reference")}
    if _pick(full, "c4", "eight_frames_in_flight"):
        line["c4"]["eight_frames_us"] = g("c4", "eight_frames_in_flight", "us_per_batch_host_to_host")
        line["c4"]["eight_frames_value"] = g("c4", "eight_frames_in_flight", "value")
    if "c5" in full:
        c5 = full["c5"]
        line["c5"] = {"frames_in_flight": g("c5", "frames_in_flight"), "value": g("c5", "value"),
                      "frac": g("c5", "roofline_whole_iteration", "frac"),
                      "blur_pass_frac": g("c5", "roofline", "frac"),
                      "traffic_per_iteration": g("c5", "roofline_whole_iteration", "traffic"),
                      "label_match": g("c5", "label_match_vs_cpu_reference"), "max_abs_dQ": g("c5", "max_abs_dQ_vs_cpu_reference")}
        if "single_frame" in c5:
            line["c5_single_frame"] = {"value": g("c5", "single_frame", "value"),
                                       "frac_events": g("c5", "single_frame", "roofline_whole_iteration", "frac"),
                                       "frac_wall": g("c5", "single_frame", "frac_wall"),
                                       "object_api_frame_ms": g("c5", "single_frame", "object_api", "frame_ms_host_to_host"),
                                       "label_match": g("c5", "single_frame", "label_match_vs_cpu_reference")}
    for k, name in (("single_frame_latency_us", "single_frame_latency_us"), ("single_frame_latency_us_n500", "single_frame_latency_us_n500")):
        if k in full:
            line[name] = {"hip": g(k, "hip"), "hip_p90": g(k, "hip_p90"), "cpu_reference": g(k, "cpu_reference")}
    if "image_demo" in full:
        line["image_demo_ms"] = g("image_demo", "crf_ms_host_to_host")
        line["image_demo_known_answer"] = g("image_demo", "known_answer_reproduced")
    if "multi_gpu" in full:
        mg = full["multi_gpu"]
        line["multi_gpu"] = {"ms_per_step_by_rank": _r(mg.get("ms_per_step_by_rank")), "ranks_in_collective": mg.get("ranks_in_collective"),
                             "backend": mg.get("backend"), "label_gather_exposed_ms_per_step": g("multi_gpu", "label_gather", "exposed_ms_per_step")}
    if "full_record" in full:
        line["full_record"] = full["full_record"]
    return line


def emit(full, path=None, stream=None):
    """Prints the sub-records as their own lines, writes the complete record to `path`, and prints the compact line LAST.
    Returns the compact line's text."""
    stream = stream or sys.stdout
    nested = ("two_handles", "c1", "c3", "c4", "n500", "c5", "image_demo", "single_frame_latency_us", "single_frame_latency_us_n500",
              "end_to_end", "multi_gpu", "cpu_baseline")
    for k in nested:
        if isinstance(full.get(k), dict):
            stream.write(json.dumps(dict({"record": k}, **full[k])) + "\n")
    if isinstance(full.get("roofline"), dict):
        stream.write(json.dumps({"record": "roofline_detail", **full["roofline"]}) + "\n")
    if path:
        try:
            with open(path, "w") as fh:
                json.dump(full, fh)
            full = dict(full, full_record=os.path.relpath(path, ROOT))
        except OSError:
            pass
    text = json.dumps(compact_line(full), separators=(",", ":"))
    if len(text) >= COMPACT_LIMIT:
        raise SystemExit("bench.py: the compact line is %d bytes (limit %d)" % (len(text), COMPACT_LIMIT))
    stream.write(text + "\n")
    stream.flush()
    return text


class CudaView:
    """Zero-copy torch view of a device buffer owned by the C library."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False),
                                             version=2)


def check_distinct_frames(pbs, idx, M, Q, n_iter):
    """Parity gate as a COVER: every distinct frame of the batch against the CPU checker -- labels identical,
    max |dQ|.  Returns (frames checked, label match, max |dQ|)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    same = tot = 0
    max_dq = 0.0
    for i, pb in enumerate(pbs):
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
    return len(pbs), same / max(tot, 1), max_dq


def longest_rows(pkg, pbs, kernel=0, limit=8):
    """Mean over (up to `limit`) distinct frames of the longest splat row of `kernel` -- the number of products ONE lane adds
    strictly left to right per label and iteration (quirk Q6) -- from this library's own lattice probe (lccrf_get_lattice)."""
    import numpy as np
    rows = []
    for pb in pbs[:limit]:
        h = pkg.DenseCRFHIP(pb["N"], pb["L"])
        h.set_unary_from_label(pb["label"], pb["conf"])
        for f, w in pb["kernels"]:
            h.add_pairwise(f, w)
        rows.append(int(np.bincount(h.kernel(kernel)["offset"].ravel()).max()))
        h.close()
    return float(np.mean(rows))


def chain_floor(row, n_iter, F, N=2000):
    """Latency floor of the ordered row sums: one dependent fp32 add per product, 5.1 cycles each
    (scripts/ubench/chain.hip), the longest row once per mean-field iteration, one frame per CU at a time (two for the
    512-lane shapes of frames up to 1024 points: their chains overlap)."""
    return 5.1 * row * n_iter / CLK_HZ * F / (N_CU * (2 if N <= 1024 else 1)) * 1e3


def tiles_identical(torch, b, dev, F, N, idx):
    """... and every tiled slot of the batch against its source slot ON THE DEVICE, bit for bit (Q as int32 patterns, labels):
    with the check above this covers all F frames of the timed batch, not a sample."""
    map_ptr, prob_ptr = b.device_buffers()
    Q = torch.as_tensor(CudaView(prob_ptr, (F, N * 2), "<i4"), device=dev)
    M = torch.as_tensor(CudaView(map_ptr, (F, N), "<i2"), device=dev)
    first = {}
    for f, i in enumerate(idx):
        first.setdefault(i, f)
    src = torch.tensor([first[i] for i in idx], dtype=torch.int64, device=dev)
    ok = True
    for lo in range(0, F, 2048):                       # (chunks: the gathered copy stays small)
        hi = min(lo + 2048, F)
        ok = ok and bool(torch.equal(Q[lo:hi], Q[src[lo:hi]])) and bool(torch.equal(M[lo:hi], M[src[lo:hi]]))
    return ok


def make_batch(wl, name, F, rank, distinct):
    """F frames for this rank: `distinct` different synthetic frames, tiled."""
    import numpy as np
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


# ---------------------------------------------------------------------------------------------
# CPU baseline: the reference's own CPU path (oracle/_ref, prebuilt from the reference headers) or,
# if that is absent, the oracle port -- the checker, timed; never the product.
# ---------------------------------------------------------------------------------------------
def _cpu_frames(pbs, n_iter, budget_s, core):
    """Run whole frames (construct, unary, two PottsPotential ctors, inference, destroy) for budget_s
    seconds on one pinned core.  Returns (frames, seconds in inference, seconds in total, kind)."""
    if core is not None:
        try:
            os.sched_setaffinity(0, {core})
        except OSError:
            pass
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
    return frames, t_inf, t_all, kind


def _cpu_worker(args):
    return _cpu_frames(*args)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pbs, n_iter, budget_one=10.0, budget_all=8.0):
    """(i) one pinned core -- the reference path is single-threaded; (ii) all cores -- independent frames
    on every core this process may use, the fair comparator of the frames-in-flight number."""
    import multiprocessing as mp
    cores = sorted(os.sched_getaffinity(0))
    slim = [dict(N=pb["N"], L=pb["L"], label=pb["label"], conf=pb["conf"], kernels=pb["kernels"]) for pb in pbs[:4]]
    saved = os.sched_getaffinity(0)
    frames, t_inf, t_all, kind = _cpu_frames(slim, n_iter, budget_one, cores[-1])
    os.sched_setaffinity(0, saved)
    out = dict(value=frames * n_iter / t_inf, unit="iters/s", cores=1, kind=kind, pinned_core=cores[-1],
               nproc=len(cores), cpu_model=cpu_model(),
               sample="%d frames x %d iters of the same workload on one pinned core, inference only, %.1f s of CPU work; "
                      "end-to-end incl. lattice build: %.1f frames/s" % (frames, n_iter, t_all, frames / t_all),
               frames_per_s_end_to_end=frames / t_all, sample_frames=frames, sample_seconds=t_all)
    try:
        ctx = mp.get_context("spawn")             # fresh interpreters: nothing of this process (torch, HIP) is inherited
        with ctx.Pool(len(cores)) as pool:
            res = pool.map(_cpu_worker, [(slim, n_iter, budget_all, c) for c in cores])
        out["all_cores"] = dict(cores=len(cores),
                                value=sum(r[0] * n_iter / r[1] for r in res),       # iters/s, inference only
                                frames_per_s_end_to_end=sum(r[0] / r[2] for r in res),
                                sample="%d frames in %.0f s on %d pinned worker processes" % (sum(r[0] for r in res), budget_all, len(cores)))
    except Exception as e:                        # the 1-core figure stands on its own
        out["all_cores"] = dict(error=repr(e))
    return out


# ---------------------------------------------------------------------------------------------
# extra records
# ---------------------------------------------------------------------------------------------
def single_frame_latency(pkg, pbs, n_iter, reps=240):
    """The plug-in surface as Tracking::DynamicDetectionWithCRF uses it (src/Tracking.cc:1920-1930):
    construct, setUnaryEnergyFromLabel, two kernels, inference(5, true), getMap, destroy -- host buffers in,
    host buffers out, one frame at a time.  Median over `reps` frames, through the ctypes binding, for this
    library and for the reference's CPU path; for this library also where the time goes: host time of every call
    (`api`: everything that only stages inputs; `launch`: inference(), which queues the one kernel; `wait`: getMap(),
    i.e. launch latency + kernel + completion signal as seen by the host) and the kernel's own duration by HIP events."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po

    def one(cls, pb, native, stamps=None):
        t0 = time.perf_counter()
        c = cls(pb["N"], pb["L"])
        c.set_unary_from_label(pb["label"], pb["conf"])
        for f, w in pb["kernels"]:
            c.add_pairwise(f, w)
        t1 = time.perf_counter()
        (c.inference_native if native else c.inference)(n_iter, True)
        t2 = time.perf_counter()
        m = c.map()
        t3 = time.perf_counter()
        c.close()
        t4 = time.perf_counter()
        if stamps is not None:
            stamps.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        return (t4 - t0) * 1e6, m

    out = {"n_points": pbs[0]["N"], "n_iters": n_iter, "reps": reps,
           "what": "construct + unary + 2 kernels (lattice + norm) + inference + getMap + destroy, host to host, via ctypes"}
    ref_cls = po.RefCRF if po.have_ref() else po.OracleCRF
    for key, cls, native in (("hip", pkg.DenseCRFHIP, False), ("cpu_reference", ref_cls, True)):
        ts, stamps = [], []
        for r in range(reps + 10):
            t, m = one(cls, pbs[r % len(pbs)], native, stamps if key == "hip" else None)
            if r >= 10:
                ts.append(t)
        out[key] = float(np.median(ts))
        out[key + "_p90"] = float(np.percentile(ts, 90))
        if key == "hip":
            st = np.array(stamps[10:]) * 1e6
            med = np.median(st, axis=0)
            out["hip_breakdown_us"] = {"api_create_unary_kernels": float(med[0]), "launch_inference_call": float(med[1]),
                                       "wait_getMap": float(med[2]), "api_destroy": float(med[3])}
    # the kernel alone: the same frame as a 1-frame batch through lccrf_batch_run, HIP events around the launch
    pb = pbs[0]
    b = pkg.BatchCRF(1, pb["N"], 2, [f.shape[1] for f, _ in pb["kernels"]], [float(w) for _, w in pb["kernels"]])
    b.set_inputs_host([pb["N"]], [f[None] for f, _ in pb["kernels"]], label=pb["label"][None], conf=pb["conf"])
    ks = []
    for r in range(30):
        b.run(n_iter, True)
        ks.append(b.last_timing()["inference_ms"] * 1e3)
    b.close()
    out["hip_breakdown_us"]["kernel_k_frame_hip_events"] = float(np.median(ks[5:]))
    out["cpu_reference_kind"] = "reference" if po.have_ref() else "port"
    # The tracker is a C++ program: its figure is the headline of this record; the ctypes figure (the same calls + Python
    # and numpy time) stays beside it.
    cpp = cpp_caller_latency(pbs, reps)
    out["hip_cpp_caller"] = cpp
    out["hip_ctypes"], out["hip_ctypes_p90"] = out["hip"], out["hip_p90"]
    if "median" in cpp:
        out["hip"], out["hip_p90"] = cpp["median"], cpp["p90"]
        out["hip_measured_from"] = "C++ caller (tools/latency_cpp.cpp through include/lccrf_densecrf.hpp); hip_ctypes = the same calls through the Python binding"
    else:
        out["hip_measured_from"] = "ctypes binding (no C++ compiler found: %s)" % cpp.get("error", "?")
    return out


def cpp_caller_latency(pbs, reps):
    """The same call site from C++ (tools/latency_cpp.cpp through include/lccrf_densecrf.hpp): what the tracker -- a C++
    program -- sees, without the ctypes / numpy time of the Python figures above.  Compiled here with g++ against the
    in-tree library; product code only.  None (with the reason) if no compiler is around."""
    import shutil
    import tempfile
    import numpy as np
    if shutil.which("g++") is None:
        return {"error": "g++ not found"}
    tmp = tempfile.mkdtemp(prefix="lccrf_lat_")
    try:
        exe, inp = os.path.join(tmp, "latency_cpp"), os.path.join(tmp, "frames.bin")
        libdir = os.path.join(ROOT, "lc-crf-slam_amd")
        subprocess.run(["g++", "-std=c++14", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "latency_cpp.cpp"),
                        "-o", exe, os.path.join(libdir, "liblccrf_hip.so"), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                       check=True, capture_output=True, timeout=300)
        N = pbs[0]["N"]
        with open(inp, "wb") as f:
            f.write(np.int32(len(pbs)).tobytes() + np.int32(N).tobytes())
            for pb in pbs:
                fr = pb["frame"]
                for a, dt in ((fr["obs"], np.float32), (fr["err"], np.float32), (fr["uv"], np.float32), (fr["init_label"], np.int16)):
                    f.write(np.ascontiguousarray(a, dt).tobytes())
        r = subprocess.run([exe, inp, str(reps)], capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}
    except Exception as e:                          # the Python figures stand on their own
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def slam_subrecord(pkg, wl, torch, dev, name, steps=10, warmup=3, distinct=32, handles=1):
    """One more SLAM-shaped configuration, timed exactly like the headline (lccrf_batch_inference over F frames resident
    in HBM, wall clock over `steps` back-to-back batches, HIP events for the launch) -- the default line's C1 / C3 / C4 /
    N500 sub-records, so that every configuration of BASELINE.json is driver-timed."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    N, n_iter, _, desc = WORKLOADS[name]
    F = SUB_FRAMES[name]
    pbs, idx, feats, label, dims, weights = make_batch(wl, name, F, 0, distinct)
    d_feats = [torch.from_numpy(f).to(dev) for f in feats]
    d_label = torch.from_numpy(label).to(dev)
    d_np = torch.full((F,), N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    H = handles                                          # (as the main record: handles of F / H frames, a stream each)
    Fh = F // H
    bs, build_ms = [], 0.0
    for h in range(H):
        lo, hi = h * Fh, (h + 1) * Fh
        bh = pkg.BatchCRF(Fh, N, 2, dims, weights, device=dev.index)
        bh.bind_inputs_device(Fh, d_np[lo:hi].data_ptr(), [t[lo:hi].data_ptr() for t in d_feats], d_label=d_label[lo:hi].data_ptr(), conf=pbs[0]["conf"])
        bh.build(); bh.synchronize(); bh.build(); bh.synchronize()
        build_ms += bh.last_timing()["build_ms"]
        bh.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 0)
        bs.append(bh)
    b = bs[0]
    engine = b.engine()
    Vs = [float(np.concatenate([bh.lattice_sizes(k) for bh in bs]).astype(np.float64).mean()) for k in range(len(dims))]

    def all_of(what):                                    # (every handle on its own stream: include/lccrf.h, lccrf_batch_get_stream)
        for bh in bs:
            (bh.inference if what == "inference" else bh.run)(n_iter, True)
    for _ in range(warmup):
        all_of("inference")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        all_of("inference")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 1)
    ms = []
    for _ in range(5):
        b.inference(n_iter, True)                       # (two back to back, the SECOND one event-timed: a launch that starts on a busy GPU,
        b.inference(n_iter, True)                       #  not on one that has just idled through a host-side synchronisation)
        ms.append(b.last_timing()["inference_ms"])
    inf_ms = float(np.median(ms))                       # ONE handle's launch: Fh frames
    lanes, per_cu = b.fused_shape()
    prepare_ms = sum(bh.last_prepare()[0] for bh in bs)
    M = np.concatenate([bh.map() for bh in bs])
    Q = np.concatenate([bh.probability() for bh in bs])
    frames_checked, label_match, max_dq = check_distinct_frames(pbs, idx, M, Q, n_iter)
    tiles_ok = all(tiles_identical(torch, bh, dev, Fh, N, idx[h * Fh:(h + 1) * Fh]) for h, bh in enumerate(bs))
    b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 0)
    for _ in range(2):
        all_of("run")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(steps // 2, 3)):
        all_of("run")
    torch.cuda.synchronize()
    run_ms = (time.perf_counter() - t0) / max(steps // 2, 3) * 1e3
    b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 1)
    run_engine, fb = b.engine(), sum(bh.fallback_frames() for bh in bs)
    run_lanes, run_per_cu = b.fused_shape()
    # (the one-launch results through the same gate: every distinct frame against the CPU checker)
    _, run_match, run_dq = check_distinct_frames(pbs, idx, np.concatenate([bh.map() for bh in bs]), np.concatenate([bh.probability() for bh in bs]), n_iter)
    lds_bytes, lds_clocks, _ = fused_lds_model(N, dims, Vs, True, per_cu == 2 and N > 1024)
    row = longest_rows(pkg, pbs)
    t_floor = lds_clocks * n_iter * Fh / N_CU / CLK_HZ
    achieved = lds_bytes * n_iter * Fh / (inf_ms * 1e-3) / 1e9
    peak = lds_bytes * n_iter * Fh / t_floor / 1e9
    rec = {"workload": desc, "frames_in_flight": F, "handles": H, "value": F * n_iter / dt, "unit": "iters/s", "ms_per_step": dt * 1e3,
           "engine": {1: "streaming", 2: "fused"}.get(engine, str(engine)), "mean_lattice_vertices": Vs,
           "roofline": {"bound": "lds" if engine == 2 else "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s",
                        "frac": achieved / peak, "launch_ms": inf_ms, "lds_bytes_per_iteration_frame": lds_bytes,
                        "lanes_per_frame": lanes, "frames_per_cu": per_cu,
                        "frames_per_launch": Fh, "lds_floor_ms": t_floor * 1e3, "longest_row": row, "chain_floor_ms": chain_floor(row, n_iter, Fh, N),
                        "algorithmic_hbm_bytes_per_iteration_frame": algorithmic_bytes_per_iter(N, 2, dims, Vs)},
           "build_ms_per_batch": build_ms, "prepare_ms_per_batch": prepare_ms,
           "end_to_end": {"one_launch_ms_per_batch": run_ms, "one_launch_engine": run_engine, "fallback_frames": fb,
                          "one_launch_lanes_per_frame": run_lanes, "one_launch_frames_per_cu": run_per_cu,
                          "one_launch_label_match_vs_cpu_reference": run_match, "one_launch_max_abs_dQ_vs_cpu_reference": run_dq,
                          "frames_per_s": F / (run_ms * 1e-3), "two_kernel_ms_per_batch": build_ms + prepare_ms + inf_ms},
           "label_match_vs_cpu_reference": label_match, "max_abs_dQ_vs_cpu_reference": max_dq,
           "frames_checked": frames_checked, "tiles_identical": tiles_ok}
    # the reference's CPU path on this configuration (one pinned core, a short sample: the headline's cpu_baseline is the long one)
    try:
        cores = sorted(os.sched_getaffinity(0))
        saved = os.sched_getaffinity(0)
        slim = [dict(N=pb["N"], L=pb["L"], label=pb["label"], conf=pb["conf"], kernels=pb["kernels"]) for pb in pbs[:4]]
        frames, t_inf, t_all, kind = _cpu_frames(slim, n_iter, 2.0, cores[-1])
        os.sched_setaffinity(0, saved)
        rec["cpu_baseline"] = {"value": frames * n_iter / t_inf, "unit": "iters/s", "cores": 1, "kind": kind,
                               "frames_per_s_end_to_end": frames / t_all, "sample": "%d frames, one pinned core, %.1f s" % (frames, t_all)}
    except Exception as e:                              # (the headline's baseline stands on its own)
        rec["cpu_baseline"] = {"error": repr(e)}
    if name == "c4":
        # BASELINE config 4 as written: EIGHT 3000-keypoint frames in flight (one per GPU of an 8-GPU node; here all eight on this GPU):
        # one lccrf_batch_run (both lattice builds + inference; two workgroups per frame) + the labels, host to host
        small = pkg.BatchCRF(8, N, 2, dims, weights, device=dev.index)
        sf = [np.ascontiguousarray(f[:8]) for f in feats]
        small.set_inputs_host([N] * 8, sf, label=np.ascontiguousarray(label[:8]), conf=pbs[0]["conf"])
        for _ in range(5):
            small.run(n_iter, True); small.map()
        ts = []
        for _ in range(40):
            t0 = time.perf_counter(); small.run(n_iter, True); m8 = small.map(); ts.append(time.perf_counter() - t0)
        q8 = small.probability()
        small.close()
        _, lm8, dq8 = check_distinct_frames(pbs[:8], idx[:8], m8, q8, n_iter)
        med = float(np.median(ts))
        rec["eight_frames_in_flight"] = {"frames": 8, "us_per_batch_host_to_host": med * 1e6, "value": 8 * n_iter / med, "unit": "iters/s",
                                         "frames_per_s_end_to_end": 8 / med, "label_match_vs_cpu_reference": lm8, "max_abs_dQ_vs_cpu_reference": dq8,
                                         "what": "lccrf_batch_run (lattice builds + inference, one launch, two workgroups per frame) + lccrf_batch_get_map_host, median of 40"}
    ptag = latest_profile({"c1": "small_c1", "c4": "fused_c4"}.get(name, "none"))      # committed rocprofv3 summary of `bench.py --workload <name>`
    if ptag:
        tr = pmc_traffic(ptag, "k_fused")
        rec["roofline"].update({"profile": "profiles/%s (kernel_stats.csv: the launch duration; pmc_summary.csv: FETCH x2 + WRITE)" % ptag,
                                "traffic": tr, "hbm_counter_frac": (tr / (inf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None})
    for bh in bs:
        bh.close()
    del d_feats, d_label, d_np
    torch.cuda.empty_cache()
    return rec


def two_handles_record(pkg, wl, torch, dev, name, F, steps=20, warmup=5, distinct=64):
    """The same frames in flight held by TWO handles of F / 2 frames, each on its own stream (include/lccrf.h: lccrf_batch_get_stream):
    the tail of one launch runs under the head of the other handle's.  Wall clock over `steps` steps of two launches each; every
    distinct frame of both handles against the CPU checker."""
    import numpy as np
    N, n_iter, _, desc = WORKLOADS[name]
    pbs, idx, feats, label, dims, weights = make_batch(wl, name, F, 0, distinct)
    d_feats = [torch.from_numpy(f).to(dev) for f in feats]
    d_label = torch.from_numpy(label).to(dev)
    d_np = torch.full((F,), N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    Fh, bs = F // 2, []
    for h in range(2):
        lo, hi = h * Fh, (h + 1) * Fh
        bh = pkg.BatchCRF(Fh, N, 2, dims, weights, device=dev.index)
        bh.bind_inputs_device(Fh, d_np[lo:hi].data_ptr(), [t[lo:hi].data_ptr() for t in d_feats], d_label=d_label[lo:hi].data_ptr(), conf=pbs[0]["conf"])
        bh.build(); bh.synchronize()
        bh.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 0)
        bs.append(bh)
    dts = []
    for rep in range(2):
        for _ in range(warmup):
            for bh in bs:
                bh.inference(n_iter, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for bh in bs:
                bh.inference(n_iter, True)
        torch.cuda.synchronize()
        dts.append((time.perf_counter() - t0) / steps)
    dt = min(dts)
    M = np.concatenate([bh.map() for bh in bs])
    Q = np.concatenate([bh.probability() for bh in bs])
    fc, lm, dq = check_distinct_frames(pbs[:distinct], idx, M, Q, n_iter)
    # ... and end to end (one launch per frame: both lattice builds + inference) on the same two handles
    n_run = max(steps // 2, 3)
    for _ in range(2):
        for bh in bs:
            bh.run(n_iter, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_run):
        for bh in bs:
            bh.run(n_iter, True)
    torch.cuda.synchronize()
    run_dt = (time.perf_counter() - t0) / n_run
    _, rlm, rdq = check_distinct_frames(pbs[:distinct], idx, np.concatenate([bh.map() for bh in bs]), np.concatenate([bh.probability() for bh in bs]), n_iter)
    for bh in bs:
        bh.close()
    del d_feats, d_label, d_np
    torch.cuda.empty_cache()
    return {"handles": 2, "frames_per_handle": Fh, "value": F * n_iter / dt, "unit": "iters/s", "ms_per_step": dt * 1e3,
            "ms_per_step_both_runs": [d * 1e3 for d in dts], "label_match_vs_cpu_reference": lm, "max_abs_dQ_vs_cpu_reference": dq,
            "frames_per_s_end_to_end": F / run_dt, "one_launch_ms_per_step": run_dt * 1e3,
            "end_to_end_label_match_vs_cpu_reference": rlm, "end_to_end_max_abs_dQ_vs_cpu_reference": rdq}


def image_demo_record(pkg, wl, reps=5):
    """The reference library's own demo (Thirdparty/DenseCRF/examples/example_cpu.cpp:79-103; its README quotes "320x240, 21 classes,
    10 iters: 225 ms" for the whole process on the author's CPU): im1 + anno1 -> res1_cpu.ppm -- 76 800 pixels, L = 21, a 2-D
    smoothness and a 5-D appearance kernel, 10 iterations -- through the object API, host arrays in / labels out, checked against the
    reference's known answer byte for byte.  The generic (L-label) kernels of the streaming engine; the appearance kernel puts
    uniformly coloured regions on single vertices (rows of up to ~50 000 entries, summed in order by a workgroup each)."""
    import numpy as np
    fn = os.path.join(ROOT, "tests", "golden", "example_im1.npz")
    if not os.path.exists(fn):
        return None
    z = np.load(fn)
    im, res, lab, colors = z["im"], z["res"], z["label"], z["colors"]
    H, W, _ = im.shape
    f_smooth, f_app = wl.image_features(W, H, 3.0), wl.image_features(W, H, 60.0, im, 20.0)
    whole, inf = [], []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        c = pkg.DenseCRFHIP(W * H, 21)
        c.set_unary_from_label(lab, 0.5)
        c.add_pairwise(f_smooth, 3.0)
        c.add_pairwise(f_app, 10.0)
        c.inference(10, True)
        m = c.map()
        whole.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        c.inference(10, True)
        m2 = c.map()
        inf.append(time.perf_counter() - t0)
        c.close()
    col = colors[m]
    out = np.stack([col & 255, (col >> 8) & 255, (col >> 16) & 255], -1).astype(np.uint8).reshape(H, W, 3)
    return {"workload": "DenseCRF example: 320x240 image, 21 classes, 2-D smoothness + 5-D appearance kernel, 10 iterations",
            "crf_ms_host_to_host": float(np.median(whole[1:])) * 1e3, "inference_ms_host_to_host": float(np.median(inf[1:])) * 1e3,
            "known_answer_reproduced": bool(np.array_equal(out, res) and np.array_equal(m, m2)),
            "reference_readme_ms_whole_process": 225.0,
            "note": "constructor + unaries from the annotation + both kernels (lattices, normalisation) + 10 iterations + map, object "
                    "API; the README's 225 ms is the whole process (image I/O included) on the author's machine -- context, not a baseline"}


def c5_object_api(pkg, pb, n_iter, reps=5):
    """BASELINE config 5 through the reference's OWN interface (include/lccrf_densecrf.hpp / DenseCRFHIP): host arrays in, labels
    out -- constructor, setUnaryEnergyFromLabel, addPairwiseEnergy (the lattice + normalisation), inference(20), map() -- the
    call sequence of src/Tracking.cc:1911-1923 on a 100 000-point frame, host to host; and the inference alone (a second call on
    the resident lattices).  Handles of >= 8192 points run inference() in locality mode (sorted build, blur passes in the splat)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    whole, inf = [], []
    M = None
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        h = pkg.DenseCRFHIP(pb["N"], pb["L"])
        h.set_unary_from_label(pb["label"], pb["conf"])
        for f, w in pb["kernels"]:
            h.add_pairwise(f, w)
        h.inference(n_iter, True)
        M = h.map()
        whole.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        h.inference(n_iter, True)
        M2 = h.map()
        inf.append(time.perf_counter() - t0)
        h.close()
    o = po.OracleCRF(pb["N"], pb["L"])
    o.set_unary_from_label(pb["label"], pb["conf"])
    for f, w in pb["kernels"]:
        o.add_pairwise(f, w)
    t0 = time.perf_counter()
    o.inference_native(n_iter, True)
    cpu_inf = time.perf_counter() - t0
    match = float((M == o.map()).mean()) * float((M2 == o.map()).mean())
    o.close()
    w, i = float(np.median(whole[1:])), float(np.median(inf[1:]))
    return {"frame_ms_host_to_host": w * 1e3, "inference_ms_host_to_host": i * 1e3, "us_per_iteration": i * 1e6 / n_iter,
            "cpu_reference_inference_ms": cpu_inf * 1e3, "label_match_vs_cpu_reference": match,
            "note": "DenseCRF object API, one 100 000-point frame, host arrays in / labels out: whole frame (constructors + lattice + "
                    "%d iterations + map) and the inference alone on the resident lattices; the CPU figure is the reference's inference "
                    "on one core of this box" % n_iter}


def c5_record(pkg, wl, torch, dev, steps=6, frames=8):
    """Config C5 (100 000 points, one 6-D kernel, V ~ 5.9e5, 20 iterations): the configuration whose working set lives in
    HBM, i.e. the one where the 8 TB/s roof is the applicable one.  `frames` frames in flight (distinct buffers: ~340 MB
    of lattice values and neighbour tables per iteration, beyond the 256 MB Infinity Cache) and, for the latency view, one."""
    import numpy as np
    N, n_iter, _, desc = WORKLOADS["c5"]
    pbs = [wl.bilateral_problem(N, s + 1) for s in range(2)]
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    out = {"workload": desc, "unit": "iters/s"}
    for F in (frames, 1, -frames):                        # (-frames: the same frames once more on the HASH build, LCCRF_OPT_VERTEX_ORDER = 2)
        hash_build = F < 0
        F = abs(F)
        idx = [i % len(pbs) for i in range(F)]
        f = torch.from_numpy(np.stack([pbs[i]["kernels"][0][0] for i in idx])).to(dev)
        lab = torch.from_numpy(np.stack([pbs[i]["label"] for i in idx])).to(dev)
        npt = torch.full((F,), N, dtype=torch.int32, device=dev)
        b = pkg.BatchCRF(F, N, 2, [6], [float(pbs[0]["kernels"][0][1])], device=dev.index)
        if hash_build:
            b.set_option(pkg.OPT_VERTEX_ORDER, 2)
        b.bind_inputs_device(F, npt.data_ptr(), [f.data_ptr()], d_label=lab.data_ptr(), conf=pbs[0]["conf"])
        b.build(); b.synchronize(); b.build(); b.synchronize()
        build_ms = b.last_timing()["build_ms"]
        V = float(b.lattice_sizes(0).astype(np.float64).mean())
        for _ in range(2):
            b.inference(n_iter, True)
        b.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            b.inference(n_iter, True)
        b.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ms = []
        for _ in range(5):
            for _ in range(3 if F < 8 else 2):          # back to back, the LAST one event-timed (HIP events of the library on its stream): steady
                b.inference(n_iter, True)               # state, as in the wall-clock loop above -- an isolated 0.8 ms inference of ONE frame right
            ms.append(b.last_timing()["inference_ms"])  # behind a host-side synchronisation reads 15-20 % long (idle -> busy ramp)
        inf_ms = float(np.median(ms))
        bytes_iter = algorithmic_bytes_per_iter(N, 2, [6], [V])
        achieved = bytes_iter * n_iter * F / (inf_ms * 1e-3) / 1e9
        rec = {"frames_in_flight": F, "value": F * n_iter / dt, "us_per_iteration_per_frame": dt * 1e6 / n_iter / F,
               "frac_wall": bytes_iter * n_iter * F / dt / 1e9 / HBM_PEAK_GBS,      # the same bytes over the WALL clock of back-to-back inferences
               "lattice_vertices": V, "build_ms_per_batch": build_ms,
               "roofline_whole_iteration": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": bytes_iter,
                                            "inference_ms": inf_ms,
                                            "note": "algorithmic bytes (SURVEY 8d: splat, 7 blur passes of 24 B per vertex, slice + softmax) x frames / "
                                                    "HIP-event time of the 20-iteration inference.  With the sorted build the first three passes "
                                                    "ride in the splat's LDS window and move no bytes of their own (6 launches per iteration with "
                                                    "many frames in flight, 4 with one): the engine moves less than the algorithmic bytes -- "
                                                    "`traffic` (PMC, per frame and iteration) says how much"}}
        if hash_build:
            # what the sorted build of locality mode (default with 8 or more frames in flight) changes: the same frames on the hash build
            blur_ms, nv = b.time_blur_pass(0, 40)
            M, Q = b.map(), b.probability()
            fc, lm, dq = check_distinct_frames(pbs[:min(F, len(pbs))], idx, M, Q, n_iter)
            out["with_hash_build"] = {
                "frames_in_flight": F, "value": rec["value"], "us_per_iteration_per_frame": rec["us_per_iteration_per_frame"],
                "build_ms_per_batch": build_ms, "roofline_whole_iteration_frac": rec["roofline_whole_iteration"]["frac"],
                "blur_pass_ms": blur_ms, "blur_pass_frac": 24.0 * nv / (blur_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "label_match_vs_cpu_reference": lm, "max_abs_dQ_vs_cpu_reference": dq,
                "note": "LCCRF_OPT_VERTEX_ORDER = 2: hash-table build, vertices numbered by first occurrence along the points' Z-order curve "
                        "(round 3's locality mode); the default above sorts the entries on the row-major code of their vertex instead"}
        elif F == frames:
            blur_ms, nv = b.time_blur_pass(0, 40)
            # (d+1) x (8 V L + 8 V) per iteration = 24 B per vertex and pass at L = 2 (values read + written once,
            # the neighbour pair read once; "gathers assumed cached").  What the lanes REQUEST is 40 B per vertex and pass
            # (centre, two neighbour values, the pair, the store): kept beside it, never as `frac`.
            blur_bytes = 24.0 * nv
            tag = latest_profile("stream_c5")
            traffic = pmc_traffic(tag, "k_blur2") if F == 8 else None
            traffic_raw = pmc_traffic(tag, "k_blur2", corrected=False) if F == 8 else None
            gbs = blur_bytes / (blur_ms * 1e-3) / 1e9
            rec["roofline"] = {"bound": "hbm", "kernel": "k_blur2 (one Jacobi blur pass over every frame), HIP events, 40 launches",
                               "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                               "traffic": traffic, "traffic_fetch_as_reported": traffic_raw,
                               "traffic_over_algorithmic": (traffic / blur_bytes) if traffic else None,
                               "traffic_source": ("profiles/%s/pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                                  "`bench.py --workload c5 --frames 8`; FETCH_SIZE x2 per the guide's gfx950 "
                                                  "correction, calibration of that factor for this access mix: profiles/%s/fetch_calibration.md)"
                                                  % (tag, tag)) if traffic else None,
                               "launch_ms": blur_ms, "algorithmic_bytes_per_launch": blur_bytes,
                               "requested_bytes_per_launch_gathers_included": 40.0 * nv,
                               "requested_gbs_gathers_included": 40.0 * nv / (blur_ms * 1e-3) / 1e9}
            M, Q = b.map(), b.probability()
            fc, lm, dq = check_distinct_frames(pbs[:min(F, len(pbs))], idx, M, Q, n_iter)
            rec["label_match_vs_cpu_reference"], rec["max_abs_dQ_vs_cpu_reference"] = lm, dq
            rec["frames_checked"], rec["tiles_identical"] = fc, tiles_identical(torch, b, dev, F, N, idx)
            tr = iteration_traffic(tag) if F == 8 else None
            if tr:                                      # (per frame, like the algorithmic bytes)
                tr["traffic"] /= F
                rec["roofline_whole_iteration"].update(tr, traffic_over_algorithmic=tr["traffic"] / bytes_iter)
            out.update(rec)
        else:
            M, Q = b.map(), b.probability()
            fc, lm, dq = check_distinct_frames(pbs[:1], idx, M, Q, n_iter)
            rec["label_match_vs_cpu_reference"], rec["max_abs_dQ_vs_cpu_reference"], rec["frames_checked"] = lm, dq, fc
            tr = iteration_traffic(latest_profile("stream_c5_f1"))
            if tr:
                rec["roofline_whole_iteration"].update(tr, traffic_over_algorithmic=tr["traffic"] / bytes_iter)
            rec["object_api"] = c5_object_api(pkg, pbs[0], n_iter)
            out["single_frame"] = rec
        b.close()
        del f, lab, npt
        torch.cuda.empty_cache()
    return out



# ---------------------------------------------------------------------------------------------
# SURVEY 8(d)'s end to end, host to host: feature upload + lattice build + norm + inference + label download, pipelined.
# The reference pays its per-frame cost from host arrays to host labels (src/Tracking.cc:1919-1930); here batches of B frames go
# through lccrf_batch_set_inputs_host_async -> lccrf_batch_run -> lccrf_batch_download_async on three handles used round-robin, so
# that batch i+1 is staged and uploaded under batch i's kernel while batch i-1's labels travel back.
# ---------------------------------------------------------------------------------------------
def link_peak(torch, dev, mb=256, reps=5):
    """Measured rate of the host link with large pinned copies, GB/s (h2d, d2h)."""
    host = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    devb = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    out = []
    for up in (True, False):
        best = 0.0
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if up:
                devb.copy_(host, non_blocking=True)
            else:
                host.copy_(devb, non_blocking=True)
            e1.record()
            e1.synchronize()
            best = max(best, (mb << 20) / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        out.append(best)
    return out


def host_to_host_record(pkg, wl, torch, dev, name="c2", batch_sizes=(256, 4096), distinct=64, target_s=1.0):
    """end_to_end.host_to_host: frames/s from host arrays to host label bits.  The figures are those of a C++ host
    (tools/host_pipeline.cpp, compiled here with g++ against the in-tree library: the tracker / a replay tool is a C++ program);
    the same pipeline driven from Python through ctypes is reported beside them, and the label bits of both are compared with
    the synchronous path's."""
    import shutil
    import tempfile
    import numpy as np
    N, n_iter, cap, _ = WORKLOADS[name]
    pbs = [wl.slam_problem(N, 1 + i, obs_cap=cap) for i in range(distinct)]
    dims = [2, 2]
    weights = [float(pbs[0]["kernels"][k][1]) for k in range(2)]
    conf = float(pbs[0]["conf"])
    h2d, d2h = link_peak(torch, dev)
    BC = pkg.BatchCRF
    words = (N + 63) // 64
    rec = {"link_peak_GBs": {"h2d": h2d, "d2h": d2h, "how": "256 MB pinned copies, best of 5, HIP events"},
           "note": "frames/s, host arrays in -> host label bits out.  Per batch: lccrf_batch_set_inputs_host_async (features 2 x 16 KB + labels "
                   "4 KB per frame) -> lccrf_batch_run (ONE launch per frame: both lattices, norms, %d iterations, MAP) -> "
                   "lccrf_batch_download_async (label bits, 256 B per frame), on several handles round-robin so that batch i+1 is staged "
                   "and uploaded under batch i's kernel.  pageable = the caller's arrays are ordinary memory (copied into the batch's pinned "
                   "staging by a pool of 16 host threads; the caller's buffers are free on return), pinned = LCCRF_HOST_PINNED (the DMA reads "
                   "the caller's pinned arrays), serial = one handle, the synchronous lccrf_batch_set_inputs_host + run + "
                   "lccrf_batch_get_map_host.  pcie_bound = link_peak h2d / upload bytes per frame." % n_iter}
    tmp = tempfile.mkdtemp(prefix="lccrf_h2h_")
    exe = None
    try:
        if shutil.which("g++"):
            exe, libdir = os.path.join(tmp, "host_pipeline"), os.path.join(ROOT, "lc-crf-slam_amd")
            r = subprocess.run(["g++", "-std=c++14", "-O2", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                                os.path.join(ROOT, "tools", "host_pipeline.cpp"), "-o", exe, "-L" + libdir, "-l:liblccrf_hip.so",
                                "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                               capture_output=True, text=True, timeout=300)
            if r.returncode:
                rec["cpp_build_error"], exe = r.stderr[-300:], None
        inp = os.path.join(tmp, "frames.bin")
        with open(inp, "wb") as f:
            f.write(np.array([distinct, N, n_iter], np.int32).tobytes() + np.array([weights[0], weights[1], conf], np.float32).tobytes())
            for pb in pbs:
                f.write(np.ascontiguousarray(pb["kernels"][0][0], np.float32).tobytes())
                f.write(np.ascontiguousarray(pb["kernels"][1][0], np.float32).tobytes())
                f.write(np.ascontiguousarray(pb["label"], np.int16).tobytes())
        for B in batch_sizes:
            idx = [i % distinct for i in range(B)]
            feats = [np.ascontiguousarray(np.stack([pbs[i]["kernels"][k][0] for i in idx])) for k in range(2)]
            label = np.ascontiguousarray(np.stack([pbs[i]["label"] for i in idx]))
            npts = np.full(B, N, np.int32)
            up_bytes = sum(f.nbytes for f in feats) + label.nbytes + npts.nbytes
            nh, copy_threads = 4, 16                         # (scripts/gpu_h2h.sh matrix: handles x staging threads; 3-4 handles, 16 threads)
            ref = BC(B, N, 2, dims, weights)                 # the label bits of this batch from the synchronous path
            ref.set_inputs_host(npts, feats, label=label, conf=conf)
            ref.run(n_iter, True)
            m_ref = ref.map()
            ref.close()
            ref_bits = np.packbits((m_ref == 1).astype(np.uint8).reshape(B, -1), axis=1, bitorder="little")
            ref_bits = np.pad(ref_bits, ((0, 0), (0, words * 8 - ref_bits.shape[1]))).view(np.uint64)
            r = {"frames_per_batch": B, "handles": nh, "upload_bytes_per_frame": up_bytes / B, "download_bytes_per_frame": words * 8,
                 "pcie_bound_frames_per_s": h2d * 1e9 / (up_bytes / B)}
            if exe:
                for mode in ("serial", "pageable", "pinned"):
                    per = max(up_bytes / (h2d * 1e9), 1e-4) * (3 if mode == "serial" else 1.5)
                    nb = int(min(max(target_s / per, 8), 2000))
                    bits_out = os.path.join(tmp, "bits_%s.bin" % mode)
                    q = subprocess.run([exe, inp, str(B), str(nb), str(nh), mode, bits_out, str(copy_threads)], capture_output=True, text=True, timeout=600)
                    line = [l for l in q.stdout.splitlines() if l.startswith("{")]
                    d = json.loads(line[-1]) if line else {"error": (q.stderr or q.stdout)[-300:]}
                    if "frames_per_s" in d:
                        got = np.fromfile(bits_out, np.uint64).reshape(B, words)
                        d["labels_identical_to_synchronous_path"] = bool(np.array_equal(got, ref_bits))
                        d["frac_of_link_peak"] = d["upload_GBs"] / h2d
                        d["frac_of_pcie_bound"] = d["frames_per_s"] / r["pcie_bound_frames_per_s"]
                    r[mode] = d
            # the same pipeline from Python (ctypes): what bench-style callers see
            handles = [BC(B, N, 2, dims, weights) for _ in range(nh)]
            for h in handles:
                h.set_option(BC.OPT_COPY_THREADS, copy_threads)
            t_feats = [torch.from_numpy(f).pin_memory() for f in feats]
            t_label = torch.from_numpy(label).pin_memory()
            py = {}
            for mode in ("pageable", "pinned"):
                src_f = feats if mode == "pageable" else [t.numpy() for t in t_feats]
                src_l = label if mode == "pageable" else t_label.numpy()

                def pump(nb):
                    got = None
                    for i in range(nb + nh):
                        h = handles[i % nh]
                        if i >= nh:
                            got = h.wait_download(copy=False)["bits"]
                        if i < nb:
                            h.set_inputs_host_async(npts, src_f, label=src_l, conf=conf, pinned=(mode == "pinned"))
                            h.run(n_iter, True)
                            h.download_async(BC.DOWNLOAD_LABEL_BITS)
                    return got
                pump(2 * nh)                                     # staging buffers allocated, kernels loaded
                t0 = time.perf_counter()
                pump(nh)
                per = (time.perf_counter() - t0) / nh
                nb = int(min(max(0.3 / max(per, 1e-6), 2 * nh), 400))
                t0 = time.perf_counter()
                bits = pump(nb)
                dt = time.perf_counter() - t0
                py[mode] = {"frames_per_s": nb * B / dt, "batches": nb, "upload_GBs": nb * up_bytes / dt / 1e9,
                            "labels_identical_to_synchronous_path": bool(np.array_equal(bits, ref_bits))}
            for h in handles:
                h.close()
            r["python_ctypes"] = py
            rec["B%d" % B] = r
            for mode in ("pageable", "pinned"):                  # the compact line's figure: the best pipelined C++ rate (ctypes when g++ is absent)
                d = r.get(mode) if isinstance(r.get(mode), dict) and "frames_per_s" in r.get(mode, {}) else py.get(mode)
                if d and d.get("labels_identical_to_synchronous_path") and d["frames_per_s"] > rec.get("best_frames_per_s", 0.0):
                    rec["best_frames_per_s"], rec["best_mode"] = d["frames_per_s"], "%s B%d" % (mode, B)
                    rec["best_frac_of_link"] = d["frames_per_s"] / r["pcie_bound_frames_per_s"]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return rec


def latest_profile(suffix):
    """profiles/r<NN>_<suffix> of the latest round that committed one (e.g. 'fused_c2' -> 'r3_fused_c2'), or None."""
    import re
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for d in os.listdir(pdir) if os.path.isdir(pdir) else []:
        m = re.fullmatch(r"r(\d+)_" + re.escape(suffix), d)
        if m and os.path.exists(os.path.join(pdir, d, "pmc_summary.csv")) and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), d)
    return best[1] if best else None


def iteration_traffic(tag):
    """HBM bytes ONE mean-field iteration of the streaming engine moves, from a committed profile of this command line: per kernel of
    the iteration (splat, blur passes, slice) the PMC bytes per launch (pmc_summary.csv) x its launches per iteration (kernel_stats.csv:
    calls relative to the slice's, which runs once per iteration).  None unless the profile is there."""
    import csv
    d = os.path.join(ROOT, "profiles", tag or "")
    if not tag or not os.path.exists(os.path.join(d, "kernel_stats.csv")) or not os.path.exists(os.path.join(d, "pmc_summary.csv")):
        return None
    calls = {}
    for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
        calls[r["Name"]] = calls.get(r["Name"], 0) + int(r["Calls"])
    per_iter = max([c for n, c in calls.items() if "k_slice2" in n] or [0])
    if not per_iter:
        return None
    total, parts = 0.0, {}
    for r in csv.DictReader(open(os.path.join(d, "pmc_summary.csv"))):
        short = next((k for k in ("k_splat2", "k_blur2", "k_slice2") if k in r["kernel"]), None)
        n = calls.get(r["kernel"], 0)                   # (both files carry rocprofv3's full kernel name)
        if short and n:
            b = float(r["bytes_corrected"]) * n / per_iter
            total += b
            nm = r["kernel"][r["kernel"].index(short):].split("(")[0]
            parts[nm] = {"launches_per_iteration": n / per_iter, "bytes": parts.get(nm, {}).get("bytes", 0.0) + b}
    return {"traffic": total, "traffic_by_kernel": parts,
            "profile": "profiles/%s (kernel_stats.csv, pmc_summary.csv: FETCH x2 + WRITE per launch)" % tag} if total else None


def sq_counter(tag, kernel, counter):
    """Mean per launch of an SQ counter of `kernel` from the committed profiles/<tag>/sq_counters.txt (scripts/pmc_sq.sh), or None."""
    fn = os.path.join(ROOT, "profiles", tag or "", "sq_counters.txt")
    if not tag or not os.path.exists(fn):
        return None
    for line in open(fn):
        w = line.split()
        if len(w) >= 4 and w[0] == kernel and w[1] == counter and w[2] == "mean/dispatch":
            return float(w[3])
    return None


def pmc_traffic(tag, kernel="k_fused", corrected=True):
    """HBM bytes per launch of `kernel` from a committed rocprofv3 --pmc profile of THIS command line
    (profiles/<tag>/pmc_summary.csv: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE; corrected=False:
    FETCH_SIZE as reported).  Counters cannot be collected inside this process; None unless such a profile is committed."""
    import csv
    fn = os.path.join(ROOT, "profiles", tag or "", "pmc_summary.csv")
    if not tag or not os.path.exists(fn):
        return None
    # (several instances of a kernel template may match -- the self-contained first inference, the once-per-build prepare launch, the
    # steady-state kernel: the instance with the most dispatches is the one the timed region runs)
    by_name = {}
    for r in csv.DictReader(open(fn)):
        if kernel in r["kernel"]:
            e = by_name.setdefault(r["kernel"], [0, 0.0])
            e[0] = max(e[0], int(r["dispatches"]))
            e[1] += float(r["bytes_corrected" if corrected else "bytes_per_dispatch"])
    if not by_name:
        return None
    return max(by_name.values(), key=lambda e: e[0])[1] or None


# ---------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """`python bench.py --gpus N`: start the N ranks as CHILD processes.  Nothing in this (parent) process has
    imported torch or touched HIP, and it only waits: no process that initialised the GPU ever execs."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def rehearse_cpu(args, world, rank):
    """--rehearse-cpu: the rank plumbing (spawn, process group, per-step label gather, max-over-ranks timing)
    with gloo on CPU tensors and NO compute -- a test of the launcher, prints no metric."""
    import torch
    import torch.distributed as dist
    sh = importlib.import_module("lc-crf-slam_amd.sharding")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    S, W = 8, 32
    bits = torch.full((S, W), rank + 1, dtype=torch.int64)
    out = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = sh.gather_label_bits(bits)
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ok = all(bool((out[r] == r + 1).all()) for r in range(world))
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "steps": args.steps, "gathers": args.steps, "gather_ok": ok}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


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
    ap.add_argument("--no-extras", action="store_true", help="skip the c5 and single-frame-latency records")
    ap.add_argument("--serial-gather", action="store_true",
                    help="N > 1: wait for each step's label gather before the next launch instead of overlapping them")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic frames tiled into the batch")
    ap.add_argument("--rehearse-cpu", action="store_true", help="launcher test: gloo, no GPU, no compute, no metric")
    ap.add_argument("--host-to-host", action="store_true", help="only the pipelined host-to-host record (end_to_end.host_to_host)")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "bench_full.json"),
                    help="where the complete nested record goes (the LAST stdout line is the compact one the driver parses)")
    ap.add_argument("--handles", type=int, default=0,
                    help="batch handles (each on its own stream) that share the frames in flight (default 1; the default line reports the two-handle schedule as `two_handles`)")
    ap.add_argument("--lite", action="store_true",
                    help="counter-collection runs (rocprofv3 --pmc serialises every dispatch): one event-timed launch instead of "
                         "five, two one-launch batches instead of many -- the timed region itself is unchanged")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))          # before torch / HIP are touched
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to report a number for a different rank count"
                         % (args.gpus, world))
    if args.rehearse_cpu:
        sys.exit(rehearse_cpu(args, world, rank))

    import numpy as np
    import torch
    import torch.distributed as dist

    # Rehearsal knobs for a 1-GPU box (scripts/rehearse_multi.sh): every rank on one device and a
    # backend that tolerates that.  The driver's runs never set them.
    if "LCCRF_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["LCCRF_BENCH_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)                 # before the process group: RCCL binds the rank to this device
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LCCRF_BENCH_BACKEND", "nccl")                # "nccl" is RCCL on ROCm
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        assert dist.get_world_size() == args.gpus

    pkg = importlib.import_module("lc-crf-slam_amd")
    wl = importlib.import_module("lc-crf-slam_amd.workloads")
    sh = importlib.import_module("lc-crf-slam_amd.sharding")

    if args.host_to_host:
        print(json.dumps({"host_to_host": host_to_host_record(pkg, wl, torch, dev, args.workload if args.workload != "c5" else "c2")}))
        return
    name = args.workload
    N, n_iter, _, desc = WORKLOADS[name]
    F = args.frames or (1 if name == "c5" else DEFAULT_FRAMES)   # frames in flight per GPU: 64 full waves of workgroups on 256 CUs, ~11 GB
    distinct = 1 if name == "c5" else min(F, args.distinct)
    pbs, idx, feats, label, dims, weights = make_batch(wl, name, F, rank, distinct)
    L = 2
    # --handles H: the batch held by H handles of F / H frames each, every one on its own stream: a step launches all of them, so the
    # tail of one launch (its last workgroups, CUs going idle one by one: half a workgroup's 53 us on average) runs under the head of
    # the next handle's launch -- what a replay loop over many batches does anyway (tools/host_pipeline.cpp: 3-4 handles round-robin).
    # C2, same box: 1 x 16384 frames 4.83e7, 2 x 8192 5.02e7, 4 x 4096 4.88e7 iterations/s (scripts/two_handles_probe.py).  The
    # DEFAULT stays one handle: `roofline` is then the duration of the one kernel the timed region launches, and the committed rocprofv3
    # kernel stats of this command show that same duration (two overlapping launches stretch each other in a kernel trace); the
    # two-handle schedule is reported beside it (`two_handles`).
    H = args.handles or 1
    if F % H:
        raise SystemExit("--frames %d is not a multiple of --handles %d" % (F, H))
    Fh = F // H

    # inputs resident in HBM before the timed region (torch = allocator plumbing)
    d_feats = [torch.from_numpy(f).to(dev) for f in feats]
    d_label = torch.from_numpy(label).to(dev)
    d_np = torch.full((F,), N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    class Lane:                                      # one handle: its slice of the frames, its stream, its label gather
        pass
    lanes = []
    for h in range(H):
        ln = Lane()
        lo, hi = h * Fh, (h + 1) * Fh
        ln.lo, ln.hi = lo, hi
        ln.b = pkg.BatchCRF(Fh, N, L, dims, weights, device=local_rank)
        ln.b.set_engine(args.engine)
        ln.b.bind_inputs_device(Fh, d_np[lo:hi].data_ptr(), [t[lo:hi].data_ptr() for t in d_feats], d_label=d_label[lo:hi].data_ptr(),
                                conf=pbs[0]["conf"])
        ln.b.build()                                 # first build: also allocates and zeroes the lattice arrays
        ln.b.synchronize()
        ln.b.build()                                 # steady state (what a replay loop pays per batch)
        ln.b.synchronize()
        ln.build_ms = ln.b.last_timing()["build_ms"]
        # The handle's OWN stream (wrapped for torch: the label gather's copy and the collective's wait are ordered on it).  Two torch-made
        # streams may land on one of HIP's four hardware queues and then run their kernels one after the other (1.73 instead of 1.63 ms
        # per step, depending on what else the process has created: scripts/two_handles_probe.py); the handles' streams do not.
        ln.stream = torch.cuda.ExternalStream(ln.b.own_stream(), device=dev)
        bits_ptr, words = ln.b.device_label_bits()
        ln.bits_view = torch.as_tensor(CudaView(bits_ptr, (Fh, words), "<i8"), device=dev)
        # The label gather of step i overlaps the inference of step i+1 (lc-crf-slam_amd/sharding.py: OverlappedLabelGather):
        # the bits a launch wrote are copied (device to device) into one of two staging buffers on the compute stream and the
        # all_gather of that buffer runs asynchronously on the collective's own stream.
        ln.gather = sh.OverlappedLabelGather(ln.bits_view, world, serial=args.serial_gather) if world > 1 else None
        lanes.append(ln)
    b = lanes[0].b
    build_ms = sum(ln.build_ms for ln in lanes)
    engine = b.engine()
    Vs = [float(np.concatenate([ln.b.lattice_sizes(k) for ln in lanes]).astype(np.float64).mean()) for k in range(len(dims))]

    def barrier():
        for ln in lanes:
            if ln.gather is not None:               # every gather of the timed region has finished when the clock stops
                with torch.cuda.stream(ln.stream):
                    ln.gather.wait_all()
        # The host polls for the end of the queued work before the (contractual) barrier + synchronize: a thread that SLEEPS in
        # hipDeviceSynchronize wakes up tens of microseconds to a millisecond after the GPU has finished, which a 38 ms timed
        # region of 20 steps reads as up to 4 % (profiles/r5_fused_c2/bench_repeat.txt, first run of a process).
        evs = []
        for ln in lanes:
            ev = torch.cuda.Event()
            ev.record(ln.stream)
            evs.append(ev)
        while not all(ev.query() for ev in evs):
            pass
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        for ln in lanes:
            with torch.cuda.stream(ln.stream):      # the library's kernels, the staging copy and the collective's wait all go through this stream
                ln.b.inference(n_iter, True, stream=ln.stream.cuda_stream)
                if ln.gather is not None:           # the one collective of the path: the label gather, every batch (RCCL)
                    ln.gather.push()

    # the timed region records no HIP events between its launches (LCCRF_OPT_EVENT_TIMING = 0: what a replay loop that does not read
    # per-batch timings sets); the kernel's own duration is event-timed afterwards, on the same stream
    for ln in lanes:
        ln.b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 0)
    for _ in range(args.warmup):                    # (RCCL sets its rings up on first use)
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    for ln in lanes:
        ln.b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 1)
    dt = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    multi = None
    words = lanes[0].bits_view.shape[1]
    if world > 1:
        # what every rank measured (the value uses the slowest), what the collective library itself spans, and the label
        # gather's own cost: the same steps once more with gather-then-launch instead of the overlapped form
        per_rank = [torch.zeros_like(dt) for _ in range(world)]
        dist.all_gather(per_rank, dt)
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)                           # a sum over the communicator of the timed region: N iff it spans N ranks
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        main_gathers = [ln.gather for ln in lanes]
        for ln in lanes:
            ln.gather = sh.OverlappedLabelGather(ln.bits_view, world, serial=not args.serial_gather)
        for _ in range(2):
            step()
        barrier()
        t0s = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dts = torch.tensor([time.perf_counter() - t0s], dtype=torch.float64, device=dev)
        dist.all_reduce(dts, op=dist.ReduceOp.MAX)
        for ln, g in zip(lanes, main_gathers):
            ln.gather = g
        ser, ovl = (float(dt.item()), float(dts.item())) if args.serial_gather else (float(dts.item()), float(dt.item()))
        multi = {"ms_per_step_by_rank": [float(x.item()) / args.steps * 1e3 for x in per_rank],
                 "ranks_in_collective": int(ones.item()), "backend": dist.get_backend(),
                 "label_gather": {"serial_ms_per_step": ser / args.steps * 1e3, "overlapped_ms_per_step": ovl / args.steps * 1e3,
                                  "exposed_ms_per_step": (ser - ovl) / args.steps * 1e3,
                                  "bytes_per_rank": int(F * words * 8), "collectives_per_step": H}}
        if multi["ranks_in_collective"] != world:
            raise SystemExit("the collective spans %d ranks, not %d" % (multi["ranks_in_collective"], world))
    dt = float(dt.item())

    gather_ok = None
    if world > 1:                                   # every rank's slot of the gathered buffer holds that rank's labels
        gather_ok = True
        for ln in lanes:
            mine = sh.unpack_label_bits(ln.gather.last()[rank], N)
            map_ptr, _ = ln.b.device_buffers()
            gather_ok = gather_ok and bool(torch.equal(mine, torch.as_tensor(CudaView(map_ptr, (Fh, N), "<i2"), device=dev)))

    # HIP-event duration of ONE handle's inference launch, on the stream it is launched on, with nothing else on the GPU
    stream = lanes[0].stream.cuda_stream
    kernel_ms = []
    for _ in range(1 if args.lite else 5):
        if not args.lite:
            b.inference(n_iter, True, stream=stream)   # (back to back: the event-timed launch starts on a busy GPU)
        b.inference(n_iter, True, stream=stream)
        kernel_ms.append(b.last_timing()["inference_ms"])
    inf_ms = float(np.median(kernel_ms))
    inf_shape = b.fused_shape()            # (lanes per frame, frames per CU) of the inference kernel just timed
    # the fused engine's prepared launch records (include/lccrf.h: lccrf_batch_last_prepare): written once behind the build, by the
    # second inference on its lattices (inside the warm-up here) -- part of the lattice construction, reported beside build_ms
    prepare_ms = sum(ln.b.last_prepare()[0] for ln in lanes)
    prepare_runs = sum(ln.b.last_prepare()[1] for ln in lanes)

    # end to end per frame = PottsPotential ctors + inference, as the reference pays per frame: ONE launch per frame
    run_ms = None
    if name != "c5":
        n_run = 1 if args.lite else max(args.steps // 2, 3)

        def run_all():
            for ln in lanes:
                ln.b.run(n_iter, True, stream=ln.stream.cuda_stream)
        for ln in lanes:
            ln.b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 0)     # (as the timed region above: a replay loop records no events)
        for _ in range(1 if args.lite else 2):
            run_all()
        torch.cuda.synchronize()
        t0r = time.perf_counter()
        for _ in range(n_run):
            run_all()
        for ln in lanes:
            ln.b.synchronize()
        run_ms = (time.perf_counter() - t0r) / n_run * 1e3
        for ln in lanes:
            ln.b.set_option(pkg.BatchCRF.OPT_EVENT_TIMING, 1)
        run_engine, run_fallback = b.engine(), sum(ln.b.fallback_frames() for ln in lanes)
        run_shape = b.fused_shape()

    # parity gate on the timed configuration: labels vs the CPU reference path
    label_match = None
    max_dq = None
    frames_checked = tiles_ok = None
    if rank == 0 and not args.no_check:
        M = np.concatenate([ln.b.map() for ln in lanes])
        Q = np.concatenate([ln.b.probability() for ln in lanes])
        frames_checked, label_match, max_dq = check_distinct_frames(pbs[:distinct], idx, M, Q, n_iter)   # EVERY distinct frame ...
        tiles_ok = all(tiles_identical(torch, ln.b, dev, Fh, N, idx[ln.lo:ln.hi]) for ln in lanes)      # ... and every tile of each
        if H > 1:                                        # (a lane's tiles are compared among themselves: one frame across the lanes too)
            tiles_ok = tiles_ok and all(np.array_equal(Q[ln.lo + idx[ln.lo:ln.hi].index(idx[0])].view(np.int32), Q[0].view(np.int32)) for ln in lanes if idx[0] in idx[ln.lo:ln.hi])

    if rank == 0:
        total_iters = float(F) * n_iter * args.steps * world
        value = total_iters / dt
        bytes_iter = algorithmic_bytes_per_iter(N, L, dims, Vs)
        alg_launch = bytes_iter * n_iter * Fh            # (per LAUNCH: one handle's frames)
        launch_s = inf_ms * 1e-3
        if engine == 2:
            # The one-workgroup-per-frame engine keeps the mean-field state in registers and LDS: HBM carries the
            # per-frame records once per launch.  Its roof is the CU's LDS pipe.
            chain0 = True                                           # SLAM frames: the appearance kernel takes the chain path
            wg_lanes, per_cu = inf_shape
            lean = per_cu == 2 and N > 1024                         # (frames of up to 1024 points share a CU on the 137 KB plan's small form)
            lds_bytes, lds_clocks, by = fused_lds_model(N, dims, Vs, chain0, lean)
            row = longest_rows(pkg, pbs)
            lds_launch = lds_bytes * n_iter * Fh
            t_floor = lds_clocks * n_iter * Fh / N_CU / CLK_HZ      # every CU streaming at the per-instruction peak
            achieved = lds_launch / launch_s / 1e9
            peak = lds_launch / t_floor / 1e9
            ptag = latest_profile("fused_c2") if (name, F) == ("c2", DEFAULT_FRAMES) else None
            traffic = pmc_traffic(ptag)
            valu = sq_counter(ptag, "k_fused", "SQ_INSTS_VALU")      # wavefront-level vector instructions per launch (committed SQ counters)
            valu_floor = (valu * 4.0 / (N_CU * 4 * CLK_HZ)) if valu else None   # a SIMD issues one wave64 vector instruction per 4 clocks
            roof = {"bound": "lds", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                    "traffic": traffic,
                    "valu_issue": ({"instructions_per_launch": valu, "floor_ms": valu_floor * 1e3, "frac": valu_floor / launch_s,
                                    "note": "with two frames per CU the vector ALU is the busiest unit: SQ_INSTS_VALU of the committed profile "
                                            "x 4 clocks / (256 CUs x 4 SIMDs x 2.4 GHz) against the launch -- context beside the LDS roof, "
                                            "which stays the reported bound"} if valu else None),
                    "kernel": "inference launch (start + %d mean-field iterations + map), HIP events" % n_iter,
                    "launch_ms": inf_ms, "frames_per_launch": Fh, "lds_bytes_per_iteration_frame": lds_bytes, "lanes_per_frame": wg_lanes, "frames_per_cu": per_cu,
                    "lds_floor_ms": t_floor * 1e3, "longest_row": row, "chain_floor_ms": chain_floor(row, n_iter, Fh, N),
                    "chain_floor_note": "the appearance kernel's longest row is a strictly sequential fp32 sum (one lane per label): 5.1 cycles "
                                        "x longest row x n_iter per frame / 2.4 GHz x frames / 256 CUs -- the latency floor of ONE frame per CU next to the LDS "
                                        "floor (lds_floor_ms); with frames_per_cu = 2 one frame's chain runs under the other frame's point phases",
                    "lds_bytes_by_instruction": by,
                    "peak_note": "instruction-mix-weighted LDS peak: bytes / sum(bytes_i / rate_i), rates per CU and clock "
                                 "from MI355X_MICROARCH.md (ds_read_b64/b128 256, ds_read_b32 128, ds_write_b32 64, "
                                 "ds_write_b64 85 B/clk), 256 CUs at 2.4 GHz",
                    "hbm_counter_frac": (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    "traffic_source": ("profiles/%s (rocprofv3 --pmc of this command line)" % ptag) if traffic else None,
                    "algorithmic_hbm_bytes_per_launch": alg_launch,
                    "note": "SURVEY 8(d)'s byte model counts arrays that this engine keeps in LDS/registers; against HBM "
                            "the honest figure is hbm_counter_frac"}
        else:
            achieved = alg_launch / launch_s / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "kernel": "inference (start + %d mean-field iterations + map), HIP events" % n_iter,
                    "algorithmic_bytes_per_launch": alg_launch, "launch_ms": inf_ms}
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
            "config": {"workload": desc, "frames_in_flight_per_gpu": F, "handles": H, "distinct_frames": distinct, "n_points": N,
                       "n_iters": n_iter, "n_labels": L, "kernel_dims": dims, "mean_lattice_vertices": Vs,
                       "engine": {1: "streaming", 2: "fused"}.get(engine, str(engine)),
                       "sharding": "frames over ranks, no data-path collective; one RCCL all_gather of the bit-packed "
                                   "labels per handle and step (%d bytes per rank and step), %s" % (F * words * 8, "serial" if args.serial_gather else "overlapped with the next step's launch (double-buffered)")},
            "roofline": roof,
            "build_ms_per_batch": build_ms,
            "prepare_ms_per_batch": prepare_ms,      # once per build: the launch records every later inference on these lattices starts from
            "prepare_runs": prepare_runs,
            "frames_per_s_end_to_end": (F * world / (run_ms * 1e-3)) if (run_ms and run_engine == 3) else F * world / ((build_ms + prepare_ms + inf_ms) * 1e-3),
            "end_to_end": {"one_launch_ms_per_batch": run_ms, "one_launch_engine": (run_engine if run_ms else None),
                           "fallback_frames": (run_fallback if run_ms else None),
                           "one_launch_lanes_per_frame": (run_shape[0] if run_ms else None),
                           "one_launch_frames_per_cu": (run_shape[1] if run_ms else None),
                           "two_kernel_ms_per_batch": build_ms + prepare_ms + inf_ms,
                           "one_launch_hbm_bytes_per_frame": (pmc_traffic(latest_profile("fused_c2"), "k_frame") / F) if ((name, F) == ("c2", DEFAULT_FRAMES) and pmc_traffic(latest_profile("fused_c2"), "k_frame")) else None,
                           "note": "per frame: both PottsPotential3D ctors (lattice + norm) + inference(n, true); one_launch = "
                                   "lccrf_batch_run (frame_lean.hip / frame_engine.hip), wall clock over back-to-back batches; two_kernel = HIP "
                                   "events of lccrf_batch_build + lccrf_batch_inference"},
            "label_match_vs_cpu_reference": label_match,
            "max_abs_dQ_vs_cpu_reference": max_dq,
            "frames_checked": frames_checked,        # distinct frames compared with the CPU checker (all of them)
            "tiles_identical": tiles_ok,             # every one of the F slots equals its source frame bit for bit (Q and labels, on the device)
        }
        # a scaling curve needs the 1/2/4/8-GPU lines of one node side by side: this line alone never is one
        out["scaling_measured"] = False
        props = torch.cuda.get_device_properties(dev)        # which box a number comes from (profiles carry the same string)
        out["device"] = "%s pci %04x:%02x:%02x uuid %s" % (props.name, getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                                          getattr(props, "pci_device_id", 0), str(getattr(props, "uuid", ""))[:13])
        if gather_ok is not None:
            out["label_gather_ok"] = gather_ok
        if multi is not None:
            out["multi_gpu"] = multi
        if world == 1 and not args.no_extras:
            for ln in lanes:
                ln.b.close()
            del d_feats, d_label
            torch.cuda.empty_cache()
            if name == "c2":                            # every other configuration of BASELINE.json (+ the live-SLAM size) in the same line
                for sub in ("c1", "c3", "c4", "n500"):
                    out[sub] = slam_subrecord(pkg, wl, torch, dev, sub)
            if name != "c5":
                out["c5"] = c5_record(pkg, wl, torch, dev)
            if name == "c2":
                out["two_handles"] = two_handles_record(pkg, wl, torch, dev, "c2", F)
                out["image_demo"] = image_demo_record(pkg, wl)
                out["end_to_end"]["host_to_host"] = host_to_host_record(pkg, wl, torch, dev, "c2")
            lat_pbs = pbs[:8] if name not in ("c5", "n500") else [wl.slam_problem(2000, s) for s in range(1, 9)]
            out["single_frame_latency_us"] = single_frame_latency(pkg, lat_pbs, 5)
            out["single_frame_latency_us_n500"] = single_frame_latency(pkg, [wl.slam_problem(500, s) for s in range(1, 9)], 5, reps=160)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pbs, n_iter)
        emit(out, args.full_json)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
