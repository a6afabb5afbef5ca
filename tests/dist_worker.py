"""Worker for tests/test_sharding.py: one rank of a gloo world on CPU.

The GPU compute is replaced by the oracle here (this is a test of the distribution logic:
frame -> rank mapping, padded label gather, unsharding), which tests are allowed to do."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def labels_of(pb):
    import pyoracle as po
    import crf_cases as cc
    o = cc.setup(po.OracleCRF, pb)
    o.inference_native(5, True)
    return o.map()


def main():
    out_path = sys.argv[1]
    sizes = [int(x) for x in sys.argv[2].split(",")]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = importlib.import_module("lc-crf-slam_amd.workloads")
    sh = importlib.import_module("lc-crf-slam_amd.sharding")
    n_frames, maxN = len(sizes), max(sizes)
    S = sh.frames_per_rank(n_frames, world)
    local = torch.full((S, maxN), -7, dtype=torch.int16)
    counts = torch.full((S,), -1, dtype=torch.int32)
    for s, f in enumerate(sh.frames_of_rank(n_frames, rank, world)):
        lab = labels_of(wl.slam_problem(sizes[f], seed=500 + f))
        local[s, :sizes[f]] = torch.from_numpy(lab)
        counts[s] = sizes[f]
    labels, cnt = sh.gather_labels(local, counts)                     # one byte per label on the wire
    per_frame = sh.unshard(labels, cnt, n_frames)
    labels2, cnt2 = sh.gather_labels(local, counts, n_labels=2)       # one bit per label (the SLAM CRF is binary)
    for a, b in zip(per_frame, sh.unshard(labels2, cnt2, n_frames)):
        assert torch.equal(a, b)
    np.savez(out_path + ".rank%d.npz" % rank, **{"f%d" % i: t.numpy() for i, t in enumerate(per_frame)})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
