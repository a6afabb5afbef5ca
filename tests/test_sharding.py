"""N > 1 path on CPU: world_size-2 gloo run of the frame sharding + label gather
(lc-crf-slam_amd/sharding.py), compared with a single-process pass over all frames."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import crf_cases as cc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sh = importlib.import_module("lc-crf-slam_amd.sharding")


def test_round_robin_partition_is_a_partition():
    for n, w in ((8, 8), (9, 2), (5, 4), (1, 8), (0, 2), (17, 3)):
        owned = [sh.frames_of_rank(n, r, w) for r in range(w)]
        assert sorted(sum(owned, [])) == list(range(n))
        assert max(len(o) for o in owned) <= sh.frames_per_rank(n, w)
        for r, o in enumerate(owned):
            assert all(f % w == r for f in o)


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_label_gather_matches_single_process(tmp_path, po, wl, world):
    sizes = [300, 0, 257, 1, 64, 199, 300]            # ragged, one empty frame, odd count
    out = str(tmp_path / "gather")
    port = 29500 + (os.getpid() % 1000) + world
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), out,
                                       ",".join(map(str, sizes))], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    expected = []
    for f, n in enumerate(sizes):
        o = cc.setup(po.OracleCRF, wl.slam_problem(n, seed=500 + f))
        o.inference_native(5, True)
        expected.append(o.map())
    for r in range(world):                              # every rank ends up with every frame's labels
        z = np.load(out + ".rank%d.npz" % r)
        for f, e in enumerate(expected):
            assert np.array_equal(z["f%d" % f], e), (r, f)
