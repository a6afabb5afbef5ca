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


@pytest.mark.parametrize("world,count,batch", [(2, 37, 4), (3, 20, 3), (2, 8, 8), (2, 1, 2)])
def test_replay_sharding_arithmetic_under_a_gloo_gather(tmp_path, world, count, batch):
    """include/lccrf_sharding.h -- frame -> rank, slot, round and the word of the gathered label bits, the index code of
    tools/replay_multi.cpp -- compiled with gcc and run by `world` gloo ranks on CPU: ragged frame counts (short last round, ranks
    with nothing to do), every frame found exactly once and exactly where the header says, on every rank."""
    so = str(tmp_path / "libshardmap.so")
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "sharding_map.c"),
                    "-o", so], check=True)
    port = 29700 + (os.getpid() % 1000) + 7 * world + count
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), so, str(count), str(batch), "5"], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0


def test_bench_gpus_flag_spawns_that_many_ranks():
    """VERDICT r1 / ADVICE: `python bench.py --gpus N` must produce N ranks itself (children started before
    anything touches the GPU) and gather labels every step.  --rehearse-cpu runs exactly that plumbing with
    gloo on CPU tensors, no compute and no metric."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--rehearse-cpu"],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d == {"rehearsal": True, "n_gpus": 2, "steps": 3, "gathers": 3, "gather_ok": True}
    assert "value" not in d


def test_bench_refuses_a_rank_count_that_differs_from_gpus_flag():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearse-cpu"],
                       capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "refusing" in r.stderr


def test_label_bit_packing_round_trip():
    import torch
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 2, (3, 5, 2000)).astype(np.int16)
    words = (2000 + 63) // 64
    padded = np.zeros((3, 5, words * 64), np.uint64)
    padded[..., :2000] = lab
    bits = (padded.reshape(3, 5, words, 64) << np.arange(64, dtype=np.uint64)).sum(-1).astype(np.uint64)
    t = torch.from_numpy(bits.view(np.int64))
    assert np.array_equal(sh.unpack_label_bits(t, 2000).numpy(), lab)
    assert torch.equal(sh.gather_label_bits(t[0]), t[0][None])          # world 1: a copy


def test_overlapped_gather_without_a_process_group_is_a_copy():
    import torch
    bits = torch.arange(12, dtype=torch.int64).view(3, 4)
    g = sh.OverlappedLabelGather(bits, 1)
    assert not g.collective
    for step in range(3):
        bits += 1
        g.push()
    g.wait_all()
    assert torch.equal(g.last()[0], bits) and g.steps == 3


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["overlapped", "serial"])
def test_rccl_label_gather_runs_with_world_size_one(mode):
    """the RCCL code of `bench.py --gpus N` (process group on the device, async all_gather_into_tensor on a
    view of library memory, double-buffered staging) executed once, with world size 1, in a child process."""
    port = 29700 + (os.getpid() % 200) + (0 if mode == "overlapped" else 1)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), mode], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "rccl world-1 gather ok" in r.stdout
