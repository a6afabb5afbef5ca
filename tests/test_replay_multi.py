"""tools/replay_multi.cpp -- the C++ host of the multi-GPU split: frames sharded over the
GPUs of a node (frame f -> GPU f mod G, one host thread per GPU), lccrf_batch_run per shard, ONE ncclAllGather of the bit-packed
labels per batch on librccl directly, labels compared with what the reference recorded.  Here: it compiles against
/opt/rocm/include/rccl/rccl.h and the in-tree library (CPU), and with G = 1 -- the only world size a 1-GPU box has -- it
reproduces the recorded labels and probabilities of the committed capture sample (GPU)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "lc-crf-slam_amd")
SAMPLE = os.path.join(ROOT, "tests", "golden", "sample_frames.lccrfrec")


def build_tool(out):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "replay_multi.cpp"),
           "-o", out, "-L" + LIBDIR, "-l:liblccrf_hip.so", "-lrccl", "-lpthread", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    if not os.path.exists(os.path.join(LIBDIR, "liblccrf_hip.so")):
        pytest.skip("library not built")
    return build_tool(str(tmp_path_factory.mktemp("replay_multi") / "replay_multi"))


def test_replay_multi_builds_against_rccl_and_refuses_to_run_without_a_gpu(tool):
    assert os.path.exists("/opt/rocm/include/rccl/rccl.h")
    r = subprocess.run([tool], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:                                     # no CPU path: loud failure, not a fallback
        r = subprocess.run([tool, SAMPLE], capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("batch,serial", [(3, False), (64, False), (1, False), (3, True)])
def test_replay_multi_reproduces_the_recorded_labels_with_one_gpu(tool, batch, serial):
    """batch = 3 / 1: three / eight rounds, i.e. the three-deep pipeline (upload + launch of round t, gather of t-1, check of t-2 on
    three handles) wraps around; --serial is the same work one round at a time."""
    r = subprocess.run([tool, SAMPLE, "--gpus", "1", "--batch", str(batch)] + (["--serial"] if serial else []), capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["frames"] == 8 and out["checked_frames"] == 8 and out["gpus"] == 1 and out["batch"] == batch
    assert out["label_mismatches"] == 0 and out["prob_mismatches"] == 0 and out["max_abs_dQ"] == 0.0
    assert out["label_gathers"] == (8 + batch - 1) // batch and out["points"] > 0 and 0 < out["dynamic_points"] < out["points"]
    assert out["rccl_comm_ranks"] == 1 and len(out["rank_seconds"]) == 1 and out["pipeline"].startswith("serial" if serial else "three rounds")
    # the same keys (and the same counts) as the single-process Python tool
    import importlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    rp = importlib.import_module("replay").replay(SAMPLE, batch=batch)
    for k in ("frames", "points", "checked_frames", "label_mismatches", "prob_mismatches", "dynamic_points"):
        assert out[k] == rp[k], k


@pytest.mark.gpu
def test_replay_multi_default_takes_every_gpu_of_the_node(tool):
    """G defaults to hipGetDeviceCount(): the first 8-GPU box needs no flag (here: whatever this box has)."""
    import torch
    r = subprocess.run([tool, SAMPLE, "--single-workgroup"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["gpus"] == torch.cuda.device_count() and out["label_mismatches"] == 0 and out["checked_frames"] == 8
