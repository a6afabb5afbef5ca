"""Worker for tests/test_sharding.py::test_rccl_label_gather_runs_with_world_size_one: ONE rank, backend "nccl" (= RCCL
on ROCm), on the GPU.  It runs the exact collective code path of `bench.py --gpus N` -- process group bound to the
device, the library's bit buffer viewed zero-copy, OverlappedLabelGather's double-buffered async
all_gather_into_tensor behind every lccrf_batch_inference launch -- so that the first multi-GPU run is not the first
time that code executes.  It proves nothing about scaling."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class CudaView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)


def main():
    serial = len(sys.argv) > 1 and sys.argv[1] == "serial"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    pkg = importlib.import_module("lc-crf-slam_amd")
    wl = importlib.import_module("lc-crf-slam_amd.workloads")
    sh = importlib.import_module("lc-crf-slam_amd.sharding")
    F, N = 96, 1500
    pbs = [wl.slam_problem(N, seed=800 + (i % 5)) for i in range(F)]
    feats = [torch.from_numpy(np.stack([pb["kernels"][k][0] for pb in pbs])).to(dev) for k in range(2)]
    label = torch.from_numpy(np.stack([pb["label"] for pb in pbs])).to(dev)
    npt = torch.full((F,), N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    b = pkg.BatchCRF(F, N, 2, [2, 2], [10.0, 30.0])
    b.bind_inputs_device(F, npt.data_ptr(), [t.data_ptr() for t in feats], d_label=label.data_ptr(), conf=0.7)
    b.build()
    ptr, words = b.device_label_bits()
    bits = torch.as_tensor(CudaView(ptr, (F, words), "<i8"), device=dev)
    g = sh.OverlappedLabelGather(bits, 1, serial=serial)
    assert g.collective
    stream = torch.cuda.current_stream(dev).cuda_stream
    for step in range(6):
        b.inference(5, True, stream=stream)
        g.push()
    g.wait_all()
    dist.barrier()
    torch.cuda.synchronize()
    got = sh.unpack_label_bits(g.last()[0], N).cpu().numpy()
    assert np.array_equal(got, b.map()), "gathered bits differ from the labels"
    assert g.steps == 6 and got.sum() > 0
    dist.destroy_process_group()
    print("rccl world-1 gather ok: %d steps, %d bytes per step, %s" % (g.steps, F * words * 8, "serial" if serial else "overlapped"))


if __name__ == "__main__":
    main()
