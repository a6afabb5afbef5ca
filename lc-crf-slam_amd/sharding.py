"""Frames-in-flight across the GPUs of a node.

One frame's CRF is globally coupled through its lattices and fits one GPU (one workgroup
at SLAM sizes), so a frame is never split.  The unit of distribution is the FRAME:

    frame f  ->  rank  f mod world_size                       (round robin)

No inter-GPU traffic happens during lattice build or inference.  The only collective is the
final label gather: every rank contributes its [frames_per_rank, max_points] int16 label
block (padded; per-frame point counts travel alongside) in ONE all_gather, one bit per label
for the binary SLAM CRF (one byte otherwise; RCCL has no 16-bit integer type).  The payload is
a few hundred bytes per frame -- latency-bound, so no ring/tree tuning and no bucketing: one call.

torch.distributed is plumbing here (backend "nccl" is RCCL on ROCm; "gloo" in CPU tests).
"""
import torch
import torch.distributed as dist


def frames_of_rank(n_frames, rank, world):
    """Global frame ids owned by `rank`."""
    return list(range(rank, n_frames, world))


def frames_per_rank(n_frames, world):
    """Slots every rank reserves (the last ranks may leave one unused)."""
    return (n_frames + world - 1) // world


def gather_labels(local_labels, local_counts, group=None, n_labels=None):
    """All-gather the per-rank label blocks.

    local_labels : int16 [S, max_points]  (S = frames_per_rank slots, unused slots arbitrary)
    local_counts : int32 [S]              points per local frame, -1 for an unused slot
    n_labels     : label count of the CRF if known.  The wire format is one byte per label
                   (labels are < 64 < 128; RCCL has no 16-bit integer type anyway), and one BIT
                   per label for the binary static/dynamic CRF of the SLAM path (n_labels == 2):
                   250 bytes per 2000-keypoint frame.
    returns (labels [world, S, max_points] int16, counts [world, S] int32), on every rank.
    Entries beyond a frame's point count are unspecified.
    """
    world = dist.get_world_size(group)
    S, P = local_labels.shape
    dev = local_labels.device
    if n_labels == 2:
        Pb = (P + 7) // 8
        bits = torch.zeros((S, Pb * 8), dtype=torch.uint8, device=dev)
        bits[:, :P] = (local_labels & 1).to(torch.uint8)
        weights = (2 ** torch.arange(8, device=dev, dtype=torch.int32)).to(torch.uint8)       # 1, 2, ..., 128
        wire = (bits.view(S, Pb, 8) * weights).sum(-1, dtype=torch.int32).to(torch.uint8)
    else:
        wire = local_labels.to(torch.int8).contiguous().view(torch.uint8)
    # outputs are the rank blocks concatenated along dim 0 (the layout both RCCL and gloo accept)
    out = torch.empty((world * S, wire.shape[1]), dtype=torch.uint8, device=dev)
    counts = torch.empty((world * S,), dtype=torch.int32, device=local_counts.device)
    if world == 1:
        out.copy_(wire)
        counts.copy_(local_counts)
    else:
        dist.all_gather_into_tensor(out, wire.contiguous(), group=group)
        dist.all_gather_into_tensor(counts, local_counts.contiguous(), group=group)
    if n_labels == 2:
        shifts = torch.arange(8, device=dev, dtype=torch.uint8)
        labels = ((out.unsqueeze(-1) >> shifts) & 1).view(world * S, -1)[:, :P].to(torch.int16)
    else:
        labels = out.view(torch.int8).to(torch.int16)
    return labels.reshape(world, S, P), counts.view(world, S)


def gather_label_bits(local_bits, out=None, group=None):
    """The per-step collective of the multi-GPU bench: ONE all_gather of the bit-packed labels the
    inference kernel wrote itself (lccrf_batch_device_label_bits: int64 [S, words], point i = bit
    i%64 of word i/64).  No packing kernels on the way: a 2000-keypoint frame is 32 words.
    returns int64 [world, S, words] on every rank (`out`, if given, is reused)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    S, W = local_bits.shape
    if out is None:
        out = torch.empty((world * S, W), dtype=local_bits.dtype, device=local_bits.device)
    if world == 1:
        out.copy_(local_bits)
    else:
        dist.all_gather_into_tensor(out, local_bits, group=group)
    return out.view(world, S, W)


class OverlappedLabelGather:
    """The per-step label gather of `bench.py --gpus N`, overlapped with the next step's launch.

    The inference kernel writes the bit-packed labels into a buffer the library owns (`bits_view`, int64 [S, words]); the
    next launch overwrites it.  push() copies the bits (device to device, on the CURRENT stream, i.e. behind the launch
    that produced them) into one of two staging buffers and starts an asynchronous all_gather_into_tensor of that
    buffer on the collective's own stream; a staging buffer is reused only after the gather that read it has been
    waited for (a stream-side wait, no host round trip).  wait_all() before the clock stops.  `serial=True` waits for
    every gather right away (gather-then-launch).  The collective is issued whenever a process group exists -- also
    with world size 1, which is how the RCCL path is exercised on a one-GPU box (tests/test_sharding.py)."""

    def __init__(self, bits_view, world, group=None, serial=False):
        self.bits, self.world, self.group, self.serial = bits_view, int(world), group, bool(serial)
        self.collective = dist.is_available() and dist.is_initialized()
        n = 1 if (serial or not self.collective) else 2
        S, W = bits_view.shape
        self.stage = [torch.empty((S, W), dtype=bits_view.dtype, device=bits_view.device) for _ in range(n)]
        self.out = [torch.empty((self.world * S, W), dtype=bits_view.dtype, device=bits_view.device) for _ in range(n)]
        self.pending = [None] * n
        self.steps = 0

    def push(self):
        k = self.steps % len(self.stage)
        if self.pending[k] is not None:
            self.pending[k].wait()
            self.pending[k] = None
        if not self.collective:
            self.out[k].copy_(self.bits, non_blocking=True)
        elif self.serial:
            dist.all_gather_into_tensor(self.out[k], self.bits, group=self.group)
        else:
            self.stage[k].copy_(self.bits, non_blocking=True)
            self.pending[k] = dist.all_gather_into_tensor(self.out[k], self.stage[k], group=self.group, async_op=True)
        self.steps += 1

    def wait_all(self):
        for k, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[k] = None

    def last(self):
        """int64 [world, S, words]: what the most recent push() gathered (call wait_all() first)."""
        S, W = self.bits.shape
        return self.out[(self.steps - 1) % len(self.out)].view(self.world, S, W)


def unpack_label_bits(bits, max_points):
    """int64 [..., words] -> int16 [..., max_points] labels (0 / 1)."""
    shifts = torch.arange(64, device=bits.device, dtype=torch.int64)
    lab = (bits.unsqueeze(-1) >> shifts) & 1
    return lab.reshape(*bits.shape[:-1], -1)[..., :max_points].to(torch.int16)


def unshard(labels, counts, n_frames):
    """Back to global frame order: list of 1-D int16 tensors (trimmed to each frame's size)."""
    world = labels.shape[0]
    out = []
    for f in range(n_frames):
        r, s = f % world, f // world
        n = int(counts[r, s])
        out.append(labels[r, s, :n].clone())
    return out
