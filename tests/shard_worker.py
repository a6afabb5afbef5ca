"""Worker for tests/test_sharding.py::test_replay_sharding_arithmetic_under_a_gloo_gather: one rank of a gloo world on CPU walks
the rounds of tools/replay_multi.cpp with the index functions of include/lccrf_sharding.h (through tests/cpp/sharding_map.c),
puts a tag for every frame it owns where that frame's label bits would go, all-gathers the blocks the way the tool's ncclAllGather
does, and checks on EVERY rank that every frame is found exactly where the header says."""
import ctypes as C
import os
import sys

import torch
import torch.distributed as dist


def main():
    lib = C.CDLL(sys.argv[1])
    count, B, words = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    for name in ("t_shard_round", "t_shard_rounds", "t_shard_frame", "t_gather_word"):
        getattr(lib, name).restype = C.c_long
    lib.t_shard_rank.argtypes = [C.c_long, C.c_int]
    lib.t_shard_slot.argtypes = [C.c_long, C.c_int, C.c_int]
    lib.t_shard_round.argtypes = [C.c_long, C.c_int, C.c_int]
    lib.t_shard_rounds.argtypes = [C.c_long, C.c_int, C.c_int]
    lib.t_shard_frame.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long]
    lib.t_gather_word.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    rank, G = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=G)
    rounds = lib.t_shard_rounds(count, G, B)
    seen = set()
    for t in range(rounds):
        mine = torch.zeros(B * words, dtype=torch.int64)
        for i in range(B):
            k = lib.t_shard_frame(t, rank, i, G, B, count)
            if k >= 0:
                # the inverse maps agree with the forward one
                assert lib.t_shard_rank(k, G) == rank and lib.t_shard_slot(k, G, B) == i and lib.t_shard_round(k, G, B) == t, (k, rank, i, t)
                mine[i * words:(i + 1) * words] = torch.arange(words, dtype=torch.int64) + (k + 1) * 1000
        allw = torch.empty(G * B * words, dtype=torch.int64)
        dist.all_gather_into_tensor(allw, mine)             # rank r's block at r * B * words, as ncclAllGather lays it out
        for r in range(G):
            for i in range(B):
                k = lib.t_shard_frame(t, r, i, G, B, count)
                w0 = lib.t_gather_word(r, i, B, words)
                got = allw[w0:w0 + words]
                if k < 0:
                    assert int(got.abs().sum()) == 0, (t, r, i)
                else:
                    assert torch.equal(got, torch.arange(words, dtype=torch.int64) + (k + 1) * 1000), (t, r, i, k)
                    assert k not in seen
                    seen.add(k)
    assert seen == set(range(count)), (len(seen), count)      # every frame exactly once, on every rank's view of the gathers
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
