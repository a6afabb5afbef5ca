/* lccrf_sharding.h -- where a frame of a replay lives when the frames of a sequence are dealt to the GPUs of a node
 * (BASELINE north_star: "sharded over independent sequence frames ... RCCL only for the final label gather").
 *
 * A frame's CRF is one globally coupled problem (reference src/Tracking.cc:1919-1930: one DenseCRF3D per frame) and is never
 * split.  The frames k = 0, 1, ... of a group go round-robin to G ranks, B at a time per rank:
 *
 *     rank  = k mod G                    which GPU runs the frame
 *     slot  = (k div G) mod B            which frame of that GPU's batch it is
 *     round = k div (G * B)              which batch
 *
 * The only collective of the path is ONE all-gather per round of every rank's bit-packed MAP labels (lccrf_batch_device_label_bits:
 * uint64 [B][words]); in the gathered buffer rank r's block sits at r * B * words.  This header is that arithmetic and nothing else
 * (plain C, no HIP, no RCCL), shared by tools/replay_multi.cpp and the CPU tests (tests/test_sharding.py runs it under a world-size-2
 * gloo all-gather), so that the first multi-GPU run does not depend on index code that has only ever seen one rank.              */
#ifndef LCCRF_SHARDING_H
#define LCCRF_SHARDING_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

static inline int lccrf_shard_rank(size_t k, int n_ranks) { return (int)(k % (size_t)n_ranks); }
static inline int lccrf_shard_slot(size_t k, int n_ranks, int batch) { return (int)((k / (size_t)n_ranks) % (size_t)batch); }
static inline size_t lccrf_shard_round(size_t k, int n_ranks, int batch) { return k / ((size_t)n_ranks * (size_t)batch); }
/* rounds every rank walks for `count` frames (a rank whose share of the last round is short pads with empty frames, so that the
 * collectives of all ranks line up) */
static inline size_t lccrf_shard_rounds(size_t count, int n_ranks, int batch)
{
    const size_t per_round = (size_t)n_ranks * (size_t)batch;
    return (count + per_round - 1) / per_round;
}
/* the frame in (round, rank, slot), or -1 for padding */
static inline long lccrf_shard_frame(size_t round, int rank, int slot, int n_ranks, int batch, size_t count)
{
    const size_t k = (round * (size_t)batch + (size_t)slot) * (size_t)n_ranks + (size_t)rank;
    return k < count ? (long)k : -1L;
}
/* first uint64 word of (rank, slot) in the gathered label bits */
static inline size_t lccrf_gather_word(int rank, int slot, int batch, int words_per_frame)
{
    return ((size_t)rank * (size_t)batch + (size_t)slot) * (size_t)words_per_frame;
}

#ifdef __cplusplus
}
#endif
#endif
