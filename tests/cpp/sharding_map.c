/* The index arithmetic of include/lccrf_sharding.h behind plain C symbols, so that tests/test_sharding.py can run it from
 * Python (ctypes) inside a world-size-2/3 gloo all-gather -- without a GPU.  Test infrastructure, not product. */
#include "lccrf_sharding.h"

int t_shard_rank(long k, int G) { return lccrf_shard_rank((size_t)k, G); }
int t_shard_slot(long k, int G, int B) { return lccrf_shard_slot((size_t)k, G, B); }
long t_shard_round(long k, int G, int B) { return (long)lccrf_shard_round((size_t)k, G, B); }
long t_shard_rounds(long count, int G, int B) { return (long)lccrf_shard_rounds((size_t)count, G, B); }
long t_shard_frame(long round, int rank, int slot, int G, int B, long count) { return lccrf_shard_frame((size_t)round, rank, slot, G, B, (size_t)count); }
long t_gather_word(int rank, int slot, int B, int words) { return (long)lccrf_gather_word(rank, slot, B, words); }
