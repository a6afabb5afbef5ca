// build_small.hip -- the PottsPotential3D constructor (pairwise3d.h:20-28: lattice + normalisation)
// for SLAM-size frames as ONE kernel launch: one 1024-lane workgroup per (frame, kernel).
//
// The streaming build needs 19 dependent launches per kernel; for one live frame that is
// ~70 us of launch latency per kernel, and for a batch it is 13x the inference time.  Here
// the phases are separated by workgroup barriers instead of kernel boundaries, the hash table,
// the CSR counters/unsorted rows and the normalisation's lattice values live in LDS, and no
// global atomic is issued.  Every output array is the streaming build's, bit for bit: the same
// device functions produce the point records, keys and weights; vertex ids come from the same
// first-occurrence scan; CSR rows are ordered by point; the normalisation runs the same
// splat(ones) / blur / slice arithmetic in the same order.
//
// ref: permutohedral_cpu.h:241-424 (init), :634-699 (compute), pairwise3d.h:20-28 (norm).
#include "engine.h"
#include "device_math.h"
#include "lattice_device.h"

#include <algorithm>

namespace lccrf {

namespace {

constexpr int kBT = 1024;
constexpr size_t kBuildLdsLimit = 158 * 1024;   // dynamic part; the kernel also has a few static LDS words

// exclusive scan of n values over the whole workgroup; `get(i)` yields the value, `put(i, x)`
// receives the exclusive prefix.  Returns the grand total (uniform).
template <typename Get, typename Put>
__device__ __forceinline__ int block_scan(int n, Get get, Put put)
{
    __shared__ int wave_sum[kBT / 64];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += kBT) {
        const int i = base + tid;
        const int x = (i < n) ? get(i) : 0;
        int incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wave_sum[w];
        const int carry = carry_s;
        if (i < n) put(i, carry + wbase + incl - x);
        __syncthreads();
        if (tid == kBT - 1) carry_s = carry + wbase + incl;
        __syncthreads();
    }
    return carry_s;
}

template <int D>
__global__ void __launch_bounds__(kBT) k_build_small(KernelDev kd0, KernelDev kd1, CrfDev c, int lds_ints)
{
    constexpr int D1 = D + 1;
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const KernelDev &kd = blockIdx.y == 0 ? kd0 : kd1;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int N = c.n_points[f];
    const int Npad = (N + 3) & ~3;                       // blocks of four, permutohedral_cpu.h:294 (quirk Q1)
    const int live = Npad * D1, E = N * D1;
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const unsigned mask = (unsigned)kd.cap - 1u;

    // ---- 0/1: empty hash table; point records (elevate, round, rank, barycentric) ------------
    int *slot = lds;
    for (int i = tid; i < kd.cap; i += kBT) slot[i] = kEmpty;
    for (int n = tid; n < Npad; n += kBT) {
        float feat[D];
        const float *fp = kd.feat + ((size_t)f * kd.maxN + n) * D;
#pragma unroll
        for (int j = 0; j < D; ++j) feat[j] = (n < N) ? fp[j] : 0.0f;      // phantom lanes, :299
        int16_t r0[D];
        uint8_t rk[D];
        float b[D1];
        point_record<D>(feat, kd.scale, kd.inv_dp1, r0, rk, b);
        int16_t *r0p = kd.rem0 + ((size_t)f * kd.maxNpad + n) * D;
        uint8_t *rkp = kd.rank + ((size_t)f * kd.maxNpad + n) * D;
        float *bp = kd.bary + fe + (size_t)n * D1;
#pragma unroll
        for (int i = 0; i < D; ++i) { r0p[i] = r0[i]; rkp[i] = rk[i]; }
#pragma unroll
        for (int i = 0; i < D1; ++i) bp[i] = b[i];
    }
    __syncthreads();

    // ---- 2: insert every entry's vertex key; a slot keeps the LOWEST entry id with that key ----
    for (int e = tid; e < live; e += kBT) {
        int16_t key[D];
        load_entry_key<D>(kd, f, e, key);
        unsigned h = hash_key<D>(key) & mask;
        for (;;) {
            const int prev = atomicCAS(&slot[h], kEmpty, e);
            if (prev == kEmpty || prev == e) break;
            int16_t other[D];
            load_entry_key<D>(kd, f, prev, other);
            bool same = true;
#pragma unroll
            for (int i = 0; i < D; ++i) same &= (other[i] == key[i]);
            if (same) { atomicMin(&slot[h], e); break; }
            h = (h + 1u) & mask;
        }
        kd.slot_of[fe + e] = (int)h;
    }
    __syncthreads();

    // ---- 3: dense vertex ids = exclusive scan of "first occurrence" flags in entry order -----
    int *prefix = kd.prefix + f1;
    const int V = block_scan(
        live, [&](int e) { return (int)(slot[kd.slot_of[fe + e]] == e); }, [&](int e, int x) { prefix[e] = x; });
    if (tid == 0) kd.V[f] = V;
    __syncthreads();

    // ---- 4: offset[e] = id of e's vertex; first entries register as representatives ------------
    for (int e = tid; e < live; e += kBT) {
        const int r = slot[kd.slot_of[fe + e]];
        const int id = prefix[r];
        kd.offset[fe + e] = id;
        if (r == e) kd.rep[fe + id] = e;
    }
    __syncthreads();

    // ---- 5: blur neighbours of every (axis, vertex), permutohedral_cpu.h:408-421 ----------------
    for (int idx = tid; idx < V * D1; idx += kBT) {
        const int j = idx / V, v = idx - j * V;
        int16_t key[D], n1[D], n2[D];
        load_entry_key<D>(kd, f, kd.rep[fe + v], key);
#pragma unroll
        for (int t = 0; t < D; ++t) {
            n1[t] = (int16_t)(key[t] - 1);
            n2[t] = (int16_t)(key[t] + 1);
        }
#pragma unroll
        for (int t = 0; t < D; ++t)
            if (t == j) { n1[t] = (int16_t)(key[t] + D); n2[t] = (int16_t)(key[t] - D); }
        int2 r;
        for (int side = 0; side < 2; ++side) {
            const int16_t(&q)[D] = side ? n2 : n1;
            unsigned h = hash_key<D>(q) & mask;
            int found = -1;
            for (;;) {
                const int e = slot[h];
                if (e == kEmpty) break;
                int16_t other[D];
                load_entry_key<D>(kd, f, e, other);
                bool same = true;
#pragma unroll
                for (int i = 0; i < D; ++i) same &= (other[i] == q[i]);
                if (same) { found = prefix[e]; break; }
                h = (h + 1u) & mask;
            }
            if (side) r.y = found; else r.x = found;
        }
        reinterpret_cast<int2 *>(kd.nbr)[((size_t)f * D1 + j) * kd.Epad + v] = r;
    }
    __syncthreads();                                      // the hash table is dead from here on

    // ---- 6-8: CSR of splat contributions, rows ordered by point (LDS counters, no global atomics)
    int *rowstart = lds;                                  // [V+1]
    int *cnt = lds + (live + 2);                          // [V+1]
    int *unsorted = lds + 2 * (live + 2);                 // [E]
    for (int v = tid; v <= V; v += kBT) cnt[v] = 0;
    __shared__ int rowmax_s;
    if (tid == 0) rowmax_s = 0;
    __syncthreads();
    for (int e = tid; e < E; e += kBT) atomicAdd(&cnt[kd.offset[fe + e]], 1);      // real points only
    __syncthreads();
    int *rowptr = kd.rowptr + f1;
    block_scan(V + 1, [&](int v) { return cnt[v]; }, [&](int v, int x) { rowstart[v] = x; rowptr[v] = x; });
    __syncthreads();
    for (int e = tid; e < E; e += kBT) {
        const int v = kd.offset[fe + e];
        unsorted[rowstart[v] + atomicSub(&cnt[v], 1) - 1] = e;
    }
    for (int v = tid; v < V; v += kBT) atomicMax(&rowmax_s, rowstart[v + 1] - rowstart[v]);
    __syncthreads();
    for (int p = tid; p < E; p += kBT) {
        const int e = unsorted[p];
        const int v = kd.offset[fe + e];
        const int s = rowstart[v], t = rowstart[v + 1];
        int rank = 0;
        for (int q = s; q < t; ++q) rank += (unsorted[q] < e);
        kd.csr_pt[fe + s + rank] = e / D1;
        kd.csr_w[fe + s + rank] = kd.bary[fe + e];
    }
    if (tid == 0) kd.rowmax[f] = rowmax_s;
    __syncthreads();

    // ---- 9: norm = 1 / (compute(ones) + 1e-20), pairwise3d.h:22-27; lattice values in LDS -------
    float *val = reinterpret_cast<float *>(cnt);          // [V+1], slot 0 = absent neighbour
    float *nxt = reinterpret_cast<float *>(unsorted);     // [V+1]
    if (tid == 0) { val[0] = 0.0f; nxt[0] = 0.0f; }
    for (int v = tid; v < V; v += kBT) {
        float acc = 0.0f;
        for (int p = rowstart[v]; p < rowstart[v + 1]; ++p) acc += kd.csr_w[fe + p] * 1.0f;
        val[v + 1] = acc;
    }
    __syncthreads();
    const int2 *nbr = reinterpret_cast<const int2 *>(kd.nbr) + (size_t)f * D1 * kd.Epad;
    for (int j = 0; j < D1; ++j) {
        for (int v = tid; v < V; v += kBT) {
            const int2 nb = nbr[(size_t)j * kd.Epad + v];
            nxt[v + 1] = val[v + 1] + 0.5f * (val[nb.x + 1] + val[nb.y + 1]);
        }
        __syncthreads();
        float *t = val; val = nxt; nxt = t;
    }
    for (int i = tid; i < N; i += kBT) {
        float t = 0.0f;
#pragma unroll
        for (int j = 0; j < D1; ++j)
            t += (kd.bary[fe + (size_t)i * D1 + j] * kd.alpha) * val[kd.offset[fe + (size_t)i * D1 + j] + 1];
        kd.norm[(size_t)f * kd.maxN + i] = 1.0f / (t + 1e-20f);
    }
}

size_t build_small_lds(const KernelDev &kd)
{
    const size_t ints = std::max<size_t>((size_t)kd.cap, 3 * ((size_t)kd.Epad + 2));
    return ints * sizeof(int);
}

}  // namespace

// Can kernels kds[0..n) (same d) of this batch be built by the fused kernel?
bool build_small_supported(const KernelDev *kds, int n)
{
    if (n < 1 || n > 2) return false;
    for (int k = 0; k < n; ++k)
        if (kds[k].d != kds[0].d || kds[k].d > 3 || build_small_lds(kds[k]) > kBuildLdsLimit) return false;
    return true;
}

void launch_build_small(const KernelDev *kds, int n, const CrfDev &c, hipStream_t s)
{
    const size_t lds = std::max(build_small_lds(kds[0]), build_small_lds(kds[n - 1]));
    const dim3 grid(c.F, n);
    const KernelDev &k0 = kds[0], &k1 = kds[n - 1];
#define BUILD_CASE(DD)                                                                                  \
    case DD: {                                                                                          \
        auto fn = k_build_small<DD>;                                                                    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn),                                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBuildLdsLimit);     \
        fn<<<grid, dim3(kBT), lds, s>>>(k0, k1, c, (int)(lds / sizeof(int)));                           \
    } break;
    switch (kds[0].d) {
        BUILD_CASE(1)
        BUILD_CASE(2)
        BUILD_CASE(3)
    default: break;
    }
#undef BUILD_CASE
}

}  // namespace lccrf
