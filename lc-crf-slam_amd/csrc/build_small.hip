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
#include <cstdio>
#include <cstdlib>

namespace lccrf {

namespace {

__device__ long long g_build_stamps[32];    // debug: shader-clock stamps of workgroup (0,0) (LCCRF_BUILD_TIMING=1)
#define BSTAMP(i)                                                                              \
    do {                                                                                       \
        if (stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_build_stamps[i] = clock64(); \
    } while (0)

constexpr int kBT = 1024;
constexpr size_t kBuildLdsLimit = 158 * 1024;   // dynamic part; the kernel also has a few static LDS words

// exclusive scan of n values over the whole workgroup; `get(i)` yields the value, `put(i, x)`
// receives the exclusive prefix.  Returns the grand total (uniform).
template <typename Get, typename Put>
__device__ __forceinline__ int block_scan(int n, Get get, Put put)
{
    // every lane owns `per` consecutive elements: local sum, one workgroup scan of the 1024
    // partial sums (wave shuffles + 16 wave totals), then a local running prefix.
    __shared__ int wave_sum[kBT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + kBT - 1) / kBT, base = tid * per;
    int sum = 0;
    for (int i = 0; i < per; ++i)
        if (base + i < n) sum += get(base + i);
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o, 64);
        if (lane >= o) incl += y;
    }
    __syncthreads();                                      // wave_sum may still be read by a previous scan
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBT / 64; ++w) {
        const int x = wave_sum[w];
        if (w < wave) wbase += x;
        total += x;
    }
    int run = wbase + incl - sum;
    for (int i = 0; i < per; ++i) {
        if (base + i < n) {
            const int x = get(base + i);
            put(base + i, run);
            run += x;
        }
    }
    return total;
}

template <int D>
__global__ void __launch_bounds__(kBT) k_build_small(KernelDev kd0, KernelDev kd1, CrfDev c, int lds_ints, int hcap,
                                                    int stamps)
{
    constexpr int D1 = D + 1;
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const KernelDev &kd = blockIdx.y == 0 ? kd0 : kd1;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int N = c.n_points[f];
    const int Npad = (N + 3) & ~3;                       // blocks of four, permutohedral_cpu.h:294 (quirk Q1)
    const int live = Npad * D1, E = N * D1;
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const unsigned mask = (unsigned)hcap - 1u;          // LDS hash table, sized for the frames' real size

    BSTAMP(0);
    // ---- 0/1: empty hash table; point records (elevate, round, rank, barycentric) ------------
    // The records (from which every vertex key is recomputed) stay in LDS: key compares during
    // hash probing are the build's inner loop and must not pay a global-memory round trip each.
    int *slot = lds;
    int16_t *r0_s = reinterpret_cast<int16_t *>(lds + lds_ints);            // [Npad*D]
    uint8_t *rk_s = reinterpret_cast<uint8_t *>(r0_s + (size_t)Npad * D);         // [Npad*D]
    auto entry_key = [&](int e, int16_t(&key)[D]) {
        const int pt = e / D1, rem = e - pt * D1;
#pragma unroll
        for (int i = 0; i < D; ++i) key[i] = vertex_coord<D>(r0_s[pt * D + i], rk_s[pt * D + i], rem);
    };
    for (int i = tid; i < hcap; i += kBT) slot[i] = kEmpty;
    for (int n = tid; n < Npad; n += kBT) {
        float feat[D];
        const float *fp = kd.feat + ((size_t)f * kd.maxN + n) * D;
#pragma unroll
        for (int j = 0; j < D; ++j) feat[j] = (n < N) ? fp[j] : 0.0f;      // phantom lanes, :299
        int16_t r0[D];
        uint8_t rk[D];
        float b[D1];
        point_record<D>(feat, kd.scale, kd.inv_dp1, r0, rk, b);
        float *bp = kd.bary + fe + (size_t)n * D1;
#pragma unroll
        for (int i = 0; i < D; ++i) { r0_s[n * D + i] = r0[i]; rk_s[n * D + i] = rk[i]; }
#pragma unroll
        for (int i = 0; i < D1; ++i) bp[i] = b[i];
    }
    __syncthreads();
    BSTAMP(1);

    // ---- 2: insert every entry's vertex key; a slot keeps the LOWEST entry id with that key ----
    for (int e = tid; e < live; e += kBT) {
        int16_t key[D];
        entry_key(e, key);
        unsigned h = hash_key<D>(key) & mask;
        for (;;) {
            const int prev = atomicCAS(&slot[h], kEmpty, e);
            if (prev == kEmpty || prev == e) break;
            int16_t other[D];
            entry_key(prev, other);
            bool same = true;
#pragma unroll
            for (int i = 0; i < D; ++i) same &= (other[i] == key[i]);
            if (same) { atomicMin(&slot[h], e); break; }
            h = (h + 1u) & mask;
        }
        kd.slot_of[fe + e] = (int)h;
    }
    __syncthreads();
    BSTAMP(2);

    // ---- 3: dense vertex ids = exclusive scan of "first occurrence" flags in entry order -----
    int *prefix = kd.prefix + f1;
    const int V = block_scan(
        live, [&](int e) { return (int)(slot[kd.slot_of[fe + e]] == e); }, [&](int e, int x) { prefix[e] = x; });
    if (tid == 0) kd.V[f] = V;
    __syncthreads();
    BSTAMP(3);

    // ---- 4: offset[e] = id of e's vertex; first entries register as representatives ------------
    for (int e = tid; e < live; e += kBT) {
        const int r = slot[kd.slot_of[fe + e]];
        const int id = prefix[r];
        kd.offset[fe + e] = id;
        if (r == e) kd.rep[fe + id] = e;
    }
    __syncthreads();
    BSTAMP(4);

    // ---- 5: blur neighbours of every (axis, vertex), permutohedral_cpu.h:408-421 ----------------
    for (int idx = tid; idx < V * D1; idx += kBT) {
        const int j = idx / V, v = idx - j * V;
        int16_t key[D], n1[D], n2[D];
        entry_key(kd.rep[fe + v], key);
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
                entry_key(e, other);
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
    BSTAMP(5);

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
    BSTAMP(9);
    int *rowptr = kd.rowptr + f1;
    block_scan(V + 1, [&](int v) { return cnt[v]; }, [&](int v, int x) { rowstart[v] = x; rowptr[v] = x; });
    __syncthreads();
    BSTAMP(10);
    for (int e = tid; e < E; e += kBT) {
        const int v = kd.offset[fe + e];
        unsorted[rowstart[v] + atomicSub(&cnt[v], 1) - 1] = e;
    }
    for (int v = tid; v < V; v += kBT) atomicMax(&rowmax_s, rowstart[v + 1] - rowstart[v]);
    __syncthreads();
    BSTAMP(6);
    float *wsorted = reinterpret_cast<float *>(lds + lds_ints);                  // [E] weights in CSR order
    for (int p = tid; p < E; p += kBT) {
        const int e = unsorted[p];
        const int v = kd.offset[fe + e];
        const int s = rowstart[v], t = rowstart[v + 1];
        int rank = 0, q = s;                               // rank of e inside its row, 4 compares per LDS read
        for (; q < t && (q & 3); ++q) rank += (unsorted[q] < e);
        for (; q + 4 <= t; q += 4) {
            const int4 u = *reinterpret_cast<const int4 *>(unsorted + q);
            rank += (u.x < e) + (u.y < e) + (u.z < e) + (u.w < e);
        }
        for (; q < t; ++q) rank += (unsorted[q] < e);
        const float w = kd.bary[fe + e];
        kd.csr_pt[fe + s + rank] = e / D1;
        kd.csr_w[fe + s + rank] = w;
        wsorted[s + rank] = w;
    }
    if (tid == 0) kd.rowmax[f] = rowmax_s;
    __syncthreads();
    BSTAMP(7);

    // ---- 9: norm = 1 / (compute(ones) + 1e-20), pairwise3d.h:22-27; lattice values in LDS -------
    float *val = reinterpret_cast<float *>(cnt);          // [V+1], slot 0 = absent neighbour
    float *nxt = reinterpret_cast<float *>(unsorted);     // [V+1]
    if (tid == 0) { val[0] = 0.0f; nxt[0] = 0.0f; }
    for (int v = tid; v < V; v += kBT) {                  // splat of ones: the row's weights, left to right
        float acc = 0.0f;
        int p = rowstart[v];
        const int t = rowstart[v + 1];
        for (; p + 8 <= t; p += 8) {
            const float x0 = wsorted[p], x1 = wsorted[p + 1], x2 = wsorted[p + 2], x3 = wsorted[p + 3];
            const float x4 = wsorted[p + 4], x5 = wsorted[p + 5], x6 = wsorted[p + 6], x7 = wsorted[p + 7];
            acc += x0 * 1.0f; acc += x1 * 1.0f; acc += x2 * 1.0f; acc += x3 * 1.0f;
            acc += x4 * 1.0f; acc += x5 * 1.0f; acc += x6 * 1.0f; acc += x7 * 1.0f;
        }
        for (; p < t; ++p) acc += wsorted[p] * 1.0f;
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
    BSTAMP(8);
}

// LDS plan for frames of at most NA points: hash table (load factor <= 2/3), later reused for the
// CSR counters / row starts / unsorted rows and the normalisation's values; then the point records.
struct SmallPlan { int hcap, ints; size_t bytes; };
SmallPlan small_plan(const KernelDev &kd, int NA)
{
    const long live = (long)((NA + 3) & ~3) * kd.D1;
    SmallPlan p;
    p.hcap = 1024;
    while (p.hcap < live + live / 2) p.hcap <<= 1;
    p.ints = (int)((std::max<long>(p.hcap, 3 * (live + 2)) + 3) & ~3L);
    // after the ints: the point records (dead once the neighbours are known), overlaid later by the
    // E sorted weights of the normalisation's splat
    p.bytes = (size_t)p.ints * sizeof(int) + std::max<size_t>((size_t)((NA + 3) & ~3) * kd.d * 3 + 16, (size_t)live * sizeof(float));
    return p;
}

}  // namespace

// Can kernels kds[0..n) (same d) of this batch be built by the fused kernel?
bool build_small_supported(const KernelDev *kds, int n, int NA)
{
    if (n < 1 || n > 2 || NA < 0) return false;
    for (int k = 0; k < n; ++k)
        if (kds[k].d != kds[0].d || kds[k].d > 3 || small_plan(kds[k], NA).bytes > kBuildLdsLimit) return false;
    return true;
}

void launch_build_small(const KernelDev *kds, int n, int NA, const CrfDev &c, hipStream_t s)
{
    const SmallPlan p = small_plan(kds[0], NA);          // same d and same capacities for all n
    static const bool want_stamps = getenv("LCCRF_BUILD_TIMING") != nullptr;
    const dim3 grid(c.F, n);
    const KernelDev &k0 = kds[0], &k1 = kds[n - 1];
#define BUILD_CASE(DD)                                                                                  \
    case DD: {                                                                                          \
        auto fn = k_build_small<DD>;                                                                    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn),                                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBuildLdsLimit);     \
        fn<<<grid, dim3(kBT), p.bytes, s>>>(k0, k1, c, p.ints, p.hcap, (int)want_stamps);                \
    } break;
    switch (kds[0].d) {
        BUILD_CASE(1)
        BUILD_CASE(2)
        BUILD_CASE(3)
    default: break;
    }
#undef BUILD_CASE
    if (want_stamps) {
        long long h[32];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_build_stamps), sizeof(h));
        fprintf(stderr, "[lccrf build timing] lds=%zu B hcap=%d; phase deltas (shader clocks):", p.bytes, p.hcap);
        for (int i = 1; i <= 8; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
        fprintf(stderr, " | csr: count %lld scan %lld fill %lld", h[9] - h[5], h[10] - h[9], h[6] - h[10]);
        fprintf(stderr, "\n");
    }
}

}  // namespace lccrf
