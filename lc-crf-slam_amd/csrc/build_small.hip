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

#ifndef LCCRF_INSTRUMENT
#define LCCRF_INSTRUMENT 0
#endif
constexpr bool kInstr = LCCRF_INSTRUMENT != 0;   // `make INSTRUMENT=1`: shader-clock stamps; the release library has none
__device__ long long g_build_stamps[32];    // debug: shader-clock stamps of workgroup (0,0) (LCCRF_BUILD_TIMING=1)
#define BSTAMP(i)                                                                              \
    do {                                                                                       \
        if (kInstr && stamps && blockIdx.x == 0 && (int)blockIdx.y == stamps - 1 && threadIdx.x == 0) g_build_stamps[i] = clock64(); \
    } while (0)

constexpr int kBT = 1024;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global store
// of the wave to be acknowledged (a full round trip to L2); where the data behind the barrier lives
// in LDS that wait buys nothing.  Global data written here and read by OTHER lanes later is always
// separated from its readers by at least one full __syncthreads().
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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

    // ---- LDS plan (see small_plan): [ints region][tail region] ---------------------------------
    //   ints region : hash table, later CSR counters / row starts / (unsorted rows | count matrix),
    //                 later the normalisation's lattice values
    //   tail region : point records + slot index + vertex-id prefix of every entry (all 16-bit),
    //                 later the splat weights in CSR order
    int *slot = lds;
    unsigned short *rep_s = reinterpret_cast<unsigned short *>(lds + hcap);      // [V] vertex -> its first entry (phases 4-5)
    unsigned char *tail = reinterpret_cast<unsigned char *>(lds + lds_ints);
    int16_t *r0_s = reinterpret_cast<int16_t *>(tail);                            // [Npad*D]
    uint8_t *rk_s = reinterpret_cast<uint8_t *>(r0_s + (size_t)Npad * D);         // [Npad*D]
    unsigned short *sof = reinterpret_cast<unsigned short *>(tail + (((size_t)Npad * D * 3 + 15) & ~(size_t)15));  // [live] slot -> vertex id
    unsigned short *pfx = sof + ((live + 7) & ~7);                                // [live+1]
    auto entry_key = [&](int e, int16_t(&key)[D]) {
        const int pt = e / D1, rem = e - pt * D1;
#pragma unroll
        for (int i = 0; i < D; ++i) key[i] = vertex_coord<D>(r0_s[pt * D + i], rk_s[pt * D + i], rem);
    };

    // ---- 0/1: empty hash table; point records (elevate, round, rank, barycentric) ------------
    BSTAMP(0);
    for (int i = tid; i < hcap; i += kBT) slot[i] = kEmpty;
    {
        constexpr int kPR = 12 / D1;                      // point rounds: live <= 12 * 1024
        float feat[kPR][D];
#pragma unroll
        for (int r = 0; r < kPR; ++r) {                   // every round's features requested up front
            const int n = tid + r * kBT;
            const float *fp = kd.feat + ((size_t)f * kd.maxN + min(n, max(N - 1, 0))) * D;
#pragma unroll
            for (int j = 0; j < D; ++j) feat[r][j] = fp[j];
        }
#pragma unroll
        for (int r = 0; r < kPR; ++r) {
            const int n = tid + r * kBT;
            if (n < Npad) {
                float ft[D];
#pragma unroll
                for (int j = 0; j < D; ++j) ft[j] = (n < N) ? feat[r][j] : 0.0f;   // phantom lanes, :299
                int16_t r0[D];
                uint8_t rk[D];
                float b[D1];
                point_record<D>(ft, kd.scale, kd.inv_dp1, r0, rk, b);
                float *bp = kd.bary + fe + (size_t)n * D1;
#pragma unroll
                for (int i = 0; i < D; ++i) { r0_s[n * D + i] = r0[i]; rk_s[n * D + i] = rk[i]; }
#pragma unroll
                for (int i = 0; i < D1; ++i) bp[i] = b[i];
            }
        }
    }
    lds_barrier();
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
        sof[e] = (unsigned short)h;
    }
    __syncthreads();
    BSTAMP(2);

    // ---- 3: dense vertex ids = exclusive scan of "first occurrence" flags in entry order -----
    const int V = block_scan(
        live, [&](int e) { return (int)(slot[sof[e]] == e); }, [&](int e, int x) { pfx[e] = (unsigned short)x; });
    if (tid == 0) {
        kd.V[f] = V;
        if (kd.V_host) kd.V_host[f] = V;                 // straight into pinned host memory: no read-back copy
    }
    __syncthreads();
    BSTAMP(3);

    // ---- 4: offset[e] = id of e's vertex (kept in LDS too, over the slot index) ------------------
    for (int e = tid; e < live; e += kBT) {
        const int first = slot[sof[e]];
        const int id = pfx[first];
        kd.offset[fe + e] = id;
        sof[e] = (unsigned short)id;                      // from here on sof[e] is the vertex id of entry e
        if (first == e) rep_s[id] = (unsigned short)e;
    }
    lds_barrier();
    BSTAMP(4);

    // ---- 5: blur neighbours of every (axis, vertex), permutohedral_cpu.h:408-421.  Walk the entries;
    //         a vertex's key is rebuilt from its first entry.
    for (int v = tid; v < V; v += kBT) {                   // one lane per vertex (dense: no idle lanes)
        const int e = rep_s[v];
        int16_t key[D];
        entry_key(e, key);
        for (int j = 0; j < D1; ++j) {
            int16_t n1[D], n2[D];
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
                    const int o = slot[h];
                    if (o == kEmpty) break;
                    int16_t other[D];
                    entry_key(o, other);
                    bool same = true;
#pragma unroll
                    for (int i = 0; i < D; ++i) same &= (other[i] == q[i]);
                    if (same) { found = pfx[o]; break; }
                    h = (h + 1u) & mask;
                }
                if (side) r.y = found; else r.x = found;
            }
            reinterpret_cast<int2 *>(kd.nbr)[((size_t)f * D1 + j) * kd.Epad + v] = r;
            kd.nbr16[((size_t)f * D1 + j) * kd.Epad + v] = (unsigned)(r.x + 1) | ((unsigned)(r.y + 1) << 16);
        }
    }
    __syncthreads();                                      // hash table, records and pfx are dead from here on
    BSTAMP(5);

    // ---- 6-8: CSR of splat contributions, rows ordered by point (LDS only, no global atomics) ------
    int *cnt = lds;                                       // [V+1] counters, later each entry's CSR position [E]
    int *work = lds + (live + 2);                         // [E]: unsorted rows, or the count matrix
    unsigned short *rowstart = reinterpret_cast<unsigned short *>(lds + (live + 2) + ((E + 3) & ~3));   // [V+1]
    for (int v = tid; v <= V; v += kBT) cnt[v] = 0;
    __shared__ int rowmax_s;
    if (tid == 0) rowmax_s = 0;
    __syncthreads();
    for (int e = tid; e < E; e += kBT) atomicAdd(&cnt[sof[e]], 1);                 // real points only
    __syncthreads();
    BSTAMP(9);
    int *rowptr = kd.rowptr + f1;
    block_scan(V + 1, [&](int v) { return cnt[v]; }, [&](int v, int x) { rowstart[v] = (unsigned short)x; rowptr[v] = x; });
    lds_barrier();
    BSTAMP(10);
    {                                                     // longest row: wave maximum first, one LDS atomic per wavefront
        int m = 0;
        for (int v = tid; v < V; v += kBT) m = max(m, (int)rowstart[v + 1] - (int)rowstart[v]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        if ((tid & 63) == 0 && m > 0) atomicMax(&rowmax_s, m);
    }
    // sorted weights overlay the (dead) records / slot / prefix arrays
    float *wsorted = reinterpret_cast<float *>(tail);                            // [E]
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kBT / 64;
    if (V > 0 && V * NW <= E) {
        // Few vertices, long rows (the appearance kernel): ONE stable counting pass.  Wavefront w owns
        // the contiguous entries [w*C, (w+1)*C); mat[w][v] counts its entries per vertex, turned into
        // each wavefront's first position inside every row; then 64 entries at a time find their rank
        // among equal-vertex lanes below them with one ballot per vertex-id bit.
        int *mat = work;                                  // [NW][V]
        const int C = (E + NW - 1) / NW, e0 = wave * C, e1 = min(E, e0 + C);
        for (int i = tid; i < NW * V; i += kBT) mat[i] = 0;
        __syncthreads();
        for (int e = e0 + lane; e < e1; e += 64) atomicAdd(&mat[wave * V + sof[e]], 1);
        __syncthreads();
        for (int v = tid; v < V; v += kBT) {
            int run = rowstart[v];
            for (int w = 0; w < NW; ++w) {
                const int t = mat[w * V + v];
                mat[w * V + v] = run;
                run += t;
            }
        }
        __syncthreads();
        BSTAMP(6);
        int nbits = 1;
        while ((1 << nbits) < V) ++nbits;
        for (int eb = e0; eb < e1; eb += 64) {
            const int e = eb + lane;
            const bool valid = e < e1;
            const int v = valid ? (int)sof[e] : 0;
            // peers = lanes holding the same vertex; 32-bit halves, v_mbcnt for "how many below me"
            const unsigned long long all = __ballot(valid);
            unsigned plo = (unsigned)all, phi = (unsigned)(all >> 32);
            for (int b = 0; b < nbits; ++b) {
                const bool bit = (v >> b) & 1;
                const unsigned long long bal = __ballot(valid && bit);
                const unsigned blo = (unsigned)bal, bhi = (unsigned)(bal >> 32);
                plo &= bit ? blo : ~blo;
                phi &= bit ? bhi : ~bhi;
            }
            const int below = (int)__builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            int pos = 0;
            if (valid) pos = mat[wave * V + v] + below;
            __builtin_amdgcn_wave_barrier();              // every lane has read its base before any leader bumps it
            if (valid && below == 0) mat[wave * V + v] += __popc(plo) + __popc(phi);
            __builtin_amdgcn_wave_barrier();
            if (valid) cnt[e] = pos;                      // CSR position of entry e
        }
    } else {
        // Many vertices, short rows: scatter, then rank every entry inside its (short) row.
        int *unsorted = work;
        for (int e = tid; e < E; e += kBT) {
            const int v = sof[e];
            unsorted[rowstart[v] + atomicSub(&cnt[v], 1) - 1] = e;
        }
        __syncthreads();
        BSTAMP(6);
        for (int p = tid; p < E; p += kBT) {
            const int e = unsorted[p];
            const int v = sof[e];
            const int s = rowstart[v], t = rowstart[v + 1];
            int rank = 0, q = s;                           // 4 compares per LDS read
            for (; q < t && (q & 3); ++q) rank += (unsorted[q] < e);
            for (; q + 4 <= t; q += 4) {
                const int4 u = *reinterpret_cast<const int4 *>(unsorted + q);
                rank += (u.x < e) + (u.y < e) + (u.z < e) + (u.w < e);
            }
            for (; q < t; ++q) rank += (unsorted[q] < e);
            cnt[e] = s + rank;                             // (cnt's counters have all run down to zero by now)
        }
    }
    __syncthreads();
    BSTAMP(11);
    // scatter (point, weight) to their CSR positions -- in LDS, so that the global copies below are
    // coalesced (a scattered 4-byte global store touches one cache line per lane).  The sorted
    // weights overlay the dead records / vertex-id arrays, the sorted points the dead work area.
    unsigned short *pts = reinterpret_cast<unsigned short *>(work);               // [E]
    {
        constexpr int kMaxEPT = 12;                       // E <= 12 * 1024 for every frame this kernel accepts
        float wv[kMaxEPT];
        int pv[kMaxEPT];
        unsigned vv[kMaxEPT];
#pragma unroll
        for (int u = 0; u < kMaxEPT; ++u) {
            const int e = min(tid + u * kBT, max(E - 1, 0));
            wv[u] = kd.bary[fe + e];
            pv[u] = cnt[e];
            vv[u] = sof[e];
        }
        __syncthreads();                                  // every position is in registers: cnt / work / tail may be overwritten
        BSTAMP(12);
#pragma unroll
        for (int u = 0; u < kMaxEPT; ++u) {
            const int e = tid + u * kBT;
            if (e < E) {
                wsorted[pv[u]] = wv[u];
                pts[pv[u]] = (unsigned short)(e / D1);
                kd.csr_pos[fe + e] = pv[u];
                kd.pk[fe + e] = (vv[u] + 1u) | ((unsigned)pv[u] << 16);
            }
        }
    }
    lds_barrier();
    BSTAMP(13);
    for (int p = tid; p < E; p += kBT) {
        kd.csr_pt[fe + p] = pts[p];
        kd.csr_w[fe + p] = wsorted[p];
    }
    if (tid == 0) {
        kd.rowmax[f] = rowmax_s;
        if (kd.rowmax_host) kd.rowmax_host[f] = rowmax_s;
    }
    __syncthreads();
    BSTAMP(7);

    // ---- 9: norm = 1 / (compute(ones) + 1e-20), pairwise3d.h:22-27; lattice values in LDS -------
    float *val = reinterpret_cast<float *>(cnt);          // [V+1], slot 0 = absent neighbour
    float *nxt = reinterpret_cast<float *>(work);         // [V+1]
    if (tid == 0) { val[0] = 0.0f; nxt[0] = 0.0f; }      // (the barrier above: nobody reads pts any more)
    // what the blur and the slice will need from global memory (this workgroup wrote it earlier) is
    // requested now, so that it arrives while the rows are summed
    constexpr int kVR = 2;                                // vertex rounds held in registers (V <= 2048)
    constexpr int kPR = 12 / D1;                          // point rounds: live <= 12 * 1024
    const int2 *nbr = reinterpret_cast<const int2 *>(kd.nbr) + (size_t)f * D1 * kd.Epad;
    int2 nb[D1][kVR];
#pragma unroll
    for (int j = 0; j < D1; ++j)
#pragma unroll
        for (int r = 0; r < kVR; ++r) nb[j][r] = nbr[(size_t)j * kd.Epad + min(tid + r * kBT, max(V - 1, 0))];
    float bw[kPR][D1];
    int bo[kPR][D1];
#pragma unroll
    for (int r = 0; r < kPR; ++r) {
        const size_t e0 = fe + (size_t)min(tid + r * kBT, max(N - 1, 0)) * D1;
#pragma unroll
        for (int j = 0; j < D1; ++j) { bw[r][j] = kd.bary[e0 + j]; bo[r][j] = kd.offset[e0 + j]; }
    }
    for (int v = tid; v < V; v += kBT) {                  // splat of ones: the row's weights, left to right
        float acc = 0.0f;
        int p = rowstart[v];
        const int t = rowstart[v + 1];
        for (; p + 16 <= t; p += 16) {                    // 16 loads in flight, then 16 ordered adds
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = wsorted[p + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += x[u] * 1.0f;
        }
        if (p < t) {                                      // the rest of the row, padded with +0 (x + 0 is exact here)
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = (p + u < t) ? wsorted[p + u] * 1.0f : 0.0f;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += x[u];
        }
        val[v + 1] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < D1; ++j) {
#pragma unroll
        for (int r = 0; r < kVR; ++r) {
            const int v = tid + r * kBT;
            if (v < V) nxt[v + 1] = val[v + 1] + 0.5f * (val[nb[j][r].x + 1] + val[nb[j][r].y + 1]);
        }
        for (int v = tid + kVR * kBT; v < V; v += kBT) {  // lattices beyond the register rounds
            const int2 n2 = nbr[(size_t)j * kd.Epad + v];
            nxt[v + 1] = val[v + 1] + 0.5f * (val[n2.x + 1] + val[n2.y + 1]);
        }
        __syncthreads();
        float *t = val; val = nxt; nxt = t;
    }
#pragma unroll
    for (int r = 0; r < kPR; ++r) {
        const int i = tid + r * kBT;
        if (i < N) {
            float t = 0.0f;
#pragma unroll
            for (int j = 0; j < D1; ++j) t += (bw[r][j] * kd.alpha) * val[bo[r][j] + 1];
            kd.norm[(size_t)f * kd.maxN + i] = 1.0f / (t + 1e-20f);
        }
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
    p.ints = (int)((std::max<long>(p.hcap + (live + 1) / 2, 2 * (live + 2) + 4 + (live + 4) / 2) + 3) & ~3L);   // hash table + rep[] | CSR arrays
    // after the ints: the point records (dead once the neighbours are known), overlaid later by the
    // E sorted weights of the normalisation's splat
    const size_t rec = (((size_t)((NA + 3) & ~3) * kd.d * 3 + 15) & ~(size_t)15) + (size_t)((live + 7) & ~7) * 2 + (size_t)(live + 8) * 2;
    p.bytes = (size_t)p.ints * sizeof(int) + std::max<size_t>(rec, (size_t)live * sizeof(float)) + 16;
    return p;
}

}  // namespace

// Can kernels kds[0..n) (same d) of this batch be built by the fused kernel?
bool build_small_supported(const KernelDev *kds, int n, int NA)
{
    if (n < 1 || n > 2 || NA < 0) return false;
    for (int k = 0; k < n; ++k)
        if (kds[k].d != kds[0].d || kds[k].d > 3 || (long)((NA + 3) & ~3) * kds[k].D1 > 12 * kBT ||
            small_plan(kds[k], NA).bytes > kBuildLdsLimit)
            return false;
    return true;
}

void launch_build_small(const KernelDev *kds, int n, int NA, const CrfDev &c, hipStream_t s)
{
    const SmallPlan p = small_plan(kds[0], NA);          // same d and same capacities for all n
    static const int want_stamps = (kInstr && ab_env("LCCRF_BUILD_TIMING")) ? std::max(atoi(ab_env("LCCRF_BUILD_TIMING")), 1) : 0;   // 1 + kernel index
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
        fprintf(stderr, " | csr: count %lld scan %lld fill %lld | order %lld load %lld scatter %lld copy %lld", h[9] - h[5], h[10] - h[9],
                h[6] - h[10], h[11] - h[6], h[12] - h[11], h[13] - h[12], h[7] - h[13]);
        fprintf(stderr, "\n");
    }
}

}  // namespace lccrf
