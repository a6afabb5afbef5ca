// frame_engine.hip -- one frame's WHOLE CRF as ONE kernel launch (SURVEY.md section 7's design stance):
// one 1024-lane workgroup per frame builds both permutohedral lattices, normalises both kernels, runs
// startInference, every mean-field iteration and buildMap.  HBM is touched twice: the frame's inputs
// (features of every kernel + labels or unary energies, ~36 KB at N = 2000) are read once, Q and the MAP
// labels (~20 KB) are written once.  Nothing else leaves the CU: the reference's per-frame sequence
//     DenseCRF3D crf(N); crf.setUnaryEnergyFromLabel(..); crf.addPairwiseEnergy(appearanceKernel(..));
//     crf.addPairwiseEnergy(smoothKernel(..)); crf.inference(5, true);            (src/Tracking.cc:1920-1929)
// costs one launch.  L = 2 labels, 2-D kernels, K <= 2, N <= 4096 -- the SLAM configuration; anything else
// (and any frame whose lattices do not fit the LDS plan below) runs on the two-kernel / streaming path with
// identical results.
//
// Lattice construction here is NOT the streaming build's algorithm (stream_engine.hip / build_small.hip keep
// the reference's vertex numbering because the parity probes compare offset_/blur_neighbors_ verbatim).  The
// mean-field result does not depend on how vertices are numbered -- a vertex's value is a sum over its OWN
// row in ascending point order, and blur neighbours are found by key -- so this kernel uses whatever
// numbering is cheapest:
//   keys in the table   a 2-D vertex key is two int16 = one 32-bit word: the LDS hash table stores the key
//                       itself (ds_cmpst claims a slot or finds the key; no representative entry, no key
//                       recomputation while probing), permutohedral_cpu.h:66-167,371-377
//   ids by slot order   dense vertex ids = exclusive scan of "slot occupied" (no first-occurrence machinery)
//   neighbours          one lane per vertex probes the same table for key +- 1, permutohedral_cpu.h:408-421
//   row order           a product's place in its vertex's row must follow ascending point index (quirk Q6).
//                       Short rows (the smoothness kernel, ~5 entries per vertex): entries are dropped into
//                       their row in arrival order and every entry counts the smaller entry ids of its row
//                       (one ds_read_b64 per 4 entries).  Long rows (the appearance kernel, ~50 entries per
//                       vertex): a bitmap per vertex over the points; rank = prefix popcount.
//   normalisation       pairwise3d.h:20-28 is one splat/blur/slice of all-ones: the loop's own phases run
//                       once with Q = 1 before startInference (ordered row sums included).
// What must match the reference exactly does: the point records (point_record<2>, quirks Q1-Q4, phantom
// points of the last block of four included), the set of vertices (V is reported and tested against the
// reference's M_), the order of every sum.  ref: permutohedral_cpu.h:241-424,634-699; densecrf_base.h:65-91.
#include "engine.h"
#include "device_math.h"
#include "fused_loop.h"
#include "frame_build.h"
#include "fused_lean.h"
#include <type_traits>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace lccrf {

namespace {

using namespace fl;
using namespace fb;

struct Hdr {                              // lives at smem + 192
    int wave_sum[16];
    int fail;
    int rowmax;
    int dual_ok, dual_V;
};

// DUAL: a single frame (the live tracker's case: one CU busy, 255 idle) is given TWO workgroups -- workgroup 0 builds
// kernel 0's lattice and runs everything else, workgroup 1 (on another CU) builds kernel 1's lattice meanwhile and hands
// its tables (neighbour table, row starts) and per-point records (vertex | row place words, barycentric weights) over
// through this area: 29k of the frame's 157k cycles leave the critical path for ~10k of hand-off.
constexpr int kDualVcap = 8192;                           // vertices of kernel 1 the area can carry (more: the frame falls back)
constexpr int kDualHdr = 16, kDualNbr = kDualHdr, kDualRow = kDualNbr + kD1 * kDualVcap, kDualRec = kDualRow + kDualVcap / 2 + 8,
              kDualWn = kDualRec + 2 * 4 * kD1 * kNT,      // records: (vertex | row place word, barycentric weight) pairs, 8 bytes each
              kDualWords = kDualWn + 4 * kNT;              // w * norm of kernel 1, per point

__device__ __forceinline__ unsigned dual_load(const unsigned *p)       // device-coherent load (the producer ran on another CU / XCD)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long dual_load2(const unsigned *p)   // the same, 8 bytes (p 8-byte aligned)
{
    return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// NT = 1024 lanes, or 512 for frames of up to 1024 points whose plan fits half the CU's LDS: two frames per CU
// (fused_loop.h: kNTSmall).  Second launch bound: 128 registers per lane in both shapes.
template <int NT, int PPT, int K, bool DUAL = false>
__global__ void __launch_bounds__(NT, 4) k_frame(CrfDev c, FrameArgs a)
{
    constexpr int D1 = kD1;
    static_assert(!DUAL || (K == 2 && NT == kNT), "the two-workgroup form splits a two-kernel frame");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int f = DUAL ? (int)(blockIdx.x >> 1) : (int)blockIdx.x, tid = threadIdx.x;
    const int role = DUAL ? (int)(blockIdx.x & 1) : 0;    // DUAL: 0 = the frame's main workgroup, 1 = the helper that builds kernel 1
    const int N = a.n_single >= 0 ? a.n_single : c.n_points[f];     // (a single frame's count rides in the kernel arguments:
                                                                    //  c.n_points is pinned host memory there, a PCIe round trip away)
    Instr ins{a.timing, a.timing_block, 8, 0, a.timing_lane};
    FL_STAMP();
    // The last thing the frame's main workgroup does (uniform; every lane): results first, then the word the host polls.
    auto publish_done = [&]() {
        if (a.done) {
            __threadfence_system();                       // this lane's stores have reached host memory ...
            __syncthreads();                              // ... every lane's have
            if (tid == 0) __hip_atomic_store(a.done, a.done_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    // A frame that does not fit this kernel's LDS plan is left to the fallback path.  (A host that reads the labels of a
    // single frame as they arrive -- lccrf_get_map -- finds -2 in place of the first one.)
    auto flag_unfit = [&]() {
        if (tid == 0) {
            if (a.status) *a.status = 1;
            if (a.frame_status) a.frame_status[f] = 1;
            if (a.done && c.map) c.map[(size_t)f * c.maxN] = (int16_t)-2;
        }
    };
    if (N <= 0) {                                         // an empty frame has no lattice (V = 0) and nothing to infer
        if (role == 0) {
            if (a.with_map) clear_label_bits<NT>(c, f, 0, tid);
            if (tid < K && a.V_out[tid]) a.V_out[tid][f] = 0;
            if (tid == 0 && a.frame_status) a.frame_status[f] = 0;
            publish_done();
        }
        return;
    }
    unsigned *xs = DUAL ? a.dual + (size_t)f * kDualWords : nullptr;
    // the helper's way out when its lattice cannot be handed over: tell the main workgroup (which then flags the frame)
    auto helper_fail = [&]() {
        if (tid == 0) {
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(xs), (unsigned long long)a.dual_epoch | (0x80000000ull << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // (nothing to publish but the word itself)
        }
    };
    const int Npad = (N + 3) & ~3;                       // blocks of four, permutohedral_cpu.h:294 (quirk Q1)
    Hdr *hdr = reinterpret_cast<Hdr *>(smem + 192);
    const int hcap = a.hcap;
    const unsigned mask = (unsigned)hcap - 1u;
    const int hk_off = a.lds_total - hcap * 4, ido_off = hk_off - hcap * 2;
    unsigned *hk = reinterpret_cast<unsigned *>(smem + hk_off);                  // hash table: the keys themselves
    unsigned short *ido = reinterpret_cast<unsigned short *>(smem + ido_off);    // slot -> vertex id

    // ---- inputs: every global load of the launch is issued here --------------------------------
    // (3-4 points per lane: a kernel's features are fetched when its build starts instead -- holding both kernels'
    //  features across the first build is what pushed those shapes into scratch)
    constexpr bool kFeatUpFront = PPT <= 2;
    PointRegs<PPT, K> pr;
    float2 ft[PPT][K];
    // Kernel 0's features first: its build starts when THEY have arrived (loads return in order; for a single frame
    // through the object API they come from pinned host memory, ~7 us for all of a 2000-point frame's 36 KB), with the
    // other kernels' features and the labels / unaries still on their way.
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if ((kFeatUpFront || k == 0) && (!DUAL || k == role)) {
#pragma unroll
            for (int s = 0; s < PPT; ++s)
                ft[s][k] = reinterpret_cast<const float2 *>(a.feat[k])[(size_t)f * a.maxN + min(tid + s * NT, N - 1)];
        }
    }
    int lab_raw[PPT];
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int ic = min(tid + s * NT, N - 1);
        lab_raw[s] = 0;
        if (a.label) lab_raw[s] = a.label[(size_t)f * a.maxN + ic];
        else pr.un[s] = reinterpret_cast<const float2 *>(c.unary)[(size_t)f * c.maxN + ic];
    }
    if (tid < 16) reinterpret_cast<float *>(smem + 128)[tid] = 0.0f;
    if (tid == 0) { hdr->fail = 0; hdr->rowmax = 0; }

    FusedLayout lay{};
    lay.zero = 128;
    int V[K], row0max = 0;
    unsigned pk[PPT][K][D1];              // (vertex id + 1) | place in the row << 16, as the HBM records of k_fused
    int cursor = kHdr;

#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (DUAL && k != role) continue;                  // (uniform per workgroup)
        // ---- A: point records (elevate, round, rank, barycentric) and the keys of their three corners ----
        unsigned key[PPT][D1];
        if (!kFeatUpFront && k > 0) {
#pragma unroll
            for (int s = 0; s < PPT; ++s)
                ft[s][k] = reinterpret_cast<const float2 *>(a.feat[k])[(size_t)f * a.maxN + min(tid + s * NT, N - 1)];
        }
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * NT;
            float feat[2] = {i < N ? ft[s][k].x : 0.0f, i < N ? ft[s][k].y : 0.0f};   // phantom lanes, :299
            int16_t r0[2];
            uint8_t rk[2];
            float b[D1];
            point_record<2>(feat, a.scale, a.inv_dp1, r0, rk, b);
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                pr.bary[s][k][j] = b[j];
                key[s][j] = corner_key(r0, rk, j);
            }
        }
        for (int u = tid; u < hcap; u += NT) hk[u] = kEmptyKey;
        __syncthreads();
        FL_PSTAMP();

        // ---- B: insert.  ds_cmpst either claims an empty slot for the key (this entry CREATES the vertex) or returns
        //      the key that lives there.  Every entry's first probe is issued before any result is looked at (the
        //      table is at most ~3/4 full and usually ~1/7: the first probe nearly always settles it).
        unsigned slot[PPT][D1], got[PPT][D1];
        bool bad = false;
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                slot[s][j] = hash32(key[s][j]) & mask;
                got[s][j] = key[s][j];
                bad |= key[s][j] == kEmptyKey;
                if (tid + s * NT < Npad) got[s][j] = atomicCAS(&hk[slot[s][j]], kEmptyKey, key[s][j]);
            }
        int ncreated = 0;
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                const unsigned kk = key[s][j];
                if (got[s][j] != kEmptyKey && got[s][j] != kk) {           // somebody else's key lives there: linear probing
                    unsigned h = slot[s][j];
                    for (int probes = 0;; ++probes) {
                        h = (h + 1u) & mask;
                        got[s][j] = atomicCAS(&hk[h], kEmptyKey, kk);
                        if (got[s][j] == kEmptyKey || got[s][j] == kk) break;
                        if (probes >= hcap) { bad = true; break; }
                    }
                    slot[s][j] = h;
                }
                ncreated += got[s][j] == kEmptyKey;
            }
        if (bad) hdr->fail = 1;
        FL_PSTAMP();

        // ---- C: dense vertex ids, handed out to the entries that created their vertex (any numbering will do: a
        //      vertex's value is a sum over its own row, its neighbours are found by key); carve this kernel's tables
        int Vk;
        int id = block_excl_scan<NT>(ncreated, tid, hdr->wave_sum, Vk);       // (its barriers also close phase B)
        V[k] = Vk;
        FL_PSTAMP();
        const int W = (((Npad + 31) >> 5) + 3) & ~3;      // bitmap words per vertex, a multiple of 4
        auto take = [&](int &o, int bytes) { const int r = o; o += (bytes + 15) & ~15; return r; };
        lay.val[k][0] = take(cursor, (Vk + 1) * 8);
        lay.val[k][1] = take(cursor, (Vk + 1) * 8);
        lay.nbr[k] = take(cursor, D1 * Vk * 4);
        lay.row[k] = take(cursor, (Vk + 2) * 2);
        int vs = cursor;                                  // V-dependent scratch behind the persistent tables
        const int vkey_off = take(vs, Vk * 4);
        const int cnt_off = take(vs, (Vk + 1) * 4);       // short mode: arrival counters, then start | length of every row list
        const int E = N * D1;
        // long rows (few vertices, many entries each): rank by bitmap, if the bitmap fits
        const bool bitmap = (long)E >= 16L * Vk && vs + 5 * Vk * W + 64 <= ido_off;
        const int bm_off = vs, pre_off = bm_off + Vk * W * 4;             // pre: entries before every 4-word group (u16)
        // short mode: u16 entry lists, rows padded to 8.  They are written after phase D, when the hash table (keys + slot ids,
        // at the end of LDS) is dead: if they fit there they take its place instead of more scratch -- what lets a
        // 1000-point frame build inside half the CU's LDS
        const int list_cap = (E + 7 * Vk + 8) & ~7;
        const bool list_in_hash = !bitmap && list_cap * 2 <= a.lds_total - ido_off;
        const int list_off = list_in_hash ? ido_off : vs;
        const int vs_end = bitmap ? pre_off + Vk * W / 2 : (list_in_hash ? vs : list_off + list_cap * 2);
        if (vs_end > ido_off || Vk >= 32767 || E + 7 * Vk >= 65535 || hdr->fail || (DUAL && role == 1 && Vk > kDualVcap)) {   // does not fit: leave the frame to the fallback path
            if (DUAL && role == 1) {
                helper_fail();
                return;
            }
            flag_unfit();
            publish_done();
            return;                                       // uniform: every lane read the same V and the same flag
        }
        unsigned *vkey = reinterpret_cast<unsigned *>(smem + vkey_off);
        unsigned *cnt = reinterpret_cast<unsigned *>(smem + cnt_off);
        unsigned *bm = reinterpret_cast<unsigned *>(smem + bm_off);
        unsigned short *pre = reinterpret_cast<unsigned short *>(smem + pre_off);
        unsigned short *list = reinterpret_cast<unsigned short *>(smem + list_off);
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int j = 0; j < D1; ++j)
                if (got[s][j] == kEmptyKey) {
                    ido[slot[s][j]] = (unsigned short)id;
                    vkey[id] = key[s][j];
                    ++id;
                }
        for (int v = tid; v <= Vk; v += NT) cnt[v] = 0u;
        {
            unsigned *nbz = reinterpret_cast<unsigned *>(smem + lay.nbr[k]);   // absent neighbours stay 0
            for (int u = tid; u < D1 * Vk; u += NT) nbz[u] = 0u;
        }
        if (bitmap) {
            uint4 *b4 = reinterpret_cast<uint4 *>(bm);
            for (int u = tid; u < Vk * W / 4; u += NT) b4[u] = make_uint4(0u, 0u, 0u, 0u);
        } else if (!list_in_hash) {
            uint4 *l4 = reinterpret_cast<uint4 *>(list);   // pads compare greater than every entry (entries are < 0x7fff)
            for (int u = tid; u < list_cap / 8; u += NT) l4[u] = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
        }
        if (tid == 0) {
            reinterpret_cast<float2 *>(smem + lay.val[k][0])[0] = make_float2(0.f, 0.f);
            reinterpret_cast<float2 *>(smem + lay.val[k][1])[0] = make_float2(0.f, 0.f);
        }
        FL_PSTAMP();
        __syncthreads();
        FL_PSTAMP();

        // ---- D: every entry learns its vertex and joins the vertex's row; blur neighbours per vertex ------
        unsigned vid[PPT][D1], arr[PPT][D1];              // vertex id, arrival index inside the row (short mode)
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int j = 0; j < D1; ++j) vid[s][j] = (tid + s * NT < Npad) ? (unsigned)ido[slot[s][j]] : 0u;
        FL_PSTAMP();
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * NT;
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                arr[s][j] = 0u;
                if (i < N) {                              // real points only: phantoms add vertices, not products
                    if (bitmap) atomicOr(&bm[vid[s][j] * W + (i >> 5)], 1u << (i & 31));
                    else arr[s][j] = atomicAdd(&cnt[vid[s][j]], 1u);
                }
            }
        }
        FL_PSTAMP();
        {
            // permutohedral_cpu.h:408-421 with d = 2: n1 = key - 1 (coordinate `axis`: + d), n2 = key + 1 (axis: - d).
            // The relation is mutual -- B = n2_j(A) iff A = n1_j(B) -- so one probe per (axis, vertex) finds n2 and
            // fills both halves; absent neighbours keep the 0 the table was cleared to.
            unsigned short *nb16 = reinterpret_cast<unsigned short *>(smem + lay.nbr[k]);   // [axis][vertex][n1+1, n2+1]
            for (int t = tid; t < D1 * Vk; t += NT) {
                const int j = t >= 2 * Vk ? 2 : (t >= Vk ? 1 : 0), v = t - j * Vk;
                const unsigned kk = vkey[v];
                const unsigned qx = ((kk & 0xffffu) + (j == 0 ? 0xfffeu : 1u)) & 0xffffu, qy = ((kk >> 16) + (j == 1 ? 0xfffeu : 1u)) & 0xffffu;
                const unsigned q = qx | (qy << 16);
                unsigned h = hash32(q) & mask;
                unsigned o = hk[h];
                for (int probes = 0; o != kEmptyKey && o != q && probes < hcap; ++probes) {
                    h = (h + 1u) & mask;
                    o = hk[h];
                }
                if (o == q) {
                    const unsigned b = ido[h];
                    nb16[2 * t + 1] = (unsigned short)(b + 1u);                       // my n2
                    nb16[2 * (j * Vk + (int)b)] = (unsigned short)(v + 1);          // its n1
                }
            }
        }
        FL_PSTAMP();
        __syncthreads();
        FL_PSTAMP();
        if (list_in_hash) {                               // the hash table is dead now (the scans of phase E separate this from F's writes)
            uint4 *l4 = reinterpret_cast<uint4 *>(list);
            for (int u = tid; u < list_cap / 8; u += NT) l4[u] = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
        }

        // ---- E: row lengths -> row starts (and, short mode, the start of every padded entry list) -------
        if (bitmap) {
            // prefix popcounts of every vertex's bitmap, one 16-lane group per vertex (W <= 128 words = 32 groups of 4
            // words: 2 groups per lane at most): pre[v][g] = entries of v in points before word group g
            const int lane16 = tid & 15, ng = W >> 2, gpl = (ng + 15) >> 4;
            for (int v = tid >> 4; v < Vk; v += NT / 16) {
                int pc[2] = {0, 0}, sum = 0;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int g = lane16 * gpl + u;
                    if (u < gpl && g < ng) {
                        const uint4 b = *reinterpret_cast<const uint4 *>(bm + v * W + 4 * g);
                        pc[u] = __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
                    }
                    sum += pc[u];
                }
                int incl = sum;                            // scan inside the 16-lane row only
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);
                int run = incl - sum;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int g = lane16 * gpl + u;
                    if (u < gpl && g < ng) pre[v * ng + g] = (unsigned short)run;
                    run += pc[u];
                }
                if (lane16 == 15) cnt[v] = (unsigned)incl;
            }
            __syncthreads();
        }
        {
            // packed scan: low half = products before the row, high half = padded list entries before it
            const int vper = (Vk + 1 + NT - 1) / NT, v0 = tid * vper;
            unsigned sum = 0u;
            int mx = 0;
            for (int u = 0; u < vper; ++u) {
                const int v = v0 + u;
                if (v < Vk) {
                    const unsigned n = cnt[v];
                    sum += n | (((n + 7u) & ~7u) << 16);
                    mx = max(mx, (int)n);
                }
            }
            int tot;
            unsigned run = (unsigned)block_excl_scan<NT>((int)sum, tid, hdr->wave_sum, tot);
            unsigned short *row = reinterpret_cast<unsigned short *>(smem + lay.row[k]);
            for (int u = 0; u < vper; ++u) {
                const int v = v0 + u;
                if (v <= Vk) {
                    const unsigned n = v < Vk ? cnt[v] : 0u;
                    row[v] = (unsigned short)(run & 0xffffu);
                    cnt[v] = (run >> 16) | (n << 16);      // list start | row length
                    run += n | (((n + 7u) & ~7u) << 16);
                }
            }
            if (k == 0) {                                 // longest row of kernel 0 decides the chain path
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
                if ((tid & 63) == 0 && mx > 0) atomicMax(&hdr->rowmax, mx);
            }
        }
        __syncthreads();
        FL_PSTAMP();

        // ---- F/G: the place of every entry in its row = number of smaller entries of the same vertex ----
        if constexpr (PPT >= 3) {
            // 3-4 points per lane: one point slot at a time, nothing kept across slots but vid / arr (the all-slots-at-once
            // form below keeps 27 more registers live and spilled them: 56-208 bytes of scratch per lane)
            const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
            if (!bitmap) {
#pragma unroll
                for (int s = 0; s < PPT; ++s) {
                    const int i = tid + s * NT;
                    if (i < N) {
#pragma unroll
                        for (int j = 0; j < D1; ++j) list[(cnt[vid[s][j]] & 0xffffu) + arr[s][j]] = (unsigned short)(i * D1 + j);
                    }
                }
                __syncthreads();
            }
            FL_PSTAMP();
            auto below = [](const uint4 &x, unsigned e) {
                const unsigned e2 = e | (e << 16), hi = 0x80008000u;
                return 8u - (unsigned)(__popc(((x.x | hi) - e2) & hi) + __popc(((x.y | hi) - e2) & hi) + __popc(((x.z | hi) - e2) & hi) +
                                       __popc(((x.w | hi) - e2) & hi));
            };
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const int i = tid + s * NT, ic = min(i, N - 1);
                const bool real = i < N;
#pragma unroll
                for (int j = 0; j < D1; ++j) {
                    const unsigned v = vid[s][j];
                    unsigned r;
                    if (bitmap) {
                        const uint4 bq = *reinterpret_cast<const uint4 *>(bm + v * W + ((ic >> 5) & ~3));
                        const int wq = (ic >> 5) & 3;
                        const unsigned low = (1u << (ic & 31)) - 1u;
                        r = pre[v * (W >> 2) + (ic >> 7)] + (wq > 0 ? __popc(bq.x) : 0) + (wq > 1 ? __popc(bq.y) : 0) + (wq > 2 ? __popc(bq.z) : 0) +
                            __popc((wq == 0 ? bq.x : wq == 1 ? bq.y : wq == 2 ? bq.z : bq.w) & low);
                    } else {
                        const unsigned lcv = cnt[v], e = (unsigned)(i * D1 + j);
                        const uint4 *lp = reinterpret_cast<const uint4 *>(list + (lcv & 0xffffu));
                        r = below(lp[0], e);
                        const int n8 = (int)(((lcv >> 16) + 7u) >> 3);
                        for (int u = 1; u < n8; ++u) r += below(lp[u], e);
                    }
                    pk[s][k][j] = (v + 1u) | (((unsigned)row[v] + (real ? r : 0u)) << 16);
                }
            }
        } else {
            const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
            unsigned lc[PPT][D1], rw[PPT][D1];
    #pragma unroll
            for (int s = 0; s < PPT; ++s)
    #pragma unroll
                for (int j = 0; j < D1; ++j) {
                    lc[s][j] = cnt[vid[s][j]];
                    rw[s][j] = row[vid[s][j]];
                }
            if (!bitmap) {
    #pragma unroll
                for (int s = 0; s < PPT; ++s) {
                    const int i = tid + s * NT;
                    if (i < N) {
    #pragma unroll
                        for (int j = 0; j < D1; ++j) list[(lc[s][j] & 0xffffu) + arr[s][j]] = (unsigned short)(i * D1 + j);
                    }
                }
                __syncthreads();
            }
            FL_PSTAMP();
            unsigned rank[PPT][D1];
            if (bitmap) {
    #pragma unroll
                for (int s = 0; s < PPT; ++s) {               // (one point at a time: three 16-byte reads in flight, 12 registers)
                    const int i = min(tid + s * NT, N - 1);
                    uint4 bw[D1];
                    unsigned pv[D1];
    #pragma unroll
                    for (int j = 0; j < D1; ++j) {
                        bw[j] = *reinterpret_cast<const uint4 *>(bm + vid[s][j] * W + ((i >> 5) & ~3));   // the 4-word group of my word
                        pv[j] = pre[vid[s][j] * (W >> 2) + (i >> 7)];
                    }
                    const int wq = (i >> 5) & 3;
                    const unsigned low = (1u << (i & 31)) - 1u;
    #pragma unroll
                    for (int j = 0; j < D1; ++j) {
                        const uint4 b = bw[j];
                        rank[s][j] = pv[j] + (wq > 0 ? __popc(b.x) : 0) + (wq > 1 ? __popc(b.y) : 0) + (wq > 2 ? __popc(b.z) : 0) +
                                     __popc((wq == 0 ? b.x : wq == 1 ? b.y : wq == 2 ? b.z : b.w) & low);
                    }
                }
            } else {
                // eight 16-bit entries against e at once: entries and e are < 0x8000, so (x | 0x8000) - e keeps bit 15 of a
                // half exactly when that half is >= e, and no half ever borrows from its neighbour
                auto below = [](const uint4 &x, unsigned e) {
                    const unsigned e2 = e | (e << 16), hi = 0x80008000u;
                    return 8u - (unsigned)(__popc(((x.x | hi) - e2) & hi) + __popc(((x.y | hi) - e2) & hi) + __popc(((x.z | hi) - e2) & hi) +
                                           __popc(((x.w | hi) - e2) & hi));
                };
    #pragma unroll
                for (int s = 0; s < PPT; ++s) {
                    const int i = tid + s * NT;
                    uint4 first[D1];
    #pragma unroll
                    for (int j = 0; j < D1; ++j) first[j] = *reinterpret_cast<const uint4 *>(list + (lc[s][j] & 0xffffu));
    #pragma unroll
                    for (int j = 0; j < D1; ++j) {
                        const unsigned e = (unsigned)(i * D1 + j);
                        unsigned r = below(first[j], e);
                        const uint4 *lp = reinterpret_cast<const uint4 *>(list + (lc[s][j] & 0xffffu));
                        const int n8 = (int)(((lc[s][j] >> 16) + 7u) >> 3);
                        for (int u = 1; u < n8; ++u) r += below(lp[u], e);    // rows of more than 8 entries
                        rank[s][j] = r;
                    }
                }
            }
    #pragma unroll
            for (int s = 0; s < PPT; ++s)
    #pragma unroll
                for (int j = 0; j < D1; ++j) {
                    const bool real = tid + s * NT < N;
                    pk[s][k][j] = (vid[s][j] + 1u) | ((rw[s][j] + (real ? rank[s][j] : 0u)) << 16);
                }
        }
        if (k == 0) row0max = hdr->rowmax;
        FL_PSTAMP();
        __syncthreads();                                  // the next kernel's build (or the loop's product buffers) reuses the scratch
        FL_PSTAMP();
    }
    FL_STAMP();

    // norm = 1 / (compute(ones) + 1e-20), pairwise3d.h:22-27, of the kernels in KM: one pass of the loop's own splat / blur /
    // slice with Q = 1.  Fills pr.wn (= w * norm, pairwise3d.h:77).
    ChainLane cl{0u, 0u};
    auto normalise = [&](auto km) {
        constexpr int KM = decltype(km)::value;
#pragma unroll
        for (int s = 0; s < PPT; ++s) pr.q[s] = make_float2(1.0f, 1.0f);
        splat_blur<PPT, K, 2, true, NT, KM>(smem, lay, V, N, tid, pr, cl, ins);
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (!((KM >> k) & 1)) continue;
                pr.wn[s][k] = 0.0f;
                if (tid + s * NT < N) {
                    const float t = slice_point(smem, lay, pr, s, k, a.alpha).x;
                    pr.wn[s][k] = a.w[k] * (1.0f / (t + 1e-20f));             // pairwise3d.h:26-27,77
                }
            }
        }
    };

    // unary energies (densecrf3d.h:116-129 with L = 2; the labels arrived long ago) and Q0 = softmax(-unary)
    int first_p = -1;
    auto begin_inference = [&]() {
        if (a.label) {
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const int t = lab_raw[s];
                const bool known = t >= 0 && t < 2;
                pr.un[s].x = !known ? a.tbl[0] : (t == 0 ? a.tbl[3] : a.tbl[1 + t]);
                pr.un[s].y = !known ? a.tbl[0] : (t == 1 ? a.tbl[4] : a.tbl[1 + t]);
            }
        }
        start_inference<PPT, K, NT>(pr, N, tid);
    };

    if (DUAL && role == 1) {
        // ---- helper: normalise kernel 1 on its own lattice, then its tables (neighbour table, row starts) and per-point
        //      records (vertex | row place words, barycentric weights, w * norm) into the hand-off area, then the flag ------
        const int Vk = V[1];
        V[0] = 0;
        lay.chain0 = 0;
        lay.prod_all = 1;
        lay.Ecap[1] = plane_floats(N, Vk, false);
        lay.prod[1] = cursor;
        if (cursor + lay.Ecap[1] * 8 > a.lds_total) {     // (uniform)
            helper_fail();
            return;
        }
        place_products<PPT, K, 2, NT, 2>(smem, lay, N, tid, pk, pr);
        normalise(std::integral_constant<int, 2>{});
        FL_STAMP();
        if (kInstr && a.drop_helper) return;              // (fault injection, instrumented builds only: the main workgroup must time out and fall back)
        const unsigned *nb = reinterpret_cast<const unsigned *>(smem + lay.nbr[1]);
        for (int u = tid; u < D1 * Vk; u += NT) xs[kDualNbr + u] = nb[u];
        const unsigned *rw = reinterpret_cast<const unsigned *>(smem + lay.row[1]);      // (Vk + 2) u16, the area is 4-byte aligned
        for (int u = tid; u < (Vk + 3) / 2; u += NT) xs[kDualRow + u] = rw[u];
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
#pragma unroll
            for (int j = 0; j < D1; ++j)
                reinterpret_cast<uint2 *>(xs + kDualRec)[(s * D1 + j) * NT + tid] = make_uint2(pk[s][1][j], __float_as_uint(pr.bary[s][1][j]));
            xs[kDualWn + s * NT + tid] = __float_as_uint(pr.wn[s][1]);
        }
        __syncthreads();
        if (tid == 0) {                                   // plain stores -> barrier -> one agent-scope release -> drained -> relaxed flag store
            __threadfence();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ONE 8-byte word carries the epoch and the vertex count (bit 63: "could not be handed over"): the reader's
            // poll is a single round trip to memory, not three
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(xs), (unsigned long long)a.dual_epoch | ((unsigned long long)(unsigned)Vk << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        FL_STAMP();
        if (kInstr && a.timing && (int)blockIdx.x == a.timing_block && tid == a.timing_lane) a.timing[63] = ins.n;
        return;
    }
    if (DUAL) {
        // ---- main: kernel 0's product buffer right behind its tables (sized for either kernel: it is the shared buffer when the
        //      two do not fit side by side), rank its chain rows and normalise it -- all of that while the helper is still
        //      building; then wait for the helper (bounded: a helper that never comes makes this frame fall back, it cannot
        //      hang the launch) and take kernel 1's tables into LDS behind the buffer, its records and norms into registers ---
        lay.chain0 = chain_wanted(N, V[0], row0max, NT) ? 1 : 0;
        lay.Ecap[0] = plane_floats(N, V[0], lay.chain0 != 0);
        lay.Ecap[1] = plane_floats(N, 0, false);          // (a short-row kernel's plane does not depend on its V)
        lay.prod[0] = cursor;
        cursor += (max(lay.Ecap[0], lay.Ecap[1]) * 8 + 15) & ~15;
        lay.prod_all = 1;
        if (cursor > a.lds_total) {
            flag_unfit();
            publish_done();
            return;
        }
        V[1] = 0;
        place_products<PPT, K, 2, NT, 1>(smem, lay, N, tid, pk, pr);
        if (lay.chain0) cl = chain_setup(smem, lay, V[0], tid);
        normalise(std::integral_constant<int, 1>{});
        // still nothing from the helper is needed: Q0 = softmax(-unary) and kernel 0's first products
        begin_inference();
        if (kFuseXP && PPT <= 2 && a.n_iter > 0) {
#pragma unroll
            for (int s = 0; s < PPT; ++s)                 // (the normalisation's row sums finished with the buffer four barriers ago)
                if (tid + s * NT < N) point_products<PPT, K, 2>(smem, lay, pr, s, 0);
            first_p = 2;
        }
        FL_STAMP();
        if (tid == 0) {
            int ok = 0;
            unsigned long long word = 0;
            for (int spin = 0; spin < (1 << 17); ++spin) {     // (~0.1 s at a memory round trip per poll: then the frame falls back)
                word = dual_load2(xs);
                if ((unsigned)word == a.dual_epoch) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            hdr->dual_ok = (ok && !(word >> 63)) ? 1 : 0;
            hdr->dual_V = (int)((word >> 32) & 0x7fffffffu);
        }
        __syncthreads();
        const int Vk = __builtin_amdgcn_readfirstlane(hdr->dual_V);        // (scalar: every LDS offset below derives from it)
        const int dual_ok = __builtin_amdgcn_readfirstlane(hdr->dual_ok);
        auto take = [&](int &o, int bytes) { const int r = o; o += (bytes + 15) & ~15; return r; };
        lay.val[1][0] = take(cursor, (Vk + 1) * 8);
        lay.val[1][1] = take(cursor, (Vk + 1) * 8);
        lay.nbr[1] = take(cursor, D1 * Vk * 4);
        lay.row[1] = take(cursor, (Vk + 2) * 2);
        if (!dual_ok || Vk < 0 || Vk > kDualVcap || cursor > a.lds_total) {
            flag_unfit();
            publish_done();
            return;
        }
        V[1] = Vk;
#pragma unroll
        for (int s = 0; s < PPT; ++s) {                   // (8-byte loads: half the instructions of the hand-off's import)
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                const unsigned long long r = dual_load2(xs + kDualRec + 2 * ((s * D1 + j) * NT + tid));
                pk[s][1][j] = (unsigned)r;
                pr.bary[s][1][j] = __uint_as_float((unsigned)(r >> 32));
            }
            pr.wn[s][1] = __uint_as_float(dual_load(xs + kDualWn + s * NT + tid));
        }
        unsigned long long *nb = reinterpret_cast<unsigned long long *>(smem + lay.nbr[1]);      // (16-byte aligned, sizes rounded up to 16)
        for (int u = tid; u < (D1 * Vk + 1) / 2; u += NT) nb[u] = dual_load2(xs + kDualNbr + 2 * u);
        unsigned *rw = reinterpret_cast<unsigned *>(smem + lay.row[1]);
        for (int u = tid; u < (Vk + 3) / 2; u += NT) rw[u] = dual_load(&xs[kDualRow + u]);
        if (tid == 0) {
            reinterpret_cast<float2 *>(smem + lay.val[1][0])[0] = make_float2(0.f, 0.f);
            reinterpret_cast<float2 *>(smem + lay.val[1][1])[0] = make_float2(0.f, 0.f);
        }
        if (cursor + lay.Ecap[1] * 8 <= a.lds_total) {    // own product buffer, else kernel 0's serves both (P / S one kernel at a time)
            lay.prod[1] = cursor;
            lay.total = cursor + lay.Ecap[1] * 8;
        } else {
            lay.prod[1] = lay.prod[0];
            lay.prod_all = 0;
            lay.total = cursor;
        }
        __syncthreads();
        place_products<PPT, K, 2, NT, 2>(smem, lay, N, tid, pk, pr);
        FL_STAMP();
    } else {
        // ---- loop-phase LDS plan: product buffers behind the persistent tables ---------------------------
        lay.chain0 = chain_wanted(N, V[0], row0max, NT) ? 1 : 0;
        {
            bool ok = false;
            for (int all = 1; all >= 0 && !ok; --all) {
                int o = cursor, shared = 0;
                for (int k = 0; k < K; ++k) {
                    lay.Ecap[k] = plane_floats(N, V[k], k == 0 && lay.chain0);
                    const int pb = lay.Ecap[k] * 8;
                    if (all) { lay.prod[k] = o; o += (pb + 15) & ~15; }
                    else shared = max(shared, pb);
                }
                if (!all) {
                    for (int k = 0; k < K; ++k) lay.prod[k] = o;
                    o += (shared + 15) & ~15;
                }
                lay.prod_all = all;
                lay.total = o;
                ok = o <= a.lds_total;
            }
            if (!ok) {
                flag_unfit();
                publish_done();
                return;
            }
        }
        place_products<PPT, K, 2, NT>(smem, lay, N, tid, pk, pr);
        if (lay.chain0) cl = chain_setup(smem, lay, V[0], tid);
        // every kernel at once
        normalise(std::integral_constant<int, (1 << K) - 1>{});
    }
    FL_STAMP();                                           // (no barrier: the next writer of val[.][1] is two barriers away)

    if (!DUAL) begin_inference();
    float alpha[K];
#pragma unroll
    for (int k = 0; k < K; ++k) alpha[k] = a.alpha;
    mean_field<PPT, K, 2, NT>(smem, lay, V, N, tid, pr, cl, alpha, a.n_iter, a.relax, ins, first_p);
    store_results<PPT, K, NT>(c, f, N, tid, pr, a.with_map);
    if (tid < K && a.V_out[tid]) a.V_out[tid][f] = tid == 0 ? V[0] : V[K - 1];
    if (tid == 0 && a.frame_status) a.frame_status[f] = 0;
    publish_done();
    FL_STAMP();
    if (kInstr && a.timing && (int)blockIdx.x == a.timing_block && tid == a.timing_lane) a.timing[63] = ins.n;
}

int frame_hcap(int NA)
{
    const int live = 3 * ((NA + 3) & ~3);
    int h = 1024;
    while (h < live && h < 8192) h <<= 1;
    return h;
}

template <int NT, int PPT, int K>
void launch_frame_ppt(const CrfDev &c, const FrameArgs &a, hipStream_t s)
{
    auto fn = k_frame<NT, PPT, K>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
    fn<<<dim3(c.F), dim3(NT), a.lds_total, s>>>(c, a);
}

template <int PPT>
void launch_frame_dual(const CrfDev &c, const FrameArgs &a, hipStream_t s)
{
    auto fn = k_frame<kNT, PPT, 2, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
    fn<<<dim3(2 * c.F), dim3(kNT), a.lds_total, s>>>(c, a);
}

}  // namespace

bool frame_supported(const CrfDev &c, const KernelDev *kds)
{
    const int NA = c.activeN > 0 ? c.activeN : c.maxN;
    if (c.L != 2 || c.K < 1 || c.K > kMaxFusedK || NA < 1 || NA > 4 * kNT) return false;
    for (int k = 0; k < c.K; ++k)
        if (kds[k].d != 2) return false;
    return true;
}

size_t frame_dual_bytes(int frames) { return (size_t)frames * kDualWords * sizeof(unsigned); }

bool frame_lean_wanted(const CrfDev &c)
{
    const int NA = c.activeN > 0 ? c.activeN : c.maxN;
    return frame_lean_plausible(NA, c.K, c.F);
}
size_t frame_lean_rec_bytes(int frames) { return (size_t)frames * kLeanRecBytes; }

int launch_frame(const CrfDev &c, const KernelDev *kds, int n_iter, int with_map, float relax, int *status, int *frame_status,
                 const int16_t *label, const float *tbl5, hipStream_t s, bool allow_small, unsigned *dual, unsigned dual_epoch,
                 unsigned *done, unsigned done_epoch, unsigned char *lean_rec)
{
    FrameArgs a{};
    for (int k = 0; k < c.K; ++k) {
        a.feat[k] = kds[k].feat;
        a.w[k] = kds[k].w;
        a.V_out[k] = kds[k].V;
    }
    a.scale[0] = kds[0].scale[0];
    a.scale[1] = kds[0].scale[1];
    a.inv_dp1 = kds[0].inv_dp1;
    a.alpha = kds[0].alpha;
    a.maxN = c.maxN;
    a.label = label;
    if (label)
        for (int i = 0; i < 5; ++i) a.tbl[i] = tbl5[i];
    a.n_iter = n_iter;
    a.with_map = with_map;
    a.relax = relax;
    a.omr = 1 - relax;
    a.rec = lean_rec;
    const int NA = c.activeN > 0 ? c.activeN : c.maxN;
    a.hcap = frame_hcap(NA);
    a.lds_total = (int)kLdsLimit;
    a.status = status;
    a.frame_status = frame_status;
    a.dual = dual;
    a.dual_epoch = dual_epoch;
    a.done = c.F == 1 ? done : nullptr;
#if LCCRF_INSTRUMENT
    static const bool drop_helper = ab_env("LCCRF_DUAL_DROP_HELPER") != nullptr;   // fault injection: not compiled into the release library
    a.drop_helper = drop_helper ? 1 : 0;
#endif
    a.n_single = (c.F == 1 && done && c.activeN > 0) ? c.activeN : -1;   // (object API: activeN IS the frame's count)
    a.done_epoch = done_epoch;
    static long long *timing_buf = nullptr;
    static const bool want_timing = kInstr && ab_env("LCCRF_FRAME_TIMING") != nullptr;
    if (want_timing && !timing_buf) (void)hipMalloc(&timing_buf, 64 * sizeof(long long));
    a.timing = want_timing ? timing_buf : nullptr;
    a.timing_block = want_timing ? std::max(atoi(ab_env("LCCRF_FRAME_TIMING")) - 1, 0) : 0;
    if (a.timing_block >= (dual ? 2 : 1) * c.F) a.timing_block = 0;     // (two-workgroup form: block 2f is frame f's main workgroup, 2f + 1 its helper)
    a.timing_lane = (want_timing && ab_env("LCCRF_FRAME_TIMING_LANE")) ? atoi(ab_env("LCCRF_FRAME_TIMING_LANE")) & (kNT - 1) : 0;
    // Small frames: 512 lanes and half the CU's LDS per frame, so that two frames share a CU.  A frame whose lattices
    // do not fit that plan (or whose long rows need more chain lanes than four wavefront pairs have) flags itself
    // and is re-run like any other frame that does not fit.
    // The lattice sizes are not known before the kernel has built them: the small shape is chosen when a frame of NA points
    // with lattices of the usual SLAM proportions fits half the LDS in both phases -- an appearance kernel of ~112
    // vertices, a smoothness kernel of min(NA + 350, 1150) (734 vertices at 400 points, 985 at 700, 1071 at 1000 on
    // 640x480 images with an 18-pixel kernel) -- i.e. up to 1024 points.  Frames with larger lattices flag themselves and
    // are re-run; an engine that sees more than 1/8 of a batch flagged stops asking for this shape (allow_small).
    static const bool no_small = ab_env("LCCRF_NO_SMALL_WG") != nullptr;   // A/B switch: same results either way
    bool small = allow_small && !no_small && NA <= 2 * kNTSmall && c.F >= kSmallMinFrames;
    if (small) {
        int vest[kMaxFusedK], tables = 0;
        for (int k = 0; k < c.K; ++k) {
            vest[k] = (c.K > 1 && k == 0) ? 112 : std::min(std::min(3 * NA, NA + 350), 1150);
            tables += 2 * (vest[k] + 1) * 8 + kD1 * vest[k] * 4 + (vest[k] + 2) * 2 + 64;
        }
        FusedLayout est;
        const int build_bytes = kHdr + tables + 8 * vest[c.K - 1] + 6 * frame_hcap(NA) + 64;   // tables + vertex keys + counters + hash table
        small = layout_core(NA, c.K, vest, 1 << 20, &est, kNTSmall, kLdsHalf) && build_bytes <= (int)kLdsHalf;
    }
#define FRAME_CASE(NT, P)                                      \
    case P:                                                    \
        if (c.K == 1) launch_frame_ppt<NT, P, 1>(c, a, s);     \
        else launch_frame_ppt<NT, P, 2>(c, a, s);              \
        break;
    static const bool no_dual = ab_env("LCCRF_NO_DUAL") != nullptr;         // A/B switch: same results either way
    // Full-size frames (1025 .. 2048 points, the two-kernel SLAM configuration): 512 lanes and half the CU's LDS per frame as well
    // (frame_lean.hip) -- flagged frames and the engine's give-up rule as for the small shape.
    static const bool no_lean = ab_env("LCCRF_NO_FRAME_LEAN") != nullptr;  // A/B switch: same results either way
    const bool lean = lean_rec && !no_lean && frame_lean_plausible(NA, c.K, c.F);
    if (lean) {
        small = false;
        a.lds_total = (int)kLdsHalf;
        launch_frame_lean(c, a, NA, s);
    } else if (dual && !no_dual && c.K == 2 && !small) {                           // a frame alone: two workgroups, one per lattice build
        switch ((NA + kNT - 1) / kNT) {
        case 1: launch_frame_dual<1>(c, a, s); break;
        case 2: launch_frame_dual<2>(c, a, s); break;
        case 3: launch_frame_dual<3>(c, a, s); break;
        case 4: launch_frame_dual<4>(c, a, s); break;
        default: break;
        }
    } else if (small) {
        a.lds_total = (int)kLdsHalf;
        switch ((NA + kNTSmall - 1) / kNTSmall) {
            FRAME_CASE(kNTSmall, 1)
            FRAME_CASE(kNTSmall, 2)
        default: break;
        }
    } else {
        switch ((NA + kNT - 1) / kNT) {
            FRAME_CASE(kNT, 1)
            FRAME_CASE(kNT, 2)
            FRAME_CASE(kNT, 3)
            FRAME_CASE(kNT, 4)
        default: break;
        }
    }
#undef FRAME_CASE
    if (a.timing) {                       // instrumented builds: synchronous read-back of one workgroup's stamps
        long long h[64];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, a.timing, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[lccrf frame timing] %lld stamps, deltas (shader clocks):", h[63]);
        for (int i = 1; i < h[63] && i < 63; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
        fprintf(stderr, "\n");
    }
    return lean ? 2 : small ? 1 : 0;
}

}  // namespace lccrf
