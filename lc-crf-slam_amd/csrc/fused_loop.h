// fused_loop.h -- the mean-field loop of one frame inside one 1024-lane workgroup, shared by the two
// one-workgroup-per-frame kernels:
//     k_fused (fused_engine.hip)   inference on lattices a build kernel left in HBM
//     k_frame (frame_engine.hip)   lattice build + normalisation + inference in ONE launch
//
// State of a frame while it is being iterated:
//     registers  per point (lane t owns points t, t+1024, ...): unary, Q, and per kernel the three vertex
//                ids, product slots, barycentric weights and w*norm                       (PointRegs)
//     LDS        lattice side only: the splat products of every kernel, both ping-pong value arrays, the
//                blur neighbour table (u16 pairs) and the row pointers                    (FusedLayout)
//
// Bit-exactness: the reference splats sequentially over points (permutohedral_cpu.h:653-661), so a
// vertex's value is a left-to-right fp32 sum in ascending point order.  Phase P writes every product
// bary*Q into its row at the place the build recorded for it (exact, order-free); phase S then adds each
// row strictly left to right -- chain_rows for the long rows of the appearance kernel.  Nothing is
// re-associated, nothing is fused (-ffp-contract=off).  L = 2 labels, 2-D kernels only.
#pragma once

#include "engine.h"
#include "device_math.h"

namespace lccrf {
namespace fl {

constexpr int kNT = 1024;                 // lanes per workgroup (16 wavefronts)
constexpr int kMaxFusedK = 2;
constexpr int kD1 = 3;                    // 2-D kernels: three simplex corners per point
constexpr int kChainGap = 14;             // product slots reserved per chain row beyond its products (see pst)
#ifndef LCCRF_CHAIN_TOP
#define LCCRF_CHAIN_TOP 16                // A/B (scripts/gpu_ab_build.sh "" "-DLCCRF_CHAIN_TOP=32"): 8, 16 or 32
#endif
constexpr int kChainTop = LCCRF_CHAIN_TOP; // rows of the first chain wavefront pair (see chain_setup)
// which of the first pair's kChainTop rows lane `ln` of its wavefront sums (-1: none): the same number of lanes in each of the four
// 16-lane groups a ds_read_b128 is served in
__host__ __device__ constexpr int chain_top_rank(int ln)
{
    return kChainTop == 32 ? (((ln & 8) == 0) ? ((ln & 7) | ((ln >> 4) << 3)) : -1)
         : kChainTop == 8  ? (((ln & 0x1c) == 0) ? ((ln & 3) | ((ln >> 5) << 2)) : -1)
                           : (((ln & 0x18) == 0) ? ((ln & 7) | ((ln >> 5) << 3)) : -1);
}
static_assert(kChainTop == 8 || kChainTop == 16 || kChainTop == 32, "chain_top_rank");
constexpr size_t kLdsLimit = 160 * 1024;  // MI355X: 160 KiB LDS per CU, one workgroup may own it all
constexpr int kChainMinRow = 64;          // kernel 0 runs chain_rows when its longest splat row has at least this many products ...
// ... and it has at most this many vertices (one lane per (vertex,label) row: wavefront pair 0 takes the kChainTop longest
// rows, every further pair of the workgroup's nt / 128 pairs 64 rows)
__host__ __device__ constexpr int chain_max_v(int nt) { return kChainTop + (nt / 128 - 1) * 64; }
constexpr int kChainMaxV = chain_max_v(kNT);
// Small frames: a 512-lane workgroup needs half the registers of the CU and, up to ~1000 points, less than half its LDS,
// so TWO frames share a CU and one frame's ordered row sums (one wavefront, latency-bound) run under the other's
// point phases (VALU / LDS-bound) -- across workgroups, where the hardware arbitrates, not inside one.
constexpr int kNTSmall = 512;
constexpr int kSmallMinFrames = 256;       // fewer frames than CUs: nothing to share a CU with, 1024 lanes finish a frame sooner
constexpr size_t kLdsHalf = kLdsLimit / 2;

#ifndef LCCRF_FUSE_XP
#define LCCRF_FUSE_XP 1
#endif
constexpr bool kFuseXP = LCCRF_FUSE_XP != 0;     // A/B switch (scripts/gpu_ab_build.sh): products written right behind each point's softmax
#ifndef LCCRF_INSTRUMENT
#define LCCRF_INSTRUMENT 0
#endif
constexpr bool kInstr = LCCRF_INSTRUMENT != 0;   // `make INSTRUMENT=1`; the release library has no stamps and no phase-skipping switches

struct FusedLayout {                      // byte offsets into dynamic LDS
    int zero;                             // 64 bytes of zeros (what a finished chain lane keeps reading); must be >= 128
    int prod[kMaxFusedK];                 // float  [2][Ecap_k] (label-major) for a chain kernel, float2 [Ecap_k] otherwise; aliased when !prod_all
    int val[kMaxFusedK][2];               // float2 [V_k+1]     slot 0 = absent neighbour = 0
    int nbr[kMaxFusedK];                  // u32    [D1][V_k]   (n1+1) | (n2+1)<<16
    int row[kMaxFusedK];                  // u16    [V_k+1]
    int Ecap[kMaxFusedK];                 // floats per label plane of prod
    int prod_all;                         // 1: every kernel has its own product buffer
    int chain0;                           // 1: kernel 0 has long splat rows, S runs chain_rows on them
    int pstart;                           // lean plan only (fused_lean.h): u16 [V_0+2] first product slot of every chain row
    int total;
};

// Instrumented builds only: shader-clock stamps of one workgroup and the timing experiments of LCCRF_FUSED_DBG
// (1 skip short-row S, 2 skip chain S -- wrong results, timing only; 4 poison LDS; 8 prologue stamps).
struct Instr {
    long long *timing;
    int timing_block;
    int dbg;
    int n;
    int lane;                             // the lane whose clock is stamped (LCCRF_*_TIMING_LANE, default 0)
};
#define FL_DBG(bit) (::lccrf::fl::kInstr && (ins.dbg & (bit)))
#define FL_STAMP()                                                                                                   \
    do {                                                                                                             \
        if (::lccrf::fl::kInstr && ins.timing && (int)blockIdx.x == ins.timing_block && (int)threadIdx.x == ins.lane) ins.timing[ins.n++] = clock64(); \
    } while (0)
#define FL_PSTAMP()                   \
    do {                              \
        if (FL_DBG(8)) FL_STAMP();    \
    } while (0)

// Where row v of a chain kernel starts in its product plane: rows are re-placed at ceil4(row[v] + 14 v) --
// starts are multiples of 4 and pst(v+1) - pst(v) is a multiple of 4 that is >= the row's length + 11, i.e.
// there is room for the row padded to 4 products plus the eight zeros chain_rows wants behind it, without a scan.
__host__ __device__ inline int pst(int r0, int v) { return (r0 + kChainGap * v + 3) & ~3; }

// Size of the product plane of kernel k (floats per label) for frames of NA points.
__host__ __device__ inline int plane_floats(int NA, int V, bool chain) { return ((chain ? NA * kD1 + kChainGap * V + 16 : NA * kD1) + 63) & ~63; }

__host__ __device__ inline bool chain_wanted(int NA, int V0, int row0, int nt = kNT)
{
    return row0 >= kChainMinRow && V0 <= chain_max_v(nt) && NA * kD1 + kChainGap * V0 + 64 < 65535;
}

// LDS plan of one workgroup for frames of at most NA points whose K lattices (all 2-D) have at most
// V[k] vertices and kernel 0's longest row has `row0` products.  Shared by the host (batch API: sizes
// maximised over the frames) and k_fused itself (late-bound launches: each frame sizes its own).
__host__ __device__ inline bool layout_core(int NA, int K, const int *V, int row0, FusedLayout *lay, int nt = kNT,
                                            size_t lds_limit = kLdsLimit)
{
    constexpr int D1 = kD1;
    if (NA < 1 || NA > 4 * kNT || K < 1 || K > kMaxFusedK) return false;
    for (int k = 0; k < K; ++k)
        if (V[k] >= 65535) return false;
    const int chain0 = chain_wanted(NA, V[0], row0, nt);
    for (int all = 1; all >= 0; --all) {                  // own product buffers, else one shared buffer
        FusedLayout L{};
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return (int)r; };
        L.prod_all = all;
        L.chain0 = chain0;
        (void)take(128);                                   // chain_rows wants the zero block at an address >= 128
        L.zero = take(64);
        size_t shared_prod = 0;
        for (int k = 0; k < K; ++k) {
            L.Ecap[k] = plane_floats(NA, V[k], k == 0 && chain0);
            L.val[k][0] = take((size_t)(V[k] + 1) * sizeof(float2));
            L.val[k][1] = take((size_t)(V[k] + 1) * sizeof(float2));
            L.nbr[k] = take((size_t)D1 * V[k] * sizeof(unsigned));
            L.row[k] = take((size_t)(V[k] + 2) * sizeof(unsigned short));
            const size_t pb = (size_t)L.Ecap[k] * 2 * sizeof(float);
            if (all) L.prod[k] = take(pb);
            else shared_prod = pb > shared_prod ? pb : shared_prod;
        }
        if (!all) {
            const int p = take(shared_prod);
            for (int k = 0; k < K; ++k) L.prod[k] = p;
        }
        L.total = (int)o;
        if (o <= lds_limit) {
            *lay = L;
            return true;
        }
    }
    return false;
}

// ---- ordered row sums for kernels with long splat rows -----------------------------------
// The appearance kernel of a SLAM frame puts ~2000 points on ~120 lattice vertices: a few rows
// hold 300-600 products that must be added strictly left to right, one lane per (vertex,label)
// row.  The floor is the dependent v_add_f32 latency, 5.1 cycles per product; the compiler's
// schedule of the plain loop (8 ds_read_b32, s_waitcnt 0, 8 adds) runs at ~30, and its schedule
// of a software-pipelined C++ loop at ~13 (scripts/ubench/rowchain.hip, feedcost.hip).
// chain_rows is that loop written out by hand, ~8 cycles per product:
//   * a ring of four 8-product units in v96..v127 (two ds_read_b128 each): while unit u is added,
//     units u+1..u+3 are in flight; LDS data returns in order, so lgkmcnt(6) right after a unit
//     was issued means the oldest of the four has landed;
//   * a row is stored as [products][+0 up to a multiple of 4][eight +0]: a lane that has run out
//     of row keeps reading its own eight zeros (address clamp = one v_min_u32 per unit; no
//     compare/select, no EXEC games), and a row whose padded length is 8n+4 needs no tail code.
// Adding +0 is exact here: the accumulator starts at +0, and x + (+0) == x bit for bit for every
// x != -0, which cannot arise from +0 + ... (x + -x rounds to +0).
//   addr  LDS byte address of the lane's row (16-byte aligned)    units  ceil(row length / 8)
//   end   LDS byte address of the row's eight zeros                 trips  ceil(max units of the wavefront / 4)
#define LCCRF_ASM_ADD8(a, b, c, d, e, f, g, h)                                                        \
    "v_add_f32_e32 %[acc], %[acc], " #a "\n\tv_add_f32_e32 %[acc], %[acc], " #b "\n\t"               \
    "v_add_f32_e32 %[acc], %[acc], " #c "\n\tv_add_f32_e32 %[acc], %[acc], " #d "\n\t"               \
    "v_add_f32_e32 %[acc], %[acc], " #e "\n\tv_add_f32_e32 %[acc], %[acc], " #f "\n\t"               \
    "v_add_f32_e32 %[acc], %[acc], " #g "\n\tv_add_f32_e32 %[acc], %[acc], " #h "\n\t"

#define LCCRF_CHAIN_ROWS_ASM                                                                                         \
        "v_min_u32_e32 %[sel], %[ad], %[e0]\n\t"                                                                     \
        "ds_read_b128 v[96:99], %[sel]\n\tds_read_b128 v[100:103], %[sel] offset:16\n\t"                             \
        "v_min_u32_e32 %[sel], %[ad], %[e1]\n\t"                                                                     \
        "ds_read_b128 v[104:107], %[sel] offset:32\n\tds_read_b128 v[108:111], %[sel] offset:48\n\t"                 \
        "v_min_u32_e32 %[sel], %[ad], %[e2]\n\t"                                                                     \
        "ds_read_b128 v[112:115], %[sel] offset:64\n\tds_read_b128 v[116:119], %[sel] offset:80\n\t"                 \
        "1:\n\t"                                                                                                     \
        "v_min_u32_e32 %[sel], %[ad], %[e3]\n\t"                                                                     \
        "ds_read_b128 v[120:123], %[sel] offset:96\n\tds_read_b128 v[124:127], %[sel] offset:112\n\t"                \
        "s_waitcnt lgkmcnt(6)\n\t"                                                                                   \
        LCCRF_ASM_ADD8(v96, v97, v98, v99, v100, v101, v102, v103)                                                   \
        "v_add_u32_e32 %[ad], 0x80, %[ad]\n\t"                                                                       \
        "v_min_u32_e32 %[sel], %[ad], %[e0]\n\t"                                                                     \
        "ds_read_b128 v[96:99], %[sel]\n\tds_read_b128 v[100:103], %[sel] offset:16\n\t"                             \
        "s_waitcnt lgkmcnt(6)\n\t"                                                                                   \
        LCCRF_ASM_ADD8(v104, v105, v106, v107, v108, v109, v110, v111)                                               \
        "v_min_u32_e32 %[sel], %[ad], %[e1]\n\t"                                                                     \
        "ds_read_b128 v[104:107], %[sel] offset:32\n\tds_read_b128 v[108:111], %[sel] offset:48\n\t"                 \
        "s_waitcnt lgkmcnt(6)\n\t"                                                                                   \
        LCCRF_ASM_ADD8(v112, v113, v114, v115, v116, v117, v118, v119)                                               \
        "v_min_u32_e32 %[sel], %[ad], %[e2]\n\t"                                                                     \
        "ds_read_b128 v[112:115], %[sel] offset:64\n\tds_read_b128 v[116:119], %[sel] offset:80\n\t"                 \
        "s_waitcnt lgkmcnt(6)\n\t"                                                                                   \
        LCCRF_ASM_ADD8(v120, v121, v122, v123, v124, v125, v126, v127)                                               \
        "s_sub_u32 %[n], %[n], 1\n\t"                                                                                \
        "s_cmp_lg_u32 %[n], 0\n\t"                                                                                   \
        "s_cbranch_scc1 1b\n\t"                                                                                      \
        "s_waitcnt lgkmcnt(0)\n\t"
#define LCCRF_CHAIN_RING_CLOBBERS                                                                                      \
    "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",   \
        "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

__device__ __forceinline__ float chain_rows(unsigned addr, unsigned end, unsigned trips)
{
    float acc = 0.0f;
    if (trips == 0) return acc;
    const unsigned e1 = end - 32u, e2 = end - 64u, e3 = end - 96u;      // min(addr, end - k) + k == min(addr + k, end)
    unsigned sel;
    asm volatile(LCCRF_CHAIN_ROWS_ASM
                 : [acc] "+v"(acc), [ad] "+v"(addr), [n] "+s"(trips), [sel] "=&v"(sel)
                 : [e0] "v"(end), [e1] "v"(e1), [e2] "v"(e2), [e3] "v"(e3)
                 : "scc", "memory", LCCRF_CHAIN_RING_CLOBBERS);
    return acc;
}

// Everything a lane keeps about its points for the whole launch.
template <int PPT, int K>
struct PointRegs {
    float2 un[PPT], q[PPT];
    unsigned ix[PPT][K][3];               // (id0+1) | (id1+1) << 16,  (id2+1) | slot0 << 16,  slot1 | slot2 << 16
                                          //   id+1 = index into val (0 = absent), slot = index into prod
    float bary[PPT][K][kD1];
    float wn[PPT][K];                     // w * norm                            pairwise3d.h:77
};

// What a chain lane knows about the row it sums:  a = row address | wavefront max units << 18 ;
// b = 8-product units | (padded length is 8n+4) << 13 | pad slots << 14 | output index << 16
struct ChainLane {
    unsigned a, b;
};

// CH = 0 / 1: kernel 0 short-row / chain, fixed at launch;  CH = 2: decided per frame (lay.chain0)
template <int CH>
__device__ __forceinline__ bool chain_k(const FusedLayout &lay, int k)
{
    return CH == 2 ? (k == 0 && lay.chain0 != 0) : (((CH >> k) & 1) != 0);
}

// ---- where does each of my products go?  (once per launch) -----------------------------------
// pk[s][k][j] = (vertex id + 1) | place in the vertex's row << 16 -- the place of entry (i, j) in its
// vertex's row in ascending point order, the reference's splat order, counted from the start of the CSR
// (row[v] + rank).  A plain kernel stores its products at exactly that position; the chain kernel re-places
// row v at pst(row[v], v).  Fills ix.  The row tables must be in LDS and visible (barrier before).
// KMASK: the kernels to place (bit k); the frame kernel's two-workgroup form places them one at a time.
template <int PPT, int K, int CH, int NT = kNT, int KMASK = (1 << K) - 1>
__device__ __forceinline__ void place_products(unsigned char *smem, const FusedLayout &lay, int N, int tid,
                                               const unsigned (&pk)[PPT][K][kD1], PointRegs<PPT, K> &pr)
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (!((KMASK >> k) & 1)) continue;
        const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            pr.ix[s][k][0] = (pk[s][k][0] & 0xffffu) | (pk[s][k][1] << 16);
            pr.ix[s][k][1] = pk[s][k][2] & 0xffffu;
            pr.ix[s][k][2] = 0;
            if (tid + s * NT < N) {
                unsigned sl[kD1];
#pragma unroll
                for (int j = 0; j < kD1; ++j) {
                    sl[j] = pk[s][k][j] >> 16;
                    if (chain_k<CH>(lay, k)) {
                        const int v = (int)(pk[s][k][j] & 0xffffu) - 1;
                        const int r0 = row[v];
                        sl[j] = (unsigned)(pst(r0, v) + ((int)(pk[s][k][j] >> 16) - r0));
                    }
                }
                pr.ix[s][k][1] |= sl[0] << 16;
                pr.ix[s][k][2] = sl[1] | (sl[2] << 16);
            }
        }
    }
}

// Chain lanes: one lane per (vertex,label) row of kernel 0, long rows first so that whole wavefronts
// retire early.  Rank = counting sort on the row's 16-product block count (64 buckets, longest first;
// order inside a bucket is whatever the LDS atomics give -- it only decides which lane sums which row,
// never the order inside a row).  Uses the head of prod[0] as scratch; ends with a workgroup barrier.
// Every lane of the workgroup must call it (uniformly) when chain_k(0) holds.
__device__ __forceinline__ ChainLane chain_setup(unsigned char *smem, const FusedLayout &lay, int V0, int tid)
{
    constexpr int k = 0;
    ChainLane cl{0u, 0u};
    int *hist = reinterpret_cast<int *>(smem + lay.prod[k]);            // [64] counts, [64] bases
    if (tid < 128) hist[tid] = 0;
    __syncthreads();
    const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
    unsigned short *srt = reinterpret_cast<unsigned short *>(smem + lay.prod[k]) + 256;   // [V] vertex of rank r
    int key = 0, len = 0;
    if (tid < V0) {
        len = (int)row[tid + 1] - (int)row[tid];
        key = 63 - min((len + 3) >> 4, 63);
        atomicAdd(&hist[key], 1);
    }
    __syncthreads();
    if (tid < 64) {                                   // exclusive scan of the 64 bucket counts
        const int x = hist[tid];
        int incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o, 64);
            if (tid >= o) incl += y;
        }
        hist[64 + tid] = incl - x;
    }
    __syncthreads();
    if (tid < V0) srt[atomicAdd(&hist[64 + key], 1)] = (unsigned short)tid;
    __syncthreads();
    // Wavefront pair p = (2p, 2p+1) owns labels 0 and 1 of a rank range, so a wavefront reads one
    // label plane only.  Pair 0 takes just the kChainTop longest rows, four lanes in each of the
    // four 16-lane groups a ds_read_b128 is served in: the cost of a ring unit grows with the
    // bank conflicts among the ACTIVE lanes (~75 cycles with 16 rows, ~110 with 64), and it is
    // the longest rows' wavefront that everybody waits for.  Pair p >= 1: 64 rows each.
    const int l = (tid >> 6) & 1, pr = tid >> 7, ln = tid & 63;
    const int r = pr == 0 ? (chain_top_rank(ln) >= 0 ? chain_top_rank(ln) : V0) : kChainTop + ((pr - 1) << 6) + ln;
    unsigned nblk = 0, addr = 0;
    if (r < V0) {
        const int v = srt[r];
        const int r0 = row[v], rl = (int)row[v + 1] - r0, len4 = (rl + 3) & ~3;
        addr = (unsigned)(lay.prod[k] + 4 * (l * lay.Ecap[k] + pst(r0, v)));               // < 2^18
        nblk = (unsigned)((len4 + 7) >> 3);                                                    // 8-product units, < 2^13
        cl.b = nblk | ((unsigned)((len4 >> 2) & 1) << 13) | ((unsigned)(len4 - rl) << 14) | ((unsigned)((v + 1) * 2 + l) << 16);
    }
    unsigned m = nblk;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    cl.a = addr | (m << 18);
    __syncthreads();                      // the ranking scratch becomes the product buffer
    return cl;
}

// ---- splat = products (P) + ordered row sums (S), then the d+1 Jacobi blur passes ---------------
// On return val[k][kD1 & 1] holds the blurred lattice values of every kernel and every lane has passed
// the barrier behind the last blur pass.  permutohedral_cpu.h:653-679.
// Phase P for one point slot: the 3 products bary * Q per label of point (tid + s * 1024) into their rows of kernel k.
template <int PPT, int K, int CH>
__device__ __forceinline__ void point_products(unsigned char *smem, const FusedLayout &lay, const PointRegs<PPT, K> &pr, int s, int k)
{
    float *p0 = reinterpret_cast<float *>(smem + lay.prod[k]);
    float *p1 = p0 + lay.Ecap[k];
    float2 *p2 = reinterpret_cast<float2 *>(p0);
    const unsigned s0 = pr.ix[s][k][1] >> 16, s1 = pr.ix[s][k][2] & 0xffffu, s2 = pr.ix[s][k][2] >> 16;
    if (chain_k<CH>(lay, k)) {                             // chain kernel: one plane per label
        p0[s0] = pr.bary[s][k][0] * pr.q[s].x;
        p1[s0] = pr.bary[s][k][0] * pr.q[s].y;
        p0[s1] = pr.bary[s][k][1] * pr.q[s].x;
        p1[s1] = pr.bary[s][k][1] * pr.q[s].y;
        p0[s2] = pr.bary[s][k][2] * pr.q[s].x;
        p1[s2] = pr.bary[s][k][2] * pr.q[s].y;
    } else {                                              // short rows: labels interleaved
        p2[s0] = make_float2(pr.bary[s][k][0] * pr.q[s].x, pr.bary[s][k][0] * pr.q[s].y);
        p2[s1] = make_float2(pr.bary[s][k][1] * pr.q[s].x, pr.bary[s][k][1] * pr.q[s].y);
        p2[s2] = make_float2(pr.bary[s][k][2] * pr.q[s].x, pr.bary[s][k][2] * pr.q[s].y);
    }
}

// behind a chain row: +0 up to a multiple of 4, then eight +0 (what chain_rows reads past the end of the row)
__device__ __forceinline__ void chain_pads(unsigned char *smem, const ChainLane &cl)
{
    if (cl.b >> 16) {
        float *e = reinterpret_cast<float *>(smem + (cl.a & 0x3ffffu)) + ((cl.b & 0x1fffu) * 8u - ((cl.b >> 13) & 1u) * 4u);
        const unsigned npad = (cl.b >> 14) & 3u;
        for (unsigned z = 1; z <= npad; ++z) e[-(int)z] = 0.0f;
        reinterpret_cast<float4 *>(e)[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4 *>(e)[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// WITH_P = false (own product buffers only): the products are already in place -- mean_field writes the next
// iteration's products point by point right behind each point's softmax.
// KMASK: the kernels that take part (bit k) -- all of them in the loop; the frame kernel's two-workgroup form normalises
// its kernels one at a time (frame_engine.hip).
// REV: see the blur passes.
template <int PPT, int K, int CH, bool WITH_P = true, int NT = kNT, int KMASK = (1 << K) - 1, bool REV = false>
__device__ __forceinline__ void splat_blur(unsigned char *smem, const FusedLayout &lay, const int (&V)[K], int N, int tid,
                                           const PointRegs<PPT, K> &pr, const ChainLane &cl, Instr &ins)
{
    constexpr int D1 = kD1;
    auto phase_P = [&](int k) {
#pragma unroll
        for (int s = 0; s < PPT; ++s)
            if (tid + s * NT < N) point_products<PPT, K, CH>(smem, lay, pr, s, k);
        if (chain_k<CH>(lay, k)) chain_pads(smem, cl);    // (the buffer may have held another kernel's products)
    };
    // lanes [s_lo, NT) share the short-row kernels; the wavefronts that own the chain kernel's
    // longest rows keep out of them
    auto phase_S = [&](int k, int s_lo) {
        float *val = reinterpret_cast<float *>(smem + lay.val[k][0]);
        if (chain_k<CH>(lay, k)) {
            const int npairs = 1 + ((max(V[k] - kChainTop, 0) + 63) >> 6);
            if ((tid >> 7) < npairs && !FL_DBG(2)) {                                // whole wavefronts
                FL_PSTAMP();
                __builtin_amdgcn_s_setprio(3);            // a chain wavefront issues ahead of its SIMD's other wavefronts (+0.6 % on C2)
                if ((cl.b >> 16) != 0) {                  // lanes without a row sit the chain out (EXEC is set once, outside the ring)
                    const unsigned row_addr = cl.a & 0x3ffffu;
                    const float acc = chain_rows(row_addr, row_addr + ((cl.b & 0x1fffu) * 8u - ((cl.b >> 13) & 1u) * 4u) * 4u,
                                                 ((unsigned)__builtin_amdgcn_readfirstlane((int)(cl.a >> 18)) + 3u) >> 2);
                    val[cl.b >> 16] = acc;
                }
                __builtin_amdgcn_s_setprio(0);
                FL_PSTAMP();
            }
            return;
        }
        // short rows: one lane per vertex sums both labels (products are stored label-interleaved),
        // 8 at a time with all loads issued before the first add; a lane past the end of its row
        // reads the zero block (x + 0 is exact, see chain_rows)
        const float2 *pl = reinterpret_cast<const float2 *>(smem + lay.prod[k]);
        const float2 *zero = reinterpret_cast<const float2 *>(smem + lay.zero);
        const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
        if (tid < s_lo || FL_DBG(1)) return;
        for (int v = tid - s_lo; v < V[k]; v += NT - s_lo) {
            const int t = row[v + 1];
            float a0 = 0.0f, a1 = 0.0f;
            for (int p = row[v]; p < t; p += 8) {
                float2 x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = *((p + u < t) ? pl + p + u : zero);
#pragma unroll
                for (int u = 0; u < 8; ++u) { a0 += x[u].x; a1 += x[u].y; }   // strictly left to right
            }
            reinterpret_cast<float2 *>(val)[v + 1] = make_float2(a0, a1);
        }
    };
    constexpr auto on = [](int k) { return ((KMASK >> k) & 1) != 0; };
    if (lay.prod_all) {
        if (WITH_P) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (on(k)) phase_P(k);
        }
        __syncthreads();
        FL_STAMP();
        const int s_lo = (K > 1 && on(0) && on(1) && chain_k<CH>(lay, 0)) ? 128 : 0;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (on(k)) phase_S(k, s_lo);                               // the chain kernel is kernel 0: it starts first
        __syncthreads();
        FL_STAMP();
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (!on(k)) continue;
            phase_P(k);
            __syncthreads();
            phase_S(k, 0);
            __syncthreads();
        }
    }

    // ---- d+1 Jacobi blur passes, permutohedral_cpu.h:663-679 -----------------------
#pragma unroll
    for (int j = 0; j < D1; ++j) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (!on(k)) continue;
            const float2 *src = reinterpret_cast<const float2 *>(smem + lay.val[k][j & 1]);
            float2 *dst = reinterpret_cast<float2 *>(smem + lay.val[k][(j & 1) ^ 1]);
            const unsigned *nbr = reinterpret_cast<const unsigned *>(smem + lay.nbr[k]) + j * V[k];
            // REV (k_fused, 1024 lanes, two kernels): the small lattice (kernel 0, ~120 vertices) goes to the LAST lanes -- the first
            // ones have a second vertex of the large lattice (~1160), and a pass is as long as its busiest lane (+0.6 % C2, +1 % C4;
            // with 512 lanes every lane has two or three vertices anyway and the reversal lost 5 % on C1; in k_frame the extra
            // index pushed one variant into scratch)
            int v0 = tid;
            if (REV && K > 1 && k == 0 && NT == kNT) {
                asm volatile("" : "+v"(v0));              // (recomputed per pass: hoisted, the reversed index would cost a register for the whole loop)
                v0 = NT - 1 - v0;
            }
            for (int v = v0; v < V[k]; v += NT) {
                const unsigned n = nbr[v];
                const float2 o = src[v + 1], x = src[n & 0xffffu], y = src[n >> 16];
                float2 r;
                r.x = o.x + 0.5f * (x.x + y.x);
                r.y = o.y + 0.5f * (x.y + y.y);
                dst[v + 1] = r;
            }
        }
        __syncthreads();
    }
    FL_STAMP();
}

// slice of kernel k at point slot s: sum_j (bary_j * alpha) * val[vertex_j], permutohedral_cpu.h:684-694
template <int PPT, int K>
__device__ __forceinline__ float2 slice_point(const unsigned char *smem, const FusedLayout &lay, const PointRegs<PPT, K> &pr,
                                              int s, int k, float alpha)
{
    const float2 *val = reinterpret_cast<const float2 *>(smem + lay.val[k][kD1 & 1]);
    const float2 x0 = val[pr.ix[s][k][0] & 0xffffu], x1 = val[pr.ix[s][k][0] >> 16], x2 = val[pr.ix[s][k][1] & 0xffffu];
    const float w0 = pr.bary[s][k][0] * alpha, w1 = pr.bary[s][k][1] * alpha, w2 = pr.bary[s][k][2] * alpha;   // permutohedral_cpu.h:689
    // permutohedral_cpu.h:686-693 starts its sum at 0.  0 + x == x bit for bit unless x is -0, and a -0 here could only turn
    // the sign of a sum that is zero anyway -- which neither the energies nor the softmax downstream can tell apart
    float t0 = w0 * x0.x, t1 = w0 * x0.y;
    t0 += w1 * x1.x; t1 += w1 * x1.y;
    t0 += w2 * x2.x; t1 += w2 * x2.y;
    return make_float2(t0, t1);
}

// The packed ids / slots and the weights are loop invariants.  Left alone, the compiler hoists every LDS
// address and every bary*alpha out of the loop, which costs ~50 more live registers than the 128 a
// 1024-lane workgroup has, and spills.  Make them opaque per trip.
template <int PPT, int K>
__device__ __forceinline__ void opaque(PointRegs<PPT, K> &pr)
{
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            asm volatile("" : "+v"(pr.ix[s][k][0]), "+v"(pr.ix[s][k][1]), "+v"(pr.ix[s][k][2]));
            asm volatile("" : "+v"(pr.bary[s][k][0]), "+v"(pr.bary[s][k][1]), "+v"(pr.bary[s][k][2]));
        }
    }
}

// startInference: Q = softmax(-unary), densecrf_base.h:78-80
template <int PPT, int K, int NT = kNT>
__device__ __forceinline__ void start_inference(PointRegs<PPT, K> &pr, int N, int tid)
{
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        pr.q[s] = make_float2(0.f, 0.f);
        if (tid + s * NT < N) {
            pr.q[s] = softmax2(-pr.un[s].x, -pr.un[s].y, make_float2(0.f, 0.f), 1.0f);   // scale = -1 is an exact negation
        }
    }
}

// n_iter x stepInference (densecrf_base.h:82-91): splat, blur, then slice + apply + softmax per point.
// first_p (bit k): with own product buffers the loop starts by writing every kernel's products of Q0; a caller that has
// already put kernel k's in place (the frame kernel's two-workgroup form, while it waits for the other lattice) clears bit k.
// No barrier is needed behind X: the next P only writes the product buffers, whose readers finished
// two barriers ago.
template <int PPT, int K, int CH, int NT = kNT, bool REV = false>
__device__ __forceinline__ void mean_field(unsigned char *smem, const FusedLayout &lay, const int (&V)[K], int N, int tid,
                                           PointRegs<PPT, K> &pr, const ChainLane &cl, const float (&alpha)[K], int n_iter,
                                           float relax, Instr &ins, int first_p = -1)
{
    // slice + apply + softmax of point slot s (X)
    auto point_update = [&](int s) {
        float nx[2] = {-pr.un[s].x, -pr.un[s].y};                 // stepInit, densecrf3d.h:154-158
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float2 t = slice_point(smem, lay, pr, s, k, alpha[k]);
            nx[0] += pr.wn[s][k] * t.x;                           // pairwise3d.h:77
            nx[1] += pr.wn[s][k] * t.y;
        }
        pr.q[s] = softmax2(nx[0], nx[1], pr.q[s], relax);        // densecrf3d.h:70-98 with L = 2: one exp, not two
    };
    if (kFuseXP && PPT <= 2 && lay.prod_all) {            // (3-4 points per lane: the fused form costs registers the loop does not have)
        // Every kernel owns its product buffer: a point's next products go out right behind its softmax, so one
        // point's LDS stores drain while the next point's slice and softmax occupy the VALU (X is VALU-bound, P is
        // bound by the LDS store path; back to back they cost the sum).  The barrier that used to follow P now
        // opens splat_blur; the pads behind the chain rows are written once (nothing else ever writes there).
        if (n_iter > 0) {                                  // (first_p: the kernels whose first products are not in place yet)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (!((first_p >> k) & 1)) continue;
#pragma unroll
                for (int s = 0; s < PPT; ++s)
                    if (tid + s * NT < N) point_products<PPT, K, CH>(smem, lay, pr, s, k);
                if (chain_k<CH>(lay, k)) chain_pads(smem, cl);
            }
        }
        for (int it = 0; it < n_iter; ++it) {
            opaque(pr);
            splat_blur<PPT, K, CH, false, NT, (1 << K) - 1, REV>(smem, lay, V, N, tid, pr, cl, ins);
            const bool more = it + 1 < n_iter;
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                if (tid + s * NT < N) {
                    point_update(s);
                    if (more) {
#pragma unroll
                        for (int k = 0; k < K; ++k) point_products<PPT, K, CH>(smem, lay, pr, s, k);
                    }
                }
            }
            FL_STAMP();
        }
        return;
    }
    for (int it = 0; it < n_iter; ++it) {
        opaque(pr);
        splat_blur<PPT, K, CH, true, NT, (1 << K) - 1, REV>(smem, lay, V, N, tid, pr, cl, ins);
#pragma unroll
        for (int s = 0; s < PPT; ++s)
            if (tid + s * NT < N) point_update(s);
        FL_STAMP();
    }
}

// The label bits beyond a frame's points are 0 (include/lccrf.h: lccrf_batch_device_label_bits): the words from `first_word` on are
// cleared, so that a batch handle used again with smaller frames does not hand the gather the last batch's bits.
template <int NT>
__device__ __forceinline__ void clear_label_bits(const CrfDev &c, int f, int first_word, int tid)
{
    if (c.map_bits)
        for (int w = first_word + tid; w < c.bits_stride; w += NT) c.map_bits[(size_t)f * c.bits_stride + w] = 0ull;
}

// Q and the MAP labels (densecrf3d.h:136-151: first maximum wins, ties -> label 0) of this lane's points.
// NTS: non-temporal stores (the results leave the XCD's L2 working set alone: fused_lean.h, LCCRF_LEAN_NT)
template <int PPT, int K, int NT = kNT, bool NTS = false>
__device__ __forceinline__ void store_results(const CrfDev &c, int f, int N, int tid, const PointRegs<PPT, K> &pr, int with_map)
{
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int i = tid + s * NT;
        if (i < N) {
            if (NTS) {
                typedef float nts_f2 __attribute__((ext_vector_type(2)));
                nts_f2 q;
                q.x = pr.q[s].x;
                q.y = pr.q[s].y;
                __builtin_nontemporal_store(q, reinterpret_cast<nts_f2 *>(c.Q) + (size_t)f * c.maxN + i);
                if (with_map) __builtin_nontemporal_store((int16_t)((pr.q[s].x < pr.q[s].y) ? 1 : 0), c.map + (size_t)f * c.maxN + i);
            } else {
                reinterpret_cast<float2 *>(c.Q)[(size_t)f * c.maxN + i] = pr.q[s];
                if (with_map) c.map[(size_t)f * c.maxN + i] = (pr.q[s].x < pr.q[s].y) ? 1 : 0;   // densecrf3d.h:145
            }
        }
        if (with_map && c.map_bits && (i & ~63) < N) {                   // the same labels, one bit each (label gather payload)
            const unsigned long long m = __ballot(i < N && pr.q[s].x < pr.q[s].y);
            if ((tid & 63) == 0) c.map_bits[(size_t)f * c.bits_stride + (i >> 6)] = m;
        }
    }
    if (with_map) clear_label_bits<NT>(c, f, (N + 63) >> 6, tid);
}

}  // namespace fl
}  // namespace lccrf
