// fused_engine.hip -- SLAM-size inference as ONE kernel launch: one 1024-lane workgroup per
// frame runs startInference, every mean-field iteration and buildMap without leaving the CU.
//
// Why: at SLAM sizes (N ~ 2000 keypoints, V ~ 1.2k lattice vertices) one iteration moves
// < 0.5 MB; a launch-per-phase design is bound by launch gaps, not by memory (SURVEY.md
// section 7).  Here the mean-field state lives on chip:
//     registers  per point (lane t owns points t, t+1024, ...): unary, Q, and per kernel the
//                d+1 vertex ids, product slots, barycentric weights and w*norm -- loaded once
//     LDS        lattice side only: the splat products of every kernel, both ping-pong value
//                arrays, the blur neighbour table (u16 pairs) and the row pointers
//     HBM        touched by the prologue (~150 KB of records per frame) and the final store
//
// Bit-exactness: the reference splats sequentially over points (permutohedral_cpu.h:653-661),
// so a vertex's value is a left-to-right fp32 sum in ascending point order.  Phase P writes
// every product bary*Q into its row at the place the build recorded for it (exact, order-free);
// phase S then adds each row strictly left to right -- chain_rows for the long rows of the
// appearance kernel.  Nothing is re-associated, nothing is fused (-ffp-contract=off).
//
// Specialised for L = 2 labels (the SLAM configuration, src/Tracking.cc:1919) and 2-D kernels;
// anything else runs on the streaming engine with identical results.  DESIGN.md section 4.2.
#include "engine.h"
#include "device_math.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace lccrf {

namespace {

constexpr int kNT = 1024;                 // lanes per workgroup (16 wavefronts)
constexpr int kMaxFusedK = 2;
constexpr int kChainGap = 14;               // product slots reserved per chain row beyond its products (see pst)
constexpr int kChainTop = 16;               // rows of the first chain wavefront pair (see the prologue)
constexpr size_t kLdsLimit = 160 * 1024;  // MI355X: 160 KiB LDS per CU, one workgroup may own it all

struct FusedLayout {                      // byte offsets into dynamic LDS
    int zero;                             // 64 bytes of zeros (what a finished chain lane keeps reading)
    int prod[kMaxFusedK];                 // float  [2][Ecap_k] (label-major), aliased when !prod_all
    int val[kMaxFusedK][2];               // float2 [V_k+1]     slot 0 = absent neighbour = 0
    int nbr[kMaxFusedK];                  // u32    [D1][V_k]   (n1+1) | (n2+1)<<16
    int row[kMaxFusedK];                  // u16    [V_k+1]
    int perm[kMaxFusedK];                 // u16    [V_k]       vertex -> index of its lattice values (KernelDev::vperm)
    int Ecap[kMaxFusedK];                 // floats per label plane of prod
    int Vcap[kMaxFusedK];
    int prod_all;                         // 1: every kernel has its own product buffer
    int chain0;                           // 1: kernel 0 has long splat rows, S runs chain_rows on them
    int total;
};

struct FusedArgs {
    KernelDev kd[kMaxFusedK];
    FusedLayout lay;
    int n_iter, with_map;
    float relax;
    long long *timing;                    // debug: shader-clock stamps of one workgroup (LCCRF_FUSED_TIMING=<block index + 1>)
    int timing_block;
    int late;                             // 1: the kernel sizes its own LDS from its frame's V / longest row (lay is ignored)
    int *status;                          // late-bound launches: set to 1 (pinned host memory) when the frame does not fit
    const int16_t *label;                 // non-null: unary energies come from these labels and `tbl` (setUnaryEnergyFromLabel,
    float tbl[5];                         //   densecrf3d.h:100-130, L = 2) -- computed here and stored to c.unary, no separate launch
    int dbg;                              // LCCRF_FUSED_DBG: 1 skip short-row S, 2 skip chain S (timing only, wrong results); 4 poison LDS
};

// Instrumentation (shader-clock stamps, phase-skipping experiments) exists only in builds made with
// `make INSTRUMENT=1` (-DLCCRF_INSTRUMENT=1): the release library reads no debug switch that could change a
// result.  prologue breakdown: LCCRF_FUSED_DBG=8
#ifndef LCCRF_INSTRUMENT
#define LCCRF_INSTRUMENT 0
#endif
constexpr bool kInstr = LCCRF_INSTRUMENT != 0;
#define DBG(bit) (kInstr && (a.dbg & (bit)))
#define PSTAMP()                      \
    do {                              \
        if (DBG(8)) STAMP();          \
    } while (0)
#define STAMP()                                                        \
    do {                                                               \
        if (kInstr && a.timing && blockIdx.x == a.timing_block && tid == 0) a.timing[n_stamp++] = clock64(); \
    } while (0)

constexpr int kChainMinRow = 64;          // kernel 0 runs chain_rows when its longest splat row has at least this many products ...
constexpr int kChainMaxV = kChainTop + 7 * 64;           // ... and it has at most this many vertices (one lane per (vertex,label) row)

// LDS plan of one workgroup for frames of at most NA points whose K lattices (all 2-D) have at most
// V[k] vertices and kernel 0's longest row has `row0` products.  Shared by the host (batch API: sizes
// maximised over the frames) and the kernel itself (late-bound launches: each frame sizes its own).
__host__ __device__ inline bool layout_core(int NA, int K, const int *V, int row0, FusedLayout *lay)
{
    constexpr int D1 = 3;
    if (NA < 1 || NA > 4 * kNT || K < 1 || K > kMaxFusedK) return false;
    for (int k = 0; k < K; ++k)
        if (V[k] >= 65535) return false;
    const int chain0 = row0 >= kChainMinRow && V[0] <= kChainMaxV && NA * D1 + kChainGap * V[0] + 64 < 65535;
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
            const int E = NA * D1;
            // chain rows carry kChainGap extra slots each; every plane is a multiple of 64 floats
            L.Ecap[k] = ((k == 0 && chain0 ? E + kChainGap * V[k] + 16 : E) + 63) & ~63;
            L.Vcap[k] = V[k];
            L.val[k][0] = take((size_t)(V[k] + 1) * sizeof(float2));
            L.val[k][1] = take((size_t)(V[k] + 1) * sizeof(float2));
            L.nbr[k] = take((size_t)D1 * V[k] * sizeof(unsigned));
            L.row[k] = take((size_t)(V[k] + 2) * sizeof(unsigned short));
            L.perm[k] = take((size_t)(V[k] + 2) * sizeof(unsigned short));
            const size_t pb = (size_t)L.Ecap[k] * 2 * sizeof(float);
            if (all) L.prod[k] = take(pb);
            else shared_prod = pb > shared_prod ? pb : shared_prod;
        }
        if (!all) {
            const int p = take(shared_prod);
            for (int k = 0; k < K; ++k) L.prod[k] = p;
        }
        L.total = (int)o;
        if (o <= kLdsLimit) {
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

__device__ __forceinline__ float chain_rows(unsigned addr, unsigned end, unsigned trips)
{
    float acc = 0.0f;
    if (trips == 0) return acc;
    const unsigned e1 = end - 32u, e2 = end - 64u, e3 = end - 96u;      // min(addr, end - k) + k == min(addr + k, end)
    unsigned sel;
    asm volatile(
        "v_min_u32_e32 %[sel], %[ad], %[e0]\n\t"
        "ds_read_b128 v[96:99], %[sel]\n\tds_read_b128 v[100:103], %[sel] offset:16\n\t"
        "v_min_u32_e32 %[sel], %[ad], %[e1]\n\t"
        "ds_read_b128 v[104:107], %[sel] offset:32\n\tds_read_b128 v[108:111], %[sel] offset:48\n\t"
        "v_min_u32_e32 %[sel], %[ad], %[e2]\n\t"
        "ds_read_b128 v[112:115], %[sel] offset:64\n\tds_read_b128 v[116:119], %[sel] offset:80\n\t"
        "1:\n\t"
        "v_min_u32_e32 %[sel], %[ad], %[e3]\n\t"
        "ds_read_b128 v[120:123], %[sel] offset:96\n\tds_read_b128 v[124:127], %[sel] offset:112\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v96, v97, v98, v99, v100, v101, v102, v103)
        "v_add_u32_e32 %[ad], 0x80, %[ad]\n\t"
        "v_min_u32_e32 %[sel], %[ad], %[e0]\n\t"
        "ds_read_b128 v[96:99], %[sel]\n\tds_read_b128 v[100:103], %[sel] offset:16\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v104, v105, v106, v107, v108, v109, v110, v111)
        "v_min_u32_e32 %[sel], %[ad], %[e1]\n\t"
        "ds_read_b128 v[104:107], %[sel] offset:32\n\tds_read_b128 v[108:111], %[sel] offset:48\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v112, v113, v114, v115, v116, v117, v118, v119)
        "v_min_u32_e32 %[sel], %[ad], %[e2]\n\t"
        "ds_read_b128 v[112:115], %[sel] offset:64\n\tds_read_b128 v[116:119], %[sel] offset:80\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v120, v121, v122, v123, v124, v125, v126, v127)
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [acc] "+v"(acc), [ad] "+v"(addr), [n] "+s"(trips), [sel] "=&v"(sel)
        : [e0] "v"(end), [e1] "v"(e1), [e2] "v"(e2), [e3] "v"(e3)
        : "scc", "memory", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",
          "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120",
          "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    return acc;
}

// One workgroup per frame.  Lane t owns points t, t+1024, ... (PPT of them) and keeps everything
// they need in registers for the whole launch: unary, Q, and per kernel the three vertex ids,
// barycentric weights, product slots and w*norm.  LDS holds only lattice-side data.
//   P  every point writes its 3 products bary*Q per kernel into its vertices' rows (slot = the
//      point's place in the row, ascending point order -- found once per launch)
//   S  row sums, strictly left to right (chain_rows for a long-row kernel 0: CH = 1)
//   B  three Jacobi blur passes            X  slice + apply + softmax, all in registers
//   CH = 0 / 1: kernel 0 short-row / chain, decided by the host;  CH = 2: decided here (late-bound layout)
template <int PPT, int K, int CH>
__global__ void __launch_bounds__(kNT) k_fused(CrfDev c, FusedArgs a)
{
    constexpr int D1 = 3;
    int n_stamp = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int N = c.n_points[f];
    STAMP();

    // ---- per-thread point state (registers) ------------------------------------------
    float2 un[PPT], q[PPT];
    unsigned offp[PPT][K][2];             // (id0+1) | (id1+1) << 16,  id2+1     (index into val, 0 = absent)
    unsigned slp[PPT][K][2];              // slot0 | slot1 << 16,  slot2         (index into prod)
    float bary[PPT][K][D1];
    float wn[PPT][K];
    int V[K];
#pragma unroll
    for (int k = 0; k < K; ++k) V[k] = a.kd[k].V[f];

    if (N <= 0) return;                   // nothing to infer (and nothing below may index an empty frame)
    FusedLayout lay = a.lay;
    if constexpr (CH == 2) {              // late-bound launch: this frame sizes its own LDS (the host has not seen V yet)
        if (!layout_core(N, K, V, a.kd[0].rowmax[f], &lay)) {
            if (tid == 0 && a.status) *a.status = 1;      // the host falls back to the streaming engine
            return;
        }
    }
    auto chain_k = [&](int k) -> bool { return CH == 2 ? (k == 0 && lay.chain0 != 0) : (((CH >> k) & 1) != 0); };
    if (DBG(4)) {                      // debugging aid: NaN-poison the LDS so that reads of unwritten LDS show up
        for (int i = tid; i < lay.total / 4; i += kNT) reinterpret_cast<unsigned *>(smem)[i] = 0x7fc00000u + (unsigned)i;
        __syncthreads();
    }

    // All global loads of the prologue are issued before anything waits on them: the lattice
    // tables first (their LDS stores come last), then the per-point records.  Indices are clamped
    // instead of branched on, so that the loads stay back to back.
    constexpr int kNbrRounds = 4, kRowRounds = 2;         // covers V <= 1365 in registers; larger lattices finish in copy loops
    unsigned g_nbr[K][kNbrRounds];
    int g_row[K][kRowRounds], g_perm[K][kRowRounds];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const KernelDev &kd = a.kd[k];
        const unsigned *gn = kd.nbr16 + (size_t)f * D1 * kd.Epad;            // already (n1+1) | (n2+1) << 16
        const int *gr = kd.rowptr + (size_t)f * (kd.Epad + 1);
#pragma unroll
        for (int r = 0; r < kNbrRounds; ++r) {            // element idx = j*V + v, j-major like the LDS copy
            const int idx = min(tid + r * kNT, D1 * V[k] - 1);
            const int j = idx >= 2 * V[k] ? 2 : (idx >= V[k] ? 1 : 0);
            g_nbr[k][r] = gn[(size_t)j * kd.Epad + (idx - j * V[k])];
        }
#pragma unroll
        for (int r = 0; r < kRowRounds; ++r) {
            g_row[k][r] = gr[min(tid + r * kNT, V[k])];
            g_perm[k][r] = kd.vperm[(size_t)f * kd.Epad + min(tid + r * kNT, V[k] - 1)];
        }
    }
    unsigned pk[PPT][K][D1];              // (vertex id + 1) | place in the row << 16
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int ic = min(tid + s * kNT, N - 1);
        if (a.label) {                                    // densecrf3d.h:116-129 with L = 2, as k_unary_from_label
            const int t = a.label[(size_t)f * c.maxN + ic];
            const bool known = t >= 0 && t < 2;
            un[s].x = !known ? a.tbl[0] : (t == 0 ? a.tbl[3] : a.tbl[1 + t]);
            un[s].y = !known ? a.tbl[0] : (t == 1 ? a.tbl[4] : a.tbl[1 + t]);
            if (tid + s * kNT < N) reinterpret_cast<float2 *>(c.unary)[(size_t)f * c.maxN + ic] = un[s];
        } else {
            un[s] = reinterpret_cast<const float2 *>(c.unary)[(size_t)f * c.maxN + ic];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const KernelDev &kd = a.kd[k];
            const size_t e0 = (size_t)f * kd.Epad + (size_t)ic * D1;
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                pk[s][k][j] = kd.pk[e0 + j];
                bary[s][k][j] = kd.bary[e0 + j];
            }
            wn[s][k] = kd.norm[(size_t)f * kd.maxN + ic];
        }
    }
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        q[s] = make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            offp[s][k][0] = (pk[s][k][0] & 0xffffu) | (pk[s][k][1] << 16);
            offp[s][k][1] = pk[s][k][2] & 0xffffu;
            slp[s][k][0] = slp[s][k][1] = 0;
            wn[s][k] = a.kd[k].w * wn[s][k];                              // pairwise3d.h:77 (w_*norm_[i])
        }
    }
    PSTAMP();

    // ---- per-frame lattice tables into LDS --------------------------------------------
    int *hist = reinterpret_cast<int *>(smem + lay.prod[0]);            // chain ranking scratch: [64] counts, [64] bases
#pragma unroll
    for (int k = 0; k < K; ++k) {
        unsigned *nbr = reinterpret_cast<unsigned *>(smem + lay.nbr[k]);
        unsigned short *row = reinterpret_cast<unsigned short *>(smem + lay.row[k]);
#pragma unroll
        for (int r = 0; r < kNbrRounds; ++r) {
            const int idx = tid + r * kNT;
            if (idx < D1 * V[k]) nbr[idx] = g_nbr[k][r];
        }
        unsigned short *perm = reinterpret_cast<unsigned short *>(smem + lay.perm[k]);
#pragma unroll
        for (int r = 0; r < kRowRounds; ++r) {
            if (tid + r * kNT <= V[k]) row[tid + r * kNT] = (unsigned short)g_row[k][r];
            if (tid + r * kNT < V[k]) perm[tid + r * kNT] = (unsigned short)g_perm[k][r];
        }
        // lattices with more vertices than the register rounds cover (sparse frames): plain copy loops
        const KernelDev &kd = a.kd[k];
        const unsigned *gn = kd.nbr16 + (size_t)f * D1 * kd.Epad;
        for (int idx = tid + kNbrRounds * kNT; idx < D1 * V[k]; idx += kNT) {
            const int j = idx >= 2 * V[k] ? 2 : (idx >= V[k] ? 1 : 0);
            nbr[idx] = gn[(size_t)j * kd.Epad + (idx - j * V[k])];
        }
        const int *gr = kd.rowptr + (size_t)f * (kd.Epad + 1);
        for (int v = tid + kRowRounds * kNT; v <= V[k]; v += kNT) row[v] = (unsigned short)gr[v];
        for (int v = tid + kRowRounds * kNT; v < V[k]; v += kNT) perm[v] = (unsigned short)kd.vperm[(size_t)f * kd.Epad + v];
    }
    if (tid < 16) reinterpret_cast<float *>(smem + lay.zero)[tid] = 0.0f;
    if (CH && tid < 128) hist[tid] = 0;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            reinterpret_cast<float2 *>(smem + lay.val[k][0])[0] = make_float2(0.f, 0.f);
            reinterpret_cast<float2 *>(smem + lay.val[k][1])[0] = make_float2(0.f, 0.f);
        }
    }
#pragma unroll
    for (int s = 0; s < PPT; ++s) {                       // startInference: Q = softmax(-unary), densecrf_base.h:78-80
        if (tid + s * kNT < N) {
            float in[2] = {un[s].x, un[s].y}, out[2] = {0.f, 0.f};
            exp_and_normalize_reg<2>(in, out, -1.0f, 1.0f);
            q[s] = make_float2(out[0], out[1]);
        }
    }
    __syncthreads();
    PSTAMP();

    // ---- where does each of my products go?  (once per launch) -----------------------
    // The build recorded the place of entry (i, j) in its vertex's row (ascending point order,
    // the reference's splat order).  A plain kernel stores its products at exactly that CSR
    // position.  The chain kernel re-places row v at pst(v) = ceil4(row[v] + 14v): starts are
    // multiples of 4 and pst(v+1) - pst(v) is a multiple of 4 that is >= the row's length + 11,
    // i.e. there is room for the row padded to 4 products plus the eight zeros chain_rows wants
    // behind it -- without any scan.
    auto pst = [](int r0, int v) { return (r0 + kChainGap * v + 3) & ~3; };
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            if (tid + s * kNT < N) {
                unsigned sl[D1];
#pragma unroll
                for (int j = 0; j < D1; ++j) {
                    sl[j] = pk[s][k][j] >> 16;
                    if (chain_k(k)) {
                        const int v = (int)(pk[s][k][j] & 0xffffu) - 1;
                        const int r0 = row[v];
                        sl[j] = (unsigned)(pst(r0, v) + ((int)(pk[s][k][j] >> 16) - r0));
                    }
                }
                slp[s][k][0] = sl[0] | (sl[1] << 16);
                slp[s][k][1] = sl[2];
            }
        }
    }
    // Chain lanes: one lane per (vertex,label) row, long rows first so that whole wavefronts
    // retire early.  Rank = counting sort on the row's 16-product block count (64 buckets,
    // longest first; order inside a bucket is whatever the LDS atomics give -- it only decides
    // which lane sums which row, never the order inside a row).
    unsigned ch_a = 0, ch_b = 0;          // row address | wavefront max units << 18 ;
                                          // 8-product units | (padded length is 8n+4) << 13 | pad slots << 14 | output index << 16
    if (CH != 0 && chain_k(0)) {
        constexpr int k = 0;
        const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
        unsigned short *srt = reinterpret_cast<unsigned short *>(smem + lay.prod[k]) + 256;   // [V] vertex of rank r
        int key = 0, len = 0;
        if (tid < V[k]) {
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
        if (tid < V[k]) srt[atomicAdd(&hist[64 + key], 1)] = (unsigned short)tid;
        __syncthreads();
        // Wavefront pair p = (2p, 2p+1) owns labels 0 and 1 of a rank range, so a wavefront reads one
        // label plane only.  Pair 0 takes just the kChainTop longest rows, four lanes in each of the
        // four 16-lane groups a ds_read_b128 is served in: the cost of a ring unit grows with the
        // bank conflicts among the ACTIVE lanes (~75 cycles with 16 rows, ~110 with 64), and it is
        // the longest rows' wavefront that everybody waits for.  Pair p >= 1: 64 rows each.
        const int l = (tid >> 6) & 1, pr = tid >> 7, ln = tid & 63;
        const int r = pr == 0 ? (((ln & 0x18) == 0) ? ((ln & 7) | ((ln >> 5) << 3)) : V[k])
                              : kChainTop + ((pr - 1) << 6) + ln;
        unsigned nblk = 0, addr = 0;
        if (r < V[k]) {
            const int v = srt[r];
            const int r0 = row[v], rl = (int)row[v + 1] - r0, len4 = (rl + 3) & ~3;
            addr = (unsigned)(lay.prod[k] + 4 * (l * lay.Ecap[k] + pst(r0, v)));               // < 2^18
            nblk = (unsigned)((len4 + 7) >> 3);                                                    // 8-product units, < 2^13
            ch_b = nblk | ((unsigned)((len4 >> 2) & 1) << 13) | ((unsigned)(len4 - rl) << 14) |
                   ((unsigned)(((int)reinterpret_cast<const unsigned short *>(smem + lay.perm[k])[v] + 1) * 2 + l) << 16);
        }
        unsigned m = nblk;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        ch_a = addr | (m << 18);
        __syncthreads();                  // the ranking scratch becomes the product buffer
    }
    STAMP();

    for (int it = 0; it < a.n_iter; ++it) {
        // The packed ids / slots and the weights are loop invariants.  Left alone, the compiler
        // hoists every LDS address and every bary*alpha out of the loop, which costs ~50 more live
        // registers than the 128 a 1024-lane workgroup has, and spills.  Make them opaque per trip.
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                asm volatile("" : "+v"(slp[s][k][0]), "+v"(slp[s][k][1]), "+v"(offp[s][k][0]), "+v"(offp[s][k][1]));
                asm volatile("" : "+v"(bary[s][k][0]), "+v"(bary[s][k][1]), "+v"(bary[s][k][2]));
            }
        }
        // ---- splat = products (P) + ordered row sums (S) ------------------------------
        auto phase_P = [&](int k) {
            float *p0 = reinterpret_cast<float *>(smem + lay.prod[k]);
            float *p1 = p0 + lay.Ecap[k];
            float2 *p2 = reinterpret_cast<float2 *>(p0);
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const int i = tid + s * kNT;
                if (i < N) {
                    const unsigned s0 = slp[s][k][0] & 0xffffu, s1 = slp[s][k][0] >> 16, s2 = slp[s][k][1];
                    if (chain_k(k)) {                                  // chain kernel: one plane per label
                        p0[s0] = bary[s][k][0] * q[s].x;
                        p1[s0] = bary[s][k][0] * q[s].y;
                        p0[s1] = bary[s][k][1] * q[s].x;
                        p1[s1] = bary[s][k][1] * q[s].y;
                        p0[s2] = bary[s][k][2] * q[s].x;
                        p1[s2] = bary[s][k][2] * q[s].y;
                    } else {                                              // short rows: labels interleaved
                        p2[s0] = make_float2(bary[s][k][0] * q[s].x, bary[s][k][0] * q[s].y);
                        p2[s1] = make_float2(bary[s][k][1] * q[s].x, bary[s][k][1] * q[s].y);
                        p2[s2] = make_float2(bary[s][k][2] * q[s].x, bary[s][k][2] * q[s].y);
                    }
                }
            }
            if (chain_k(k)) {          // behind the row: +0 up to a multiple of 4, then eight +0 (the buffer may
                                          // have held another kernel's products)
                if (ch_b >> 16) {
                    float *e = reinterpret_cast<float *>(smem + (ch_a & 0x3ffffu)) + ((ch_b & 0x1fffu) * 8u - ((ch_b >> 13) & 1u) * 4u);
                    const unsigned npad = (ch_b >> 14) & 3u;
                    for (unsigned z = 1; z <= npad; ++z) e[-(int)z] = 0.0f;
                    reinterpret_cast<float4 *>(e)[0] = make_float4(0.f, 0.f, 0.f, 0.f);
                    reinterpret_cast<float4 *>(e)[1] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        // lanes [s_lo, kNT) share the short-row kernels; the wavefront that owns the chain
        // kernel's longest rows keeps out of them
        auto phase_S = [&](int k, int s_lo) {
            float *val = reinterpret_cast<float *>(smem + lay.val[k][0]);
            if (chain_k(k)) {
                const int npairs = 1 + ((max(V[k] - kChainTop, 0) + 63) >> 6);
                if ((tid >> 7) < npairs && !DBG(2)) {                                // whole wavefronts
                    PSTAMP();
                    const unsigned row_addr = ch_a & 0x3ffffu;
                    const float acc = chain_rows(row_addr, row_addr + ((ch_b & 0x1fffu) * 8u - ((ch_b >> 13) & 1u) * 4u) * 4u,
                                                 ((unsigned)__builtin_amdgcn_readfirstlane((int)(ch_a >> 18)) + 3u) >> 2);
                    if ((ch_b >> 16) != 0) val[ch_b >> 16] = acc;
                    PSTAMP();
                }
                return;
            }
            // short rows: one lane per vertex sums both labels (products are stored label-interleaved),
            // 8 at a time with all loads issued before the first add; a lane past the end of its row
            // reads the zero block (x + 0 is exact, see chain_rows)
            const float2 *pl = reinterpret_cast<const float2 *>(smem + lay.prod[k]);
            const float2 *zero = reinterpret_cast<const float2 *>(smem + lay.zero);
            const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
            const unsigned short *perm = reinterpret_cast<const unsigned short *>(smem + lay.perm[k]);
            if (tid < s_lo || DBG(1)) return;
            for (int v = tid - s_lo; v < V[k]; v += kNT - s_lo) {
                const int t = row[v + 1];
                float a0 = 0.0f, a1 = 0.0f;
                for (int p = row[v]; p < t; p += 8) {
                    float2 x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = *((p + u < t) ? pl + p + u : zero);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { a0 += x[u].x; a1 += x[u].y; }   // strictly left to right
                }
                reinterpret_cast<float2 *>(val)[(int)perm[v] + 1] = make_float2(a0, a1);
            }
        };
        if (lay.prod_all) {
#pragma unroll
            for (int k = 0; k < K; ++k) phase_P(k);
            __syncthreads();
            STAMP();
            const int s_lo = (K > 1 && chain_k(0)) ? 128 : 0;
#pragma unroll
            for (int k = 0; k < K; ++k) phase_S(k, s_lo);                 // the chain kernel is kernel 0: it starts first
            __syncthreads();
            STAMP();
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
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
                const float2 *src = reinterpret_cast<const float2 *>(smem + lay.val[k][j & 1]);
                float2 *dst = reinterpret_cast<float2 *>(smem + lay.val[k][(j & 1) ^ 1]);
                const unsigned *nbr = reinterpret_cast<const unsigned *>(smem + lay.nbr[k]) + j * V[k];
                for (int v = tid; v < V[k]; v += kNT) {
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
        STAMP();

        // ---- slice + apply + softmax per point (no barrier needed before the next P: it only
        //      writes the product buffers, whose readers finished two barriers ago) ---------
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * kNT;
            if (i < N) {
                float nx[2] = {-un[s].x, -un[s].y};                       // stepInit, densecrf3d.h:154-158
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float2 *val = reinterpret_cast<const float2 *>(smem + lay.val[k][D1 & 1]);
                    const float alpha = a.kd[k].alpha;
                    const float2 x0 = val[offp[s][k][0] & 0xffffu], x1 = val[offp[s][k][0] >> 16], x2 = val[offp[s][k][1]];
                    const float w0 = bary[s][k][0] * alpha, w1 = bary[s][k][1] * alpha, w2 = bary[s][k][2] * alpha;   // permutohedral_cpu.h:689
                    float t0 = 0.0f, t1 = 0.0f;
                    t0 += w0 * x0.x; t1 += w0 * x0.y;
                    t0 += w1 * x1.x; t1 += w1 * x1.y;
                    t0 += w2 * x2.x; t1 += w2 * x2.y;
                    nx[0] += wn[s][k] * t0;                               // pairwise3d.h:77
                    nx[1] += wn[s][k] * t1;
                }
                float out[2] = {q[s].x, q[s].y};
                exp_and_normalize_reg<2>(nx, out, 1.0f, a.relax);
                q[s] = make_float2(out[0], out[1]);
            }
        }
        STAMP();
    }

    // ---- results ------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int i = tid + s * kNT;
        if (i < N) {
            reinterpret_cast<float2 *>(c.Q)[(size_t)f * c.maxN + i] = q[s];
            if (a.with_map) c.map[(size_t)f * c.maxN + i] = (q[s].x < q[s].y) ? 1 : 0;   // densecrf3d.h:145
        }
        if (a.with_map && c.map_bits && (i & ~63) < N) {                   // the same labels, one bit each (label gather payload)
            const unsigned long long m = __ballot(i < N && q[s].x < q[s].y);
            if ((tid & 63) == 0) c.map_bits[(size_t)f * c.bits_stride + (i >> 6)] = m;
        }
    }
    STAMP();
    if (kInstr && a.timing && blockIdx.x == a.timing_block && tid == 0) a.timing[63] = n_stamp;
}

bool make_layout(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, FusedLayout *lay)
{
    if (c.L != 2 || c.K < 1 || c.K > kMaxFusedK) return false;
    const int NA = c.activeN > 0 ? c.activeN : c.maxN;   // size LDS and the points-per-lane variant by the frames' real size
    for (int k = 0; k < c.K; ++k)
        if (kds[k].d != 2 || kds[k].Epad >= 65535) return false;       // u16 row pointers / neighbour ids / slots
    return layout_core(NA, c.K, maxV, maxRow ? maxRow[0] : 0, lay);
}

template <int PPT, int K, int CH>
void launch_fused(const CrfDev &c, const FusedArgs &a, hipStream_t s)
{
    auto fn = k_fused<PPT, K, CH>;
    // per (function, device); cheap enough to repeat and safe with several devices in one process
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kLdsLimit);
    fn<<<dim3(c.F), dim3(kNT), a.late ? (int)kLdsLimit : a.lay.total, s>>>(c, a);
}

template <int PPT>
void launch_fused_ppt(const CrfDev &c, const FusedArgs &a, hipStream_t s)
{
    if (a.late) {
        if (c.K == 1) launch_fused<PPT, 1, 2>(c, a, s);
        else launch_fused<PPT, 2, 2>(c, a, s);
        return;
    }
    if (c.K == 1) {
        if (a.lay.chain0) launch_fused<PPT, 1, 1>(c, a, s);
        else launch_fused<PPT, 1, 0>(c, a, s);
    } else {
        if (a.lay.chain0) launch_fused<PPT, 2, 1>(c, a, s);
        else launch_fused<PPT, 2, 0>(c, a, s);
    }
}

}  // namespace

bool fused_supported(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, size_t *lds_bytes)
{
    FusedLayout lay;
    const bool ok = make_layout(c, kds, maxV, maxRow, &lay);
    if (lds_bytes) *lds_bytes = ok ? (size_t)lay.total : 0;
    return ok;
}

void launch_inference_fused(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, int n_iter,
                            int with_map, float relax, hipStream_t s)
{
    FusedArgs a{};
    if (!make_layout(c, kds, maxV, maxRow, &a.lay)) return;
    static const bool no_chain = getenv("LCCRF_NO_CHAIN") != nullptr;     // debugging aid: compiler-scheduled S phase
    if (no_chain) a.lay.chain0 = 0;                                        // (the padded plane size is harmless)
    for (int k = 0; k < c.K; ++k) a.kd[k] = kds[k];
    a.n_iter = n_iter;
    a.with_map = with_map;
    a.relax = relax;
    static long long *timing_buf = nullptr;
    static const bool want_timing = kInstr && getenv("LCCRF_FUSED_TIMING") != nullptr;
    if (want_timing && !timing_buf) (void)hipMalloc(&timing_buf, 64 * sizeof(long long));
    a.timing = want_timing ? timing_buf : nullptr;
    a.timing_block = want_timing ? std::max(atoi(getenv("LCCRF_FUSED_TIMING")) - 1, 0) : 0;
    if (a.timing_block >= c.F) a.timing_block = 0;
    static const int dbg = (kInstr && getenv("LCCRF_FUSED_DBG")) ? atoi(getenv("LCCRF_FUSED_DBG")) : 0;
    a.dbg = dbg;
    const int ppt = ((c.activeN > 0 ? c.activeN : c.maxN) + kNT - 1) / kNT;
    switch (ppt) {
    case 1: launch_fused_ppt<1>(c, a, s); break;
    case 2: launch_fused_ppt<2>(c, a, s); break;
    case 3: launch_fused_ppt<3>(c, a, s); break;
    case 4: launch_fused_ppt<4>(c, a, s); break;
    default: break;
    }
    if (a.timing) {                       // debug only: synchronous read-back of workgroup 0's phase stamps
        long long h[64];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, a.timing, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[lccrf fused timing] %lld stamps, deltas (shader clocks):", h[63]);
        for (int i = 1; i < h[63] && i < 63; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
        fprintf(stderr, "\n");
    }
}

// Late-bound launch: no host knowledge of the lattice sizes is needed, so it can be queued right
// behind the build.  `status` (pinned host memory, zeroed by the caller) reads 1 afterwards if
// the frame did not fit one workgroup; the caller then runs the streaming engine.
bool fused_late_supported(const CrfDev &c, const KernelDev *kds)
{
    const int NA = c.activeN > 0 ? c.activeN : c.maxN;
    if (c.L != 2 || c.K < 1 || c.K > kMaxFusedK || NA < 1 || NA > 4 * kNT) return false;
    for (int k = 0; k < c.K; ++k)
        if (kds[k].d != 2 || kds[k].Epad >= 65535) return false;
    return true;
}

void launch_inference_fused_late(const CrfDev &c, const KernelDev *kds, int n_iter, int with_map, float relax,
                                 int *status, const int16_t *label, const float *tbl5, hipStream_t s)
{
    FusedArgs a{};
    a.label = label;
    if (label)
        for (int i = 0; i < 5; ++i) a.tbl[i] = tbl5[i];
    for (int k = 0; k < c.K; ++k) a.kd[k] = kds[k];
    a.n_iter = n_iter;
    a.with_map = with_map;
    a.relax = relax;
    a.late = 1;
    a.status = status;
    const int ppt = ((c.activeN > 0 ? c.activeN : c.maxN) + kNT - 1) / kNT;
    switch (ppt) {
    case 1: launch_fused_ppt<1>(c, a, s); break;
    case 2: launch_fused_ppt<2>(c, a, s); break;
    case 3: launch_fused_ppt<3>(c, a, s); break;
    case 4: launch_fused_ppt<4>(c, a, s); break;
    default: break;
    }
}

}  // namespace lccrf
