// frame_lean.hip -- one frame's WHOLE CRF as one kernel launch (frame_engine.hip) on HALF a CU: 512 lanes, 1-4 points per lane,
// 80 KB of LDS per frame, so that two frames share a CU in every phase -- one frame's scans, barriers and ordered row sums
// (latency) run under the other frame's point phases.  Per frame
//     both PottsPotential3D constructors (lattice + norm)  +  DenseCRF3D::inference(n, true)      (src/Tracking.cc:1920-1929)
// as in k_frame; what differs:
//   build     no hash table.  A 2-D lattice vertex is a CELL of a grid over the frame's key range (phase A below): the cells that
//             hold a vertex are marked, a scan over the cells hands out dense ids in cell order, an entry finds its vertex by its
//             cell's index and a vertex its blur neighbours at fixed offsets (permutohedral_cpu.h:66-167,371-377,408-421 without
//             probing, claiming or key compares).  The id map (2 bytes per cell, ~20 KB for a 640 x 480 image's smoothness
//             kernel) dies with phase D; the row bitmap / the entry lists that rank every entry in its vertex's row (quirk Q6:
//             ascending point order; frame_engine.hip phases E-G) are only born there -- the scratch is laid out by LIFETIME.
//             Each kernel's persistent tables (row starts; the appearance kernel's neighbour table) are built straight into their
//             places in the loop's plan (fused_lean.h: lean_tables);
//   records   what fused_lean.h's loop re-reads every iteration at 3-4 points per lane -- barycentric weights, norms, unary
//             energies -- and the smoothness lattice's neighbour table live in ONE block per frame (kLeanRec*, the engine's
//             `lean_rec` area) behind one buffer resource; they come back through L2;
//   norm      one pass of the loop's own phases with Q = 1 (mean_field_lean<.., NORM>), pairwise3d.h:20-28;
//   loop      mean_field_lean, unchanged.
// Two 2-D kernels, L = 2, up to 2048 points, batches of >= 256 frames (C1 / C2 / C3).  Anything else, and any frame that does not
// fit -- key range too wide for the id map, an appearance lattice too large for the chain lanes, coordinates near the int16 wrap --
// flags itself and runs on the other paths with identical results.  The vertex numbering is this kernel's own (cell order): the
// mean-field result does not depend on it (frame_engine.hip), V is reported and tested against the reference's M_.
#include "frame_build.h"
#include "fused_lean.h"

#include <algorithm>
#include <type_traits>

namespace lccrf {
namespace fb {

namespace {

struct LeanHdr {                          // the build's scan scratch and flags: the last bytes of the launch's LDS (dead before the loop)
    int wave_sum[16];
    int fail;
    int box[2][4];                        // per kernel: min u, min -u, min v, min -v over the frame's vertices (see phase A)
};
constexpr int kLeanHdrBytes = 128;
static_assert(sizeof(LeanHdr) <= kLeanHdrBytes, "header");
constexpr int kGridMaxCells = 32768;      // cells of the id map (u16 ids; what fits beside the other scratch is checked per frame)

// minimum over the 64 lanes of a wavefront, valid in lane 63 (DPP: inside each row of 16 lanes, then across the rows as wave_incl_scan)
__device__ __forceinline__ int wave_min(int x)
{
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x112, 0xf, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x118, 0xf, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false));
    return x;
}

// A kernel argument read where it is first needed, not at the kernel's entry: the arguments are ~70 scalar registers' worth, the compiler
// loads them up front in wide blocks and keeps what is used late -- weights, iteration count, output pointers -- alive through the
// build, which then spills scalar registers (and every scalar spill costs the whole kernel a vector register, which the loop of
// fused_lean.h does not have at 4 points per lane).  The late reads go through an opaque copy of the argument segment's address.
template <class T>
__device__ __forceinline__ const T &late_args(int offset)
{
    typedef const __attribute__((address_space(4))) unsigned char *kptr;
    kptr p = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const T *)(p + offset);
}
constexpr int kArgOffA = (int)((sizeof(CrfDev) + alignof(FrameArgs) - 1) / alignof(FrameArgs) * alignof(FrameArgs));

template <int PPT>
__global__ void __launch_bounds__(kNTSmall, 4) k_frame_lean(CrfDev c, FrameArgs a)
{
    constexpr int NT = kNTSmall, K = 2, D1 = kD1;
    constexpr bool RELOAD = PPT >= 3;     // (1-2 points per lane: weights, norms and unary energies stay in registers, fused_lean.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int N = c.n_points[f];
#ifndef LCCRF_FRAME_LEAN_DBG
#define LCCRF_FRAME_LEAN_DBG 8            // instrumented builds: 8 = a stamp per build phase; 0 + LCCRF_FRAME_LEAN_STAMP_A=1: inside phase A instead
#endif
#ifndef LCCRF_FRAME_LEAN_STAMP_A
#define LCCRF_FRAME_LEAN_STAMP_A 0
#endif
#define FL_ASTAMP()                                \
    do {                                           \
        if (LCCRF_FRAME_LEAN_STAMP_A) FL_STAMP();  \
    } while (0)
    Instr ins{a.timing, a.timing_block, LCCRF_FRAME_LEAN_DBG, 0, a.timing_lane};
    FL_STAMP();
    // (the lane id far from here -- the exits, which the compiler lays out as guarded blocks at the END of the kernel, and everything
    // behind the loops -- as scalar wave base + mbcnt: a vector register holding `tid` for those would be live across the loops, and
    // the allocator parks such values in the rings' v96..v127 and spills them around the asm; see k_fused_lean)
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    auto lane_id = [&]() {
        int z = 0;
        asm volatile("" : "+v"(z));
        return wave_base + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
    };
    auto flag_unfit = [&]() {             // left to the fallback path (frame_engine.hip: flag_unfit)
        if (lane_id() == 0) {
            if (a.status) *a.status = 1;
            if (a.frame_status) a.frame_status[f] = 1;
        }
    };
    if (N <= 0) {                                         // an empty frame has no lattice (V = 0) and nothing to infer
        const int t = lane_id();
        if (a.with_map) clear_label_bits<NT>(c, f, 0, t);
        if (t < K && a.V_out[t]) a.V_out[t][f] = 0;
        if (t == 0 && a.frame_status) a.frame_status[f] = 0;
        return;
    }
    const int Npad = (N + 3) & ~3;                       // blocks of four, permutohedral_cpu.h:294 (quirk Q1)
    constexpr int kLds = (int)kLdsHalf;                   // the launch's LDS (compile time: every offset below the header is a constant)
    constexpr int hdr_off = kLds - kLeanHdrBytes;
    LeanHdr *hdr = reinterpret_cast<LeanHdr *>(smem + hdr_off);

    // this frame's record block: what the loop re-reads every iteration (one buffer resource, compile-time offsets)
    unsigned char *rec = a.rec + (size_t)f * kLeanRecBytes;
    PointRegs<PPT, K> pr;
    // unary energies: from the labels (densecrf3d.h:116-129 with L = 2), or the caller's.  Every load first, then the stores (the
    // compiler must assume the record block aliases the inputs: a store between two loads serialises their latencies).
    {
        int lab[PPT];
        float2 unr[PPT];
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int ic = min(tid + s * NT, N - 1);
            lab[s] = 0;
            unr[s] = make_float2(0.f, 0.f);
            if (a.label) lab[s] = a.label[(size_t)f * a.maxN + ic];
            else unr[s] = reinterpret_cast<const float2 *>(c.unary)[(size_t)f * c.maxN + ic];
        }
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * NT;
            float2 un = unr[s];
            if (a.label) {
                const int t = lab[s];
                const bool known = t >= 0 && t < 2;
                un.x = !known ? a.tbl[0] : (t == 0 ? a.tbl[3] : a.tbl[2]);       // {u, n0, n1, p0, p1}: p_t for the point's label, n_l for the other
                un.y = !known ? a.tbl[0] : (t == 1 ? a.tbl[4] : a.tbl[1]);       // (no run-time index: that would put the argument block in scratch)
            }
            if (!RELOAD) pr.un[s] = un;
            else if (i < N) reinterpret_cast<float2 *>(rec + kLeanRecUnary)[i] = un;
        }
    }
    if (tid < 32) reinterpret_cast<float *>(smem)[tid] = 0.0f;       // LDS bytes [0, 128) of the loop's plan are zeros
    if (tid < 8) hdr->box[tid >> 2][tid & 3] = 0x7fffffff;
    if (tid == 0) hdr->fail = 0;
    __syncthreads();

    size_t plan = 128;                    // the loop's plan, kernel after kernel (fused_lean.h: layout_lean)
    int V[K];
    unsigned pk[PPT][K][D1];              // (vertex id + 1) | place in the row << 16, as the HBM records of k_fused

    // ---- one kernel's lattice: false = the frame does not fit (uniform) ---------------------------------------------------
    auto build = [&](auto kc) -> bool {
        constexpr int k = decltype(kc)::value;
        // A: point records (elevate, round, rank, barycentric); every corner's vertex as a CELL of a grid over the frame's key range.
        // A 2-D key is (x, y) with x = y (mod 3) (permutohedral_cpu.h:274-279,373: remainder-0 point + canonical offset): u = (x - y) / 3,
        // v = y is a bijection onto Z^2, and a blur neighbour along axis j (key -+ 1, coordinate j +- 2, permutohedral_cpu.h:408-421)
        // is the cell at -+ (-1, +1), (+1, -2), (0, +1).  No hash table: a vertex is found by its cell's index, a neighbour by an offset.
        unsigned cell[PPT][D1];                           // cell index of every corner's vertex (formed behind the range's barrier)
        unsigned pw[PPT], pf[PPT];                        // per point: (u of its remainder-0 vertex & 0xffff) | v << 16; corner flags (point_record2_grid)
        int umin = 0x7fffffff, umax = -0x7fffffff, vmin = 0x7fffffff, vmax = -0x7fffffff;
        bool bad = false;
        FL_ASTAMP();
        float2 ftv[PPT];                                  // (every load of the phase first, see above)
#pragma unroll
        for (int s = 0; s < PPT; ++s) ftv[s] = reinterpret_cast<const float2 *>(a.feat[k])[(size_t)f * a.maxN + min(tid + s * NT, N - 1)];
        if (LCCRF_FRAME_LEAN_STAMP_A) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FL_STAMP(); }
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * NT;
            const float2 ft = ftv[s];
            float feat[2] = {i < N ? ft.x : 0.0f, i < N ? ft.y : 0.0f};   // phantom lanes, :299
            FL_ASTAMP();
            float b[D1];
            int r0x, r0y;
            point_record2_grid(feat, a.scale, a.inv_dp1, b, r0x, r0y, pf[s], bad);
            if (!RELOAD) {
#pragma unroll
                for (int j = 0; j < D1; ++j) pr.bary[s][k][j] = b[j];
            } else if (i < N) {
                float *bo = reinterpret_cast<float *>(rec + kLeanRecBary + k * (kLeanRecPoints * D1 * 4)) + i * D1;
#pragma unroll
                for (int j = 0; j < D1; ++j) bo[j] = b[j];
            }
            const int u = (r0x - r0y) / 3;               // (exact: both are multiples of 3)
            if (i < Npad) {                               // the corners lie within u +- 1, v - 2 .. v + 2 of the remainder-0 vertex
                umin = min(umin, u); umax = max(umax, u);
                vmin = min(vmin, r0y); vmax = max(vmax, r0y);
            }
            pw[s] = ((unsigned)u & 0xffffu) | ((unsigned)r0y << 16);
        }
        FL_ASTAMP();
        {
            // the frame's key range: wavefront minima / maxima by DPP (as wave_incl_scan), one LDS atomic per wavefront and bound
            const int lo_u = wave_min(umin), hi_u = wave_min(-umax), lo_v = wave_min(vmin), hi_v = wave_min(-vmax);
            if ((tid & 63) == 63) {
                atomicMin(&hdr->box[k][0], lo_u);
                atomicMin(&hdr->box[k][1], hi_u);
                atomicMin(&hdr->box[k][2], lo_v);
                atomicMin(&hdr->box[k][3], hi_v);
            }
            if (bad) hdr->fail = 1;
        }
        FL_ASTAMP();
        __syncthreads();
        FL_ASTAMP();
        FL_PSTAMP();
        // (the corners' margin around the remainder-0 vertices, and an empty border of one neighbour step: no bounds checks)
        const int u0 = __builtin_amdgcn_readfirstlane(hdr->box[k][0]) - 2, v0 = __builtin_amdgcn_readfirstlane(hdr->box[k][2]) - 4;
        const int Wp = -__builtin_amdgcn_readfirstlane(hdr->box[k][1]) - u0 + 3, Hp = -__builtin_amdgcn_readfirstlane(hdr->box[k][3]) - v0 + 5;
        const long cells_l = (long)Wp * Hp;
        // (readfirstlane: an exit the compiler cannot prove uniform makes everything merged behind it -- V, the loop's plan -- a vector value)
        if (__builtin_amdgcn_readfirstlane(hdr->fail) || cells_l > kGridMaxCells) return false;
        const int cells = (int)cells_l, chunks = (cells + 7) >> 3;      // 8 cells (16 bytes of ids) per chunk
        const int idmap_off = hdr_off - chunks * 16;      // u16 per cell: 0 = no vertex, else id + 1.  At the end of LDS, dead after D.
        unsigned short *idmap = reinterpret_cast<unsigned short *>(smem + idmap_off);
        if (idmap_off < (int)plan + 64) return false;
        for (int u = tid; u < chunks; u += NT) reinterpret_cast<uint4 *>(idmap)[u] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        // B: mark the cells that hold a vertex (phantom points of the last block of four included, quirk Q1)
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            // corner rem: (u + [rank_y > 2 - rem] - [rank_x > 2 - rem], v + rem - 3 [rank_y > 2 - rem])
            const int u = (int)(short)(pw[s] & 0xffffu), v = (int)pw[s] >> 16;
            const int c0 = (u - u0) * Hp + (v - v0);
            cell[s][0] = (unsigned)c0;
            cell[s][1] = (unsigned)(c0 + 1 + ((pf[s] & 4u) ? Hp - 3 : 0) - ((pf[s] & 1u) ? Hp : 0));
            cell[s][2] = (unsigned)(c0 + 2 + ((pf[s] & 8u) ? Hp - 3 : 0) - ((pf[s] & 2u) ? Hp : 0));
#pragma unroll
            for (int j = 0; j < D1; ++j)
                if (tid + s * NT < Npad) idmap[cell[s][j]] = (unsigned short)1;
        }
        __syncthreads();
        FL_PSTAMP();

        // C: dense vertex ids in cell order: every lane counts the marked cells of its run of chunks, a scan over the lanes, then the
        //    marks become ids.  This kernel's tables in the loop's plan, the scratch by lifetime:
        //      up to the end of D   the id map (end of LDS), vertex -> cell, counters
        //      from the end of D    counters, row starts, row bitmap + prefixes or entry lists (over the dead id map)
        const int cpl = (chunks + NT - 1) / NT, c0 = tid * cpl;
        int mine = 0;
        for (int q = 0; q < cpl; ++q) {
            if (c0 + q < chunks) {
                const uint4 w = reinterpret_cast<const uint4 *>(idmap)[c0 + q];
                mine += __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);      // (marks are 0 / 1 per half)
            }
        }
        int Vk;
        int id = block_excl_scan<NT>(mine, tid, hdr->wave_sum, Vk);
        V[k] = Vk;
        FL_PSTAMP();
        if (Vk > lean_max_v(NT) || (k == 0 && (Vk > chain_max_v(NT) || Vk > 256))) return false;
        const int plan_before = (int)plan;
        const LeanTables lt = lean_tables(plan, Vk, k == 0);
        auto take = [&](int &o, int bytes) { const int r = o; o += (bytes + 15) & ~15; return r; };
        // (kernel 1's scratch starts at the end of kernel 0's tables: its value arrays are not in use before the loop; its row starts are
        //  written in E)
        int vs = k == 0 ? (int)plan : plan_before;
        const int vcell_off = take(vs, Vk * 2);           // vertex -> cell
        const int cnt_off = take(vs, (Vk + 1) * 4);       // arrival counters (short rows), then start | length of every row list
        const int vs_d = vs;                              // end of what lives beside the id map
        if (k != 0) vs = max(vs, (int)plan);
        const int E = N * D1;
        const int W = (((Npad + 31) >> 5) + 3) & ~3;      // bitmap words per vertex, a multiple of 4
        const bool bitmap = (long)E >= 16L * Vk && vs + 5 * Vk * W + 64 <= hdr_off;   // long rows: rank by bitmap, if it fits
        const int bm_off = vs, pre_off = bm_off + Vk * W * 4;             // pre: entries before every 4-word group (u16)
        const int list_cap = (E + 7 * Vk + 8) & ~7;       // short rows: u16 entry lists, rows padded to 8
        const int list_off = vs;
        const int vs_end = bitmap ? pre_off + Vk * W / 2 : list_off + list_cap * 2;
        if (vs_d > idmap_off || (k != 0 && (int)plan > idmap_off) || vs_end > hdr_off || Vk >= 32767 || E + 7 * Vk >= 65535) return false;
        unsigned short *vcell = reinterpret_cast<unsigned short *>(smem + vcell_off);
        unsigned *cnt = reinterpret_cast<unsigned *>(smem + cnt_off);
        unsigned *bm = reinterpret_cast<unsigned *>(smem + bm_off);
        unsigned short *pre = reinterpret_cast<unsigned short *>(smem + pre_off);
        unsigned short *list = reinterpret_cast<unsigned short *>(smem + list_off);
        for (int q = 0; q < cpl; ++q) {
            if (c0 + q < chunks) {
                uint4 w = reinterpret_cast<const uint4 *>(idmap)[c0 + q];
                unsigned h[8] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16, w.z & 0xffffu, w.z >> 16, w.w & 0xffffu, w.w >> 16};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (h[e]) {
                        vcell[id] = (unsigned short)((c0 + q) * 8 + e);
                        h[e] = (unsigned)++id;
                    }
                }
                w = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
                reinterpret_cast<uint4 *>(idmap)[c0 + q] = w;
            }
        }
        for (int v = tid; v <= Vk; v += NT) cnt[v] = 0u;
        __syncthreads();
        FL_PSTAMP();

        // D: every entry learns its vertex (short rows: and joins the vertex's row); blur neighbours per vertex
        unsigned vid[PPT][D1], arr[PPT][D1];              // vertex id, arrival index inside the row (short rows)
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int j = 0; j < D1; ++j) vid[s][j] = (tid + s * NT < Npad) ? (unsigned)idmap[cell[s][j]] - 1u : 0u;
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                arr[s][j] = 0u;
                if (!bitmap && tid + s * NT < N) arr[s][j] = atomicAdd(&cnt[vid[s][j]], 1u);   // real points only: phantoms add vertices, not products
            }
        }
        {
            // (n1 + 1) | (n2 + 1) << 16 per (axis, vertex), 0 = absent: the ids of the two cells at -+ the axis' offset.  The appearance
            // lattice's table stays in LDS (the loop's plan), the smoothness lattice's goes to HBM (the loop reads it pass by pass).
            unsigned *tbl = reinterpret_cast<unsigned *>(smem + (k == 0 ? lt.nbr : 0));
            unsigned *out = reinterpret_cast<unsigned *>(rec + kLeanRecNbr);
            for (int t = tid; t < D1 * Vk; t += NT) {
                const int j = t >= 2 * Vk ? 2 : (t >= Vk ? 1 : 0), v = t - j * Vk;
                const int c = vcell[v], dj = j == 0 ? 1 - Hp : (j == 1 ? Hp - 2 : 1);
                const unsigned word = (unsigned)idmap[c - dj] | ((unsigned)idmap[c + dj] << 16);
                if (k == 0) tbl[t] = word;
                else out[j * (kLeanRecNbrAxis / 4) + v] = word;
            }
        }
        __syncthreads();                                  // the id map and the vertices' cells are dead
        FL_PSTAMP();
        if (bitmap) {
            uint4 *b4 = reinterpret_cast<uint4 *>(bm);
            for (int u = tid; u < Vk * W / 4; u += NT) b4[u] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const int i = tid + s * NT;
                if (i < N) {
#pragma unroll
                    for (int j = 0; j < D1; ++j) atomicOr(&bm[vid[s][j] * W + (i >> 5)], 1u << (i & 31));
                }
            }
            __syncthreads();
        } else {
            uint4 *l4 = reinterpret_cast<uint4 *>(list);   // pads compare greater than every entry (entries are < 0x7fff); E's scans separate this from F's writes
            for (int u = tid; u < list_cap / 8; u += NT) l4[u] = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
        }
        FL_PSTAMP();

        // E: row lengths -> row starts (and, short rows, the start of every padded entry list)
        if (bitmap) {
            // prefix popcounts of every vertex's bitmap, one 16-lane group per vertex (frame_engine.hip phase E)
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
        unsigned short *row = reinterpret_cast<unsigned short *>(smem + lt.row);
        {
            // packed scan: low half = products before the row, high half = padded list entries before it
            const int vper = (Vk + 1 + NT - 1) / NT, v0 = tid * vper;
            unsigned sum = 0u;
            for (int u = 0; u < vper; ++u) {
                const int v = v0 + u;
                if (v < Vk) {
                    const unsigned n = cnt[v];
                    sum += n | (((n + 7u) & ~7u) << 16);
                }
            }
            int tot;
            unsigned run = (unsigned)block_excl_scan<NT>((int)sum, tid, hdr->wave_sum, tot);
            for (int u = 0; u < vper; ++u) {
                const int v = v0 + u;
                if (v <= Vk) {
                    const unsigned n = v < Vk ? cnt[v] : 0u;
                    row[v] = (unsigned short)(run & 0xffffu);
                    cnt[v] = (run >> 16) | (n << 16);      // list start | row length
                    run += n | (((n + 7u) & ~7u) << 16);
                }
            }
        }
        __syncthreads();
        FL_PSTAMP();

        // F/G: the place of every entry in its row = number of smaller entries of the same vertex; one point slot at a time
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
        // eight 16-bit entries against e at once: entries and e are < 0x8000, so (x | 0x8000) - e keeps bit 15 of a half exactly
        // when that half is >= e, and no half ever borrows from its neighbour
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
        FL_PSTAMP();
        __syncthreads();                                  // the next kernel's build (or the loop's product buffer) reuses the scratch
        return true;
    };
    if (!build(std::integral_constant<int, 0>{}) || !build(std::integral_constant<int, 1>{})) {
        flag_unfit();
        return;
    }
    FL_STAMP();

    // ---- the loop's plan for THIS frame's sizes; the tables built above are already in their places ----------------------
    FusedLayout lay;
#pragma unroll
    for (int k = 0; k < K; ++k) V[k] = __builtin_amdgcn_readfirstlane(V[k]);     // (scalar for sure: every offset of the plan derives from them)
    {
        // (kernel 0's rows go to the chain lanes however short they are -- a frame of a few points among full-size ones is not worth a
        //  second instance of the loop; its lattice is small enough for them, or the build has already left)
        // (layout_lean carves with the same lean_tables calls in the same order: the tables built above ARE where it says)
        if (!layout_lean(N, K, V, kChainMinRow, &lay, NT, (size_t)kLds) || !lay.chain0) {
            flag_unfit();
            return;
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            reinterpret_cast<float2 *>(smem + lay.val[k][0])[0] = make_float2(0.f, 0.f);
            reinterpret_cast<float2 *>(smem + lay.val[k][1])[0] = make_float2(0.f, 0.f);
        }
    }
    const FrameArgs &al = late_args<FrameArgs>(kArgOffA);
    const CrfDev &cl_ = late_args<CrfDev>(0);
    float wk[K], alpha[K];
    LeanSrc src;
    const __amdgpu_buffer_rsrc_t rsrc = lean_rsrc(rec, kLeanRecBytes);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        wk[k] = al.w[k];
        alpha[k] = al.alpha;
        src.nbr[k] = src.bary[k] = src.norm[k] = rsrc;
        src.off_nbr[k] = kLeanRecNbr;                     // (only the last kernel's table is there, and only that one is read)
        src.off_bary[k] = kLeanRecBary + k * (kLeanRecPoints * D1 * 4);
        src.off_norm[k] = kLeanRecNorm + k * (kLeanRecPoints * 4);
        src.nbr_axis_bytes[k] = kLeanRecNbrAxis;
    }
    src.unary = rsrc;
    src.off_unary = kLeanRecUnary;
    __syncthreads();                      // (also orders this workgroup's record stores before its loads: one CU, one path to L2)

    ChainLane cl = chain_setup_lean<NT>(smem, lay, V[0], tid);
    place_products_lean<PPT, K, 1, NT>(smem, lay, N, tid, pk, pr);
    FL_STAMP();

    // norm = 1 / (compute(ones) + 1e-20), pairwise3d.h:22-27: one pass of the loop's phases with Q = 1
#pragma unroll
    for (int s = 0; s < PPT; ++s) pr.q[s] = make_float2(1.0f, 1.0f);
    int t = lane_id();
    mean_field_lean<PPT, K, 1, NT, RELOAD, true>(smem, lay, V, N, t, pr, cl, alpha, wk, src, 1, 0.0f, 0.0f, ins);
    __syncthreads();
    FL_STAMP();

    t = lane_id();
    if (RELOAD) {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            typedef unsigned lean_u2 __attribute__((ext_vector_type(2)));
            const lean_u2 u = __builtin_amdgcn_raw_buffer_load_b64(src.unary, (t + s * NT) * 8, src.off_unary, 0);
            pr.un[s] = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
        }
    }
    start_inference<PPT, K, NT>(pr, N, t);
    mean_field_lean<PPT, K, 1, NT, RELOAD, false>(smem, lay, V, N, t, pr, cl, alpha, wk, src, al.n_iter, al.relax, al.omr, ins);
    t = lane_id();
    store_results<PPT, K, NT>(cl_, f, N, t, pr, al.with_map);
    if (t < K && al.V_out[t]) al.V_out[t][f] = t == 0 ? V[0] : V[K - 1];
    if (t == 0 && al.frame_status) al.frame_status[f] = 0;
    FL_STAMP();
    if (kInstr && al.timing && (int)blockIdx.x == al.timing_block && t == al.timing_lane) al.timing[63] = ins.n;
}

template <int PPT>
void launch_lean_ppt(const CrfDev &c, const FrameArgs &a, hipStream_t s)
{
    auto fn = k_frame_lean<PPT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
    fn<<<dim3(c.F), dim3(kNTSmall), a.lds_total, s>>>(c, a);
}

}  // namespace

// Is the half-CU form worth asking for?  Frames of NA points with lattices of the usual SLAM proportions (frame_engine.hip: an appearance
// kernel of ~112 vertices, a smoothness kernel of min(NA + 350, 1150)) must fit the loop's plan; frames with larger lattices flag
// themselves and are re-run, and an engine that sees more than 1/8 of a batch flagged stops asking.
#ifndef LCCRF_FRAME_LEAN_MIN_POINTS
#define LCCRF_FRAME_LEAN_MIN_POINTS 0                 // A/B (scripts/gpu_ab_build.sh): batches of larger frames than this take the kernel (C1 +12 %, N500 +10 % against k_frame<512>)
#endif
bool frame_lean_plausible(int NA, int K, int F)
{
    if (K != 2 || F < kSmallMinFrames || NA <= LCCRF_FRAME_LEAN_MIN_POINTS || NA > 4 * kNTSmall) return false;
    const int vest[2] = {112, std::min(NA + 350, 1150)};
    FusedLayout est;
    return layout_lean(NA, K, vest, kChainMinRow, &est, kNTSmall, kLdsHalf);
}

void launch_frame_lean(const CrfDev &c, const FrameArgs &a, int NA, hipStream_t s)
{
    if (NA <= kNTSmall) launch_lean_ppt<1>(c, a, s);
    else if (NA <= 2 * kNTSmall) launch_lean_ppt<2>(c, a, s);
    else if (NA <= 3 * kNTSmall) launch_lean_ppt<3>(c, a, s);
    else launch_lean_ppt<4>(c, a, s);
}

}  // namespace fb
}  // namespace lccrf
