// frame_build.h -- what the one-launch frame kernels share (frame_engine.hip: one frame per workgroup of 1024 lanes, or of 512 for
// small frames; frame_lean.hip: full-size frames on half a CU): the launch arguments and the lattice build's helpers.
#pragma once

#include "engine.h"
#include "device_math.h"
#include "fused_loop.h"

namespace lccrf {
namespace fb {

using namespace fl;

constexpr unsigned kEmptyKey = 0x80008000u;       // (-32768, -32768): not a key any sane feature produces; a frame that does is sent to the fallback path
constexpr int kHdr = 512;                          // [0,128) unused, [128,192) zero block, [192,512) scan scratch / flags

struct FrameArgs {
    const float *feat[kMaxFusedK];        // [F][maxN][2] features of kernel k (already divided by the stdev)
    float w[kMaxFusedK];                  // kernel weights (PottsPotential3D::w_)
    float scale[2], inv_dp1, alpha;       // d = 2 constants: permutohedral_cpu.h:249,282-285,681
    int maxN;                             // per-frame stride of feat / label / unary / Q / map
    const int16_t *label;                 // non-null: unary energies from labels and tbl (densecrf3d.h:100-130, L = 2)
    float tbl[5];                         //   {u, n0, n1, p0, p1}
    int n_iter, with_map;
    float relax;
    int hcap;                             // hash capacity (power of two >= 1024)
    int lds_total;                        // dynamic LDS bytes of the launch
    int *V_out[kMaxFusedK];               // [F] vertices per kernel (reference M_), or null
    int *status;                          // pinned host word: set to 1 when a frame does not fit this kernel's LDS plan
    int *frame_status;                    // device [F] or null: 1 for exactly the frames that did not fit (they alone are re-run), else 0
    unsigned *dual;                       // DUAL launches: [F][kDualWords] hand-off area between a frame's two workgroups, else null
    unsigned dual_epoch;                  //   value the helper publishes in word 0 when its tables are complete (changes every launch)
    int n_single;                         // >= 0: the point count of the launch's only frame (else c.n_points[f])
    int drop_helper;                      // instrumented builds only (LCCRF_DUAL_DROP_HELPER): the helper workgroup leaves at once -- the main one must time out and fall back
    unsigned *done;                       // single-frame launches: pinned host word that receives done_epoch when the frame's results (labels in
    unsigned done_epoch;                  //   pinned memory, status words) are visible to the host -- earlier than the runtime's completion signal
    long long *timing;                    // instrumented builds only
    int timing_block, timing_lane;
    // frame_lean.hip only: [F][kLeanRecBytes] -- the per-point records its loop re-reads every iteration (fused_lean.h: kLeanRec*)
    unsigned char *rec;
    float omr;                            // 1 - relax (fp32, densecrf3d.h:94), formed on the host
};

__device__ __forceinline__ unsigned hash32(unsigned key)
{
    unsigned h = key * 2654435761u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    return h;
}

// Inclusive scan over the 64 lanes of a wavefront with DPP: Hillis-Steele inside each row of 16 lanes (row_shr),
// then the row totals are handed on with row_bcast:15 (into rows 1, 3) and row_bcast:31 (into rows 2, 3).
__device__ __forceinline__ int wave_incl_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}

// Exclusive scan of one int per lane over the workgroup; returns the prefix, `total` the grand total.
// Two barriers; wave_sum may be reused right after the call returns only behind another barrier.
template <int NT>
__device__ __forceinline__ int block_excl_scan(int x, int tid, int *wave_sum, int &total)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int incl = wave_incl_scan(x);
    __syncthreads();                                      // a previous scan's readers are done with wave_sum
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    const int ws = lane < NT / 64 ? wave_sum[lane] : 0;  // the 16 wavefront totals, scanned again inside every wavefront
    const int wincl = wave_incl_scan(ws);
    total = __builtin_amdgcn_readlane(wincl, NT / 64 - 1);
    const int wbase = __builtin_amdgcn_readlane(wincl - ws, __builtin_amdgcn_readfirstlane(wave));
    return wbase + incl - x;
}

// key of the simplex corner with remainder `rem` of a point record, packed (x | y << 16)
__device__ __forceinline__ unsigned corner_key(const int16_t (&r0)[2], const uint8_t (&rk)[2], int rem)
{
    const unsigned x = (unsigned short)vertex_coord<2>(r0[0], rk[0], rem);
    const unsigned y = (unsigned short)vertex_coord<2>(r0[1], rk[1], rem);
    return x | (y << 16);
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
static_assert(alignof(CrfDev) == 8 && alignof(FrameArgs) == 8, "the frame kernels' arguments (CrfDev, FrameArgs) as the ABI lays them out: by value, naturally aligned, in order");

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

// ---- one kernel's lattice without a hash table (frame_lean.hip's header; both frame kernels) ---------------------------------------
// Where the caller wants one kernel's tables and scratch (LDS byte offsets), decided once the vertex count is known:
struct GridPlace {
    int row;              // u16 [V + 2] row starts (persistent)
    int scratch_a;        // vertex -> cell and the row counters: alive from phase C to G
    int scratch_b;        // the row bitmap + prefixes or the entry lists start at max(end of the former, this); born after D, they may
                          //   run over the dead id map
};
// Phase A of one kernel: this lane's point records and the workgroup's key range.  No barrier: the caller runs it for every kernel of
// the frame and then ONE barrier (one key-range reduction and one round trip of feature loads per frame instead of one per kernel).
//   ftv          this lane's features (point tid + s * NT; loaded by the caller: every load of a phase before its stores)
//   emit_bary(s, i, b)         the barycentric weights of point slot s (point i)
//   pw, pf       per point: (u of its remainder-0 vertex & 0xffff) | v << 16; corner flags (point_record2_grid) -- for grid_build
//   box, fail    LDS: four ints preset to 0x7fffffff and a flag preset to 0, a barrier ago
// A 2-D key is (x, y) with x = y (mod 3) (permutohedral_cpu.h:274-279,373: remainder-0 point + canonical offset): u = (x - y) / 3,
// v = y is a bijection onto Z^2, and a blur neighbour along axis j (key -+ 1, coordinate j +- 2, permutohedral_cpu.h:408-421)
// is the cell at -+ (-1, +1), (+1, -2), (0, +1).  No hash table: a vertex is found by its cell's index, a neighbour by an offset.
template <int NT, int PPT, class EmitBary>
__device__ __forceinline__ void grid_records(int tid, int N, const float2 (&ftv)[PPT], const float *scale, float inv_dp1, EmitBary emit_bary,
                                             unsigned (&pw)[PPT], unsigned (&pf)[PPT], int *box, int *fail)
{
    constexpr int D1 = kD1;
    const int Npad = (N + 3) & ~3;                        // blocks of four, permutohedral_cpu.h:294 (quirk Q1)
    int umin = 0x7fffffff, umax = -0x7fffffff, vmin = 0x7fffffff, vmax = -0x7fffffff;
    bool bad = false;
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int i = tid + s * NT;
        const float2 ft = ftv[s];
        float feat[2] = {i < N ? ft.x : 0.0f, i < N ? ft.y : 0.0f};   // phantom lanes, :299
        float b[D1];
        int r0x, r0y;
        point_record2_grid(feat, scale, inv_dp1, b, r0x, r0y, pf[s], bad);
        emit_bary(s, i, b);
        const int u = (r0x - r0y) / 3;                   // (exact: both are multiples of 3)
        if (i < Npad) {                                   // the corners lie within u +- 1, v - 2 .. v + 2 of the remainder-0 vertex
            umin = min(umin, u); umax = max(umax, u);
            vmin = min(vmin, r0y); vmax = max(vmax, r0y);
        }
        pw[s] = ((unsigned)u & 0xffffu) | ((unsigned)r0y << 16);
    }
    // the frame's key range: wavefront minima / maxima by DPP (as wave_incl_scan), one LDS atomic per wavefront and bound
    const int lo_u = wave_min(umin), hi_u = wave_min(-umax), lo_v = wave_min(vmin), hi_v = wave_min(-vmax);
    if ((tid & 63) == 63) {
        atomicMin(&box[0], lo_u);
        atomicMin(&box[1], hi_u);
        atomicMin(&box[2], lo_v);
        atomicMin(&box[3], hi_v);
    }
    if (bad) *fail = 1;
}

// Phases B .. G of one kernel, a barrier behind grid_records.  Every lane of the workgroup calls it (barriers inside); false = the frame
// does not fit (uniform).
//   wave_sum     LDS: 16 ints of scan scratch
//   floor_lo     LDS below this is the caller's (persistent tables);  limit: end of the scratch -- the id map sits right below it
//   place(V, GridPlace &)      the caller's plan for a lattice of V vertices (false: does not fit)
//   store_nbr(j, v, t, word)   (n1 + 1) | (n2 + 1) << 16 of vertex v along axis j (t = j * V + v), 0 = absent
//   pk[s][j]     (vertex id + 1) | place in the row << 16 of every entry, as the HBM records of k_fused
//   rowmax       LDS int that receives the longest row (atomicMax), or null
template <int NT, int PPT, class Place, class StoreNbr>
__device__ __forceinline__ bool grid_build(unsigned char *smem, int tid, int N, const unsigned (&pw)[PPT], const unsigned (&pf)[PPT],
                                           const int *box, const int *fail, int *wave_sum, int floor_lo, int limit, Place place,
                                           StoreNbr store_nbr, int &V_out, unsigned (&pk)[PPT][kD1], int *rowmax, Instr &ins)
{
    constexpr int D1 = kD1;
    const int Npad = (N + 3) & ~3;
    unsigned cell[PPT][D1];                               // cell index of every corner's vertex
    // (the corners' margin around the remainder-0 vertices, and an empty border of one neighbour step: no bounds checks)
    const int u0 = __builtin_amdgcn_readfirstlane(box[0]) - 2, v0 = __builtin_amdgcn_readfirstlane(box[2]) - 4;
    const int Wp = -__builtin_amdgcn_readfirstlane(box[1]) - u0 + 3, Hp = -__builtin_amdgcn_readfirstlane(box[3]) - v0 + 5;
    const long cells_l = (long)Wp * Hp;
    // (readfirstlane: an exit the compiler cannot prove uniform makes everything merged behind it -- V, the caller's plan -- a vector value)
    if (__builtin_amdgcn_readfirstlane(*fail) || cells_l > kGridMaxCells) return false;
    const int cells = (int)cells_l, chunks = (cells + 7) >> 3;      // 8 cells (16 bytes of ids) per chunk
    const int idmap_off = limit - chunks * 16;      // u16 per cell: 0 = no vertex, else id + 1.  At the end of LDS, dead after D.
    unsigned short *idmap = reinterpret_cast<unsigned short *>(smem + idmap_off);
    if (idmap_off < floor_lo + 64) return false;
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
    //    marks become ids.  The caller places this kernel's tables; the scratch by lifetime:
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
    int id = block_excl_scan<NT>(mine, tid, wave_sum, Vk);
    V_out = Vk;
    FL_PSTAMP();
    GridPlace gp;
    if (!place(Vk, gp)) return false;
    auto take = [&](int &o, int bytes) { const int r = o; o += (bytes + 15) & ~15; return r; };
    int vs = gp.scratch_a;
    const int vcell_off = take(vs, Vk * 2);           // vertex -> cell
    const int cnt_off = take(vs, (Vk + 1) * 4);       // arrival counters (short rows), then start | length of every row list
    const int vs_d = vs;                              // end of what lives beside the id map
    vs = max(vs, gp.scratch_b);
    const int E = N * D1;
    const int W = (((Npad + 31) >> 5) + 3) & ~3;      // bitmap words per vertex, a multiple of 4
    const bool bitmap = (long)E >= 16L * Vk && vs + 5 * Vk * W + 64 <= limit;   // long rows: rank by bitmap, if it fits
    const int bm_off = vs, pre_off = bm_off + Vk * W * 4;             // pre: entries before every 4-word group (u16)
    const int list_cap = (E + 7 * Vk + 8) & ~7;       // short rows: u16 entry lists, rows padded to 8
    const int list_off = vs;
    const int vs_end = bitmap ? pre_off + Vk * W / 2 : list_off + list_cap * 2;
    if (vs_d > idmap_off || vs_end > limit || Vk >= 32767 || E + 7 * Vk >= 65535) return false;
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
        // (n1 + 1) | (n2 + 1) << 16 per (axis, vertex), 0 = absent: the ids of the two cells at -+ the axis' offset
        for (int t = tid; t < D1 * Vk; t += NT) {
            const int j = t >= 2 * Vk ? 2 : (t >= Vk ? 1 : 0), v = t - j * Vk;
            const int c = vcell[v], dj = j == 0 ? 1 - Hp : (j == 1 ? Hp - 2 : 1);
            store_nbr(j, v, t, (unsigned)idmap[c - dj] | ((unsigned)idmap[c + dj] << 16));
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
    unsigned short *row = reinterpret_cast<unsigned short *>(smem + gp.row);
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
        unsigned run = (unsigned)block_excl_scan<NT>((int)sum, tid, wave_sum, tot);
        for (int u = 0; u < vper; ++u) {
            const int v = v0 + u;
            if (v <= Vk) {
                const unsigned n = v < Vk ? cnt[v] : 0u;
                row[v] = (unsigned short)(run & 0xffffu);
                cnt[v] = (run >> 16) | (n << 16);      // list start | row length
                run += n | (((n + 7u) & ~7u) << 16);
            }
        }
        if (rowmax) {                                 // the kernel's longest row (k_frame: decides the chain path)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
            if ((tid & 63) == 0 && mx > 0) atomicMax(rowmax, mx);
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
            pk[s][j] = (v + 1u) | (((unsigned)row[v] + (real ? r : 0u)) << 16);
        }
    }
    FL_PSTAMP();
    __syncthreads();                                      // the next kernel's build (or the loop's product buffers) reuses the scratch
    return true;
}

// frame_lean.hip: frames of 1025 .. 2048 points, two-kernel SLAM configuration, 512 lanes and half the CU's LDS per frame
// (NA = the batch's largest frame).  Frames that do not fit flag themselves as in k_frame.
bool frame_lean_plausible(int NA, int K, int F);
void launch_frame_lean(const CrfDev &c, const FrameArgs &a, int NA, hipStream_t s);

}  // namespace fb
}  // namespace lccrf
