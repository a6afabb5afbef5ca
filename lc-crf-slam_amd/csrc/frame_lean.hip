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
template <int PPT>
__global__ void __launch_bounds__(kNTSmall, 4) k_frame_lean(CrfDev c, FrameArgs a)
{
    constexpr int NT = kNTSmall, K = 2, D1 = kD1;
    constexpr bool RELOAD = PPT >= 3;     // (1-2 points per lane: weights, norms and unary energies stay in registers, fused_lean.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int N = c.n_points[f];
    Instr ins{a.timing, a.timing_block, 8, 0, a.timing_lane};
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

    // ---- A, both kernels: point records and key ranges behind ONE barrier (every load first: the record block may alias the inputs)
    unsigned pw[K][PPT], pf[K][PPT];
    {
        float2 ftv[K][PPT];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int s = 0; s < PPT; ++s) ftv[k][s] = reinterpret_cast<const float2 *>(a.feat[k])[(size_t)f * a.maxN + min(tid + s * NT, N - 1)];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            auto emit_bary = [&](int s, int i, const float (&b)[D1]) {
                if (!RELOAD) {
#pragma unroll
                    for (int j = 0; j < D1; ++j) pr.bary[s][k][j] = b[j];
                } else if (i < N) {
                    float *bo = reinterpret_cast<float *>(rec + kLeanRecBary + k * (kLeanRecPoints * D1 * 4)) + i * D1;
#pragma unroll
                    for (int j = 0; j < D1; ++j) bo[j] = b[j];
                }
            };
            grid_records<NT, PPT>(tid, N, ftv[k], a.scale, a.inv_dp1, emit_bary, pw[k], pf[k], hdr->box[k], &hdr->fail);
        }
    }
    __syncthreads();
    FL_PSTAMP();
    // ---- B .. G, one kernel's lattice: false = the frame does not fit (uniform) ---------------------------------------------
    auto build = [&](auto kc) -> bool {
        constexpr int k = decltype(kc)::value;
        const int plan_before = (int)plan;
        int lt_nbr = 0;
        // this kernel's persistent tables straight into their places in the loop's plan (fused_lean.h: lean_tables); kernel 1's scratch
        // starts at the end of kernel 0's tables -- its value arrays are not in use before the loop, its row starts are written in E
        auto place = [&](int Vk, GridPlace &gp) -> bool {
            if (Vk > lean_max_v(NT) || (k == 0 && (Vk > chain_max_v(NT) || Vk > 256))) return false;
            const LeanTables lt = lean_tables(plan, Vk, k == 0);
            lt_nbr = lt.nbr;
            gp.row = lt.row;
            gp.scratch_a = k == 0 ? (int)plan : plan_before;
            gp.scratch_b = (int)plan;
            return true;
        };
        // the appearance lattice's table stays in LDS (the loop's plan), the smoothness lattice's goes to HBM (the loop reads it pass by pass)
        auto store_nbr = [&](int j, int v, int t, unsigned word) {
            if (k == 0) reinterpret_cast<unsigned *>(smem + lt_nbr)[t] = word;
            else reinterpret_cast<unsigned *>(rec + kLeanRecNbr)[j * (kLeanRecNbrAxis / 4) + v] = word;
        };
        unsigned pkk[PPT][D1];
        const bool ok = grid_build<NT, PPT>(smem, tid, N, pw[k], pf[k], hdr->box[k], &hdr->fail, hdr->wave_sum, plan_before, hdr_off, place,
                                            store_nbr, V[k], pkk, nullptr, ins);
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int j = 0; j < D1; ++j) pk[s][k][j] = pkk[s][j];
        return ok;
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
