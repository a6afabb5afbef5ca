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
#include "fused_loop.h"
#include "fused_lean.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace lccrf {

namespace {

using namespace fl;

constexpr int kPrepMinFrames = 64;        // prepared launch records pay for batches that fill the chip; a handful of frames keeps the self-contained kernel

struct FusedArgs {
    KernelDev kd[kMaxFusedK];
    FusedLayout lay;
    int n_iter, with_map;
    float relax, omr;                     // omr = 1 - relax (fp32, densecrf3d.h:94), formed on the host
    long long *timing;                    // instrumented builds: shader-clock stamps of one workgroup (LCCRF_FUSED_TIMING=<block index + 1>)
    int timing_block, timing_lane;
    int dbg;                              // instrumented builds: LCCRF_FUSED_DBG (see fused_loop.h)
    unsigned char *prep;                  // lean plan: the frames' prepared launch records (fused_lean.h: LeanPrepPlan), prep_stride bytes each
    int prep_stride;
    int Vcap[kMaxFusedK];                 // the vertex counts the LDS plan was sized for (maxima over the batch)
};

// One workgroup per frame.  Lane t owns points t, t+NT, ... (PPT of them); the per-frame records a build
// kernel left in HBM (~150 KB) are loaded once, then fused_loop.h runs the whole inference on chip.
//   NT = 1024, or 512 for frames small enough that two workgroups share a CU (fused_loop.h: kNTSmall)
//   CH = 0 / 1: kernel 0 short-row / chain, decided by the host
// (second launch bound: four wavefronts per SIMD = 128 registers per lane in both shapes -- what lets two 512-lane
// workgroups be co-resident)
//   MODE  0 / 1 / 2: the self-contained kernel / its prologue only, into the frame's prepared block / the run from that block
//         (fused_lean.h: LeanPrepPlan -- the same blocks as k_fused_lean's: what chain_setup and place_products derive from the
//         lattices is written once behind a build)
template <int NT, int PPT, int K, int CH, int MODE>
__global__ void __launch_bounds__(NT, 4) k_fused(CrfDev c, FusedArgs a)
{
    constexpr int D1 = kD1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int N = c.n_points[f];
    Instr ins{a.timing, a.timing_block, a.dbg, 0, a.timing_lane};
    FL_STAMP();

    PointRegs<PPT, K> pr;
    int V[K];
#pragma unroll
    for (int k = 0; k < K; ++k) V[k] = a.kd[k].V[f];
    const FusedLayout &lay = a.lay;

    if constexpr (MODE == 2) {
        // every load depends on the frame index only: one round trip between the kernel's start and its first barrier
        typedef unsigned pf_u2 __attribute__((ext_vector_type(2)));
        typedef unsigned pf_u4 __attribute__((ext_vector_type(4)));
        const LeanPrepPlan pp = lean_prep_plan(lay, K, a.Vcap, NT, PPT);
        const __amdgpu_buffer_rsrc_t rp = lean_rsrc(a.prep + (size_t)f * a.prep_stride, (size_t)a.prep_stride);
        const __amdgpu_buffer_rsrc_t ru = lean_rsrc(c.unary + (size_t)f * c.maxN * 2, (size_t)c.maxN * 8);
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const pf_u2 u = __builtin_amdgcn_raw_buffer_load_b64(ru, (tid + s * NT) * 8, 0, 0);   // (a lane past the frame reads 0 or a spare row: unused)
            pr.un[s] = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const KernelDev &kd = a.kd[k];
                const __amdgpu_buffer_rsrc_t rb = lean_rsrc(kd.bary + (size_t)f * kd.Epad, (size_t)kd.Epad * 4);
                const __amdgpu_buffer_rsrc_t rn = lean_rsrc(kd.norm + (size_t)f * kd.maxN, (size_t)kd.maxN * 4);
                const lean_u3 w = __builtin_amdgcn_raw_buffer_load_b96(rp, tid * 12, ((s * K + k) * NT) * 12, 0);
                pr.ix[s][k][0] = w.x;
                pr.ix[s][k][1] = w.y;
                pr.ix[s][k][2] = w.z;
                const lean_u3 b = __builtin_amdgcn_raw_buffer_load_b96(rb, (tid + s * NT) * (D1 * 4), 0, 0);
                pr.bary[s][k][0] = __uint_as_float(b.x);
                pr.bary[s][k][1] = __uint_as_float(b.y);
                pr.bary[s][k][2] = __uint_as_float(b.z);
                pr.wn[s][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rn, (tid + s * NT) * 4, 0, 0));
            }
        }
        const pf_u2 clw = __builtin_amdgcn_raw_buffer_load_b64(rp, tid * 8, pp.cl_off, 0);
        // the LDS tables: at most two 16-byte pieces per lane and table (launch_inference_fused checks the plan), all requested before any is stored
        pf_u4 trow[K][2], tnbr[K][2];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                trow[k][r] = tnbr[k][r] = pf_u4{0u, 0u, 0u, 0u};
                const int b = (tid + r * NT) * 16;
                if (b < pp.row_bytes[k]) trow[k][r] = __builtin_amdgcn_raw_buffer_load_b128(rp, b, pp.row_off[k], 0);
                if (b < pp.nbr_bytes[k]) tnbr[k][r] = __builtin_amdgcn_raw_buffer_load_b128(rp, b, pp.nbr_off[k], 0);
            }
        FL_PSTAMP();
        if (N <= 0) {
            if (a.with_map) clear_label_bits<NT>(c, f, 0, tid);
            return;
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int b = (tid + r * NT) * 16;
                if (b < pp.row_bytes[k]) *reinterpret_cast<pf_u4 *>(smem + lay.row[k] + b) = trow[k][r];
                if (b < pp.nbr_bytes[k]) *reinterpret_cast<pf_u4 *>(smem + lay.nbr[k] + b) = tnbr[k][r];
            }
        if (tid < 16) reinterpret_cast<float *>(smem + lay.zero)[tid] = 0.0f;
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                reinterpret_cast<float2 *>(smem + lay.val[k][0])[0] = make_float2(0.f, 0.f);
                reinterpret_cast<float2 *>(smem + lay.val[k][1])[0] = make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) pr.wn[s][k] = a.kd[k].w * pr.wn[s][k];   // pairwise3d.h:77 (w_*norm_[i])
        __syncthreads();
        FL_PSTAMP();
        const ChainLane cl{clw.x, clw.y};
        start_inference<PPT, K, NT>(pr, N, tid);
        FL_STAMP();
        float alpha[K];
#pragma unroll
        for (int k = 0; k < K; ++k) alpha[k] = a.kd[k].alpha;
        mean_field<PPT, K, CH, NT, true>(smem, lay, V, N, tid, pr, cl, alpha, a.n_iter, a.relax, ins);
        store_results<PPT, K, NT>(c, f, N, tid, pr, a.with_map);
        FL_STAMP();
        if (kInstr && a.timing && (int)blockIdx.x == a.timing_block && tid == a.timing_lane) a.timing[63] = ins.n;
        return;
    }

    if (N <= 0) {                         // nothing to infer (and nothing below may index an empty frame)
        if (a.with_map && MODE != 1) clear_label_bits<NT>(c, f, 0, tid);
        return;
    }
    if (FL_DBG(4)) {                      // debugging aid: NaN-poison the LDS so that reads of unwritten LDS show up
        for (int i = tid; i < lay.total / 4; i += NT) reinterpret_cast<unsigned *>(smem)[i] = 0x7fc00000u + (unsigned)i;
        __syncthreads();
    }

    // All global loads of the prologue are issued before anything waits on them: the lattice
    // tables first (their LDS stores come last), then the per-point records.  Indices are clamped
    // instead of branched on, so that the loads stay back to back.
    constexpr int kNbrRounds = 4096 / NT, kRowRounds = 2048 / NT;   // covers V <= 1365 in registers; larger lattices finish in copy loops
    unsigned g_nbr[K][kNbrRounds];
    int g_row[K][kRowRounds];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const KernelDev &kd = a.kd[k];
        const unsigned *gn = kd.nbr16 + (size_t)f * D1 * kd.Epad;            // already (n1+1) | (n2+1) << 16
        const int *gr = kd.rowptr + (size_t)f * (kd.Epad + 1);
#pragma unroll
        for (int r = 0; r < kNbrRounds; ++r) {            // element idx = j*V + v, j-major like the LDS copy
            const int idx = min(tid + r * NT, D1 * V[k] - 1);
            const int j = idx >= 2 * V[k] ? 2 : (idx >= V[k] ? 1 : 0);
            g_nbr[k][r] = gn[(size_t)j * kd.Epad + (idx - j * V[k])];
        }
#pragma unroll
        for (int r = 0; r < kRowRounds; ++r) g_row[k][r] = gr[min(tid + r * NT, V[k])];
    }
    unsigned pk[PPT][K][D1];              // (vertex id + 1) | place in the row << 16
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int ic = min(tid + s * NT, N - 1);
        pr.un[s] = reinterpret_cast<const float2 *>(c.unary)[(size_t)f * c.maxN + ic];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const KernelDev &kd = a.kd[k];
            const size_t e0 = (size_t)f * kd.Epad + (size_t)ic * D1;
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                pk[s][k][j] = kd.pk[e0 + j];
                pr.bary[s][k][j] = kd.bary[e0 + j];
            }
            pr.wn[s][k] = kd.norm[(size_t)f * kd.maxN + ic];
        }
    }
#pragma unroll
    for (int s = 0; s < PPT; ++s)
#pragma unroll
        for (int k = 0; k < K; ++k) pr.wn[s][k] = a.kd[k].w * pr.wn[s][k];   // pairwise3d.h:77 (w_*norm_[i])
    FL_PSTAMP();

    // ---- per-frame lattice tables into LDS --------------------------------------------
#pragma unroll
    for (int k = 0; k < K; ++k) {
        unsigned *nbr = reinterpret_cast<unsigned *>(smem + lay.nbr[k]);
        unsigned short *row = reinterpret_cast<unsigned short *>(smem + lay.row[k]);
#pragma unroll
        for (int r = 0; r < kNbrRounds; ++r) {
            const int idx = tid + r * NT;
            if (idx < D1 * V[k]) nbr[idx] = g_nbr[k][r];
        }
#pragma unroll
        for (int r = 0; r < kRowRounds; ++r)
            if (tid + r * NT <= V[k]) row[tid + r * NT] = (unsigned short)g_row[k][r];
        // lattices with more vertices than the register rounds cover (sparse frames): plain copy loops
        const KernelDev &kd = a.kd[k];
        const unsigned *gn = kd.nbr16 + (size_t)f * D1 * kd.Epad;
        for (int idx = tid + kNbrRounds * NT; idx < D1 * V[k]; idx += NT) {
            const int j = idx >= 2 * V[k] ? 2 : (idx >= V[k] ? 1 : 0);
            nbr[idx] = gn[(size_t)j * kd.Epad + (idx - j * V[k])];
        }
        const int *gr = kd.rowptr + (size_t)f * (kd.Epad + 1);
        for (int v = tid + kRowRounds * NT; v <= V[k]; v += NT) row[v] = (unsigned short)gr[v];
    }
    if (tid < 16) reinterpret_cast<float *>(smem + lay.zero)[tid] = 0.0f;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            reinterpret_cast<float2 *>(smem + lay.val[k][0])[0] = make_float2(0.f, 0.f);
            reinterpret_cast<float2 *>(smem + lay.val[k][1])[0] = make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    FL_PSTAMP();

    // The lattice tables were requested first and are in LDS now; the per-point records (unary, pk, bary, norm) are still
    // on their way.  Rank the chain rows (four barriers, LDS atomics -- needs the row table only) before anything touches
    // a point record, so that this work runs under the tail of the loads instead of behind it.
    ChainLane cl{0u, 0u};
    if (CH != 0 && chain_k<CH>(lay, 0)) cl = chain_setup(smem, lay, V[0], tid);
    FL_PSTAMP();
    if (MODE != 1) start_inference<PPT, K, NT>(pr, N, tid);
    place_products<PPT, K, CH, NT>(smem, lay, N, tid, pk, pr);
    FL_STAMP();
    if constexpr (MODE == 1) {
        // ---- the prepared block of this frame: ix words, chain lanes, LDS tables (LeanPrepPlan) ---------------------------------------
        const LeanPrepPlan pp = lean_prep_plan(lay, K, a.Vcap, NT, PPT);
        unsigned char *pf = a.prep + (size_t)f * a.prep_stride;
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                unsigned *w = reinterpret_cast<unsigned *>(pf) + ((size_t)(s * K + k) * NT + tid) * 3;
                w[0] = pr.ix[s][k][0];
                w[1] = pr.ix[s][k][1];
                w[2] = pr.ix[s][k][2];
            }
        reinterpret_cast<uint2 *>(pf + pp.cl_off)[tid] = make_uint2(cl.a, cl.b);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            for (int b = tid * 16; b < pp.row_bytes[k]; b += NT * 16)
                *reinterpret_cast<uint4 *>(pf + pp.row_off[k] + b) = *reinterpret_cast<const uint4 *>(smem + lay.row[k] + b);
            for (int b = tid * 16; b < pp.nbr_bytes[k]; b += NT * 16)
                *reinterpret_cast<uint4 *>(pf + pp.nbr_off[k] + b) = *reinterpret_cast<const uint4 *>(smem + lay.nbr[k] + b);
        }
        return;
    }

    float alpha[K];
#pragma unroll
    for (int k = 0; k < K; ++k) alpha[k] = a.kd[k].alpha;
    mean_field<PPT, K, CH, NT, true>(smem, lay, V, N, tid, pr, cl, alpha, a.n_iter, a.relax, ins);

    store_results<PPT, K, NT>(c, f, N, tid, pr, a.with_map);
    FL_STAMP();
    if (kInstr && a.timing && (int)blockIdx.x == a.timing_block && tid == a.timing_lane) a.timing[63] = ins.n;
}


// The same inference on HALF a CU's LDS (fused_lean.h): frames of 1025 .. ~2300 keypoints, two workgroups per CU.
//   NT = 512 (8 wavefronts, 4 per SIMD with both workgroups resident: 128 registers per lane), 3 or 4 points per lane;
//   RELOAD: unary energies, barycentric weights and norms are re-read every iteration instead of kept in registers
//   MODE  0: the prologue derives everything from the build's records (ranking, placement, table conversion) in every launch;
//         1: that prologue ONLY -- its results go to the frame's prepared block (fused_lean.h: LeanPrepPlan), once per build;
//         2: the prologue is a handful of coalesced loads of that block (what an inference runs from the second launch on)
template <int NT, int PPT, int K, int CH, bool RELOAD, int MODE>
__global__ void __launch_bounds__(NT, NT == 384 ? 3 : 4) k_fused_lean(CrfDev c, FusedArgs a)
{
    constexpr int D1 = kD1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int fo = blockIdx.x;                            // the frame; f: whose RECORDS are read (LEAN_SKIP(1024), timing only: the first 64 frames' --
    const int f = LEAN_SKIP(1024) ? (fo & 63) : fo;       //   identical to this frame's in the bench's tiled batch -- so that they come out of L2, as if prefetched)
    const int tid = threadIdx.x;
    const int N = c.n_points[f];
    Instr ins{a.timing, a.timing_block, a.dbg, 0, a.timing_lane};
    FL_STAMP();

    PointRegs<PPT, K> pr;
    int V[K];
#pragma unroll
    for (int k = 0; k < K; ++k) V[k] = a.kd[k].V[f];
    // (the lane id as scalar wave base + mbcnt wherever it is needed far from here -- the empty-frame exit, which the compiler lays out
    // as a guarded block at the END of the kernel, and everything behind the loop: a vector register holding `tid` for those would be
    // live across the whole loop, and the allocator parks such values in the rings' v96..v127 and spills them around the asm)
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    auto lane_id = [&]() {
        int z = 0;
        asm volatile("" : "+v"(z));
        return wave_base + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
    };
    const FusedLayout &lay = a.lay;
    if constexpr (MODE == 2) {
        // ---- prologue from the prepared block: every load below depends on the frame index only, so all of them -- the words, the
        // chain lane, the tables, the unary energies and the FIRST iteration's weights and norms -- are in flight at once: one round
        // trip to HBM / L2 between the kernel's start and its first barrier (the self-contained prologue pays four or five: the point
        // count, the records, the table conversion loops one after the other, then the loop's first re-read)
        float wk[K];
        LeanSrc src;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const KernelDev &kd = a.kd[k];
            wk[k] = kd.w;
            src.nbr[k] = lean_rsrc(kd.nbr16 + (size_t)f * D1 * kd.Epad, (size_t)D1 * kd.Epad * 4);
            src.bary[k] = lean_rsrc(kd.bary + (size_t)f * kd.Epad, (size_t)kd.Epad * 4);
            src.norm[k] = lean_rsrc(kd.norm + (size_t)f * kd.maxN, (size_t)kd.maxN * 4);
            src.nbr_axis_bytes[k] = kd.Epad * 4;
            src.off_nbr[k] = src.off_bary[k] = src.off_norm[k] = 0;
        }
        src.unary = lean_rsrc(c.unary + (size_t)f * c.maxN * 2, (size_t)c.maxN * 8);
        src.off_unary = 0;
        const LeanPrepPlan pp = lean_prep_plan(lay, K, a.Vcap, NT, PPT);
        const __amdgpu_buffer_rsrc_t rp = lean_rsrc(a.prep + (size_t)f * a.prep_stride, (size_t)a.prep_stride);
        typedef unsigned lean_u2 __attribute__((ext_vector_type(2)));
        typedef unsigned lean_u4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const lean_u2 u = __builtin_amdgcn_raw_buffer_load_b64(src.unary, (tid + s * NT) * 8, 0, LEAN_AUX(1));   // (a lane past the frame reads 0 or a spare row: unused)
            pr.un[s] = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const lean_u3 w = __builtin_amdgcn_raw_buffer_load_b96(rp, tid * 12, ((s * K + k) * NT) * 12, LEAN_AUX(1));
                pr.ix[s][k][0] = w.x;
                pr.ix[s][k][1] = w.y;
                pr.ix[s][k][2] = w.z;
                // (RELOAD: these are the first iteration's; mean_field_lean re-reads them from the second on)
                const lean_u3 b = __builtin_amdgcn_raw_buffer_load_b96(src.bary[k], (tid + s * NT) * (D1 * 4), 0, LEAN_AUX(16));
                pr.bary[s][k][0] = __uint_as_float(b.x);
                pr.bary[s][k][1] = __uint_as_float(b.y);
                pr.bary[s][k][2] = __uint_as_float(b.z);
                pr.wn[s][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(src.norm[k], (tid + s * NT) * 4, 0, LEAN_AUX(4)));
            }
        }
        const lean_u2 clw = __builtin_amdgcn_raw_buffer_load_b64(rp, tid * 8, pp.cl_off, LEAN_AUX(1));
        // the LDS tables: every piece is at most 16 bytes per lane (launch_inference_fused checks the plan), all requested before any is stored
        lean_u4 trow[K], tnbr[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            trow[k] = tnbr[k] = lean_u4{0u, 0u, 0u, 0u};
            if (tid * 16 < pp.row_bytes[k]) trow[k] = __builtin_amdgcn_raw_buffer_load_b128(rp, tid * 16, pp.row_off[k], LEAN_AUX(1));
            if (tid * 16 < pp.nbr_bytes[k]) tnbr[k] = __builtin_amdgcn_raw_buffer_load_b128(rp, tid * 16, pp.nbr_off[k], LEAN_AUX(1));
        }
        FL_PSTAMP();                                      // (everything requested)
        if (N <= 0) {                                     // (the point count was requested first and is only needed here)
            if (a.with_map) clear_label_bits<NT>(c, fo, 0, lane_id());
            return;
        }
        if (FL_DBG(8)) {                                  // (fine stamps only: ... landed)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FL_STAMP();
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (tid * 16 < pp.row_bytes[k]) *reinterpret_cast<lean_u4 *>(smem + lay.row[k] + tid * 16) = trow[k];
            if (tid * 16 < pp.nbr_bytes[k]) *reinterpret_cast<lean_u4 *>(smem + lay.nbr[k] + tid * 16) = tnbr[k];
        }
        if (tid < 32) reinterpret_cast<float *>(smem + lay.zero)[tid] = 0.0f;
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                reinterpret_cast<float2 *>(smem + lay.val[k][0])[0] = make_float2(0.f, 0.f);
                reinterpret_cast<float2 *>(smem + lay.val[k][1])[0] = make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) pr.wn[s][k] = wk[k] * pr.wn[s][k];                    // pairwise3d.h:77 (w_*norm_[i])
        FL_PSTAMP();                                      // (this wavefront's tables stored)
        __syncthreads();
        FL_PSTAMP();
        ChainLane cl{clw.x, clw.y};
        start_inference<PPT, K, NT>(pr, N, tid);
        FL_STAMP();
        float alpha[K];
#pragma unroll
        for (int k = 0; k < K; ++k) alpha[k] = a.kd[k].alpha;
        int t = lane_id();
        mean_field_lean<PPT, K, CH, NT, RELOAD, false, true>(smem, lay, V, N, t, pr, cl, alpha, wk, src, a.n_iter, a.relax, a.omr, ins);
        t = lane_id();
        store_results<PPT, K, NT, (LCCRF_LEAN_NT & 8) != 0>(c, fo, N, t, pr, a.with_map);
        FL_STAMP();
        if (kInstr && a.timing && (int)blockIdx.x == a.timing_block && t == a.timing_lane) a.timing[63] = ins.n;
        return;
    }
    if (N <= 0) {
        if (a.with_map) clear_label_bits<NT>(c, fo, 0, lane_id());
        return;
    }


    // per-point records first (the long pole of the prologue), then the lattice tables
    unsigned pk[PPT][K][D1];              // (vertex id + 1) | place in the row << 16
    const float *gbary[K], *gnorm[K];
    const unsigned *gnbr[K];
    float wk[K];
    LeanSrc src;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const KernelDev &kd = a.kd[k];
        gbary[k] = kd.bary + (size_t)f * kd.Epad;
        gnorm[k] = kd.norm + (size_t)f * kd.maxN;
        gnbr[k] = kd.nbr16 + (size_t)f * D1 * kd.Epad;
        wk[k] = kd.w;
        src.nbr[k] = lean_rsrc(gnbr[k], (size_t)D1 * kd.Epad * 4);
        src.bary[k] = lean_rsrc(gbary[k], (size_t)kd.Epad * 4);
        src.norm[k] = lean_rsrc(gnorm[k], (size_t)kd.maxN * 4);
        src.nbr_axis_bytes[k] = kd.Epad * 4;
        src.off_nbr[k] = src.off_bary[k] = src.off_norm[k] = 0;
    }
    src.unary = lean_rsrc(c.unary + (size_t)f * c.maxN * 2, (size_t)c.maxN * 8);
    src.off_unary = 0;
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int ic = min(tid + s * NT, N - 1);
        pr.un[s] = reinterpret_cast<const float2 *>(c.unary)[(size_t)f * c.maxN + ic];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const KernelDev &kd = a.kd[k];
            const size_t e0 = (size_t)f * kd.Epad + (size_t)ic * D1;
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                pk[s][k][j] = kd.pk[e0 + j];
                if (!RELOAD && MODE != 1) pr.bary[s][k][j] = gbary[k][(size_t)ic * D1 + j];
            }
            if (!RELOAD && MODE != 1) pr.wn[s][k] = gnorm[k][ic];
        }
    }
    if (!RELOAD && MODE != 1) {
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) pr.wn[s][k] = wk[k] * pr.wn[s][k];   // pairwise3d.h:77 (w_*norm_[i])
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const KernelDev &kd = a.kd[k];
        unsigned short *row = reinterpret_cast<unsigned short *>(smem + lay.row[k]);
        const int *gr = kd.rowptr + (size_t)f * (kd.Epad + 1);
        for (int v = tid; v <= V[k]; v += NT) row[v] = (unsigned short)gr[v];
        if (lay.nbr[k] >= 0) {
            unsigned *nbr = reinterpret_cast<unsigned *>(smem + lay.nbr[k]);
            for (int idx = tid; idx < D1 * V[k]; idx += NT) {
                const int j = idx >= 2 * V[k] ? 2 : (idx >= V[k] ? 1 : 0);
                nbr[idx] = gnbr[k][(size_t)j * kd.Epad + (idx - j * V[k])];
            }
        }
    }
    if (tid < 32) reinterpret_cast<float *>(smem + lay.zero)[tid] = 0.0f;   // LDS bytes [0, 128): chain_rows_sel reads them by absolute address
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            reinterpret_cast<float2 *>(smem + lay.val[k][0])[0] = make_float2(0.f, 0.f);
            reinterpret_cast<float2 *>(smem + lay.val[k][1])[0] = make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    FL_PSTAMP();

    ChainLane cl{0u, 0u};
    if (CH != 0 && chain_k<CH>(lay, 0)) cl = chain_setup_lean<NT>(smem, lay, V[0], tid);
    FL_PSTAMP();
    if (MODE != 1) start_inference<PPT, K, NT>(pr, N, tid);
    place_products_lean<PPT, K, CH, NT>(smem, lay, N, tid, pk, pr);
    FL_STAMP();
    if constexpr (MODE == 1) {
        // ---- the prepared block of this frame: ix words, chain lanes, LDS tables (LeanPrepPlan) ---------------------------------------
        const LeanPrepPlan pp = lean_prep_plan(lay, K, a.Vcap, NT, PPT);
        unsigned char *pf = a.prep + (size_t)fo * a.prep_stride;
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                unsigned *w = reinterpret_cast<unsigned *>(pf) + ((size_t)(s * K + k) * NT + tid) * 3;
                w[0] = pr.ix[s][k][0];
                w[1] = pr.ix[s][k][1];
                w[2] = pr.ix[s][k][2];
            }
        reinterpret_cast<uint2 *>(pf + pp.cl_off)[tid] = make_uint2(cl.a, cl.b);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            for (int b = tid * 16; b < pp.row_bytes[k]; b += NT * 16)
                *reinterpret_cast<uint4 *>(pf + pp.row_off[k] + b) = *reinterpret_cast<const uint4 *>(smem + lay.row[k] + b);
            for (int b = tid * 16; b < pp.nbr_bytes[k]; b += NT * 16)
                *reinterpret_cast<uint4 *>(pf + pp.nbr_off[k] + b) = *reinterpret_cast<const uint4 *>(smem + lay.nbr[k] + b);
        }
        return;
    }

    float alpha[K];
#pragma unroll
    for (int k = 0; k < K; ++k) alpha[k] = a.kd[k].alpha;
    int t = lane_id();                    // (the loop re-forms it per phase the same way)
    mean_field_lean<PPT, K, CH, NT, RELOAD>(smem, lay, V, N, t, pr, cl, alpha, wk, src, a.n_iter, a.relax, a.omr, ins);
    t = lane_id();

    store_results<PPT, K, NT>(c, fo, N, t, pr, a.with_map);
    FL_STAMP();
    if (kInstr && a.timing && (int)blockIdx.x == a.timing_block && t == a.timing_lane) a.timing[63] = ins.n;
}

bool make_layout(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, FusedLayout *lay, int nt = kNT,
                 size_t lds_limit = kLdsLimit)
{
    if (c.L != 2 || c.K < 1 || c.K > kMaxFusedK) return false;
    const int NA = c.activeN > 0 ? c.activeN : c.maxN;   // size LDS and the points-per-lane variant by the frames' real size
    for (int k = 0; k < c.K; ++k)
        if (kds[k].d != 2 || kds[k].Epad >= 65535) return false;       // u16 row pointers / neighbour ids / slots
    return layout_core(NA, c.K, maxV, maxRow ? maxRow[0] : 0, lay, nt, lds_limit);
}

// Two frames per CU?  Up to 2 points per lane of a 512-lane workgroup, the whole plan in half the CU's LDS, and -- if
// kernel 0 has long rows -- few enough vertices for the chain lanes of four wavefront pairs.
bool small_layout(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, FusedLayout *lay)
{
    static const bool off = ab_env("LCCRF_NO_SMALL_WG") != nullptr;      // A/B switch: same results either way
    const int NA = c.activeN > 0 ? c.activeN : c.maxN;
    if (off || c.F < kSmallMinFrames || NA > 2 * kNTSmall || maxV[0] > kNTSmall) return false;
    const int row0 = maxRow ? maxRow[0] : 0;
    if (row0 >= kChainMinRow && !chain_wanted(NA, maxV[0], row0, kNTSmall)) return false;   // long rows need the chain path: keep 1024 lanes
    FusedLayout L;
    if (!make_layout(c, kds, maxV, maxRow, &L, kNTSmall, kLdsHalf)) return false;
    *lay = L;
    return true;
}


// Two FULL-SIZE frames per CU (fused_lean.h)?  More than two points per lane of the 512-lane small shape, the lean plan in half the CU's LDS.
// (measured and dropped: 384 lanes x 6 points with 168 registers per lane, with and without the re-read -- 2.1e7 iterations/s on C2
// against 3.67e7 for one 1024-lane frame per CU: twelve wavefronts per CU do not keep the LDS pipe busy)
int lean_shape()
{
    static const char *e = ab_env("LCCRF_LEAN_SHAPE");                   // A/B switch: 0 = one 1024-lane frame per CU (same results)
    static const int v = e ? atoi(e) : 2;
    return v;
}

bool lean_layout(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, FusedLayout *lay, int *nt_out)
{
    const int shape = lean_shape();
    if (shape != 2) return false;
    const int nt = kNTSmall, max_ppt = 4;
    const int NA = c.activeN > 0 ? c.activeN : c.maxN;
    // Frames of 513 .. 1024 points take the plan too (two points per lane, everything in registers -- no re-reads): C1 6.47 -> 7.03e7
    // iterations/s against the half-CU form of the 137 KB plan (shared product buffer there as well, but no fused X + P and no
    // overlapped blur schedule).  Up to 512 points both kernels' products fit half a CU with buffers of their OWN: 9.94e7 against 9.05e7
    // on this plan -- those keep small_layout() (LCCRF_LEAN_SMALL, instrumented library, forces the plan for the A/B).
    static const bool lean_small = ab_env("LCCRF_LEAN_SMALL") != nullptr;
    if (c.L != 2 || c.K < 1 || c.K > kMaxFusedK || c.F < kSmallMinFrames || (NA <= kNTSmall && !lean_small) || NA > max_ppt * nt) return false;
    for (int k = 0; k < c.K; ++k)
        if (kds[k].d != 2 || kds[k].Epad >= 65535) return false;
    FusedLayout L;
    if (!layout_lean(NA, c.K, maxV, maxRow ? maxRow[0] : 0, &L, nt, kLdsHalf)) return false;
    *lay = L;
    *nt_out = nt;
    return true;
}

// (Measured and dropped: the same schedule in ONE 1024-lane workgroup per CU for frames beyond 2048 points -- C4, whose products share a
// buffer in fused_loop.h's loop as well: 2.04e7 against 2.42e7 iterations/s.  With nobody else on the CU the re-reads' latency and the
// seven barriers are all exposed; the plan pays through co-residency, not through its schedule alone.)

template <int NT, int PPT, int K, int CH, bool RELOAD, int MODE>
void launch_lean_mode(const CrfDev &c, const FusedArgs &a, hipStream_t s)
{
    auto fn = k_fused_lean<NT, PPT, K, CH, RELOAD, MODE>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
    fn<<<dim3(c.F), dim3(NT), a.lay.total, s>>>(c, a);
}

// mode: 0 the self-contained kernel, 1 prepare, 2 run from the prepared blocks
template <int NT, int PPT, int K, int CH, bool RELOAD>
void launch_lean(const CrfDev &c, const FusedArgs &a, hipStream_t s, int mode)
{
    if (mode == 2) launch_lean_mode<NT, PPT, K, CH, RELOAD, 2>(c, a, s);
    else if (mode == 1) launch_lean_mode<NT, PPT, K, CH, RELOAD, 1>(c, a, s);
    else launch_lean_mode<NT, PPT, K, CH, RELOAD, 0>(c, a, s);
}

template <int NT, int PPT, bool RELOAD>
void launch_lean_ppt(const CrfDev &c, const FusedArgs &a, hipStream_t s, int mode)
{
    if (c.K == 1) {
        if (a.lay.chain0) launch_lean<NT, PPT, 1, 1, RELOAD>(c, a, s, mode);
        else launch_lean<NT, PPT, 1, 0, RELOAD>(c, a, s, mode);
    } else {
        if (a.lay.chain0) launch_lean<NT, PPT, 2, 1, RELOAD>(c, a, s, mode);
        else launch_lean<NT, PPT, 2, 0, RELOAD>(c, a, s, mode);
    }
}

void launch_lean_any(const CrfDev &c, const FusedArgs &a, hipStream_t s, int mode, int NAp)
{
    if (NAp <= 512) launch_lean_ppt<512, 1, false>(c, a, s, mode);
    else if (NAp <= 2 * 512) launch_lean_ppt<512, 2, false>(c, a, s, mode);
    else if (NAp <= 3 * 512) launch_lean_ppt<512, 3, true>(c, a, s, mode);
    else launch_lean_ppt<512, 4, true>(c, a, s, mode);
}

template <int NT, int PPT, int K, int CH, int MODE>
void launch_fused_mode(const CrfDev &c, const FusedArgs &a, hipStream_t s)
{
    auto fn = k_fused<NT, PPT, K, CH, MODE>;
    // per (function, device); cheap enough to repeat and safe with several devices in one process
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kLdsLimit);
    fn<<<dim3(c.F), dim3(NT), a.lay.total, s>>>(c, a);
}

template <int NT, int PPT, int K, int CH>
void launch_fused(const CrfDev &c, const FusedArgs &a, hipStream_t s, int mode)
{
    if (mode == 2) launch_fused_mode<NT, PPT, K, CH, 2>(c, a, s);
    else if (mode == 1) launch_fused_mode<NT, PPT, K, CH, 1>(c, a, s);
    else launch_fused_mode<NT, PPT, K, CH, 0>(c, a, s);
}

template <int NT, int PPT>
void launch_fused_ppt(const CrfDev &c, const FusedArgs &a, hipStream_t s, int mode)
{
    if (c.K == 1) {
        if (a.lay.chain0) launch_fused<NT, PPT, 1, 1>(c, a, s, mode);
        else launch_fused<NT, PPT, 1, 0>(c, a, s, mode);
    } else {
        if (a.lay.chain0) launch_fused<NT, PPT, 2, 1>(c, a, s, mode);
        else launch_fused<NT, PPT, 2, 0>(c, a, s, mode);
    }
}

// the shape launch_inference_fused has chosen, in one of the three modes
void launch_shape(const CrfDev &c, const FusedArgs &a, hipStream_t s, int mode, int NAp, bool small, bool lean)
{
    if (small) {
        if (NAp <= kNTSmall) launch_fused_ppt<kNTSmall, 1>(c, a, s, mode);
        else launch_fused_ppt<kNTSmall, 2>(c, a, s, mode);
    } else if (lean) {
        launch_lean_any(c, a, s, mode, NAp);
    } else {
        switch ((NAp + kNT - 1) / kNT) {
        case 1: launch_fused_ppt<kNT, 1>(c, a, s, mode); break;
        case 2: launch_fused_ppt<kNT, 2>(c, a, s, mode); break;
        case 3: launch_fused_ppt<kNT, 3>(c, a, s, mode); break;
        case 4: launch_fused_ppt<kNT, 4>(c, a, s, mode); break;
        default: break;
        }
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

// the shape decision of launch_inference_fused
struct FusedShape {
    bool ok, small, lean;
    int nt, ppt;
};
static FusedShape choose_shape(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, FusedLayout *lay)
{
    FusedShape sh{false, false, false, kNT, 1};
    int lean_nt = 0;
    const int NAp = c.activeN > 0 ? c.activeN : c.maxN;
    const bool lean_first = NAp <= 2 * kNTSmall && lean_layout(c, kds, maxV, maxRow, lay, &lean_nt);      // (LCCRF_LEAN_SMALL experiment)
    sh.small = !lean_first && small_layout(c, kds, maxV, maxRow, lay);
    sh.lean = lean_first || (!sh.small && lean_layout(c, kds, maxV, maxRow, lay, &lean_nt));
    if (!sh.small && !sh.lean && !make_layout(c, kds, maxV, maxRow, lay)) return sh;
    sh.ok = true;
    sh.nt = (sh.small || sh.lean) ? kNTSmall : kNT;
    sh.ppt = std::max((NAp + sh.nt - 1) / sh.nt, 1);
    return sh;
}

size_t lean_prep_bytes(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow)
{
    FusedLayout lay;
    const FusedShape sh = choose_shape(c, kds, maxV, maxRow, &lay);
    if (!sh.ok || c.F < kPrepMinFrames) return 0;
    const LeanPrepPlan pp = lean_prep_plan(lay, c.K, maxV, sh.nt, sh.ppt);
    return (size_t)pp.total * (size_t)c.F;
}

int launch_inference_fused(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, int n_iter,
                           int with_map, float relax, hipStream_t s, LeanPrep *prep)
{
    FusedArgs a{};
    const FusedShape sh = choose_shape(c, kds, maxV, maxRow, &a.lay);
    if (!sh.ok) return 0;
    const bool small = sh.small, lean = sh.lean;
    static const bool no_chain = ab_env("LCCRF_NO_CHAIN") != nullptr;     // debugging aid: compiler-scheduled S phase
    if (no_chain) a.lay.chain0 = 0;                                        // (the padded plane size is harmless)
    for (int k = 0; k < c.K; ++k) a.kd[k] = kds[k];
    a.n_iter = n_iter;
    a.with_map = with_map;
    a.relax = relax;
    a.omr = 1 - relax;
    static long long *timing_buf = nullptr;
    static const bool want_timing = kInstr && ab_env("LCCRF_FUSED_TIMING") != nullptr;
    if (want_timing && !timing_buf) (void)hipMalloc(&timing_buf, 64 * sizeof(long long));
    a.timing = want_timing ? timing_buf : nullptr;
    a.timing_block = want_timing ? std::max(atoi(ab_env("LCCRF_FUSED_TIMING")) - 1, 0) : 0;
    if (a.timing_block >= c.F) a.timing_block = 0;
    a.timing_lane = (want_timing && ab_env("LCCRF_FUSED_TIMING_LANE")) ? atoi(ab_env("LCCRF_FUSED_TIMING_LANE")) & (kNT - 1) : 0;
    static const int dbg = (kInstr && ab_env("LCCRF_FUSED_DBG")) ? atoi(ab_env("LCCRF_FUSED_DBG")) : 0;
    a.dbg = dbg;
    const int NAp = c.activeN > 0 ? c.activeN : c.maxN;
    // Prepared launch records (fused_lean.h: LeanPrepPlan): what the prologue derives from the lattices and the plan alone.  The FIRST
    // inference behind a build runs the self-contained kernel (a caller with one inference per lattice pays nothing); the second one
    // writes the blocks (MODE 1) and every inference from then on starts from them (MODE 2).
    for (int k = 0; k < c.K; ++k) a.Vcap[k] = maxV[k];
    const LeanPrepPlan pp = lean_prep_plan(a.lay, c.K, a.Vcap, sh.nt, sh.ppt);
    static const bool no_prep = ab_env("LCCRF_NO_LEAN_PREP") != nullptr;    // A/B switches (instrumented library): same results either way
    static const bool prep_now = ab_env("LCCRF_LEAN_PREP_NOW") != nullptr;  //   ... the blocks already in the first inference
    int mode = 0;
    bool pieces_ok = !no_chain;                           // (the run kernels move every table with one (lean) / two 16-byte loads per lane)
    for (int k = 0; k < c.K; ++k) pieces_ok = pieces_ok && pp.row_bytes[k] <= (lean ? 1 : 2) * sh.nt * 16 && pp.nbr_bytes[k] <= (lean ? 1 : 2) * sh.nt * 16;
    if (prep && prep->buf && !no_prep && pieces_ok && c.F >= kPrepMinFrames && (size_t)pp.total * (size_t)c.F <= prep->bytes) {
        // what the blocks depend on besides the lattices themselves: the plan, the shape, the frame count
        unsigned long long key = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t n) { for (size_t i = 0; i < n; ++i) key = (key ^ static_cast<const unsigned char *>(p)[i]) * 1099511628211ull; };
        mix(&a.lay, sizeof(a.lay));
        mix(a.Vcap, sizeof(int) * c.K);
        const int shape[6] = {c.F, c.K, sh.ppt, NAp, sh.nt, lean ? 2 : small ? 1 : 0};
        mix(shape, sizeof(shape));
        a.prep = prep->buf;
        a.prep_stride = pp.total;
        if (prep->valid && prep->key == key) {
            mode = 2;
        } else if (prep->seen_key == key || prep_now) {      // the second inference on these lattices: write the blocks, then run from them
            if (prep->ev0) (void)hipEventRecord(prep->ev0, s);
            launch_shape(c, a, s, 1, NAp, small, lean);
            if (prep->ev1) (void)hipEventRecord(prep->ev1, s);
            prep->timed = prep->ev0 && prep->ev1;
            prep->valid = true;
            prep->key = key;
            ++prep->runs;
            mode = 2;
        } else {
            prep->seen_key = key;                            // the first one
        }
    }
    launch_shape(c, a, s, mode, NAp, small, lean);
    if (a.timing) {                       // debug only: synchronous read-back of workgroup 0's phase stamps
        long long h[64];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, a.timing, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[lccrf fused timing] %lld stamps, deltas (shader clocks):", h[63]);
        for (int i = 1; i < h[63] && i < 63; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
        fprintf(stderr, "\n");
    }
    return small ? (kNTSmall | 2 << 16) : lean ? (kNTSmall | 2 << 16) : (kNT | 1 << 16);
}

}  // namespace lccrf
