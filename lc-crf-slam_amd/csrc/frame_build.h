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


// frame_lean.hip: frames of 1025 .. 2048 points, two-kernel SLAM configuration, 512 lanes and half the CU's LDS per frame
// (NA = the batch's largest frame).  Frames that do not fit flag themselves as in k_frame.
bool frame_lean_plausible(int NA, int K, int F);
void launch_frame_lean(const CrfDev &c, const FrameArgs &a, int NA, hipStream_t s);

}  // namespace fb
}  // namespace lccrf
