// engine.h -- internal structures shared by the HIP kernels and the C-ABI layer.
//
// Vocabulary follows the reference (Thirdparty/DenseCRF): a CRF over N *points*
// (keypoints) with L *labels* and K pairwise *kernels*; every kernel owns a
// permutohedral *lattice* with V *vertices*; a point touches d+1 vertices, one per
// *remainder*; the pair (point, remainder) is an *entry* e = point*(d+1)+remainder.
// A *frame* is one independent CRF; a batch holds F frames with a common stride.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include "../../include/lccrf.h"

#ifndef LCCRF_INSTRUMENT
#define LCCRF_INSTRUMENT 0
#endif

namespace lccrf {

// A/B, cross-check and fault-injection switches -- environment variables that choose between code paths with identical results --
// exist in the INSTRUMENTED library only (`make INSTRUMENT=1` -> liblccrf_hip_instr.so, which tests and scripts load through
// LCCRF_LIB).  The release library reads no environment variable: a tracker inherits its environment, and none of these should be
// able to change its speed.  What a deployment may legitimately choose is an option of the API (lccrf_set_option).
#if LCCRF_INSTRUMENT
inline const char *ab_env(const char *name) { return getenv(name); }
#else
inline const char *ab_env(const char *) { return nullptr; }
#endif

constexpr int kMaxD = LCCRF_MAX_DIMS;
constexpr int kEmpty = -1;

constexpr int kNdistAxes = 4;
constexpr int kLongRowMin = 512;    // generic (L-label) splat: rows beyond this many entries go to a workgroup of their own (KernelDev::longrow)
constexpr int kLongRowCap = 4095;   // ... as long as there are no more than this many of them per frame
constexpr int kLongRowListMinPoints = 4096;   // ... in engines for frames beyond the one-workgroup kernels' range (the lists are 16 KB per frame)
constexpr int kNbrcBlock = 64;      // vertices per base of the compact neighbour table (KernelDev::nbrc): one wavefront
constexpr int kNbrcMinFrames = 3;   // ... which is built and read with 3 to 5 frames in flight (one frame: the passes go two per launch off the
constexpr int kNbrcMaxFrames = 5;   //   two-hop table; two frames: +-0) (C5, per frame and iteration: 3 / 4 / 5 frames
                                    //   28.4 -> 26.6 / 30.0 -> 24.3 / 29.1 -> 23.7 us; 6 / 7 / 8 / 16 frames, with the table loaded
                                    //   non-temporally: +-0 / +-0 / -0.3 / -1 % -- there the pass is bound by the gathers' lines, not bytes)

// Device view of one pairwise kernel across all frames of a batch.
// All per-frame arrays are strided by the capacities below (not by the actual N / V).
struct KernelDev {
    int d, D1;            // feature dims, d+1
    int maxN;             // per-frame stride of point arrays (features, norm)
    int maxNpad;          // maxN rounded up to a multiple of 4 (phantom points, quirk Q1)
    int Epad;             // entries per frame = maxNpad * D1 (also the vertex capacity)
    int cap;              // hash capacity per frame (power of two >= 2*Epad)
    int vstride;          // per-frame stride of val0/val1 = (Epad+2)*L of the owning CRF
    int vbase;            // = 2L of the owning CRF: vertex v, label l of a width-W pass sits at
                          //   vbase + v*W + l; [0,vbase) stays zero and [vbase-W, vbase) serves as vertex -1
    float w;              // kernel weight (PottsPotential3D::w_)
    float alpha;          // 1/(1+2^-d)                     permutohedral_cpu.h:681
    float inv_dp1;        // 1.0f/(d+1)                     permutohedral_cpu.h:249
    float scale[kMaxD];   // elevation scale factors        permutohedral_cpu.h:282-285

    const float *feat;    // [F][maxN][d]      input features (already / stdev)
    int16_t *rem0;        // [F][maxNpad][d]   nearest remainder-0 lattice point (first d coords)
    uint8_t *rank;        // [F][maxNpad][d]   rank of each coordinate
    float *bary;          // [F][Epad]         barycentric weight of every entry
    int *offset;          // [F][Epad]         vertex id of every entry (reference offset_)
    int *slot;            // [F][cap]          hash slots: lowest entry id with that key, or -1
    int *slot_of;         // [F][Epad]         slot index each entry landed in
    int *flag;            // [F][Epad+1]       scratch: first-occurrence flags / row counts / unsorted rows
    int *prefix;          // [F][Epad+1]       exclusive scan of flag == dense vertex id of a first entry
    int *rep;             // [F][Epad]         vertex id -> representative (first) entry
    int *V;               // [F]               number of vertices (reference M_)
    int *rowmax;          // [F]               longest CSR row (splat contributions of one vertex)
    int *nbr;             // [F][D1][Epad][2]  blur neighbours {n1,n2} per (axis, vertex), -1 absent
    int *nbr2;            // [F][D1/2][Epad][8] or null: two-hop table of the pass pairs (2p, 2p+1) for single-frame engines (one
                          //   launch per TWO blur passes, stream_engine.hip: k_blur2x2t): {v1, v2, a, b, a1, a2, b1, b2} with
                          //   (v1, v2) = axis-2p neighbours of v, (a, b) = its axis-(2p+1) neighbours, (a1, a2) / (b1, b2) = the
                          //   axis-2p neighbours of a / b; -1 absent
    int nbr2_ok;          // the streaming build filled nbr2 for the lattices now in HBM ...
    int nbr2_first;       // ... for the pass pairs (first + 2p, first + 2p + 1): first = 1 when the splat takes pass 0 along (fast0_ok)
    // compact form of nbr for the blur passes of the streaming engine (sorted build only): with the vertices in row-major order the
    // neighbours of consecutive vertices are (nearly) consecutive, so 16-bit offsets from a base per block of kNbrcBlock vertices
    // say the same as the 32-bit ids in half the bytes (a blur pass moves 20 instead of 24 bytes per vertex)
    unsigned short *nbrc; // [F][D1][Epad][2] or null: {n1, n2} - base of the vertex's block, 0xffff = absent
    int *nbrc_base;       // [F][D1][Epad / kNbrcBlock + 1][2]: smallest n1 / n2 of the block
    int *tbl_bad;         // two pinned host words raised by the sorted build: [0] a block's neighbours span more than 16 bits (nbrc is
                          //   unusable), [1] an axis-0 neighbour is not the next / previous id (fast0_ok must not be relied on)
    int nbrc_ok;          // nbrc describes the lattices now in HBM
    int fast0_ok;         // sorted build: axis 0 is the fastest coordinate of the row-major code, so a vertex's axis-0 neighbours are
                          //   v - 1 and v + 1 or absent, and the first blur pass can ride in the splat (k_splat2<true>) ...
    uint8_t *fastn;       // [F][Epad] or null: ... which reads this instead of the table: 1 = vertex v + 1 is v's axis-0 neighbour
    // ... and the next axes' neighbours are near ids too (C5: axis 1 within ~10, axis 2 within ~100 ids; rows and planes of the sparse
    // 6-D lattice hold a few vertices each): the sorted build measures the largest id distance per axis, and when the distances of
    // axes 1 (and 2) fit a workgroup's halo those passes ride in the splat as well, on an overlapped window in LDS (k_splat2w)
    int8_t *nearoff;      // [F][2][Epad][2] or null: axes 1 and 2, {n1 - v, n2 - v} as signed bytes, 0 = absent (valid when ndist <= 127:
                          //   the window passes read 2 instead of 8 table bytes per vertex)
    // ... and everything the window splat wants to know about a vertex in ONE 16-byte record (round 5; k_pack_srec after the sorted build):
    //   .x row start s   .y first entry's point | min(row length, 127) << 24 | fastn << 31   .z first entry's weight (bits)
    //   .w the four nearoff bytes {axis 1: n1 - v, n2 - v, axis 2: n1 - v, n2 - v}
    // one load per vertex instead of seven (two row pointers, fastn, two offset pairs, the first entry's point and weight): the pass
    // is bound by vector-memory instructions, not bytes.  Frames of < 2^24 points only (the point id's 24 bits).
    uint4 *srec;          // [F][Epad] or null
    int srec_ok;          // the records describe the lattices now in HBM
    int *ndist;           // [kNdistAxes] or null: largest |neighbour id - id| along axes 0 .. kNdistAxes-1 over all frames (device)
    int splat_passes;     // blur passes the splat takes along for the lattices now in HBM: 0 (none), 1 (k_splat2<true>), 2 or 3 (k_splat2w)
    int splat_halo;       // ... the halo that takes on each side of a window (1 + distance of axis 1 [+ distance of axis 2])
    int splat_block;      // ... and the window (256 / 512 / 1024 vertices: 256 lanes x 1 / 2 / 4)
    int *rowptr;          // [F][Epad+1]       CSR: vertex -> range of splat contributions
    int *longcnt;         // [F]               number of rows longer than kLongRowMin; beyond kLongRowCap = "not listed": every row in line
    int *longrow;         // [F][kLongRowCap]  ... and their vertices (any order)
    int longrow_ok;       // the list describes the lattices now in HBM (the streaming build writes it; k_build_small does not)
    int long_mode;        // two-label splat: this kernel has listed rows or rows of more than ~4 entries on average (a coarse kernel over many
                          //   points): k_splat2l (1) or k_splat2v (2: ~16 entries per row and more) + k_splat_long instead of the fused splat (set by
                          //   the host once the sizes are known)
    int *csr_pt;          // [F][Epad]         contributing point, ascending within a row
    float *csr_w;         // [F][Epad]         its barycentric weight
    int *csr_pos;         // [F][Epad]         entry -> its position in csr_pt/csr_w (inverse of the row ordering)
    // compact copies for the fused engine's prologue (frames with Epad < 65535 only; undefined otherwise)
    unsigned *pk;         // [F][Epad]         (offset + 1) | csr_pos << 16 of every real entry
    unsigned *nbr16;      // [F][D1][Epad]     (n1 + 1) | (n2 + 1) << 16 per (axis, vertex), 0 = absent
    int *V_host, *rowmax_host;   // [F] pinned host mirrors of V / rowmax written by the fused build (or null)
    float *norm;          // [F][maxN]         1/(K*1 + 1e-20)  (PottsPotential3D::norm_)
    float *val0, *val1;   // [F][vstride]      lattice values (see vbase)
    // Locality mode (large frames on the streaming engine): the points of a frame are processed in an internal order --
    // position i holds original point perm[i] -- chosen so that points of neighbouring lattice cells sit next to each
    // other; every point-indexed array above (rem0, rank, bary, offset, norm, csr_pt ...) is in that order.  Results do
    // not depend on it: CSR rows stay ordered by ORIGINAL point index (quirk Q6), vertices are matched by key.  Null = off.
    const int *perm;      // [F][maxNpad]      position -> original point (identity on the phantom lanes)
    const int *iperm;     // [F][maxNpad]      original point -> position
    // ... and the VERTICES are numbered along the lattice's own axes (round 4; stream_engine.hip: the sorted build): in the basis of
    // the blur directions the lattice is the integer grid Z^d, and ids in row-major order of those coordinates make the two
    // neighbours a blur gather wants consecutive ids of another row.
    int vorder;           // the lattice is built by sorting the entries on the row-major code of their vertex (no hash table)
};

// Device view of the CRF state of a batch.
struct CrfDev {
    int F, maxN, L, K;    // maxN = per-frame STRIDE (capacity) of every point array
    int activeN;          // largest n_points of the batch when the host knows it, else maxN
    const int *n_points;  // [F]
    float *unary;         // [F][maxN][L]
    float *Q;             // [F][maxN][L]   current_
    float *next;          // [F][maxN][L]   next_
    int16_t *map;         // [F][maxN]
    unsigned long long *map_bits;   // [F][bits_stride] or null: the MAP labels of a binary CRF, one bit per point (bit i%64 of
    int bits_stride;                //   word i/64) -- the wire format of the multi-GPU label gather; bits_stride = ceil(maxN/64)
    const int *perm;      // [F][perm_stride] or null: locality mode (see KernelDev::perm); unary / Q / next of THIS view are then
    int perm_stride;      //   in permuted order and the caller un-permutes Q on the way out
};

// scratch of the point sort (locality mode), owned by the engine
struct SortScratch {
    int bits;             // bucket bits of the counting sort (8..16)
    int rm_points;        // order the points row-major over their cells in the lattice's own basis (with the sorted build) instead of along a Z-order curve
    int *cells;           // [F][maxNpad][kMaxD] lattice cell of every point (int32); later the unordered buckets
    int *partial;         // [F][ceil(maxNpad/256)][2*kMaxD] per-workgroup min / max of the cells
    int *plan;            // [F][3*kMaxD]        per dimension: lowest cell, span, code bits
    int *code;            // [F][maxNpad]        Z-order bucket of every point
    int *hist;            // [F][2^bits + 1]     bucket counts
    int *start;           // [F][2^bits + 1]     exclusive scan of hist, advanced to the bucket ends by the scatter
    int *tiles;           // [F][ceil((2^bits+1)/4096)] scan scratch
    int *perm, *iperm;    // [F][maxNpad]        the result
    // vertex order (per kernel, one after the other on the engine's stream)
    int vcap;                   // entries per frame the arrays below are sized for (>= Epad of every kernel)
    int vbits;                  // bucket bits of the entry sort (<= 21): ~4 buckets per entry
    unsigned long long *vcode;  // [F][vcap]           row-major code of every ENTRY's vertex
    unsigned long long *vkey;   // [F][vcap]           ... of every vertex, by id (ids are in code order bucket by bucket)
    int *vph;                   // [F][64]             positions of the phantom points' entries in the sorted order: count, then the list
    int *vbad;                  // pinned host word: raised when a frame's code space overflows 62 bits (the host then rebuilds with the hash)
    int *vhist, *vstart;        // [F][2^vbits + 1]    bucket counts, their scan (advanced to the bucket ends by the scatter)
    int *vtiles;                // [F][ceil((2^vbits+1)/4096)] scan scratch
    int *vpartial;              // [F][ceil(maxNpad/256)][2*kMaxD] per-workgroup min / max of the vertices' grid coordinates (k_points)
    long long *vplan;           // [F][2*kMaxD+3]      per dimension: lowest coordinate, stride; then bucket scale, usable, range
};

// ---- streaming engine (any size; every array in HBM) ---------------------------------
// locality mode: Z-order buckets of the points' lattice cells under kernel kd -> ss.perm / ss.iperm
void launch_sort_points(const KernelDev &kd, const CrfDev &c, const SortScratch &ss, hipStream_t s);
// dst[f][i][0..width) = src[f][perm[i]][0..width) (gather = 1) or dst[f][perm[i]][..] = src[f][i][..] (gather = 0), i < n_points[f]
void launch_permute_rows(const CrfDev &c, float *dst, const float *src, int width, int gather, hipStream_t s);
void launch_build_kernel(const KernelDev &kd, const CrfDev &c, int maxV_hint, hipStream_t s, const SortScratch *vsort = nullptr);
void launch_norm(const KernelDev &kd, const CrfDev &c, int maxV, hipStream_t s);
// unary[i][:] from labels (densecrf3d.h:116-129); the 2L+1 energies {u, n[L], p[L]} are passed by value (no table
// upload; `label` may be device or pinned host memory)
struct UnaryTable { float v[2 * LCCRF_MAX_LABELS + 1]; };
void launch_unary_from_label_tbl(const CrfDev &c, const int16_t *label, const UnaryTable &tbl, hipStream_t s);
void launch_start(const CrfDev &c, hipStream_t s);
void launch_step_stream(const CrfDev &c, const KernelDev *kds, const int *maxV, float relax,
                        hipStream_t s);
void launch_map(const CrfDev &c, hipStream_t s);
void launch_map_of(const CrfDev &c, const float *prob, int16_t *map, hipStream_t s);
void launch_exp_and_normalize(const CrfDev &c, const float *in, float *out, float scale, float relax, hipStream_t s);
void launch_step_init(const CrfDev &c, float *out, hipStream_t s);
void launch_filter(const KernelDev &kd, const CrfDev &c, int maxV, const float *in, float *out, int accumulate, hipStream_t s);
hipError_t time_blur_pass(const KernelDev &kd, int F, int maxV, int L, int reps, hipStream_t s, float *ms_per_launch);
// out[f] = clamp(in[f], 0, maxN); *bad (pinned host memory) is set to 1 if anything had to be clamped
void launch_validate_npoints(const int *in, int *out, int F, int maxN, int *bad, hipStream_t s);

// ---- fused build (SLAM sizes; one workgroup per (frame, kernel), hash table in LDS) ------
bool build_small_supported(const KernelDev *kds, int n, int max_points);
void launch_build_small(const KernelDev *kds, int n, int max_points, const CrfDev &c, hipStream_t s);

// ---- fused engine (SLAM sizes; one workgroup per frame, lattice values in LDS) --------
bool fused_supported(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow,
                     size_t *lds_bytes);
// returns the shape it launched: lanes per workgroup (= per frame) | workgroups per CU << 16 (0: nothing launched)
// prep (optional): the batch's prepared launch records of the two-frames-per-CU kernel (fused_lean.h: LeanPrepPlan) -- `buf` of
// `bytes` bytes owned by the caller (lean_prep_bytes(): what the current batch needs, 0 when the plan does not apply); the call
// fills them when !valid (the caller clears `valid` whenever a lattice changes) and records ev0 / ev1 around that launch.
struct LeanPrep {
    unsigned char *buf = nullptr;
    size_t bytes = 0;
    bool valid = false, timed = false;
    unsigned long long key = 0, seen_key = 0;   // the plan / shape the blocks were written for; ... the last inference ran with (blocks are written by the second)
    unsigned runs = 0;                    // how many times the blocks were (re)written: tests
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};
size_t lean_prep_bytes(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow);
int launch_inference_fused(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow,
                           int n_iter, int with_map, float relax, hipStream_t s, LeanPrep *prep = nullptr);

// ---- frame engine (SLAM sizes; lattice build + normalisation + inference of a frame in ONE launch) ---------
bool frame_supported(const CrfDev &c, const KernelDev *kds);
// label != nullptr (L = 2): the unary energies are derived in the kernel from the labels and the 5 table entries
// {u, n0, n1, p0, p1}; otherwise read from c.unary.  `status` (pinned host word, zeroed by the caller) reads 1
// afterwards if some frame did not fit the kernel's LDS plan; `frame_status` (device [F], or null) then holds 1 for
// exactly those frames and the caller runs them -- and only them -- on the two-kernel path.
// allow_small: frames of up to 1024 points may run as 512-lane workgroups in half the CU's LDS (two frames per CU);
// lean_rec (device memory of frame_lean_rec_bytes(F), or null): so may batches of full-size two-kernel frames (1025 .. 2048 points;
// frame_lean.hip keeps the per-point records its loop re-reads in that area).
// Returns the shape it launched: 0 one frame per CU, 1 the small shape, 2 the lean one (a caller that sees many frames of a
// half-CU shape flagged turns that shape off).
// dual (device memory of frame_dual_bytes(F), zeroed once) + a launch-unique dual_epoch != 0: two-kernel frames are given
// two workgroups each -- one per lattice build (single frames: the other 255 CUs are idle anyway).
int launch_frame(const CrfDev &c, const KernelDev *kds, int n_iter, int with_map, float relax, int *status, int *frame_status,
                 const int16_t *label, const float *tbl5, hipStream_t s, bool allow_small = true, unsigned *dual = nullptr,
                 unsigned dual_epoch = 0, unsigned *done = nullptr, unsigned done_epoch = 0, unsigned char *lean_rec = nullptr);
bool frame_lean_wanted(const CrfDev &c);              // would launch_frame take the lean shape for this batch, given the area?
size_t frame_lean_rec_bytes(int frames);
size_t frame_dual_bytes(int frames);
// rows of `bytes` bytes each between a frame-strided array and a compact one: dst[i] = src[list[i]] (gather = 1) or
// dst[list[i]] = src[i] (gather = 0); strides in bytes, everything 4-byte aligned
void launch_copy_frames(void *dst, size_t dst_stride, const void *src, size_t src_stride, const int *list, int n_list,
                        size_t bytes, int gather, hipStream_t s);

// ---- unary builder (the step before the CRF) ------------------------
hipError_t run_unary_build(int device_id, int n_points, const float *Xw, const int32_t *obs_ptr, const int32_t *obs_kf,
                           const double *obs_kp, int n_kf, const float *kf_pose, const float *kf_intr,
                           const float *kf_bounds, const double *match_prob, const lccrf_crf_params *params,
                           float *observs_out, float *error_out, float *depth_out, int16_t *label_out);

// ---- BfMatch (src/Tracking.cc:1747-1766), csrc/bf_match.hip -------------------------------
hipError_t run_bf_match(int device_id, int n_query, const uint8_t *desc_query, int n_train, const uint8_t *desc_train,
                        double ratio, int32_t *train_of_query_out, int32_t *n_matches_out);

// ---- Optimizer::PoseOptimization (src/Optimizer.cc:239-450), csrc/pose_opt.hip; every pointer device-accessible ----
hipError_t launch_pose_optimization(int F, int maxN, const int *n_points, const float *Xw, const float *kp, const float *ur,
                                    const float *is2, const uint8_t *valid, const int16_t *label, const float *K4, float bf,
                                    const float *Tcw_in, float *Tcw_out, uint8_t *outlier, int *n_inliers, int *n_initial,
                                    hipStream_t s, const int *n_crf = nullptr);

}  // namespace lccrf
