// pose_opt.hip -- Optimizer::PoseOptimization (reference src/Optimizer.cc:239-450), the step right after the CRF
// (src/Tracking.cc:1002): motion-only bundle adjustment of the current frame on the matches the CRF left standing.
// One 256-lane workgroup per frame (fp64 issue-bound: 512 lanes measured no faster); the CRF's MAP labels are read where the CRF kernel left them (device memory):
// a keypoint labelled 0 (moving) is what Tracking::DynamicDetectionWithCRF nulls at Tracking.cc:1945-1955, so it
// contributes no edge.
//
// PARITY UNPINNED.  The reference runs this on g2o (Thirdparty/g2o; needs Eigen 3, absent in this image, and the
// reference holds no test or fixture for it).  What is restated, in double precision and operation by operation:
//   edges          EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose: error, chi2, analytic Jacobians
//                  (types_six_dof_expmap.h:153-196, .cpp:266-364; the stereo projection's float 1/z included)
//   robust kernel  RobustKernelHuber and the first-order robustified quadratic form
//                  (robust_kernel_impl.cpp:78-91, base_unary_edge.hpp:43-72), dropped after the third round
//   solver         OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:33-150): lambda
//                  = 1e-5 max diag(H), gain ratio, up to 10 trials per iteration, the (iniChi - chi)*1e3 < iniChi stop
//   manifold       SE3Quat::exp, operator*, map (se3quat.h:104-110,214-256), Eigen's quaternion <-> matrix forms
//   schedule       4 rounds x 10 iterations from the SAME initial pose, chi2 re-classification with the stale-error
//                  quirk of Optimizer.cc:385-390, the < 10 edges exit (Optimizer.cc:366-440)
// Not reproduced bit for bit: Eigen's pivoted LDLT of the 6x6 system (plain LDL^T here) and the order in which the
// edges' contributions are summed (a fixed tree over lanes here, edge order in g2o).  Every lane carries the pose and
// solves the 6x6 system redundantly from the same reduced sums, so no broadcast is needed.  No FMA contraction (the
// reference builds without -march=native): the arithmetic of an edge is the restatement's, operation for operation.
//
// Shape of the kernel (the job is latency: ~58 sweeps over the edges, each followed by a 28-value reduction and a
// serial 6x6 solve; single frame 2.1 ms -> 0.45 ms with these):
//   * the frame's edges are staged in LDS once (28 B + a state word each, up to 4096 keypoints) and each round's
//     level-0 edges are compacted into an index list, so a sweep is ceil(active / 256) full-width iterations;
//   * the error pass of an LM trial and the linearisation of the next iteration happen at the same pose: one sweep
//     computes errors, Jacobians and the normal equations, and a rejected trial's system is dropped;
//   * nothing is indexed dynamically (no scratch memory): the 21 + 6 accumulators stay in registers, the Jacobian's
//     structural zeros are skipped at compile time;
//   * the 28 sums are reduced by a transposing butterfly on the VALU (v_permlane32/16_swap + DPP), 32 exchanges per
//     wavefront instead of 168 shuffles through the LDS crossbar.
#include "engine.h"

#include <cfloat>
#include <cstdio>
#include <cstdlib>

namespace lccrf {

namespace {

#ifndef LCCRF_INSTRUMENT
#define LCCRF_INSTRUMENT 0
#endif
constexpr bool kInstr = LCCRF_INSTRUMENT != 0;           // `make INSTRUMENT=1`: LCCRF_POSE_PROF prints phase clocks of frame 0

constexpr int kPT = 256;
constexpr int kStageMax = 4096;                          // frames of up to this many keypoints keep their edges in LDS
constexpr size_t kPoseLdsMax = 160 * 1024 - 2048;         // dynamic LDS: 38 B per staged keypoint; 2 KB left for the static reduction buffers

struct PQ { double w, x, y, z; };

struct PoseArgs {
    int maxN;                             // per-frame stride of every per-point array
    const int *n_points;                  // [F]
    const float *Xw, *kp, *ur, *is2;      // [F][maxN][3], [F][maxN][2], [F][maxN], [F][maxN]
    const uint8_t *valid;                 // [F][maxN] mvpMapPoints[i] != NULL, or null (all valid)
    const int16_t *label;                 // [F][maxN] CRF labels (0 = moving: no edge), or null
    const int *n_crf;                     // [F] or null: labels gate points [0, n_crf[f]) only; the points behind them are edges of map
                                          //   points the CRF never saw (observs == 0, Tracking.cc:1857-1859) and count as static
    float fx, fy, cx, cy, bf;
    const float *Tcw_in;                  // [F][16]
    float *Tcw_out;                       // [F][16]
    uint8_t *outlier;                     // [F][maxN] mvbOutlier
    int *n_inliers, *n_initial;           // [F]
    long long *prof;                      // instrumented build only
};

__device__ __forceinline__ void pq_normalize(PQ &q)
{
    if (q.w < 0) { q.w = -q.w; q.x = -q.x; q.y = -q.y; q.z = -q.z; }
    const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.w /= n; q.x /= n; q.y /= n; q.z /= n;
}

__device__ __forceinline__ PQ pq_from_matrix(const double (&m)[3][3])
{
    PQ q;
    double t = m[0][0] + m[1][1] + m[2][2];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m[2][1] - m[1][2]) * t;
        q.y = (m[0][2] - m[2][0]) * t;
        q.z = (m[1][0] - m[0][1]) * t;
    } else {
        // Eigen's branch on the largest diagonal element, the three cases written out (no dynamic register indexing)
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > (i ? m[1][1] : m[0][0])) i = 2;
        if (i == 0) {
            t = sqrt(m[0][0] - m[1][1] - m[2][2] + 1.0);
            q.x = 0.5 * t; t = 0.5 / t;
            q.w = (m[2][1] - m[1][2]) * t; q.y = (m[1][0] + m[0][1]) * t; q.z = (m[2][0] + m[0][2]) * t;
        } else if (i == 1) {
            t = sqrt(m[1][1] - m[2][2] - m[0][0] + 1.0);
            q.y = 0.5 * t; t = 0.5 / t;
            q.w = (m[0][2] - m[2][0]) * t; q.z = (m[2][1] + m[1][2]) * t; q.x = (m[0][1] + m[1][0]) * t;
        } else {
            t = sqrt(m[2][2] - m[0][0] - m[1][1] + 1.0);
            q.z = 0.5 * t; t = 0.5 / t;
            q.w = (m[1][0] - m[0][1]) * t; q.x = (m[0][2] + m[2][0]) * t; q.y = (m[1][2] + m[2][1]) * t;
        }
    }
    return q;
}

__device__ __forceinline__ void pq_rotate(const PQ &q, const double (&v)[3], double (&out)[3])
{
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}

__device__ __forceinline__ PQ pq_mul(const PQ &a, const PQ &b)
{
    PQ r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

// estimate <- SE3Quat::exp(update) * estimate      (VertexSE3Expmap::oplusImpl)
__device__ void pose_oplus(const double (&upd)[6], PQ &q, double (&t)[3])
{
    const double om[3] = {upd[0], upd[1], upd[2]}, up[3] = {upd[3], upd[4], upd[5]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[3][3] = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}};
    double O2[3][3], R[3][3], V[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[i][j] = O[i][0] * O[0][j] + O[i][1] * O[1][j] + O[i][2] * O[2][j];
    if (theta < 0.00001) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { R[i][j] = ((i == j) ? 1.0 : 0.0) + O[i][j] + O2[i][j]; V[i][j] = R[i][j]; }
    } else {
        const double sn = sin(theta), a = sn / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sn) / (theta * theta * theta);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                const double I = (i == j) ? 1.0 : 0.0;
                R[i][j] = I + a * O[i][j] + b * O2[i][j];
                V[i][j] = I + b * O[i][j] + c * O2[i][j];
            }
    }
    PQ dq = pq_from_matrix(R);
    pq_normalize(dq);
    const double dt[3] = {V[0][0] * up[0] + V[0][1] * up[1] + V[0][2] * up[2], V[1][0] * up[0] + V[1][1] * up[1] + V[1][2] * up[2],
                          V[2][0] * up[0] + V[2][1] * up[1] + V[2][2] * up[2]};
    double rt[3];
    pq_rotate(dq, t, rt);
    t[0] = dt[0] + rt[0]; t[1] = dt[1] + rt[1]; t[2] = dt[2] + rt[2];
    q = pq_mul(dq, q);
    pq_normalize(q);
}

struct Edge { float X[3], u, v, ur, is2; int state; };    // state: bit 0 = has an edge (valid, not labelled moving), bit 1 = level 1 (outlier)

__device__ __forceinline__ double edge_error(const PoseArgs &a, const Edge &in, const PQ &q, const double (&t)[3], double (&e)[3],
                                             double (&pc)[3])
{
    const double X[3] = {in.X[0], in.X[1], in.X[2]};
    pq_rotate(q, X, pc);
    pc[0] += t[0]; pc[1] += t[1]; pc[2] += t[2];
    const double w = in.is2, fx = a.fx, fy = a.fy, cx = a.cx, cy = a.cy, bf = a.bf;
    if (in.ur < 0) {
        e[0] = (double)in.u - ((pc[0] / pc[2]) * fx + cx);
        e[1] = (double)in.v - ((pc[1] / pc[2]) * fy + cy);
        e[2] = 0.0;
        return e[0] * (w * e[0]) + e[1] * (w * e[1]);
    }
    const float invz = 1.0f / (float)pc[2];
    const double u = pc[0] * invz * fx + cx;
    e[0] = (double)in.u - u;
    e[1] = (double)in.v - (pc[1] * invz * fy + cy);
    e[2] = (double)in.ur - (u - bf * invz);
    return e[0] * (w * e[0]) + e[1] * (w * e[1]) + e[2] * (w * e[2]);
}

__device__ __forceinline__ void huber(double e, double delta, double (&rho)[3])
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double s = sqrt(e); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; rho[2] = -0.5 * rho[1] / e; }
}

__device__ __forceinline__ constexpr bool jac_zero(int d, int c) { return (d == 1) ? c == 3 : c == 4; }

__device__ __forceinline__ int edge_jacobian(const PoseArgs &a, const Edge &in, const double (&pc)[3], double (&J)[3][6])
{
    const double x = pc[0], y = pc[1], invz = 1.0 / pc[2], invz_2 = invz * invz, fx = a.fx, fy = a.fy, bf = a.bf;
    J[0][0] = x * y * invz_2 * fx;          J[0][1] = -(1 + (x * x * invz_2)) * fx; J[0][2] = y * invz * fx;
    J[0][3] = -invz * fx;                   J[0][4] = 0;                            J[0][5] = x * invz_2 * fx;
    J[1][0] = (1 + y * y * invz_2) * fy;    J[1][1] = -x * y * invz_2 * fy;         J[1][2] = -x * invz * fy;
    J[1][3] = 0;                            J[1][4] = -invz * fy;                   J[1][5] = y * invz_2 * fy;
    J[2][0] = J[0][0] - bf * y * invz_2;    J[2][1] = J[0][1] + bf * x * invz_2;    J[2][2] = J[0][2];
    J[2][3] = J[0][3];                      J[2][4] = 0;                            J[2][5] = J[0][5] - bf * invz_2;
    return in.ur < 0 ? 2 : 3;
}

// (H + lambda I) x = b for a symmetric 6x6 H, LDL^T; s[0..20] = upper triangle of H row by row, s[21..26] = b.
// false if not positive definite
__device__ bool solve6(const double (&s28)[28], double lambda, double (&x)[6])
{
    double H[6][6], L[6][6], D[6];
    int p = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++, p++) H[i][j] = H[j][i] = s28[p];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = H[j][j] + lambda;
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
        if (!(d > 0)) return false;
        D[j] = d;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double s = H[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k] * D[k];
            L[i][j] = s / d;
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double s = s28[21 + i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
        y[i] = s;
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double s = y[i] / D[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s -= L[k][i] * x[k];
        x[i] = s;
    }
    return true;
}

constexpr int kWaves = kPT / 64;
constexpr int kRedDoubles = kWaves * 28 + 28;

// Sum of K doubles per lane over the workgroup, identical in every lane afterwards (fixed tree: xor-shuffles inside a
// wavefront, then the wavefront partials in index order).  For the small sums (counts, one chi2).
template <int K>
__device__ void block_sum(double (&v)[K], double *red /* [kWaves][K] */)
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        v[k] = s;
    }
    __syncthreads();                                      // the previous reduction's readers are done
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < K; ++k) red[(threadIdx.x >> 6) * K + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = red[k];
        for (int w = 1; w < kWaves; ++w) s += red[w * K + k];
        v[k] = s;
    }
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)       // a full lane permutation inside rows of 16 lanes (VALU, no LDS)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// v_permlane32_swap: lanes 0..31 end up with a(L) + a(L + 32), lanes 32..63 with b(L - 32) + b(L)
__device__ __forceinline__ double swap_add32(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// v_permlane16_swap: even rows of 16 lanes end up with a summed over the row pair, odd rows with b
__device__ __forceinline__ double swap_add16(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// The 28 sums of a linearisation (21 + 6 + 1), every lane holding all of them afterwards.  A transposing butterfly:
// each step pairs the values (k, k + C/2) and the lanes across one lane-index bit -- the lane with the bit clear keeps
// the sum of value k over the pair, the other one that of value k + C/2 -- so the wavefront does 16 + 8 + 4 + 2 + 1 + 1
// exchanges instead of 28 x 6, on the VALU (gfx950's permlane swaps, then DPP inside rows of 16) with no LDS traffic.
// Lane L ends with value L >> 1 summed over the wavefront; the wavefronts' partials meet in LDS, in index order.
__device__ void block_sum28(double (&v)[28], double *red /* [kRedDoubles] */)
{
    const int lane = threadIdx.x & 63;
    double w[16];
#pragma unroll
    for (int k = 0; k < 12; ++k) w[k] = swap_add32(v[k], v[k + 16]);
#pragma unroll
    for (int k = 12; k < 16; ++k) w[k] = swap_add32(v[k], 0.0);
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = swap_add16(w[k], w[k + 8]);
    {
        const bool up = lane & 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double keep = up ? w[k + 4] : w[k], send = up ? w[k] : w[k + 4];
            w[k] = keep + dpp_f64<0x128>(send);           // row_ror:8 = lane ^ 8
        }
    }
    {
        const bool up = lane & 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double keep = up ? w[k + 2] : w[k], send = up ? w[k] : w[k + 2];
            w[k] = keep + dpp_f64<0x141>(send);           // row_half_mirror = lane ^ 7: crosses bit 2
        }
    }
    {
        const bool up = lane & 2;
        const double keep = up ? w[1] : w[0], send = up ? w[0] : w[1];
        w[0] = keep + dpp_f64<0x4E>(send);                // quad_perm [2,3,0,1] = lane ^ 2
    }
    w[0] = w[0] + dpp_f64<0xB1>(w[0]);                    // quad_perm [1,0,3,2] = lane ^ 1
    const int wv = threadIdx.x >> 6;
    __syncthreads();                                      // the previous reduction's readers are done
    if (!(lane & 1) && lane < 56) red[wv * 28 + (lane >> 1)] = w[0];
    __syncthreads();
    if (threadIdx.x < 28) {
        double s = red[threadIdx.x];
        for (int x = 1; x < kWaves; ++x) s += red[x * 28 + threadIdx.x];
        red[kWaves * 28 + threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 28; ++k) v[k] = red[kWaves * 28 + k];
}

// (The branch must be UNIFORM: a lane-divergent `threadIdx.x == 0` next to the reductions' lane exchanges -- v_permlane*_swap, DPP -- let the
//  compiler run part of the butterfly under a partial EXEC mask in the instrumented build: the pose never moved from its initial value for
//  frames of ten edges and more (found in round 6 by running the GPU suite on the instrumented twin).  Every lane of wavefront 0 adds the same
//  delta to the same word.)
#define POSE_PROF(slot)                                                                                         \
    do {                                                                                                       \
        if (kInstr && a.prof && blockIdx.x == 0 && __builtin_amdgcn_readfirstlane((int)threadIdx.x) == 0) {    \
            const long long now_ = clock64();                                                                  \
            a.prof[slot] += now_ - prof_t; a.prof[8 + (slot)] += 1; prof_t = now_;                             \
        }                                                                                                      \
    } while (0)

// STAGED: the frame's edges (28 bytes of input + a state word each) are copied into LDS once and every pass reads them
// there; otherwise (more than kStageMax keypoints) every pass reads the caller's arrays.
template <bool STAGED>
__global__ void __launch_bounds__(kPT) k_pose_opt(PoseArgs a)
{
    __shared__ double red[kRedDoubles];
    extern __shared__ float4 dyn[];                       // STAGED: recA[cap] | recB[cap] | edge_chi2[cap]; else edge_chi2[maxN]
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = min(a.n_points[f], a.maxN);
    const int ncrf = a.n_crf ? a.n_crf[f] : n;
    const size_t fo = (size_t)f * a.maxN;
    float4 *recA = dyn, *recB = dyn + (STAGED ? a.maxN : 0);
    float *edge_chi2 = reinterpret_cast<float *>(dyn + (STAGED ? 2 * a.maxN : 0));   // chi2 of every edge's stored _error (as `const float chi2`)
    unsigned short *idx = reinterpret_cast<unsigned short *>(edge_chi2 + a.maxN);    // STAGED: the round's level-0 edges, ascending
    __shared__ int wave_cnt[kWaves];
    long long prof_t = kInstr ? clock64() : 0;

    auto has_edge = [&](int i) { return (!a.valid || a.valid[fo + i]) && (!a.label || i >= ncrf || a.label[fo + i] != 0); };
    auto fetch = [&](int i) {
        Edge in;
        if (STAGED) {
            const float4 A = recA[i], B = recB[i];
            in.X[0] = A.x; in.X[1] = A.y; in.X[2] = A.z; in.u = A.w; in.v = B.x; in.ur = B.y; in.is2 = B.z; in.state = __float_as_int(B.w);
        } else {
            in.state = (has_edge(i) ? 1 : 0) | (a.outlier[fo + i] ? 2 : 0);
            in.X[0] = a.Xw[(fo + i) * 3]; in.X[1] = a.Xw[(fo + i) * 3 + 1]; in.X[2] = a.Xw[(fo + i) * 3 + 2];
            in.u = a.kp[(fo + i) * 2]; in.v = a.kp[(fo + i) * 2 + 1];
            in.ur = a.ur[fo + i]; in.is2 = a.is2[fo + i];
        }
        return in;
    };

    double cnt[1] = {0.0};
    for (int i = tid; i < n; i += kPT) {
        const bool he = has_edge(i);
        if (he) { cnt[0] += 1.0; a.outlier[fo + i] = 0; }                        // Optimizer.cc:283-284
        if (STAGED) {
            recA[i] = make_float4(a.Xw[(fo + i) * 3], a.Xw[(fo + i) * 3 + 1], a.Xw[(fo + i) * 3 + 2], a.kp[(fo + i) * 2]);
            recB[i] = make_float4(a.kp[(fo + i) * 2 + 1], a.ur[fo + i], a.is2[fo + i], __int_as_float(he ? 1 : 0));
        }
    }
    block_sum<1>(cnt, red);
    const int n_init = (int)cnt[0];
    double R0[3][3], t0[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R0[i][j] = a.Tcw_in[f * 16 + 4 * i + j]; t0[i] = a.Tcw_in[f * 16 + 4 * i + 3]; }
    if (tid == 0) a.n_initial[f] = n_init;
    if (n_init < 3) {                                     // Optimizer.cc:361-362: return 0, pose untouched
        if (tid < 16) a.Tcw_out[f * 16 + tid] = a.Tcw_in[f * 16 + tid];
        if (tid == 0) a.n_inliers[f] = 0;
        return;
    }
    PQ q0 = pq_from_matrix(R0);                           // Converter::toSE3Quat
    pq_normalize(q0);
    const double dMono = (double)(float)sqrt(5.991), dStereo = (double)(float)sqrt(7.815);
    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
    PQ q = q0;
    double t[3] = {t0[0], t0[1], t0[2]};
    int nBad = 0;
    POSE_PROF(0);

    // STAGED: the list of the round's level-0 edges, in index order, so that a sweep runs ceil(active / 256) full-width
    // iterations instead of ceil(n / 256) part-empty ones.  Each wavefront compacts one contiguous quarter of the frame.
    auto compact_level0 = [&]() {
        const int lane = tid & 63, wv = tid >> 6;
        const int chunk = (n + kWaves * 64 - 1) / (kWaves * 64) * 64, lo = wv * chunk, hi = min(n, lo + chunk);
        int cnt = 0;
        for (int i = lo + lane; i - lane < hi; i += 64)
            cnt += __popcll(__ballot(i < hi && (__float_as_int(recB[i].w) & 3) == 1));
        if (lane == 0) wave_cnt[wv] = cnt;
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < kWaves; ++w) { const int c = wave_cnt[w]; off += w < wv ? c : 0; tot += c; }
        for (int i = lo + lane; i - lane < hi; i += 64) {
            const bool on = i < hi && (__float_as_int(recB[i].w) & 3) == 1;
            const unsigned long long m = __ballot(on);
            if (on) idx[off + __popcll(m & ((1ull << lane) - 1))] = (unsigned short)i;
            off += __popcll(m);
        }
        __syncthreads();
        return tot;
    };
    int n_act = n;

    // computeActiveErrors + linearizeSystem of the level-0 edges at (qq, tt): acc = [upper triangle of H | b | robust chi2],
    // summed over the workgroup.  g2o runs the error pass of an LM trial and the linearisation of the next iteration as two
    // sweeps at the same pose; one sweep here serves both (a rejected trial's H and b are simply dropped).
    auto linearize = [&](bool robust, const PQ &qq, const double (&tt)[3], double (&acc)[28]) {
        for (int k = 0; k < 28; ++k) acc[k] = 0.0;
        for (int j = tid; j < n_act; j += kPT) {
            const int i = STAGED ? idx[j] : j;
            const Edge in = fetch(i);
            if (!STAGED && (in.state & 3) != 1) continue;
            double e[3], pc[3], J[3][6], rho[3] = {0, 1, 0};
            const double c = edge_error(a, in, qq, tt, e, pc);
            edge_chi2[i] = (float)c;
            if (robust) { huber(c, in.ur < 0 ? dMono : dStereo, rho); acc[27] += rho[0]; }
            else acc[27] += c;
            edge_jacobian(a, in, pc, J);
            if (in.ur < 0)                                // a monocular edge has two rows: a zero third row adds (signed) zeros only
                for (int c2 = 0; c2 < 6; ++c2) J[2][c2] = 0.0;
            // b -= rho' J^T W e, H += J^T (rho' W) J (base_unary_edge.hpp:62-63), rows summed in order; products with the
            // Jacobian's structural zeros (columns 4 of rows 0 and 2, column 3 of row 1) are skipped: they add signed zeros
            const double w = in.is2, rw = rho[1] * w;
            const double we[3] = {w * e[0], w * e[1], w * e[2]};
            double wJ[3][6];
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int c2 = 0; c2 < 6; ++c2) wJ[d][c2] = rw * J[d][c2];
            int p = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                double s = 0.0;
                bool first = true;
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    if (jac_zero(d, r)) continue;
                    const double term = J[d][r] * we[d];
                    s = first ? term : s + term;
                    first = false;
                }
                acc[21 + r] -= rho[1] * s;
#pragma unroll
                for (int c2 = r; c2 < 6; c2++, p++) {
                    double h = 0.0;
                    first = true;
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        if (jac_zero(d, r) || jac_zero(d, c2)) continue;
                        const double term = J[d][r] * wJ[d][c2];
                        h = first ? term : h + term;
                        first = false;
                    }
                    if (!first) acc[p] += h;
                }
            }
        }
        POSE_PROF(1);
        block_sum28(acc, red);
        POSE_PROF(2);
    };

    for (int it = 0; it < 4; it++) {
        const bool robust = it < 3;
        q = q0; t[0] = t0[0]; t[1] = t0[1]; t[2] = t0[2];
        double lambda = 0, ni = 2;
        int nBadLM = 0;
        double cur[28];                                   // the system at the current estimate (q, t)
        if (STAGED) n_act = compact_level0();
        linearize(robust, q, t, cur);
        for (int iter = 0; iter < 10; iter++) {
            double currentChi = cur[27];
            const double iniChi = currentChi;
            if (iter == 0) {
                double mx = 0;
                for (int j = 0, p = 0; j < 6; p += 6 - j, j++) mx = fmax(fabs(cur[p]), mx);
                lambda = 1e-5 * mx; ni = 2; nBadLM = 0;
            }
            double rho_gain = 0;
            int qmax = 0;
            do {
                PQ qn = q;
                double tn[3] = {t[0], t[1], t[2]};
                double x[6] = {0, 0, 0, 0, 0, 0};
                const bool ok2 = solve6(cur, lambda, x);
                pose_oplus(x, qn, tn);
                POSE_PROF(3);
                double nw[28];
                linearize(robust, qn, tn, nw);
                const double tempChi = ok2 ? nw[27] : DBL_MAX;
                rho_gain = currentChi - tempChi;
                double scale = 0;
                for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + cur[21 + j]);
                scale += 1e-3;
                rho_gain /= scale;
                if (rho_gain > 0 && isfinite(tempChi)) {
                    const double g = 2 * rho_gain - 1;
                    double alpha = 1. - g * g * g;
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                    q = qn; t[0] = tn[0]; t[1] = tn[1]; t[2] = tn[2];
                    for (int k = 0; k < 28; ++k) cur[k] = nw[k];
                } else {
                    lambda *= ni; ni *= 2;                // the estimate is popped: (q, t) and its system stay
                }
                qmax++;
            } while (rho_gain < 0 && qmax < 10);
            if (qmax == 10 || rho_gain == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) break;
        }
        double bad[1] = {0.0};                            // Optimizer.cc:378-432
        for (int i = tid; i < n; i += kPT) {
            const Edge in = fetch(i);
            if (!(in.state & 1)) continue;
            if (in.state & 2) {                           // sat the round out: fresh computeError(); the others keep the stale one
                double e[3], pc[3];
                edge_chi2[i] = (float)edge_error(a, in, q, t, e, pc);
            }
            const bool out = edge_chi2[i] > (in.ur < 0 ? chi2Mono : chi2Stereo);
            a.outlier[fo + i] = out ? 1 : 0;
            if (STAGED) recB[i].w = __int_as_float(out ? 3 : 1);
            bad[0] += out ? 1.0 : 0.0;
        }
        block_sum<1>(bad, red);
        nBad = (int)bad[0];
        POSE_PROF(4);
        if (n_init < 10) break;
    }
    if (tid == 0) {
        const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z, twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x,
                     txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
        const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
        float *T = a.Tcw_out + (size_t)f * 16;
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[4 * i + j] = (float)R[i][j]; T[4 * i + 3] = (float)t[i]; }
        T[12] = T[13] = T[14] = 0.0f; T[15] = 1.0f;
        a.n_inliers[f] = n_init - nBad;
    }
}

}  // namespace

// Every pointer is device-accessible memory (device or pinned host).  One workgroup per frame.
hipError_t launch_pose_optimization(int F, int maxN, const int *n_points, const float *Xw, const float *kp, const float *ur,
                                    const float *is2, const uint8_t *valid, const int16_t *label, const float *K4, float bf,
                                    const float *Tcw_in, float *Tcw_out, uint8_t *outlier, int *n_inliers, int *n_initial,
                                    hipStream_t s, const int *n_crf)
{
    PoseArgs a{};
    a.n_crf = n_crf;
    a.maxN = maxN; a.n_points = n_points; a.Xw = Xw; a.kp = kp; a.ur = ur; a.is2 = is2; a.valid = valid; a.label = label;
    a.fx = K4[0]; a.fy = K4[1]; a.cx = K4[2]; a.cy = K4[3]; a.bf = bf;
    a.Tcw_in = Tcw_in; a.Tcw_out = Tcw_out; a.outlier = outlier; a.n_inliers = n_inliers; a.n_initial = n_initial;
    static const bool want_prof = kInstr && ab_env("LCCRF_POSE_PROF") != nullptr;
    if (want_prof) {
        if (hipHostMalloc(reinterpret_cast<void **>(&a.prof), 16 * sizeof(long long)) != hipSuccess) return hipErrorOutOfMemory;
        for (int i = 0; i < 16; ++i) a.prof[i] = 0;
    }
    const bool staged = maxN <= kStageMax;
    const size_t lds = (size_t)maxN * (staged ? 38 : 4) + 16;
    if (lds > kPoseLdsMax) return hipErrorInvalidValue;
    hipError_t er = hipSuccess;
    if (staged) {
        er = hipFuncSetAttribute(reinterpret_cast<const void *>(k_pose_opt<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPoseLdsMax);
        if (er != hipSuccess) return er;
        k_pose_opt<true><<<F, kPT, lds, s>>>(a);
    } else {
        er = hipFuncSetAttribute(reinterpret_cast<const void *>(k_pose_opt<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPoseLdsMax);
        if (er != hipSuccess) return er;
        k_pose_opt<false><<<F, kPT, lds, s>>>(a);
    }
    er = hipGetLastError();
    if (want_prof && er == hipSuccess) {
        (void)hipStreamSynchronize(s);
        static const char *names[5] = {"setup", "sweep", "reduce28", "solve+oplus", "classify"};
        fprintf(stderr, "[lccrf] pose prof (frame 0, lane 0; shader clocks):");
        for (int i = 0; i < 5; ++i) fprintf(stderr, " %s %lld/%lld", names[i], a.prof[i], a.prof[8 + i]);
        fprintf(stderr, "\n");
        (void)hipHostFree(a.prof);
    }
    return er;
}

}  // namespace lccrf
