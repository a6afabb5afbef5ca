// pose_opt.hip -- Optimizer::PoseOptimization (reference src/Optimizer.cc:239-450), the step right after the CRF
// (src/Tracking.cc:1002): motion-only bundle adjustment of the current frame on the matches the CRF left standing.
// One 256-lane workgroup per frame; the CRF's MAP labels are read where the CRF kernel left them (device memory):
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
// solves the 6x6 system redundantly from the same reduced sums, so no broadcast is needed.
#include "engine.h"

#include <cfloat>

namespace lccrf {

namespace {

constexpr int kPT = 256;

struct PQ { double w, x, y, z; };

struct PoseArgs {
    int maxN;                             // per-frame stride of every per-point array
    const int *n_points;                  // [F]
    const float *Xw, *kp, *ur, *is2;      // [F][maxN][3], [F][maxN][2], [F][maxN], [F][maxN]
    const uint8_t *valid;                 // [F][maxN] mvpMapPoints[i] != NULL, or null (all valid)
    const int16_t *label;                 // [F][maxN] CRF labels (0 = moving: no edge), or null
    float fx, fy, cx, cy, bf;
    const float *Tcw_in;                  // [F][16]
    float *Tcw_out;                       // [F][16]
    uint8_t *outlier;                     // [F][maxN] mvbOutlier
    int *n_inliers, *n_initial;           // [F]
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
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double v[3];
        t = sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[k][j] - m[j][k]) * t;
        v[j] = (m[j][i] + m[i][j]) * t;
        v[k] = (m[k][i] + m[i][k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
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
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
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

struct EdgeIn { float X[3], u, v, ur, is2; };

__device__ __forceinline__ double edge_error(const PoseArgs &a, const EdgeIn &in, const PQ &q, const double (&t)[3], double (&e)[3],
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

__device__ __forceinline__ int edge_jacobian(const PoseArgs &a, const EdgeIn &in, const double (&pc)[3], double (&J)[3][6])
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

// (H + lambda I) x = b for a symmetric 6x6 H (upper triangle given as 21 packed values), LDL^T; false if not positive definite
__device__ bool solve6(const double (&Hp)[21], double lambda, const double (&b)[6], double (&x)[6])
{
    double H[6][6], L[6][6], D[6];
    for (int i = 0, p = 0; i < 6; i++)
        for (int j = i; j < 6; j++, p++) H[i][j] = H[j][i] = Hp[p];
    for (int j = 0; j < 6; j++) {
        double d = H[j][j] + lambda;
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
        if (!(d > 0)) return false;
        D[j] = d;
        for (int i = j + 1; i < 6; i++) {
            double s = H[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k] * D[k];
            L[i][j] = s / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i][k] * y[k]; y[i] = s; }
    for (int i = 5; i >= 0; i--) { double s = y[i] / D[i]; for (int k = i + 1; k < 6; k++) s -= L[k][i] * x[k]; x[i] = s; }
    return true;
}

// Sum of K doubles per lane over the workgroup, identical in every lane afterwards (fixed tree: xor-shuffles inside a
// wavefront, then the four wavefront partials in index order).
template <int K>
__device__ void block_sum(double (&v)[K], double *red /* [4][K] */)
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
    for (int k = 0; k < K; ++k) v[k] = ((red[k] + red[K + k]) + red[2 * K + k]) + red[3 * K + k];
}

__global__ void __launch_bounds__(kPT) k_pose_opt(PoseArgs a)
{
    __shared__ double red[4 * 28];
    extern __shared__ float edge_chi2[];                  // [maxN]: chi2 of every edge's stored _error (as `const float chi2`)
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = a.n_points[f];
    const size_t fo = (size_t)f * a.maxN;
    auto is_valid = [&](int i) { return (!a.valid || a.valid[fo + i]) && (!a.label || a.label[fo + i] != 0); };
    auto load = [&](int i) {
        EdgeIn in;
        in.X[0] = a.Xw[(fo + i) * 3]; in.X[1] = a.Xw[(fo + i) * 3 + 1]; in.X[2] = a.Xw[(fo + i) * 3 + 2];
        in.u = a.kp[(fo + i) * 2]; in.v = a.kp[(fo + i) * 2 + 1];
        in.ur = a.ur[fo + i]; in.is2 = a.is2[fo + i];
        return in;
    };
    double cnt[1] = {0.0};
    for (int i = tid; i < n; i += kPT)
        if (is_valid(i)) { cnt[0] += 1.0; a.outlier[fo + i] = 0; }               // Optimizer.cc:283-284
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

    // computeActiveErrors + activeRobustChi2 of the level-0 edges at (q, t)
    auto active_chi2 = [&](bool robust, const PQ &qq, const double (&tt)[3]) {
        double chi[1] = {0.0};
        for (int i = tid; i < n; i += kPT) {
            if (!is_valid(i) || a.outlier[fo + i]) continue;
            const EdgeIn in = load(i);
            double e[3], pc[3], rho[3];
            const double c = edge_error(a, in, qq, tt, e, pc);
            edge_chi2[i] = (float)c;
            if (robust) { huber(c, in.ur < 0 ? dMono : dStereo, rho); chi[0] += rho[0]; }
            else chi[0] += c;
        }
        block_sum<1>(chi, red);
        return chi[0];
    };

    for (int it = 0; it < 4; it++) {
        const bool robust = it < 3;
        q = q0; t[0] = t0[0]; t[1] = t0[1]; t[2] = t0[2];
        double lambda = 0, ni = 2;
        int nBadLM = 0;
        for (int iter = 0; iter < 10; iter++) {
            double acc[28];                               // 21 (upper triangle of H) + 6 (b) + 1 (chi2)
            for (int k = 0; k < 28; ++k) acc[k] = 0.0;
            for (int i = tid; i < n; i += kPT) {
                if (!is_valid(i) || a.outlier[fo + i]) continue;
                const EdgeIn in = load(i);
                double e[3], pc[3], J[3][6], rho[3] = {0, 1, 0};
                const double c = edge_error(a, in, q, t, e, pc);
                edge_chi2[i] = (float)c;
                if (robust) { huber(c, in.ur < 0 ? dMono : dStereo, rho); acc[27] += rho[0]; }
                else acc[27] += c;
                const int D = edge_jacobian(a, in, pc, J);
                const double w = in.is2;
                for (int r = 0, p = 0; r < 6; r++) {
                    double s = 0;
                    for (int d = 0; d < D; d++) s += J[d][r] * (w * e[d]);
                    acc[21 + r] -= rho[1] * s;
                    for (int c2 = r; c2 < 6; c2++, p++) {
                        double h = 0;
                        for (int d = 0; d < D; d++) h += J[d][r] * ((rho[1] * w) * J[d][c2]);
                        acc[p] += h;
                    }
                }
            }
            block_sum<28>(acc, red);
            double Hp[21], b[6];
            for (int k = 0; k < 21; ++k) Hp[k] = acc[k];
            for (int k = 0; k < 6; ++k) b[k] = acc[21 + k];
            double currentChi = acc[27];
            const double iniChi = currentChi;
            if (iter == 0) {
                double mx = 0;
                for (int j = 0, p = 0; j < 6; p += 6 - j, j++) mx = fmax(fabs(Hp[p]), mx);
                lambda = 1e-5 * mx; ni = 2; nBadLM = 0;
            }
            double rho_gain = 0;
            int qmax = 0;
            do {
                const PQ qb = q;
                const double tb[3] = {t[0], t[1], t[2]};
                double x[6] = {0, 0, 0, 0, 0, 0};
                const bool ok2 = solve6(Hp, lambda, b, x);
                pose_oplus(x, q, t);
                double tempChi = active_chi2(robust, q, t);
                if (!ok2) tempChi = DBL_MAX;
                rho_gain = currentChi - tempChi;
                double scale = 0;
                for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho_gain /= scale;
                if (rho_gain > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow(2 * rho_gain - 1, 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                } else {
                    lambda *= ni; ni *= 2;
                    q = qb; t[0] = tb[0]; t[1] = tb[1]; t[2] = tb[2];
                }
                qmax++;
            } while (rho_gain < 0 && qmax < 10);
            if (qmax == 10 || rho_gain == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) break;
        }
        double bad[1] = {0.0};                            // Optimizer.cc:378-432
        for (int i = tid; i < n; i += kPT) {
            if (!is_valid(i)) continue;
            if (a.outlier[fo + i]) {                      // sat the round out: fresh computeError(); the others keep the stale one
                const EdgeIn in = load(i);
                double e[3], pc[3];
                edge_chi2[i] = (float)edge_error(a, in, q, t, e, pc);
            }
            const bool out = edge_chi2[i] > (a.ur[fo + i] < 0 ? chi2Mono : chi2Stereo);
            a.outlier[fo + i] = out ? 1 : 0;
            bad[0] += out ? 1.0 : 0.0;
        }
        block_sum<1>(bad, red);
        nBad = (int)bad[0];
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
                                    hipStream_t s)
{
    PoseArgs a{};
    a.maxN = maxN; a.n_points = n_points; a.Xw = Xw; a.kp = kp; a.ur = ur; a.is2 = is2; a.valid = valid; a.label = label;
    a.fx = K4[0]; a.fy = K4[1]; a.cx = K4[2]; a.cy = K4[3]; a.bf = bf;
    a.Tcw_in = Tcw_in; a.Tcw_out = Tcw_out; a.outlier = outlier; a.n_inliers = n_inliers; a.n_initial = n_initial;
    const size_t lds = (size_t)maxN * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    k_pose_opt<<<F, kPT, lds, s>>>(a);
    return hipGetLastError();
}

}  // namespace lccrf
