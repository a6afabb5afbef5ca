// unary_builder.hip -- the step right before the CRF:
// Tracking::ComputeMapPointErrAndObserv (src/Tracking.cc:1803-1839) for every candidate map point
// of a frame, followed by Tracking::RroughClassify (src/Tracking.cc:1961-2013).
//
// In the reference this is a host loop that chases MapPoint -> observations -> KeyFrame pointers
// under mutexes.  Here the caller flattens that graph once per frame into a CSR
// (point -> observations) plus per-keyframe pose / intrinsics / image bounds, and one thread per
// map point walks its observations in the order given (the reference walks a
// std::map<KeyFrame*, size_t>, i.e. pointer order; the caller passes that order).
//
// PARITY UNPINNED: the reference has no test or fixture for these two functions and
// src/Tracking.cc cannot be built here (OpenCV).  The arithmetic follows the oracle's restatement
// (oracle/lccrf_oracle.c: orc_map_point_err_observ, orc_rough_classify) operation by operation:
// `Rcw * x3Dw + tcw` is one cv::gemm(A, B, 1, C, 1) (OpenCV fuses the MatExpr), which for a 3x3 by
// 3x1 CV_32F product takes OpenCV 3.x's small-matrix path (modules/core/src/matmul.cpp, `len == 3`,
// `d_size.width <= 16` branch): the three products and their sum in fp32, left to right, then
// `(float)(t*alpha + c*beta)` in double, which rounds like the fp32 add.  `1.0 / z` in double, the
// pixel error in double (Point2d), everything else fp32, no FMA.  exp() is the one libm call: the host oracle
// uses glibc expf, the device evaluates exp in double and narrows; the two can differ in the last
// bit for ~0.06 % of arguments, which changes a label only if the probability sum then lands
// exactly on the threshold.
#include "engine.h"

#include <cstring>
#include <mutex>

namespace lccrf {

namespace {

struct UnaryArgs {
    int n_points, n_kf;
    const float *Xw;          // [n][3]
    const int *obs_ptr;       // [n+1]
    const int *obs_kf;        // [n_obs]
    const double *obs_kp;     // [n_obs][2]
    const float *kf_pose;     // [n_kf][12] row-major [R|t]
    const float *kf_intr;     // [n_kf][4]  fx fy cx cy
    const float *kf_bounds;   // [n_kf][4]  minX maxX minY maxY
    const double *match_prob; // [n] or null
    lccrf_crf_params p;
    float *observs, *error, *depth;
    int16_t *label;
};

__global__ void __launch_bounds__(256) k_unary_build(UnaryArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_points) return;
    const int o0 = a.obs_ptr[i], o1 = a.obs_ptr[i + 1];
    const int n_obs = o1 - o0;
    float err = 0.0f, dep = 0.0f;                         // Tracking.cc:1854-1855
    const float X0 = a.Xw[3 * i], X1 = a.Xw[3 * i + 1], X2 = a.Xw[3 * i + 2];
    for (int o = o0; o < o1; ++o) {                       // Tracking.cc:1813-1836
        const int kf = a.obs_kf[o];
        const float *P = a.kf_pose + (size_t)kf * 12;
        float xc3[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float t = P[r * 4 + 0] * X0 + P[r * 4 + 1] * X1 + P[r * 4 + 2] * X2;
            xc3[r] = (float)((double)t * 1.0 + (double)P[r * 4 + 3] * 1.0);
        }
        const float invzc = (float)(1.0 / (double)xc3[2]);                   // :1821
        if (invzc < 0) continue;
        const float *K = a.kf_intr + (size_t)kf * 4;
        const float u = K[0] * xc3[0] * invzc + K[2];
        const float v = K[1] * xc3[1] * invzc + K[3];
        const float *B = a.kf_bounds + (size_t)kf * 4;
        if (u < B[0] || u > B[1] || v < B[2] || v > B[3]) continue;          // :1828
        const double dx = (double)u - a.obs_kp[2 * (size_t)o], dy = (double)v - a.obs_kp[2 * (size_t)o + 1];
        err += (float)sqrt(dx * dx + dy * dy);                               // :1833
        dep += xc3[2];
    }
    float error = 0.0f, depth = 0.0f;
    if (n_obs > 0) {                                      // divides by ALL observations, :1837-1838
        error = err / (float)n_obs;
        depth = dep / (float)n_obs;
    }
    a.observs[i] = (float)n_obs;
    a.error[i] = error;
    a.depth[i] = depth;

    // RroughClassify, Tracking.cc:1964-2010
    const lccrf_crf_params &p = a.p;
    const float observ_sigma2 = p.stdev_beta * p.stdev_beta;
    const float rpjerror_sigma2 = p.stdev_alpha * p.stdev_alpha;
    const float depth_sigma2 = p.point3d_stdev * p.point3d_stdev;
    const float ob = (float)n_obs;
    const float k1 = (ob - p.u_beta) * (ob - p.u_beta) / (2 * observ_sigma2);
    const float k2 = (error - p.u_alpha) * (error - p.u_alpha) / (2 * rpjerror_sigma2);
    const float k3 = (depth - p.u_depth) * (depth - p.u_depth) / (2 * depth_sigma2);
    const float p1 = (float)exp((double)-k1), p2 = (float)exp((double)-k2), p3 = (float)exp((double)-k3);
    int16_t lab;
    if (!a.match_prob) lab = (p1 + p2 + p3 <= p.pth) ? 0 : 1;                                   // :1996
    else lab = ((double)(p1 + p2 + p3) + a.match_prob[i] <= (double)p.pth + 0.2) ? 0 : 1;       // :2004
    a.label[i] = n_obs > 0 ? lab : (int16_t)-1;           // the caller drops points without observations, :1858
}

// Grow-only staging shared by all calls (guarded by a mutex): ONE pinned host arena that holds the
// packed inputs followed by the outputs, ONE device buffer for the inputs.  A call costs one
// host-side pack, one upload command, one kernel that writes its results straight into the pinned
// arena, one stream synchronise (each separate small copy would add ~10 us of stream time).
struct Scratch {
    std::mutex m;
    unsigned char *host = nullptr, *dev = nullptr;
    size_t host_cap = 0, dev_cap = 0;
    int device = -1;
    hipStream_t stream = nullptr;
} g_scratch;

void release_scratch()
{
    if (g_scratch.host) (void)hipHostFree(g_scratch.host);
    if (g_scratch.dev) (void)hipFree(g_scratch.dev);
    if (g_scratch.stream) (void)hipStreamDestroy(g_scratch.stream);
    g_scratch.host = g_scratch.dev = nullptr;
    g_scratch.host_cap = g_scratch.dev_cap = 0;
    g_scratch.stream = nullptr;
}

}  // namespace

// returns hipSuccess or the failing call's error
hipError_t run_unary_build(int device_id, int n_points, const float *Xw, const int32_t *obs_ptr, const int32_t *obs_kf,
                           const double *obs_kp, int n_kf, const float *kf_pose, const float *kf_intr,
                           const float *kf_bounds, const double *match_prob, const lccrf_crf_params *params,
                           float *observs_out, float *error_out, float *depth_out, int16_t *label_out)
{
    std::lock_guard<std::mutex> g(g_scratch.m);
    hipError_t e;
    if (g_scratch.device != device_id) {                  // the staging belongs to one device at a time
        release_scratch();
        g_scratch.device = device_id;
    }
    if (!g_scratch.stream && (e = hipStreamCreateWithFlags(&g_scratch.stream, hipStreamNonBlocking)) != hipSuccess) return e;
    hipStream_t s = g_scratch.stream;
    if (n_points <= 0) return hipSuccess;
    const size_t n = (size_t)n_points, n_obs = (size_t)obs_ptr[n_points];
    // input blocks (8-byte aligned offsets), then output blocks
    const size_t in_sz[8] = {n_obs * 2 * sizeof(double), match_prob ? n * sizeof(double) : 0, n * 3 * sizeof(float),
                             (n + 1) * sizeof(int), n_obs * sizeof(int), (size_t)n_kf * 12 * sizeof(float),
                             (size_t)n_kf * 4 * sizeof(float), (size_t)n_kf * 4 * sizeof(float)};
    const void *in_src[8] = {obs_kp, match_prob, Xw, obs_ptr, obs_kf, kf_pose, kf_intr, kf_bounds};
    size_t in_off[8], total_in = 0;
    for (int i = 0; i < 8; ++i) { in_off[i] = total_in; total_in += (in_sz[i] + 7) & ~(size_t)7; }
    const size_t out_sz[4] = {n * sizeof(float), n * sizeof(float), n * sizeof(float), n * sizeof(int16_t)};
    size_t out_off[4], total = total_in;
    for (int i = 0; i < 4; ++i) { out_off[i] = total; total += (out_sz[i] + 7) & ~(size_t)7; }
    if (g_scratch.host_cap < total) {
        if (g_scratch.host) (void)hipHostFree(g_scratch.host);
        g_scratch.host = nullptr;
        g_scratch.host_cap = 0;
        if ((e = hipHostMalloc(reinterpret_cast<void **>(&g_scratch.host), total + total / 2 + 4096, hipHostMallocDefault)) != hipSuccess) return e;
        g_scratch.host_cap = total + total / 2 + 4096;
    }
    if (g_scratch.dev_cap < total_in) {
        if (g_scratch.dev) (void)hipFree(g_scratch.dev);
        g_scratch.dev = nullptr;
        g_scratch.dev_cap = 0;
        if ((e = hipMalloc(reinterpret_cast<void **>(&g_scratch.dev), total_in + total_in / 2 + 4096)) != hipSuccess) return e;
        g_scratch.dev_cap = total_in + total_in / 2 + 4096;
    }
    for (int i = 0; i < 8; ++i)
        if (in_sz[i]) memcpy(g_scratch.host + in_off[i], in_src[i], in_sz[i]);
    if ((e = hipMemcpyAsync(g_scratch.dev, g_scratch.host, total_in, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    UnaryArgs a;
    a.n_points = n_points;
    a.n_kf = n_kf;
    a.obs_kp = reinterpret_cast<const double *>(g_scratch.dev + in_off[0]);
    a.match_prob = match_prob ? reinterpret_cast<const double *>(g_scratch.dev + in_off[1]) : nullptr;
    a.Xw = reinterpret_cast<const float *>(g_scratch.dev + in_off[2]);
    a.obs_ptr = reinterpret_cast<const int *>(g_scratch.dev + in_off[3]);
    a.obs_kf = reinterpret_cast<const int *>(g_scratch.dev + in_off[4]);
    a.kf_pose = reinterpret_cast<const float *>(g_scratch.dev + in_off[5]);
    a.kf_intr = reinterpret_cast<const float *>(g_scratch.dev + in_off[6]);
    a.kf_bounds = reinterpret_cast<const float *>(g_scratch.dev + in_off[7]);
    a.p = *params;
    a.observs = reinterpret_cast<float *>(g_scratch.host + out_off[0]);      // pinned host memory: written by the kernel
    a.error = reinterpret_cast<float *>(g_scratch.host + out_off[1]);
    a.depth = reinterpret_cast<float *>(g_scratch.host + out_off[2]);
    a.label = reinterpret_cast<int16_t *>(g_scratch.host + out_off[3]);
    k_unary_build<<<dim3((n_points + 255) / 256), dim3(256), 0, s>>>(a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    void *dst[4] = {observs_out, error_out, depth_out, label_out};
    for (int i = 0; i < 4; ++i) memcpy(dst[i], g_scratch.host + out_off[i], out_sz[i]);
    return hipSuccess;
}

}  // namespace lccrf
