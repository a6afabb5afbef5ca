// fused_engine.hip -- SLAM-size inference as ONE kernel launch: one 1024-lane workgroup per
// frame runs startInference, every mean-field iteration and buildMap without leaving the CU.
//
// Why: at SLAM sizes (N ~ 2000 keypoints, V ~ 1.2k lattice vertices) one iteration moves
// < 0.5 MB; a launch-per-phase design is bound by launch gaps, not by memory (SURVEY.md
// section 7).  Here the mean-field state lives on chip:
//     LDS        Q[N][2], both ping-pong lattice value arrays of every kernel, the blur
//                neighbour table, the CSR row pointers, and the per-entry splat products
//     registers  everything a point needs for slice/apply/softmax (its d+1 vertex ids,
//                weights bary*alpha, w*norm, its unary) -- loaded once per launch
//     L2         the CSR (csr_w, csr_pt) streamed once per iteration, fully coalesced
//
// Bit-exactness: the reference splats sequentially over points (permutohedral_cpu.h:653-661),
// so a vertex's value is a left-to-right fp32 sum in ascending point order.  Phase P forms
// all products w*Q in parallel (exact, order-free); phase S then adds each vertex's products
// strictly left to right.  Nothing is re-associated, nothing is fused (-ffp-contract=off).
//
// Specialised for L = 2 labels (the SLAM configuration, src/Tracking.cc:1919) and kernels of
// equal dimension D; anything else runs on the streaming engine with identical results.
#include "engine.h"
#include "device_math.h"

#include <cstdio>
#include <cstdlib>

namespace lccrf {

namespace {

constexpr int kNT = 1024;                 // lanes per workgroup (16 wavefronts)
constexpr int kMaxFusedK = 2;
constexpr size_t kLdsLimit = 160 * 1024;  // MI355X: 160 KiB LDS per CU, one workgroup may own it all

struct FusedLayout {                      // byte offsets into dynamic LDS
    int q;                                // float2 [Nq]
    int prod[kMaxFusedK];                 // float  [2][E_k]   (label-major), may alias when !prod_all
    int val[kMaxFusedK][2];               // float2 [V_k+1]    slot 0 = absent neighbour = 0
    int nbr[kMaxFusedK];                  // u32    [D1][V_k]  (n1+1) | (n2+1)<<16
    int row[kMaxFusedK];                  // u16    [V_k+1]
    int Ecap[kMaxFusedK];                 // capacity of prod in entries
    int Vcap[kMaxFusedK];
    int prod_all;                         // 1: every kernel has its own product buffer
    int total;
};

struct FusedArgs {
    KernelDev kd[kMaxFusedK];
    FusedLayout lay;
    int n_iter, with_map;
    float relax;
    long long *timing;                    // debug: shader-clock stamps of workgroup 0 (LCCRF_FUSED_TIMING=1)
};

#define STAMP()                                                        \
    do {                                                               \
        if (a.timing && blockIdx.x == 0 && tid == 0) a.timing[n_stamp++] = clock64(); \
    } while (0)

template <int PPT, int K, int D, bool CSR_REG>
__global__ void __launch_bounds__(kNT) k_fused(CrfDev c, FusedArgs a)
{
    constexpr int D1 = D + 1;
    constexpr int EPT = PPT * D1;         // splat entries per lane and kernel: ceil(N*D1 / kNT)
    int n_stamp = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int N = c.n_points[f];

    float2 *Q = reinterpret_cast<float2 *>(smem + a.lay.q);

    // ---- per-thread point state (registers) ------------------------------------------
    float2 un[PPT], q[PPT];
    int off[PPT][K][D1];
    float wgt[PPT][K][D1];
    float wn[PPT][K];
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int i = tid + s * kNT;
        un[s] = make_float2(0.f, 0.f);
        q[s] = make_float2(0.f, 0.f);
        if (i < N) {
            un[s] = reinterpret_cast<const float2 *>(c.unary)[(size_t)f * c.maxN + i];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const KernelDev &kd = a.kd[k];
                const size_t e0 = (size_t)f * kd.Epad + (size_t)i * D1;
#pragma unroll
                for (int j = 0; j < D1; ++j) {
                    off[s][k][j] = kd.offset[e0 + j] + 1;
                    wgt[s][k][j] = kd.bary[e0 + j] * kd.alpha;            // permutohedral_cpu.h:689
                }
                wn[s][k] = kd.w * kd.norm[(size_t)f * kd.maxN + i];      // pairwise3d.h:77 (w_*norm_[i])
            }
            // startInference: Q = softmax(-unary), densecrf_base.h:78-80
            float in[2] = {un[s].x, un[s].y}, out[2] = {0.f, 0.f};
            exp_and_normalize_reg<2>(in, out, -1.0f, 1.0f);
            q[s] = make_float2(out[0], out[1]);
            Q[i] = q[s];
        }
    }

    // ---- per-frame lattice tables into LDS --------------------------------------------
    int V[K], E[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const KernelDev &kd = a.kd[k];
        V[k] = kd.V[f];
        E[k] = N * D1;
        unsigned *nbr = reinterpret_cast<unsigned *>(smem + a.lay.nbr[k]);
        const int2 *gn = reinterpret_cast<const int2 *>(kd.nbr) + (size_t)f * D1 * kd.Epad;
        for (int idx = tid; idx < D1 * V[k]; idx += kNT) {
            const int j = idx / V[k], v = idx - j * V[k];
            const int2 n = gn[(size_t)j * kd.Epad + v];
            nbr[idx] = (unsigned)(n.x + 1) | ((unsigned)(n.y + 1) << 16);
        }
        unsigned short *row = reinterpret_cast<unsigned short *>(smem + a.lay.row[k]);
        const int *gr = kd.rowptr + (size_t)f * (kd.Epad + 1);
        for (int v = tid; v <= V[k]; v += kNT) row[v] = (unsigned short)gr[v];
        if (tid == 0) {
            reinterpret_cast<float2 *>(smem + a.lay.val[k][0])[0] = make_float2(0.f, 0.f);
            reinterpret_cast<float2 *>(smem + a.lay.val[k][1])[0] = make_float2(0.f, 0.f);
        }
    }
    // The splat contributions (CSR order) never change between iterations: when they fit,
    // each lane keeps its EPT entries per kernel in registers and the loop touches no HBM/L2.
    float cw_r[CSR_REG ? K : 1][CSR_REG ? EPT : 1];
    int cp_r[CSR_REG ? K : 1][CSR_REG ? EPT : 1];
    if constexpr (CSR_REG) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float *cw = a.kd[k].csr_w + (size_t)f * a.kd[k].Epad;
            const int *cp = a.kd[k].csr_pt + (size_t)f * a.kd[k].Epad;
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int p = tid + u * kNT;
                cw_r[k][u] = (p < E[k]) ? cw[p] : 0.0f;
                cp_r[k][u] = (p < E[k]) ? cp[p] : 0;
            }
        }
    }
    __syncthreads();
    STAMP();

    for (int it = 0; it < a.n_iter; ++it) {
        // ---- splat = products (P) + ordered row sums (S) ------------------------------
        auto phase_P = [&](int k) {
            const KernelDev &kd = a.kd[k];
            float *p0 = reinterpret_cast<float *>(smem + a.lay.prod[k]);
            float *p1 = p0 + a.lay.Ecap[k];
            if constexpr (CSR_REG) {
                float2 x[EPT];
#pragma unroll
                for (int u = 0; u < EPT; ++u) x[u] = Q[cp_r[k][u]];
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const int p = tid + u * kNT;
                    if (p < E[k]) {
                        p0[p] = cw_r[k][u] * x[u].x;
                        p1[p] = cw_r[k][u] * x[u].y;
                    }
                }
            } else {
                const float *cw = kd.csr_w + (size_t)f * kd.Epad;
                const int *cp = kd.csr_pt + (size_t)f * kd.Epad;
#pragma unroll 1
                for (int u0 = 0; u0 < EPT; u0 += D1) {                  // D1 entries at a time: bounded registers
                    float w[D1];
                    int pt[D1];
#pragma unroll
                    for (int u = 0; u < D1; ++u) {
                        const int p = tid + (u0 + u) * kNT;
                        w[u] = (p < E[k]) ? cw[p] : 0.0f;
                        pt[u] = (p < E[k]) ? cp[p] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < D1; ++u) {
                        const int p = tid + (u0 + u) * kNT;
                        const float2 x = Q[pt[u]];
                        if (p < E[k]) {
                            p0[p] = w[u] * x.x;
                            p1[p] = w[u] * x.y;
                        }
                    }
                }
            }
        };
        auto phase_S = [&](int k) {
            const float *p0 = reinterpret_cast<const float *>(smem + a.lay.prod[k]);
            const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + a.lay.row[k]);
            float *val = reinterpret_cast<float *>(smem + a.lay.val[k][0]);
            for (int idx = tid; idx < 2 * V[k]; idx += kNT) {
                const int v = idx >> 1, l = idx & 1;
                const float *pl = p0 + l * a.lay.Ecap[k];
                const int s = row[v], t = row[v + 1];
                float acc = 0.0f;
                int p = s;
                for (; p + 8 <= t; p += 8) {                            // 8 loads in flight, then
                    const float x0 = pl[p], x1 = pl[p + 1], x2 = pl[p + 2], x3 = pl[p + 3];
                    const float x4 = pl[p + 4], x5 = pl[p + 5], x6 = pl[p + 6], x7 = pl[p + 7];
                    acc += x0; acc += x1; acc += x2; acc += x3;         // strictly left to right
                    acc += x4; acc += x5; acc += x6; acc += x7;
                }
                for (; p < t; ++p) acc += pl[p];
                val[(v + 1) * 2 + l] = acc;
            }
        };
        if (a.lay.prod_all) {
#pragma unroll
            for (int k = 0; k < K; ++k) phase_P(k);
            __syncthreads();
            STAMP();
#pragma unroll
            for (int k = 0; k < K; ++k) phase_S(k);
            __syncthreads();
            STAMP();
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                phase_P(k);
                __syncthreads();
                phase_S(k);
                __syncthreads();
            }
        }

        // ---- d+1 Jacobi blur passes, permutohedral_cpu.h:663-679 -----------------------
#pragma unroll
        for (int j = 0; j < D1; ++j) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float2 *src = reinterpret_cast<const float2 *>(smem + a.lay.val[k][j & 1]);
                float2 *dst = reinterpret_cast<float2 *>(smem + a.lay.val[k][(j & 1) ^ 1]);
                const unsigned *nbr = reinterpret_cast<const unsigned *>(smem + a.lay.nbr[k]) + j * V[k];
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

        // ---- slice + apply + softmax per point ----------------------------------------
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * kNT;
            if (i < N) {
                float nx[2] = {-un[s].x, -un[s].y};                       // stepInit, densecrf3d.h:154-158
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float2 *val = reinterpret_cast<const float2 *>(smem + a.lay.val[k][D1 & 1]);
                    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
                    for (int j = 0; j < D1; ++j) {
                        const float2 x = val[off[s][k][j]];
                        t0 += wgt[s][k][j] * x.x;
                        t1 += wgt[s][k][j] * x.y;
                    }
                    nx[0] += wn[s][k] * t0;                               // pairwise3d.h:77
                    nx[1] += wn[s][k] * t1;
                }
                float out[2] = {q[s].x, q[s].y};
                exp_and_normalize_reg<2>(nx, out, 1.0f, a.relax);
                q[s] = make_float2(out[0], out[1]);
                Q[i] = q[s];
            }
        }
        __syncthreads();
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
    }
    STAMP();
    if (a.timing && blockIdx.x == 0 && tid == 0) a.timing[63] = n_stamp;
}

bool make_layout(const CrfDev &c, const KernelDev *kds, const int *maxV, FusedLayout *lay)
{
    if (c.L != 2 || c.K < 1 || c.K > kMaxFusedK) return false;
    const int NA = c.activeN > 0 ? c.activeN : c.maxN;   // size LDS and the points-per-lane variant by the frames' real size
    if (NA < 1 || NA > 4 * kNT) return false;
    for (int k = 0; k < c.K; ++k) {
        if (kds[k].d != kds[0].d || kds[k].d != 2) return false;
        if (maxV[k] >= 65535 || kds[k].Epad >= 65535) return false;     // u16 row pointers / neighbour ids
    }
    for (int all = 1; all >= 0; --all) {
        FusedLayout L{};
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return (int)r; };
        L.prod_all = all;
        L.q = take((size_t)NA * sizeof(float2));
        size_t shared_prod = 0;
        for (int k = 0; k < c.K; ++k) {
            L.Ecap[k] = NA * kds[k].D1;
            L.Vcap[k] = maxV[k];
            L.val[k][0] = take((size_t)(maxV[k] + 1) * sizeof(float2));
            L.val[k][1] = take((size_t)(maxV[k] + 1) * sizeof(float2));
            L.nbr[k] = take((size_t)kds[k].D1 * maxV[k] * sizeof(unsigned));
            L.row[k] = take((size_t)(maxV[k] + 2) * sizeof(unsigned short));
            const size_t pb = (size_t)L.Ecap[k] * 2 * sizeof(float);
            if (all) L.prod[k] = take(pb);
            else shared_prod = pb > shared_prod ? pb : shared_prod;
        }
        if (!all) {
            const int p = take(shared_prod);
            for (int k = 0; k < c.K; ++k) L.prod[k] = p;
        }
        L.total = (int)o;
        if (o <= kLdsLimit) {
            *lay = L;
            return true;
        }
    }
    return false;
}

template <int PPT, int K>
void launch_fused(const CrfDev &c, const FusedArgs &a, hipStream_t s)
{
    auto fn = k_fused<PPT, K, 2, (PPT <= 2)>;
    // per (function, device); cheap enough to repeat and safe with several devices in one process
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kLdsLimit);
    fn<<<dim3(c.F), dim3(kNT), a.lay.total, s>>>(c, a);
}

}  // namespace

bool fused_supported(const CrfDev &c, const KernelDev *kds, const int *maxV, const int * /*maxRow*/,
                     size_t *lds_bytes)
{
    FusedLayout lay;
    const bool ok = make_layout(c, kds, maxV, &lay);
    if (lds_bytes) *lds_bytes = ok ? (size_t)lay.total : 0;
    return ok;
}

void launch_inference_fused(const CrfDev &c, const KernelDev *kds, const int *maxV, const int * /*maxRow*/,
                            int n_iter, int with_map, float relax, hipStream_t s)
{
    FusedArgs a{};
    if (!make_layout(c, kds, maxV, &a.lay)) return;
    for (int k = 0; k < c.K; ++k) a.kd[k] = kds[k];
    a.n_iter = n_iter;
    a.with_map = with_map;
    a.relax = relax;
    static long long *timing_buf = nullptr;
    static const bool want_timing = getenv("LCCRF_FUSED_TIMING") != nullptr;
    if (want_timing && !timing_buf) (void)hipMalloc(&timing_buf, 64 * sizeof(long long));
    a.timing = want_timing ? timing_buf : nullptr;
    const int ppt = ((c.activeN > 0 ? c.activeN : c.maxN) + kNT - 1) / kNT;
#define FUSED_CASE(P)                                            \
    case P:                                                      \
        if (c.K == 1) launch_fused<P, 1>(c, a, s);               \
        else launch_fused<P, 2>(c, a, s);                        \
        break;
    switch (ppt) {
        FUSED_CASE(1)
        FUSED_CASE(2)
        FUSED_CASE(3)
        FUSED_CASE(4)
    default: break;
    }
#undef FUSED_CASE
    if (a.timing) {                       // debug only: synchronous read-back of workgroup 0's phase stamps
        long long h[64];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, a.timing, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[lccrf fused timing] %lld stamps, deltas (shader clocks):", h[63]);
        for (int i = 1; i < h[63] && i < 63; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
        fprintf(stderr, "\n");
    }
}

}  // namespace lccrf
