// fused_engine.hip -- SLAM-size inference as ONE kernel launch: one 512-lane workgroup per
// frame runs startInference, every mean-field iteration and buildMap without leaving the CU.
//
// Why: at SLAM sizes (N ~ 2000 keypoints, V ~ 1.2k lattice vertices) one iteration moves
// < 0.5 MB; a launch-per-phase design is bound by launch gaps, not by memory (SURVEY.md
// section 7).  Here the mean-field state lives on chip and the loop touches no HBM:
//     LDS        Q[N][2], both ping-pong lattice value arrays of every kernel, and the
//                per-entry splat products w*Q
//     registers  for every point a lane owns: the LDS addresses of its d+1 vertices, the
//                weights bary*alpha, w*norm, its unary; for every splat entry it owns: the
//                weight and the LDS address of the contributing point's Q; for every lattice
//                vertex it owns: the LDS addresses of its blur neighbours and its CSR row.
//                All of it is loaded once per launch (8 wavefronts -> 256 VGPRs per lane).
//
// What bounds it: measured with shader-clock stamps, the loop is bound by VALU issue inside
// the CU (not HBM, not LDS bandwidth) plus one serial tail -- see below.  Hence: addresses are
// precomputed, two labels ride in one float2, softmax needs one exp (L = 2).
//
// Bit-exactness: the reference splats sequentially over points (permutohedral_cpu.h:653-661),
// so a vertex's value is a left-to-right fp32 sum in ascending point order.  Phase P forms all
// products w*Q in parallel (exact, order-free); phase S adds each vertex's products strictly
// left to right.  The appearance kernel funnels hundreds of keypoints into one vertex (rows of
// ~450 products), so its rows get a dedicated lane per (vertex, label) that streams the row
// through a 2 x 32-deep register prefetch ring: the chain then runs at the fp32 add latency
// instead of the LDS latency.  Nothing is re-associated, nothing is fused (-ffp-contract=off).
//
// Specialised for L = 2 labels (the SLAM configuration, src/Tracking.cc:1919) and 2-D kernels
// (pairwise3d.h:37-71); anything else runs on the streaming engine with identical results.
#include "engine.h"
#include "device_math.h"

#include <cstdio>
#include <cstdlib>

namespace lccrf {

namespace {

constexpr int kNT = 512;                  // lanes per workgroup: 8 wavefronts, 2 per SIMD, 256 VGPRs each
constexpr int kD1 = 3;                    // d + 1 for the 2-D kernels this engine handles
constexpr int kVPT = 4;                   // lattice vertices (all kernels together) per lane
constexpr int kMaxPPT = 7;                // points per lane -> N <= 3584
constexpr int kMaxFusedK = 2;
constexpr int kLongRow = 24;              // a kernel whose longest splat row exceeds this uses chain lanes
constexpr size_t kLdsLimit = 160 * 1024;  // MI355X: 160 KiB LDS per CU, one workgroup may own it all

struct FusedLayout {                      // byte offsets into dynamic LDS
    int q;                                // float2 [maxN]
    int prod[kMaxFusedK];                 // splat products: float2 [Ecap] (short rows) or float [2][Ecap] (long)
    int val[kMaxFusedK][2];               // float2 [V_k+1], slot 0 = absent neighbour = 0
    int Ecap;                             // entries per product array (multiple of 4, padded for over-reads)
    int long_mode[kMaxFusedK];            // 1: label-major products + chain lanes
    int chain_base[kMaxFusedK];           // first chain lane of a long kernel
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

#define STAMP()                                                                                        \
    do {                                                                                               \
        if (a.timing && blockIdx.x == 0 && tid == 0 && n_stamp < 62) a.timing[n_stamp++] = clock64();  \
    } while (0)

// expAndNormalize for two labels (densecrf3d.h:70-98).  One of the two fast_exp arguments is
// exactly 0 (value minus row maximum) and fast_exp(0) == 1, so a single exp is evaluated; the
// sum and the two IEEE divisions are the reference's.
__device__ __forceinline__ float2 softmax2(float a, float b, float2 old, float relax)
{
    const bool lt = a < b;                            // mx = b iff a < b (densecrf3d.h:76-79)
    const float e = fast_exp_nonpos(lt ? a - b : b - a);
    const float v0 = lt ? e : 1.0f, v1 = lt ? 1.0f : e;
    const float tt = v0 + v1;
    const float p0 = v0 / tt, p1 = v1 / tt;
    if (relax == 1) return make_float2(p0, p1);
    return make_float2((1 - relax) * old.x + relax * p0, (1 - relax) * old.y + relax * p1);
}

template <int PPT, int K, bool CSR_REG>
__global__ void __launch_bounds__(kNT) k_fused(CrfDev c, FusedArgs a)
{
    constexpr int EPT = PPT * kD1;        // splat entries per lane and kernel = ceil(N*D1 / kNT)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int N = c.n_points[f];
    const int E = N * kD1;
    int n_stamp = 0;
    STAMP();

    // Q and the lattice value arrays sit in the first 64 KiB of LDS (make_layout), so their byte
    // addresses fit 16 bits and are kept two per register.
    auto lo16 = [](unsigned x) { return (int)(x & 0xffffu); };
    auto hi16 = [](unsigned x) { return (int)(x >> 16); };
    auto ld1 = [&](int addr) { return *reinterpret_cast<const float *>(smem + addr); };
    auto ld2 = [&](int addr) { return *reinterpret_cast<const float2 *>(smem + addr); };
    auto ld4 = [&](int addr) { return *reinterpret_cast<const float4 *>(smem + addr); };
    auto st1 = [&](int addr, float v) { *reinterpret_cast<float *>(smem + addr) = v; };
    auto st2 = [&](int addr, float2 v) { *reinterpret_cast<float2 *>(smem + addr) = v; };

    // ---- points a lane owns: i = tid + s*kNT ----------------------------------------------
    float2 un[PPT], q[PPT];
    unsigned gaddr[PPT][K][2];            // LDS addresses of val_final[k][offset+1]: {j0 | j1<<16, j2}
    float wgt[PPT][K][kD1];               // bary * alpha               permutohedral_cpu.h:689
    float wn[PPT][K];                     // w_ * norm_[i]              pairwise3d.h:77
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int i = tid + s * kNT;
        un[s] = q[s] = make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            wn[s][k] = 0.f;
#pragma unroll
            for (int j = 0; j < kD1; ++j) wgt[s][k][j] = 0.f;
            gaddr[s][k][0] = gaddr[s][k][1] = 0;
        }
        if (i < N) {
            un[s] = reinterpret_cast<const float2 *>(c.unary)[(size_t)f * c.maxN + i];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const KernelDev &kd = a.kd[k];
                const size_t e0 = (size_t)f * kd.Epad + (size_t)i * kD1;
#pragma unroll
                for (int j = 0; j < kD1; ++j) {
                    const unsigned ga = (unsigned)(a.lay.val[k][kD1 & 1] + (kd.offset[e0 + j] + 1) * 8);
                    if (j == 0) gaddr[s][k][0] = ga;
                    if (j == 1) gaddr[s][k][0] |= ga << 16;
                    if (j == 2) gaddr[s][k][1] = ga;
                    wgt[s][k][j] = kd.bary[e0 + j] * kd.alpha;
                }
                wn[s][k] = kd.w * kd.norm[(size_t)f * kd.maxN + i];
            }
            q[s] = softmax2(-1.0f * un[s].x, -1.0f * un[s].y, q[s], 1.0f);   // startInference, base.h:78-80
            st2(a.lay.q + i * 8, q[s]);
        }
    }

    // ---- splat entries a lane owns: p = tid + u*kNT (CSR order) ----------------------------
    float cw[CSR_REG ? K : 1][CSR_REG ? EPT : 1];
    unsigned pa[CSR_REG ? K : 1][CSR_REG ? (EPT + 1) / 2 : 1];   // LDS addresses of Q[contributing point], two per register
    if constexpr (CSR_REG) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float *gw = a.kd[k].csr_w + (size_t)f * a.kd[k].Epad;
            const int *gp = a.kd[k].csr_pt + (size_t)f * a.kd[k].Epad;
#pragma unroll
            for (int u = 0; u < (EPT + 1) / 2; ++u) pa[k][u] = 0;
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int p = tid + u * kNT;
                cw[k][u] = (p < E) ? gw[p] : 0.0f;
                pa[k][u >> 1] |= (unsigned)(a.lay.q + ((p < E) ? gp[p] : 0) * 8) << ((u & 1) * 16);
            }
        }
    }

    // ---- lattice vertices a lane owns: all kernels concatenated, g = tid + t*kNT -------------
    int V[K];
#pragma unroll
    for (int k = 0; k < K; ++k) V[k] = a.kd[k].V[f];
    int own0[kVPT], own1[kVPT];           // LDS address of the vertex in ping / pong (0: no vertex)
    unsigned nb[kVPT][kD1];               // LDS addresses of its two neighbours per blur pass: n1 | n2<<16
    int rowaddr[kVPT], rowlen[kVPT];      // its splat row (short-row kernels); len < 0: none
    int rowk[kVPT];
#pragma unroll
    for (int t = 0; t < kVPT; ++t) {
        int g = tid + t * kNT, k = 0;
        if (K > 1 && g >= V[0]) { g -= V[0]; k = 1; }
        const bool valid = g < V[K > 1 ? k : 0];
        own0[t] = own1[t] = 0;
        rowaddr[t] = 0;
        rowlen[t] = -1;
        rowk[t] = k;
#pragma unroll
        for (int j = 0; j < kD1; ++j) nb[t][j] = 0;
        if (valid) {
            const KernelDev &kd = a.kd[k];
            own0[t] = a.lay.val[k][0] + (g + 1) * 8;
            own1[t] = a.lay.val[k][1] + (g + 1) * 8;
            const int2 *gn = reinterpret_cast<const int2 *>(kd.nbr) + (size_t)f * kD1 * kd.Epad;
#pragma unroll
            for (int j = 0; j < kD1; ++j) {
                const int2 n = gn[(size_t)j * kd.Epad + g];
                nb[t][j] = (unsigned)(a.lay.val[k][j & 1] + (n.x + 1) * 8) |
                           ((unsigned)(a.lay.val[k][j & 1] + (n.y + 1) * 8) << 16);
            }
            const int *gr = kd.rowptr + (size_t)f * (kd.Epad + 1);
            const int r0 = gr[g], r1 = gr[g + 1];
            rowaddr[t] = a.lay.prod[k] + r0 * 8;
            rowlen[t] = a.lay.long_mode[k] ? -1 : r1 - r0;
        }
    }

    // ---- chain lanes: one lane per (vertex, label) of a long-row kernel ---------------------
    int chain_addr = 0, chain_len = -1, chain_out = 0, chain_k = -1;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (a.lay.long_mode[k]) {
            const int idx = tid - a.lay.chain_base[k];
            if (idx >= 0 && idx < 2 * V[k]) {
                const int v = idx >> 1, l = idx & 1;
                const int *gr = a.kd[k].rowptr + (size_t)f * (a.kd[k].Epad + 1);
                const int r0 = gr[v], r1 = gr[v + 1];
                chain_addr = a.lay.prod[k] + (l * a.lay.Ecap + r0) * 4;
                chain_len = r1 - r0;
                chain_out = a.lay.val[k][0] + (v + 1) * 8 + l * 4;
                chain_k = k;
            }
        }
    }

    if (tid < K) {                        // the "absent neighbour" slot of both buffers
        st2(a.lay.val[tid][0], make_float2(0.f, 0.f));
        st2(a.lay.val[tid][1], make_float2(0.f, 0.f));
    }
    __syncthreads();
    STAMP();

    for (int it = 0; it < a.n_iter; ++it) {
        // The packed address registers are loop-invariant; without this the compiler hoists their
        // unpacked halves out of the loop and spills.  The empty asm makes each one opaque per trip.
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("" : "+v"(gaddr[s][k][0]), "+v"(gaddr[s][k][1]));
        if constexpr (CSR_REG) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int u = 0; u < (EPT + 1) / 2; ++u) asm volatile("" : "+v"(pa[k][u]));
        }
#pragma unroll
        for (int t = 0; t < kVPT; ++t)
#pragma unroll
            for (int j = 0; j < kD1; ++j) asm volatile("" : "+v"(nb[t][j]));

        // ---- P: products w * Q[point] for every splat entry --------------------------------
        auto phase_P = [&](int k) {
            const bool lm = a.lay.long_mode[k] != 0;
            const int wbase = a.lay.prod[k] + tid * (lm ? 4 : 8);
            const int lstride = a.lay.Ecap * 4;
            if constexpr (CSR_REG) {
                constexpr int CH = (EPT % 6 == 0) ? 6 : kD1;            // gathers in flight per group
#pragma unroll
                for (int u0 = 0; u0 < EPT; u0 += CH) {
                    float2 x[CH];
#pragma unroll
                    for (int u = 0; u < CH; ++u)
                        x[u] = ld2(((u0 + u) & 1) ? hi16(pa[k][(u0 + u) >> 1]) : lo16(pa[k][(u0 + u) >> 1]));
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        if (tid + (u0 + u) * kNT < E) {
                            const float2 r = make_float2(cw[k][u0 + u] * x[u].x, cw[k][u0 + u] * x[u].y);
                            if (lm) {
                                st1(wbase + (u0 + u) * kNT * 4, r.x);
                                st1(wbase + (u0 + u) * kNT * 4 + lstride, r.y);
                            } else {
                                st2(wbase + (u0 + u) * kNT * 8, r);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);                   // keep the groups apart: bounded registers
                }
            } else {
                const float *gw = a.kd[k].csr_w + (size_t)f * a.kd[k].Epad;
                const int *gp = a.kd[k].csr_pt + (size_t)f * a.kd[k].Epad;
#pragma unroll 1
                for (int u0 = 0; u0 < EPT; u0 += kD1) {                 // D1 entries at a time: bounded registers
                    float w[kD1];
                    int pt[kD1];
#pragma unroll
                    for (int u = 0; u < kD1; ++u) {
                        const int p = tid + (u0 + u) * kNT;
                        w[u] = (p < E) ? gw[p] : 0.0f;
                        pt[u] = (p < E) ? gp[p] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < kD1; ++u) {
                        const int p = tid + (u0 + u) * kNT;
                        const float2 x = ld2(a.lay.q + pt[u] * 8);
                        if (p < E) {
                            const float2 r = make_float2(w[u] * x.x, w[u] * x.y);
                            if (lm) {
                                st1(wbase + (u0 + u) * kNT * 4, r.x);
                                st1(wbase + (u0 + u) * kNT * 4 + lstride, r.y);
                            } else {
                                st2(wbase + (u0 + u) * kNT * 8, r);
                            }
                        }
                    }
                }
            }
        };

        // ---- S: ordered row sums -----------------------------------------------------------
        auto phase_S_short = [&](int k) {
            // a vertex's (short) row is fetched eight products at a time, then added in order
#pragma unroll
            for (int t = 0; t < kVPT; ++t) {
                const int n = rowlen[t];
                if (n < 0 || rowk[t] != k) continue;
                float a0 = 0.0f, a1 = 0.0f;
                for (int cidx = 0; cidx < n; cidx += 8) {
                    float2 x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = ld2(rowaddr[t] + (cidx + u) * 8);   // buffer is padded
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (cidx + u < n) { a0 += x[u].x; a1 += x[u].y; }
                }
                st2(own0[t], make_float2(a0, a1));
            }
        };
        auto phase_S_chain = [&](int k) {
            // one lane per (vertex, label): the row streams through a 2 x 32-deep register ring so
            // the strictly ordered additions never wait for LDS.
            if (chain_len < 0 || chain_k != k) return;
            const int n = chain_len;
            int addr = chain_addr;
            float acc = 0.0f;
            int p = 0;
            while (p < n && (addr & 15)) { acc += ld1(addr); addr += 4; ++p; }     // reach 16-byte alignment
            // Blocks of 16 products, two register buffers A/B used alternately: the loads of the
            // next block are issued BEFORE the (strictly ordered) additions of the current one;
            // sched_barrier keeps the compiler from folding the pipeline back into load->use.
            const int nblk = (n - p) >> 4;
            if (nblk > 0) {
                float4 A[4], B[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) A[u] = ld4(addr + u * 16);
                int b = 0;
#pragma unroll 1
                for (; b + 2 <= nblk; b += 2) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) B[u] = ld4(addr + (b + 1) * 64 + u * 16);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { acc += A[u].x; acc += A[u].y; acc += A[u].z; acc += A[u].w; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) A[u] = ld4(addr + (b + 2) * 64 + u * 16);   // may over-read into the pad
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { acc += B[u].x; acc += B[u].y; acc += B[u].z; acc += B[u].w; }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (b < nblk) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { acc += A[u].x; acc += A[u].y; acc += A[u].z; acc += A[u].w; }
                }
                addr += nblk * 64;
                p += nblk * 16;
            }
            if (p < n) {                                             // < 16 left: up to 3 aligned quads, then singles
                const int nq = (n - p) >> 2;
                float4 t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t4[u] = ld4(addr + u * 16);           // over-read stays inside the buffer pad
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (u < nq) { acc += t4[u].x; acc += t4[u].y; acc += t4[u].z; acc += t4[u].w; }
                const float4 last = nq == 0 ? t4[0] : nq == 1 ? t4[1] : nq == 2 ? t4[2] : t4[3];
                p += nq * 4;
                if (p < n) acc += last.x;
                if (p + 1 < n) acc += last.y;
                if (p + 2 < n) acc += last.z;
            }
            st1(chain_out, acc);
        };

        if (a.lay.prod_all) {
#pragma unroll
            for (int k = 0; k < K; ++k) phase_P(k);
            __syncthreads();
            STAMP();
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (a.lay.long_mode[k]) phase_S_chain(k);
                else phase_S_short(k);
                STAMP();
            }
            __syncthreads();
            STAMP();
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                phase_P(k);
                __syncthreads();
                if (a.lay.long_mode[k]) phase_S_chain(k);
                else phase_S_short(k);
                __syncthreads();
            }
            STAMP();
            STAMP();
        }

        // ---- d+1 Jacobi blur passes, permutohedral_cpu.h:663-679 ------------------------------
#pragma unroll
        for (int j = 0; j < kD1; ++j) {
#pragma unroll
            for (int t = 0; t < kVPT; ++t) {
                if (own0[t]) {
                    const float2 o = ld2((j & 1) ? own1[t] : own0[t]);
                    const float2 x = ld2(lo16(nb[t][j])), y = ld2(hi16(nb[t][j]));
                    st2((j & 1) ? own0[t] : own1[t],
                        make_float2(o.x + 0.5f * (x.x + y.x), o.y + 0.5f * (x.y + y.y)));
                }
            }
            __syncthreads();
        }
        STAMP();

        // ---- X: slice + apply + softmax per point -------------------------------------------------
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * kNT;
            if (i < N) {
                float2 x[K][kD1];
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int j = 0; j < kD1; ++j)
                        x[k][j] = ld2(j == 0 ? lo16(gaddr[s][k][0]) : j == 1 ? hi16(gaddr[s][k][0]) : (int)gaddr[s][k][1]);
                float n0 = -un[s].x, n1 = -un[s].y;                       // stepInit, densecrf3d.h:154-158
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float t0 = 0.0f, t1 = 0.0f;                           // slice, permutohedral_cpu.h:684-694
#pragma unroll
                    for (int j = 0; j < kD1; ++j) {
                        t0 += wgt[s][k][j] * x[k][j].x;
                        t1 += wgt[s][k][j] * x[k][j].y;
                    }
                    n0 += wn[s][k] * t0;                                  // apply, pairwise3d.h:77
                    n1 += wn[s][k] * t1;
                }
                q[s] = softmax2(1.0f * n0, 1.0f * n1, q[s], a.relax);     // densecrf_base.h:90
                st2(a.lay.q + i * 8, q[s]);
            }
        }
        __syncthreads();
        STAMP();
    }

    // ---- results ------------------------------------------------------------------------------
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

bool make_layout(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, FusedLayout *lay)
{
    if (c.L != 2 || c.K < 1 || c.K > kMaxFusedK) return false;
    if (c.maxN < 1 || c.maxN > kMaxPPT * kNT) return false;
    int vtot = 0;
    for (int k = 0; k < c.K; ++k) {
        if (kds[k].d != 2) return false;
        vtot += maxV[k];
    }
    if (vtot > kVPT * kNT) return false;
    // chain lanes for long-row kernels, as long as lanes remain
    int long_mode[kMaxFusedK] = {0, 0}, chain_base[kMaxFusedK] = {0, 0}, next_lane = 0;
    for (int k = 0; k < c.K; ++k) {
        if (maxRow[k] > kLongRow && next_lane + 2 * maxV[k] <= kNT) {
            long_mode[k] = 1;
            chain_base[k] = next_lane;
            next_lane += (2 * maxV[k] + 63) & ~63;                  // whole wavefronts
        }
    }
    for (int all = 1; all >= 0; --all) {
        FusedLayout L{};
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return (int)r; };
        L.prod_all = all;
        L.q = take((size_t)c.maxN * sizeof(float2));
        L.Ecap = ((c.maxN * kD1 + 3) & ~3) + 64;                     // pad: rows are over-read by < 64 floats
        for (int k = 0; k < c.K; ++k) {
            L.long_mode[k] = long_mode[k];
            L.chain_base[k] = chain_base[k];
            L.val[k][0] = take((size_t)(maxV[k] + 1) * sizeof(float2));
            L.val[k][1] = take((size_t)(maxV[k] + 1) * sizeof(float2));
        }
        if (o > 65536) return false;                                 // 16-bit LDS addresses for Q and val
        const size_t pb = (size_t)L.Ecap * sizeof(float2);
        if (all) {
            for (int k = 0; k < c.K; ++k) L.prod[k] = take(pb);
        } else {
            const int p = take(pb);
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
    auto fn = k_fused<PPT, K, (PPT <= 4)>;
    // per (function, device); cheap enough to repeat and safe with several devices in one process
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kLdsLimit);
    fn<<<dim3(c.F), dim3(kNT), a.lay.total, s>>>(c, a);
}

}  // namespace

bool fused_supported(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, size_t *lds_bytes)
{
    FusedLayout lay;
    const bool ok = make_layout(c, kds, maxV, maxRow, &lay);
    if (lds_bytes) *lds_bytes = ok ? (size_t)lay.total : 0;
    return ok;
}

void launch_inference_fused(const CrfDev &c, const KernelDev *kds, const int *maxV, const int *maxRow, int n_iter,
                            int with_map, float relax, hipStream_t s)
{
    FusedArgs a{};
    if (!make_layout(c, kds, maxV, maxRow, &a.lay)) return;
    for (int k = 0; k < c.K; ++k) a.kd[k] = kds[k];
    a.n_iter = n_iter;
    a.with_map = with_map;
    a.relax = relax;
    static long long *timing_buf = nullptr;
    static const bool want_timing = getenv("LCCRF_FUSED_TIMING") != nullptr;
    if (want_timing && !timing_buf) (void)hipMalloc(&timing_buf, 64 * sizeof(long long));
    a.timing = want_timing ? timing_buf : nullptr;
    const int ppt = (c.maxN + kNT - 1) / kNT;
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
        FUSED_CASE(5)
        FUSED_CASE(6)
        FUSED_CASE(7)
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
