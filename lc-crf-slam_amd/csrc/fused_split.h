// fused_split.h -- the 1024-lane mean-field loop for frames whose two product buffers do not fit a CU's LDS side by side (3000-keypoint
// frames: BASELINE config 4), with the chain kernel's product plane SPLIT in two (round 6).
//
// fused_loop.h gives such a frame ONE product buffer that the kernels take in turn -- P(0) | S(0): the appearance kernel's ordered row
// sums, three wavefront pairs at most, ~6.8k cycles at 3000 points | P(1) | S(1) | blur | X -- and while the chain is added the other
// thirteen wavefronts wait, the short-row sums of kernel 1 (3.2k cycles, 13 % of the launch) among the things they could be doing.
// Here kernel 1 keeps a buffer of its own (72 KB) and the chain kernel's plane holds the products of HALF the points at a time: the
// points below `lay.half` are a prefix of every row (rows are in ascending point order, quirk Q6), so
//     P(1) + P(0, first half)  |  S(1) beside S(0, first half)  |  P(0, second half)  |  S(0, second half: the ring goes on from its sum)  |  blur | X
// adds the same products in the same order -- the same bits -- with kernel 1's row sums hidden under the first half of the chain.
// A row's region is as long as the longer of its two halves (each padded to four products; eight zeros behind the 16 longest rows for
// chain_rows' address clamp): ~55 % of the whole row for keypoints in random order.  Which products go where is decided once per
// build by the prepare launch (split_prepare; the blocks of fused_lean.h: LeanPrepPlan) -- this plan only exists from prepared launch
// records; a frame whose regions do not fit the plane raises a flag and the batch keeps fused_loop.h's shared-buffer plan.
#pragma once

#include "fused_lean.h"

namespace lccrf {
namespace fl {

// LDS plan: [zeros 128][per kernel: values x 2, neighbour table, row starts][kernel 1's products][the chain kernel's half plane: what is left]
__host__ __device__ inline bool layout_split(int NA, const int *V, int row0, FusedLayout *lay, size_t lds_limit = kLdsLimit)
{
    constexpr int D1 = kD1, K = 2;
    if (NA < 1 || NA > 4 * kNT || row0 < kChainMinRow || V[0] > chain_max_v(kNT) || V[0] > 256) return false;
    for (int k = 0; k < K; ++k)
        if (V[k] >= 65535) return false;
    FusedLayout L{};
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return (int)r; };
    L.prod_all = 1;
    L.chain0 = 1;
    L.split0 = 1;
    L.half = ((NA + 1) / 2 + 63) & ~63;                  // (a multiple of 64: which half a point slot's lanes belong to is uniform per wavefront)
    L.zero = take(128);                                  // LDS bytes [0, 128) are zeros (chain_rows_sel reads them by absolute address)
    for (int k = 0; k < K; ++k) {
        L.val[k][0] = take((size_t)(V[k] + 1) * sizeof(float2));
        L.val[k][1] = take((size_t)(V[k] + 1) * sizeof(float2));
        L.nbr[k] = take((size_t)D1 * V[k] * sizeof(unsigned));
        L.row[k] = take((size_t)(V[k] + 2) * sizeof(unsigned short));
    }
    L.Ecap[1] = (NA * D1 + 63) & ~63;
    L.prod[1] = take((size_t)L.Ecap[1] * sizeof(float2));
    if (o + 4096 > lds_limit) return false;
    long cap = (long)((lds_limit - o - 16) / (2 * sizeof(float))) & ~63l;          // floats per label plane
    if (cap > 65472) cap = 65472;                                                  // (16-bit product slots)
    // not worth a prepare launch unless rows of balanced halves fit: (3 NA / 2) products + padding to four per half + the top rows' zeros
    if (cap < (long)NA * D1 / 2 + 4 * V[0] + 8 * kChainTop + 64) return false;
    L.Ecap[0] = (int)cap;
    L.prod[0] = take((size_t)cap * 2 * sizeof(float));
    L.total = (int)o;
    if (o > lds_limit) return false;
    *lay = L;
    return true;
}

// What a chain lane knows about its row in the split plan:
//   a = row address (label plane of the lane, 18 bits) | quads of the first half << 18 (11 bits) | its pad slots << 29
//   b = quads of the second half (11 bits) | its pad slots << 11 | output index << 16
// (the wavefront's longest half is formed where it is needed: five lane exchanges beside a ring of thousands of cycles)
__device__ __forceinline__ unsigned split_quads(const ChainLane &cl, int h) { return h == 0 ? ((cl.a >> 18) & 0x7ffu) : (cl.b & 0x7ffu); }
__device__ __forceinline__ unsigned split_pads(const ChainLane &cl, int h) { return h == 0 ? (cl.a >> 29) : ((cl.b >> 11) & 3u); }

// The prepare launch's part (k_fused<.., MODE 1>): ranks the chain rows, finds every row's split point, places the regions, forms the
// lanes' chain words and every point's product slots.  `pk` as in place_products: (vertex id + 1) | place in the CSR << 16.
// Returns false (uniformly) when the regions do not fit the plane.  Scratch: the head of kernel 1's product buffer.
template <int PPT, int NT>
__device__ __forceinline__ bool split_prepare(unsigned char *smem, const FusedLayout &lay, const int (&V)[2], int N, int tid,
                                              const unsigned (&pk)[PPT][2][kD1], PointRegs<PPT, 2> &pr, ChainLane &cl)
{
    constexpr int D1 = kD1;
    const int V0 = V[0], H = lay.half;
    int *hist = reinterpret_cast<int *>(smem + lay.prod[1]);                      // [64] counts, [64] bases, [1] fits
    int *n1 = hist + 160;                                                         // [V0] products of points < H in row v
    unsigned short *srt = reinterpret_cast<unsigned short *>(n1 + 256);           // [V0] vertex of rank r
    unsigned short *pstart = srt + 256;                                           // [V0] first slot of row v's region
    const unsigned short *row0 = reinterpret_cast<const unsigned short *>(smem + lay.row[0]);
    if (tid < 160) hist[tid] = 0;
    if (tid < 256) n1[tid] = 0;
    __syncthreads();
    int key = 0;
    if (tid < V0) {
        const int len = (int)row0[tid + 1] - (int)row0[tid];
        key = 63 - min((len + 3) >> 4, 63);
        atomicAdd(&hist[key], 1);
    }
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int i = tid + s * NT;
        if (i < N && i < H) {
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                const int v = (int)(pk[s][0][j] & 0xffffu) - 1;
                atomicMax(&n1[v], (int)(pk[s][0][j] >> 16) - (int)row0[v] + 1);   // rows are in point order: the first half is a prefix
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int x = hist[tid];
        int incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o, 64);
            if (tid >= o) incl += y;
        }
        hist[64 + tid] = incl - x;
    }
    __syncthreads();
    if (tid < V0) srt[atomicAdd(&hist[64 + key], 1)] = (unsigned short)tid;
    __syncthreads();
    if (tid < 64) {                                       // regions in rank order, four rows per lane (V0 <= 256)
        int sz[4], sum = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = tid * 4 + u;
            sz[u] = 0;
            if (r < V0) {
                const int v = srt[r];
                const int len = (int)row0[v + 1] - (int)row0[v], a = n1[v], b = len - a;
                sz[u] = max((a + 3) & ~3, (b + 3) & ~3) + (r < kChainTop ? 8 : 0);
            }
            sum += sz[u];
        }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o, 64);
            if (tid >= o) incl += y;
        }
        int base = incl - sum;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = tid * 4 + u;
            if (r < V0) pstart[srt[r]] = (unsigned short)min(base, 65535);
            base += sz[u];
        }
        if (tid == 63) hist[128] = incl <= lay.Ecap[0] ? 1 : 0;
    }
    __syncthreads();
    const bool fits = hist[128] != 0;
    // chain lanes as chain_setup: wavefront pair p owns labels 0 / 1 of a rank range, pair 0 the kChainTop longest rows
    const int l = (tid >> 6) & 1, pair = tid >> 7, ln = tid & 63;
    const int r = pair == 0 ? (((ln & 0x18) == 0) ? ((ln & 7) | ((ln >> 5) << 3)) : V0) : kChainTop + ((pair - 1) << 6) + ln;
    cl.a = cl.b = 0u;
    if (r < V0 && fits) {
        const int v = srt[r];
        const int len = (int)row0[v + 1] - (int)row0[v], a = n1[v], b = len - a;
        const unsigned q1 = (unsigned)((a + 3) >> 2), q2 = (unsigned)((b + 3) >> 2);
        const unsigned addr = (unsigned)(lay.prod[0] + 4 * (l * lay.Ecap[0] + (int)pstart[v]));          // < 2^18
        cl.a = addr | (q1 << 18) | ((unsigned)(((a + 3) & ~3) - a) << 29);
        cl.b = q2 | ((unsigned)(((b + 3) & ~3) - b) << 11) | ((unsigned)((v + 1) * 2 + l) << 16);
    }
    // product slots: kernel 1 at its CSR place, the chain kernel inside its row's region, counted from the start of its half
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            pr.ix[s][k][0] = (pk[s][k][0] & 0xffffu) | (pk[s][k][1] << 16);
            pr.ix[s][k][1] = pk[s][k][2] & 0xffffu;
            pr.ix[s][k][2] = 0;
            if (tid + s * NT < N) {
                unsigned sl[D1];
#pragma unroll
                for (int j = 0; j < D1; ++j) {
                    sl[j] = pk[s][k][j] >> 16;
                    if (k == 0) {
                        const int v = (int)(pk[s][k][j] & 0xffffu) - 1;
                        const int rank = (int)(pk[s][k][j] >> 16) - (int)row0[v], a = n1[v];
                        sl[j] = (unsigned)((int)pstart[v] + (rank < a ? rank : rank - a));
                    }
                }
                pr.ix[s][k][1] |= sl[0] << 16;
                pr.ix[s][k][2] = sl[1] | (sl[2] << 16);
            }
        }
    }
    __syncthreads();                                      // (the scratch becomes kernel 1's product buffer again)
    return fits;
}

// splat + blur of one iteration on the split plan (see the top of the file).  On return val[k][kD1 & 1] holds the blurred values.
template <int PPT, int NT>
__device__ __forceinline__ void splat_blur_split(unsigned char *smem, const FusedLayout &lay, const int (&V)[2], int N, int tid,
                                                 const PointRegs<PPT, 2> &pr, ChainLane &cl, Instr &ins)
{
    constexpr int D1 = kD1, K = 2;
    const int H = lay.half;
    auto chain_products = [&](int h) {                    // P(0, half h) + the zeros behind this half of the lane's row
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * NT;
            if (i < N && ((i < H) == (h == 0))) point_products<PPT, K, 1>(smem, lay, pr, s, 0);
        }
        if (cl.b >> 16) {
            float *e = reinterpret_cast<float *>(smem + (cl.a & 0x3ffffu)) + split_quads(cl, h) * 4u;
            const unsigned npad = split_pads(cl, h);
            for (unsigned z = 1; z <= npad; ++z) e[-(int)z] = 0.0f;
            if ((tid >> 7) == 0) {                        // (the first pair's ring reads past the end of its rows: chain_rows' address clamp)
                reinterpret_cast<float4 *>(e)[0] = make_float4(0.f, 0.f, 0.f, 0.f);
                reinterpret_cast<float4 *>(e)[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    const int npairs = 1 + ((max(V[0] - kChainTop, 0) + 63) >> 6);
    auto chain_sum = [&](int h, float acc) {              // S(0, half h): whole wavefronts; returns the lane's running sum
        if ((tid >> 7) < npairs) {
            FL_PSTAMP();
            if ((tid >> 7) == 0) __builtin_amdgcn_s_setprio(3);
            asm volatile("" : "+v"(cl.a), "+v"(cl.b));
            unsigned q = (cl.b >> 16) != 0 ? split_quads(cl, h) : 0u;
            unsigned m = q;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
            m = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
            if ((cl.b >> 16) != 0) {
                const unsigned row_addr = cl.a & 0x3ffffu;
                acc = (tid >> 7) == 0 ? chain_rows_keep(row_addr, row_addr + q * 16u, (((m + 1u) >> 1) + 3u) >> 2, cl.a, cl.b, acc)
                                      : chain_rows_sel(row_addr, (int)(q * 16u), (m + 7u) >> 3, cl.a, cl.b, acc);
            }
            if ((tid >> 7) == 0) __builtin_amdgcn_s_setprio(0);
            FL_PSTAMP();
        }
        return acc;
    };
    // ---- P(1) + P(0, first half) ----------------------------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < PPT; ++s)
        if (tid + s * NT < N) point_products<PPT, K, 1>(smem, lay, pr, s, 1);
    chain_products(0);
    __syncthreads();
    FL_STAMP();
    // ---- S(0, first half) on the chain wavefronts; S(1) on everybody behind the first pair --------------------------------------
    float acc = chain_sum(0, 0.0f);
    if (tid >= 128) {
        const float2 *pl = reinterpret_cast<const float2 *>(smem + lay.prod[1]);
        const float2 *zero = reinterpret_cast<const float2 *>(smem + lay.zero);
        const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[1]);
        float *val = reinterpret_cast<float *>(smem + lay.val[1][0]);
        for (int v = tid - 128; v < V[1]; v += NT - 128) {
            const int t = row[v + 1];
            float a0 = 0.0f, a1 = 0.0f;
            for (int p = row[v]; p < t; p += 8) {
                float2 x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = *((p + u < t) ? pl + p + u : zero);
#pragma unroll
                for (int u = 0; u < 8; ++u) { a0 += x[u].x; a1 += x[u].y; }   // strictly left to right
            }
            reinterpret_cast<float2 *>(val)[v + 1] = make_float2(a0, a1);
        }
    }
    __syncthreads();
    // ---- P(0, second half) | S(0, second half) ----------------------------------------------------------------------------------
    chain_products(1);
    __syncthreads();
    acc = chain_sum(1, acc);
    if ((tid >> 7) < npairs && (cl.b >> 16) != 0) reinterpret_cast<float *>(smem + lay.val[0][0])[cl.b >> 16] = acc;
    __syncthreads();
    FL_STAMP();
    // ---- d+1 Jacobi blur passes, permutohedral_cpu.h:663-679 (as fused_loop.h, the small lattice on the last lanes) ---------------------
#pragma unroll
    for (int j = 0; j < D1; ++j) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float2 *src = reinterpret_cast<const float2 *>(smem + lay.val[k][j & 1]);
            float2 *dst = reinterpret_cast<float2 *>(smem + lay.val[k][(j & 1) ^ 1]);
            const unsigned *nbr = reinterpret_cast<const unsigned *>(smem + lay.nbr[k]) + j * V[k];
            int v0 = tid;
            if (k == 0) {
                asm volatile("" : "+v"(v0));
                v0 = NT - 1 - v0;
            }
            for (int v = v0; v < V[k]; v += NT) {
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
    FL_STAMP();
}

// n_iter x stepInference on the split plan (the non-fused form of fused_loop.h's mean_field: 3-4 points per lane)
template <int PPT, int NT>
__device__ __forceinline__ void mean_field_split(unsigned char *smem, const FusedLayout &lay, const int (&V)[2], int N, int tid,
                                                 PointRegs<PPT, 2> &pr, ChainLane &cl, const float (&alpha)[2], int n_iter, float relax, Instr &ins)
{
    for (int it = 0; it < n_iter; ++it) {
        opaque(pr);
        splat_blur_split<PPT, NT>(smem, lay, V, N, tid, pr, cl, ins);
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            if (tid + s * NT < N) {
                float nx[2] = {-pr.un[s].x, -pr.un[s].y};                 // stepInit, densecrf3d.h:154-158
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float2 t = slice_point(smem, lay, pr, s, k, alpha[k]);
                    nx[0] += pr.wn[s][k] * t.x;                           // pairwise3d.h:77
                    nx[1] += pr.wn[s][k] * t.y;
                }
                pr.q[s] = softmax2(nx[0], nx[1], pr.q[s], relax);        // densecrf3d.h:70-98
            }
        }
        FL_STAMP();
    }
}

}  // namespace fl
}  // namespace lccrf
