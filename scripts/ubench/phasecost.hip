// phasecost.hip -- what ONE dependent phase of the C5 single-frame iteration costs on gfx950, by how the phase boundary is made.
//
// BASELINE config 5 as written is ONE 100 000-point CRF: V = 590 626 lattice vertices, and every mean-field iteration is a
// chain of 9 phases (splat, 7 blur passes, slice) that each need the previous phase's values from anywhere in the frame.
// With one launch per phase a blur pass over 14 MB takes ~8 us (rocprofv3, profiles/r4_stream_c5_f1) where the bytes alone
// would take 2-3.  This program prices the alternatives on a synthetic lattice with the C5 neighbour statistics
// (65 % of the neighbour slots absent, 12 % inside the vertex's own simplex = within +-6 ids, of the rest 66 % within 4096 ids):
//   L0  empty kernel, same grid, back to back                              -> the launch boundary itself
//   L1  copy (centre load + store, no neighbour table, no gathers)
//   L2  the shipped blur pass (k_blur2 shape), plain (x) grid              -> what round 3 shipped for F < 8
//   L3  the same, XCD-chunked grid (workgroup L -> XCD L % 8 -> contiguous chunk)
//   P0  persistent kernel, 1 workgroup per CU: grid barrier only (XCD-hierarchical counters, no fences)
//   P1  persistent blur: neighbour ids + centre values in registers for the whole run, values published with write-through
//       (sc1) stores, neighbours gathered with sc1 loads, one barrier per pass
//   P2  P1 + the in-chunk neighbours served from LDS (only out-of-chunk neighbours touch memory)
// Every variant runs kPasses dependent passes; L2/L3/P1/P2 must produce the same values (checked).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kV = 590626, kAxes = 7, kBlock = 256;
constexpr int kPasses = 7 * 40;                 // dependent passes per timed run

struct XcdMap { int nb, G; };
__device__ __forceinline__ int block_of(XcdMap m)
{
    if (m.nb == 0) return blockIdx.x;
    const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
    return (xcd % m.G) * m.nb + q;
}

__global__ void __launch_bounds__(kBlock) k_empty(int V, XcdMap m)
{
    const int v = 2 * (block_of(m) * kBlock + threadIdx.x);
    if (v >= V) return;
}

__global__ void __launch_bounds__(kBlock) k_copy(const float2 *__restrict__ o, float2 *__restrict__ d, int V, XcdMap m)
{
    const int v = 2 * (block_of(m) * kBlock + threadIdx.x);
    if (v + 1 >= V) return;
    *reinterpret_cast<float4 *>(d + v) = *reinterpret_cast<const float4 *>(o + v);
}

typedef int v4i __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(kBlock) k_blur(const float2 *__restrict__ o, float2 *__restrict__ d, const int *__restrict__ nbr, int V, XcdMap m)
{
    const int v = 2 * (block_of(m) * kBlock + threadIdx.x);
    if (v + 1 >= V) return;                                // (V is even here)
    v4i n;
    if (NT) n = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(nbr + 2 * (size_t)v));
    else n = *reinterpret_cast<const v4i *>(nbr + 2 * (size_t)v);
    const float4 c = *reinterpret_cast<const float4 *>(o + v);
    const float2 x0 = o[n.x], y0 = o[n.y], x1 = o[n.z], y1 = o[n.w];
    *reinterpret_cast<float4 *>(d + v) = make_float4(c.x + 0.5f * (x0.x + y0.x), c.y + 0.5f * (x0.y + y0.y),
                                                     c.z + 0.5f * (x1.x + y1.x), c.w + 0.5f * (x1.y + y1.y));
}

// two passes (axes j, j + 1) in ONE launch, no extra tables: out[v] = t[v] + 0.5 (t[a] + t[b]) with a, b = the axis-(j+1)
// neighbours of v and t[x] = s[x] + 0.5 (s[n1_j(x)] + s[n2_j(x)]) recomputed for x = v, a, b -- the same operations in the
// same order as two passes, so the same bits; the absent vertex (-1) has no neighbours and t[-1] = 0 exactly.
template <int BLK>
__global__ void __launch_bounds__(BLK) k_blur2x(const float2 *__restrict__ o, float2 *__restrict__ d, const int2 *__restrict__ nbj,
                                               const int2 *__restrict__ nbj1, int V, XcdMap m)
{
    const int v = block_of(m) * BLK + threadIdx.x;
    if (v >= V) return;
    const int2 ab = nbj1[v], nv = nbj[v];
    const float2 sv = o[v];
    const int2 na = ab.x >= 0 ? nbj[ab.x] : make_int2(-1, -1), nb = ab.y >= 0 ? nbj[ab.y] : make_int2(-1, -1);
    const float2 sa = o[ab.x], sb = o[ab.y], v1 = o[nv.x], v2 = o[nv.y];
    const float2 a1 = o[na.x], a2 = o[na.y], b1 = o[nb.x], b2 = o[nb.y];
    const float2 tv = make_float2(sv.x + 0.5f * (v1.x + v2.x), sv.y + 0.5f * (v1.y + v2.y));
    const float2 ta = make_float2(sa.x + 0.5f * (a1.x + a2.x), sa.y + 0.5f * (a1.y + a2.y));
    const float2 tb = make_float2(sb.x + 0.5f * (b1.x + b2.x), sb.y + 0.5f * (b1.y + b2.y));
    d[v] = make_float2(tv.x + 0.5f * (ta.x + tb.x), tv.y + 0.5f * (ta.y + tb.y));
}

template <int BLK, int PAIRS>
__global__ void __launch_bounds__(BLK) k_blur_b(const float2 *__restrict__ o, float2 *__restrict__ d, const int *__restrict__ nbr, int V, XcdMap m)
{
#pragma unroll
    for (int u = 0; u < PAIRS; ++u) {
        const int v = 2 * ((block_of(m) * PAIRS + u) * BLK + threadIdx.x);
        if (v + 1 >= V) return;
        const v4i n = *reinterpret_cast<const v4i *>(nbr + 2 * (size_t)v);
        const float4 c = *reinterpret_cast<const float4 *>(o + v);
        const float2 x0 = o[n.x], y0 = o[n.y], x1 = o[n.z], y1 = o[n.w];
        *reinterpret_cast<float4 *>(d + v) = make_float4(c.x + 0.5f * (x0.x + y0.x), c.y + 0.5f * (x0.y + y0.y),
                                                         c.z + 0.5f * (x1.x + y1.x), c.w + 0.5f * (x1.y + y1.y));
    }
}

// ---- persistent variants -------------------------------------------------------------------------------------------------
struct Sync {
    unsigned xcc_n[8 * 32];       // census: workgroups per XCC (one counter per 128-byte line)
    unsigned xcc_cnt[8 * 32];     // arrivals per XCC, monotonic
    unsigned top[32];             // XCC leaders' arrivals, monotonic
    unsigned gen[8 * 32];         // generation word per XCC, written by the last arriver of the top counter
    unsigned flat[32];            // census barrier
    unsigned nxcc[32];            // XCCs that hold workgroups
};

__device__ __forceinline__ unsigned ld_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct BarrierCtx { int xcc; unsigned n_local, n_xcc; };

__device__ __forceinline__ BarrierCtx barrier_census(Sync *s, int nwg)
{
    __shared__ BarrierCtx ctx;
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        const int xcc = x & 7;
        const unsigned old = atomicAdd(&s->xcc_n[xcc * 32], 1u);
        if (old == 0) atomicAdd(&s->nxcc[0], 1u);
        atomicAdd(&s->flat[0], 1u);
        for (int spin = 0; ld_u32(&s->flat[0]) < (unsigned)nwg; ++spin) {     // bounded: a workgroup that is not resident must not hang the box
            if (spin > (1 << 22)) { s->flat[1] = 1u; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ctx.xcc = xcc;
        ctx.n_local = ld_u32(&s->xcc_n[xcc * 32]);
        ctx.n_xcc = ld_u32(&s->nxcc[0]);
    }
    __syncthreads();
    return ctx;
}

// phase p = 1, 2, ...: everybody's stores before the call are visible to everybody's sc1 loads after it (the stores are
// write-through sc1 stores drained by every wave before the arrive: no cache maintenance needed)
__device__ __forceinline__ void grid_barrier(Sync *s, const BarrierCtx &c, unsigned p)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = atomicAdd(&s->xcc_cnt[c.xcc * 32], 1u);
        if (old == p * c.n_local - 1u) {
            const unsigned o2 = atomicAdd(&s->top[0], 1u);
            if (o2 == p * c.n_xcc - 1u)
                for (int x = 0; x < 8; ++x) __hip_atomic_store(&s->gen[x * 32], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int spin = 0; ld_u32(&s->gen[c.xcc * 32]) < p; ++spin) {
            if (spin > (1 << 22)) { s->flat[1] = 1u; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(1024) k_pbarrier(Sync *s, int nwg, int passes)
{
    const BarrierCtx c = barrier_census(s, nwg);
    for (int p = 1; p <= passes; ++p) grid_barrier(s, c, (unsigned)p);
}

__device__ __forceinline__ float2 ld_f2_sc1(const float2 *p)
{
    const unsigned long long x = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)x), __uint_as_float((unsigned)(x >> 32)));
}
__device__ __forceinline__ void st_f2_sc1(float2 *p, float2 v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// VPT vertices per lane, workgroup w owns the contiguous chunk [w * chunk, (w + 1) * chunk)
template <int VPT, bool LDS>
__global__ void __launch_bounds__(1024) k_pblur(Sync *s, int nwg, float2 *buf0, float2 *buf1, const int *__restrict__ nbr, int V, int chunk, int passes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *loc[2] = {reinterpret_cast<float2 *>(smem), reinterpret_cast<float2 *>(smem) + chunk + 1};    // [0] of each = the absent vertex
    const BarrierCtx c = barrier_census(s, nwg);
    const int base = blockIdx.x * chunk;
    int2 nb[kAxes][VPT];
    float2 cen[VPT];
    bool own[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = threadIdx.x + k * 1024, v = base + i;
        own[k] = i < chunk && v < V;
        cen[k] = own[k] ? buf0[v] : make_float2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < kAxes; ++j) nb[j][k] = own[k] ? reinterpret_cast<const int2 *>(nbr)[(size_t)j * V + v] : make_int2(-1, -1);
        if (LDS && own[k]) loc[0][i + 1] = cen[k];
    }
    if (LDS && threadIdx.x == 0) { loc[0][0] = make_float2(0.f, 0.f); loc[1][0] = make_float2(0.f, 0.f); }
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        const int j = p % kAxes;
        const float2 *src = (p & 1) ? buf1 : buf0;
        float2 *dst = (p & 1) ? buf0 : buf1;
        const float2 *ls = loc[p & 1];
        float2 *ld = loc[(p & 1) ^ 1];
        float2 a[VPT], b[VPT];
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            int2 n = nb[0][k];
#pragma unroll
            for (int jj = 1; jj < kAxes; ++jj) if (j == jj) n = nb[jj][k];
            if (LDS) {
                const unsigned ra = (unsigned)(n.x - base), rb = (unsigned)(n.y - base);
                a[k] = (n.x < 0 || ra < (unsigned)chunk) ? ls[n.x < 0 ? 0 : (int)ra + 1] : ld_f2_sc1(src + n.x);
                b[k] = (n.y < 0 || rb < (unsigned)chunk) ? ls[n.y < 0 ? 0 : (int)rb + 1] : ld_f2_sc1(src + n.y);
            } else {
                a[k] = ld_f2_sc1(src + n.x);
                b[k] = ld_f2_sc1(src + n.y);
            }
        }
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            cen[k] = make_float2(cen[k].x + 0.5f * (a[k].x + b[k].x), cen[k].y + 0.5f * (a[k].y + b[k].y));
            if (own[k]) {
                st_f2_sc1(dst + base + threadIdx.x + k * 1024, cen[k]);
                if (LDS) ld[threadIdx.x + k * 1024 + 1] = cen[k];
            }
        }
        grid_barrier(s, c, (unsigned)(p + 1));
    }
}

static unsigned rng_state = 12345u;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main(int argc, char **argv)
{
    const int V = kV;
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    // synthetic neighbour tables [axis][V][2] with the C5 statistics
    std::vector<int> nbr((size_t)kAxes * V * 2);
    // mode (argv[1]): 0 the C5 statistics; 1 every neighbour absent (the dependent level without memory traffic); 2 the 35 % present
    // ones all within +-6 ids; 3 all within +-4096; 4 all uniform over the frame; quick (argv[2]): launch variants only
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const bool quick = argc > 2 && atoi(argv[2]) != 0;
    printf("neighbour mode %d\n", mode);
    for (size_t i = 0; i < nbr.size(); ++i) {
        const int v = (int)((i / 2) % V);
        const unsigned r = rnd() % 100;
        int n = -1;
        if (r >= 65 && mode != 1) {
            const unsigned r2 = rnd() % 100;
            if (mode == 2 || (mode == 0 && r < 77)) n = v + (int)(rnd() % 13) - 6;
            else if (mode == 3 || (mode == 0 && r2 < 66)) n = v + (int)(rnd() % 8193) - 4096;
            else n = (int)(rnd() % V);
            n = std::min(std::max(n, 0), V - 1);
        }
        nbr[i] = n;
    }
    std::vector<float2> init(V + 2);
    init[0] = init[1] = make_float2(0.f, 0.f);
    for (int v = 0; v < V; ++v) init[v + 1] = make_float2((rnd() % 1000) * 1e-3f, (rnd() % 1000) * 1e-3f);
    int *d_nbr;
    float2 *d_a, *d_b;
    Sync *d_sync;
    CK(hipMalloc(&d_nbr, nbr.size() * 4));
    CK(hipMalloc(&d_a, (V + 2) * sizeof(float2)));
    CK(hipMalloc(&d_b, (V + 2) * sizeof(float2)));
    CK(hipMalloc(&d_sync, sizeof(Sync)));
    CK(hipMemcpy(d_nbr, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice));
    float2 *A = d_a + 1, *B = d_b + 1;                     // [-1] = the absent vertex; vertex pairs 16-byte aligned? (d_a + 1 is 8 mod 16)
    // keep pairs aligned like the engine does: vertex 0 at a 16-byte boundary
    CK(hipFree(d_a)); CK(hipFree(d_b));
    CK(hipMalloc(&d_a, (V + 4) * sizeof(float2)));
    CK(hipMalloc(&d_b, (V + 4) * sizeof(float2)));
    A = d_a + 2; B = d_b + 2;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset = [&]() {
        CK(hipMemset(d_a, 0, (V + 4) * sizeof(float2)));
        CK(hipMemset(d_b, 0, (V + 4) * sizeof(float2)));
        CK(hipMemcpy(A, init.data() + 1, V * sizeof(float2), hipMemcpyHostToDevice));
    };
    std::vector<float2> ref(V), got(V);
    auto fetch = [&](std::vector<float2> &out, int passes) { CK(hipMemcpy(out.data(), (passes & 1) ? B : A, V * sizeof(float2), hipMemcpyDeviceToHost)); };
    const long work = (V + 1) / 2, n = (work + kBlock - 1) / kBlock;
    XcdMap plain{0, 1}, chunked{(int)((n + 7) / 8), 8};
    const dim3 gp((unsigned)n), gc((unsigned)(8 * chunked.nb));
    auto time_launches = [&](const char *name, auto launch, bool check) {
        reset();
        for (int p = 0; p < 14; ++p) launch(p);
        CK(hipStreamSynchronize(st));
        reset();
        CK(hipEventRecord(e0, st));
        for (int p = 0; p < kPasses; ++p) launch(p);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %7.2f us per pass\n", name, ms * 1e3 / kPasses);
        if (check) fetch(got, kPasses);
    };
    time_launches("L0 empty kernel, plain grid", [&](int) { k_empty<<<gp, kBlock, 0, st>>>(V, plain); }, false);
    time_launches("L0 empty kernel, chunked grid", [&](int) { k_empty<<<gc, kBlock, 0, st>>>(V, chunked); }, false);
    time_launches("L1 copy, plain grid", [&](int p) { k_copy<<<gp, kBlock, 0, st>>>((p & 1) ? B : A, (p & 1) ? A : B, V, plain); }, false);
    time_launches("L1 copy, chunked grid", [&](int p) { k_copy<<<gc, kBlock, 0, st>>>((p & 1) ? B : A, (p & 1) ? A : B, V, chunked); }, false);
    time_launches("L2 blur (nt table loads), plain grid", [&](int p) { k_blur<true><<<gp, kBlock, 0, st>>>((p & 1) ? B : A, (p & 1) ? A : B, d_nbr + (size_t)(p % kAxes) * V * 2, V, plain); }, true);
    ref = got;
    time_launches("L3 blur (nt table loads), XCD-chunked grid", [&](int p) { k_blur<true><<<gc, kBlock, 0, st>>>((p & 1) ? B : A, (p & 1) ? A : B, d_nbr + (size_t)(p % kAxes) * V * 2, V, chunked); }, true);
    printf("    L3 == L2: %s\n", memcmp(ref.data(), got.data(), V * sizeof(float2)) ? "NO" : "yes");
    time_launches("L3' blur (plain table loads), XCD-chunked grid", [&](int p) { k_blur<false><<<gc, kBlock, 0, st>>>((p & 1) ? B : A, (p & 1) ? A : B, d_nbr + (size_t)(p % kAxes) * V * 2, V, chunked); }, true);
    printf("    L3' == L2: %s\n", memcmp(ref.data(), got.data(), V * sizeof(float2)) ? "NO" : "yes");

    // block-size / work-per-lane variants of the chunked pass (plain table loads)
    auto sweep = [&](const char *name, auto kern, int blk, int pairs) {
        const long nn = (work + (long)blk * pairs - 1) / ((long)blk * pairs);
        XcdMap mm{(int)((nn + 7) / 8), 8};
        const dim3 gg((unsigned)(8 * mm.nb));
        time_launches(name, [&](int p) { kern<<<gg, blk, 0, st>>>((p & 1) ? B : A, (p & 1) ? A : B, d_nbr + (size_t)(p % kAxes) * V * 2, V, mm); }, true);
        printf("    == L2: %s\n", memcmp(ref.data(), got.data(), V * sizeof(float2)) ? "NO" : "yes");
    };
    sweep("L3' 512 lanes, 1 pair per lane", k_blur_b<512, 1>, 512, 1);
    sweep("L3' 1024 lanes, 1 pair per lane", k_blur_b<1024, 1>, 1024, 1);
    sweep("L3' 256 lanes, 2 pairs per lane", k_blur_b<256, 2>, 256, 2);
    sweep("L3' 128 lanes, 1 pair per lane", k_blur_b<128, 1>, 128, 1);
    sweep("L3' 64 lanes, 1 pair per lane", k_blur_b<64, 1>, 64, 1);
    {
        // two passes per launch (kPasses / 2 launches): per PASS figures, comparable with the lines above
        auto run2x = [&](const char *name, auto kern, int blk) {
            const long nn = ((long)V + blk - 1) / blk;
            XcdMap mm{(int)((nn + 7) / 8), 8};
            const dim3 gg((unsigned)(8 * mm.nb));
            reset();
            CK(hipEventRecord(e0, st));
            for (int q = 0; q < kPasses / 2; ++q) {
                // passes 2q, 2q + 1: axes (2q) % 7 and (2q + 1) % 7
                const int j = (2 * q) % kAxes, j1 = (2 * q + 1) % kAxes;
                kern<<<gg, blk, 0, st>>>((q & 1) ? B : A, (q & 1) ? A : B, reinterpret_cast<const int2 *>(d_nbr) + (size_t)j * V,
                                        reinterpret_cast<const int2 *>(d_nbr) + (size_t)j1 * V, V, mm);
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-58s %7.2f us per pass (%.2f per launch of two passes)\n", name, ms * 1e3 / kPasses, ms * 1e3 / (kPasses / 2));
            CK(hipMemcpy(got.data(), ((kPasses / 2) & 1) ? B : A, V * sizeof(float2), hipMemcpyDeviceToHost));
            printf("    == L2: %s\n", memcmp(ref.data(), got.data(), V * sizeof(float2)) ? "NO" : "yes");
        };
        run2x("L4 two passes per launch (3-level chain), 256 lanes", k_blur2x<256>, 256);
        run2x("L4 two passes per launch (3-level chain), 512 lanes", k_blur2x<512>, 512);
    }

    if (quick) return 0;
    // persistent variants: one workgroup per CU
    const int nwg = ncu;
    auto time_persistent = [&](const char *name, auto launch, bool check) {
        for (int rep = 0; rep < 2; ++rep) {
            reset();
            CK(hipMemsetAsync(d_sync, 0, sizeof(Sync), st));
            CK(hipEventRecord(e0, st));
            launch();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        Sync hs;
        CK(hipMemcpy(&hs, d_sync, sizeof(Sync), hipMemcpyDeviceToHost));
        printf("%-58s %7.2f us per pass (one launch of %d passes: %.1f us)%s\n", name, ms * 1e3 / kPasses, kPasses, ms * 1e3, hs.flat[1] ? "  ** a wait TIMED OUT **" : "");
        if (check) {
            fetch(got, kPasses);
            printf("    == L2: %s\n", memcmp(ref.data(), got.data(), V * sizeof(float2)) ? "NO" : "yes");
        }
    };
    time_persistent("P0 grid barrier only (XCD-hierarchical counters), 1024 lanes", [&]() { k_pbarrier<<<nwg, 1024, 0, st>>>(d_sync, nwg, kPasses); }, false);
    time_persistent("P0 grid barrier only, 256 lanes", [&]() { k_pbarrier<<<nwg, 256, 0, st>>>(d_sync, nwg, kPasses); }, false);
    const int chunk = ((V + nwg - 1) / nwg + 1) & ~1;
    if (chunk <= 3 * 1024) {
        const size_t lds = 2 * (size_t)(chunk + 1) * sizeof(float2);
        time_persistent("P1 persistent blur, everything via sc1 loads/stores", [&]() { k_pblur<3, false><<<nwg, 1024, 0, st>>>(d_sync, nwg, A, B, d_nbr, V, chunk, kPasses); }, true);
        time_persistent("P2 persistent blur, in-chunk neighbours from LDS", [&]() { k_pblur<3, true><<<nwg, 1024, lds, st>>>(d_sync, nwg, A, B, d_nbr, V, chunk, kPasses); }, true);
    } else {
        printf("chunk %d does not fit 3 vertices per lane (CUs: %d)\n", chunk, ncu);
    }
    return 0;
}
