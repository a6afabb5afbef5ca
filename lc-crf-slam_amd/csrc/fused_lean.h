// fused_lean.h -- the mean-field loop of fused_loop.h on HALF a CU's LDS for full-size SLAM frames (1025 .. ~2300 keypoints), so that
// TWO frames share a CU: one frame's ordered row sums (a single wavefront adding the appearance kernel's ~450-product row left to
// right, everybody else waiting at the barrier behind it) run under the other frame's point phases.  DESIGN.md section 4.2.
//
// What the 137 KB plan of fused_loop.h keeps in LDS and this one does not:
//   * the second product buffer: the kernels share ONE (P/S run kernel after kernel: the short-row kernel first -- its products are
//     written right behind each point's softmax -- then the chain kernel);
//   * the large lattice's blur neighbour table (14 KB at V = 1160): read from HBM/L2 pass by pass, coalesced, each pass's words
//     requested while the pass before is running (a small lattice's table stays in LDS);
//   * the closed-form placement of the chain rows (14 spare slots per row): rows are placed by a scan over the ranked rows, the
//     kChainTop longest ones (first wavefront pair: the rows everybody waits for) padded to 4 products + their own eight zeros for
//     chain_rows' address clamp, every other row padded to 4 only -- chain_rows_sel picks the address of every 16-byte read
//     (row or a shared zero block) with a compare + select instead of the clamp.
// Same products, same places in the same rows, every row added strictly left to right: the same bits as fused_loop.h.
#pragma once

#include "fused_loop.h"

// Timing experiments (scripts/gpu_ab_build.sh "" "-DLCCRF_LEAN_SKIP=<bits>"; WRONG results, never in a shipped library): what a phase of
// the loop costs in throughput, as the launch time without it.  2 no chain sums, 16 no second trips of the short-row sums, 32 no short-row
// sums, 64 no blur passes, 128 no X (slice + softmax + products), 256 the per-point records are read once (no re-reads), 512 no
// neighbour words, 1024 (k_fused_lean) every workgroup reads the records of one of the first 64 frames.  A compile-time constant: as a
// run-time switch the tests alone cost the 4-points-per-lane loop 52 spilled registers.
#ifndef LCCRF_LEAN_SKIP
#define LCCRF_LEAN_SKIP 0
#endif
#define LEAN_SKIP(bit) ((LCCRF_LEAN_SKIP & (bit)) != 0)
// Cache policy of the loop's HBM / L2 accesses (bit set = non-temporal, aux = 2): 1 the prologue's once-per-launch records, 2 the
// large lattice's neighbour words, 4 norms + unary energies, 8 the results, 16 the barycentric weights.  What every iteration re-reads
// is ~94 KB per frame, 64 frames per XCD: 6 MB cycling through a 4 MB L2 misses every time (FETCH_SIZE: profiles/r5_fused_c2); the
// point of the hints is to keep a subset that FITS (the weights: 48 KB per frame) resident and stream the rest past it.
#ifndef LCCRF_LEAN_NT
#define LCCRF_LEAN_NT 0
#endif
#define LEAN_AUX(bit) ((LCCRF_LEAN_NT & (bit)) ? 2 : 0)

namespace lccrf {
namespace fl {

constexpr int kLeanMaxRounds = 4;         // vertices per lane and kernel in the blur passes (register rounds): V <= 1536 in both shapes
__host__ __device__ constexpr int lean_rounds(int nt) { return nt >= 512 ? 3 : 4; }
__host__ __device__ constexpr int lean_max_v(int nt) { return 3 * (nt - 64); }   // (the overlapped blur schedule: 3 rounds of nt - 64 lanes)

// product slots per label plane of the chain kernel: rows padded to 4, eight zeros behind each of the kChainTop longest
__host__ __device__ inline int lean_plane_floats(int NA, int V0)
{
    const int top = V0 < kChainTop ? V0 : kChainTop;
    return (NA * kD1 + 11 * top + 3 * (V0 - top) + 16 + 63) & ~63;
}

__host__ __device__ inline bool lean_chain_wanted(int NA, int V0, int row0, int nt)
{
    return row0 >= kChainMinRow && V0 <= chain_max_v(nt) && V0 <= 256 && lean_plane_floats(NA, V0) < 65535;
}

// The persistent tables of one kernel, carved at byte offset `o` of the plan (advanced past them): both value arrays, the neighbour
// table (the chain kernel's only: its lattice is small, <= chain_max_v vertices; -1 = the table stays in HBM) and the row starts.
// Kernel k's places depend on the vertex counts of kernels 0 .. k only -- the one-launch frame kernel (frame_lean.hip) builds its
// tables straight into them, kernel after kernel, before the plan as a whole is known.
struct LeanTables {
    int val0, val1, nbr, row;
};
__host__ __device__ inline LeanTables lean_tables(size_t &o, int Vk, bool chain)
{
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return (int)r; };
    LeanTables t;
    t.val0 = take((size_t)(Vk + 1) * sizeof(float2));
    t.val1 = take((size_t)(Vk + 1) * sizeof(float2));
    t.nbr = chain ? take((size_t)kD1 * Vk * sizeof(unsigned)) : -1;
    t.row = take((size_t)(Vk + 2) * sizeof(unsigned short));
    return t;
}

// LDS plan (see FusedLayout; nbr[k] < 0: kernel k's neighbour table stays in HBM -- every kernel's but the chain kernel's;
// pstart: u16 [V0 + 2] row starts of the chain kernel)
__host__ __device__ inline bool layout_lean(int NA, int K, const int *V, int row0, FusedLayout *lay, int nt, size_t lds_limit)
{
    constexpr int D1 = kD1;
    if (NA < 1 || K < 1 || K > kMaxFusedK) return false;
    for (int k = 0; k < K; ++k)
        if (V[k] >= 65535 || V[k] > lean_max_v(nt)) return false;
    if (row0 >= kChainMinRow && !lean_chain_wanted(NA, V[0], row0, nt)) return false;   // long rows that the chain lanes cannot take: not this plan
    const int chain0 = lean_chain_wanted(NA, V[0], row0, nt);
    {
        FusedLayout L{};
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) & ~(size_t)15; return (int)r; };
        L.prod_all = 0;
        L.chain0 = chain0;
        L.zero = take(128);                                // LDS bytes [0, 128) are zeros (chain_rows_sel reads them by absolute address)
        size_t shared_prod = 0;
        for (int k = 0; k < K; ++k) {
            const bool chain = k == 0 && chain0;
            L.Ecap[k] = chain ? lean_plane_floats(NA, V[k]) : ((NA * D1 + 63) & ~63);
            const LeanTables lt = lean_tables(o, V[k], chain);
            L.val[k][0] = lt.val0;
            L.val[k][1] = lt.val1;
            L.nbr[k] = lt.nbr;
            L.row[k] = lt.row;
            const size_t pb = (size_t)L.Ecap[k] * 2 * sizeof(float);
            shared_prod = pb > shared_prod ? pb : shared_prod;
        }
        if (o > 65535) return false;                       // (the vertex words hold 16-bit absolute addresses into the value arrays)
        L.pstart = chain0 ? take((size_t)(V[0] + 2) * sizeof(unsigned short)) : 0;
        const int p = take(shared_prod < 1024 ? 1024 : shared_prod);        // (the head of it is chain_setup's scratch)
        for (int k = 0; k < K; ++k) L.prod[k] = p;
        L.total = (int)o;
        if (o <= lds_limit) {
            *lay = L;
            return true;
        }
    }
    return false;
}

// chain_rows with the address of every 16-byte read chosen by compare + select: quad q of the lane's row (4 products at addr + 16 q)
// while 16 q < the bytes the row has left, LDS bytes [16 q, 16 q + 16) afterwards -- the first 128 bytes of the plan are zeros -- so
// rows need no zeros of their own behind them, only padding to a multiple of 4.  A ring of eight quads in v96..v127 (32 products
// per trip), six to eight reads in flight as in chain_rows.  (The select's other source is the inline constant 0: gfx9 allows one
// scalar operand per VALU instruction and VCC is one.)
//   addr  LDS byte address of the row (16-byte aligned)     bytes  16 * quads of the padded row
//   trips  ceil(max quads of the wavefront / 8)
#define LCCRF_ASM_SEL(off, vlo, vhi, S)                                                                             \
    "v_cmp_lt_i32_e32 vcc, " #off ", %[rem]\n\tv_cndmask_b32_e32 %[" #S "], 0, %[ad], vcc\n\t"                     \
    "ds_read_b128 v[" #vlo ":" #vhi "], %[" #S "] offset:" #off "\n\t"

// keep0 / keep1: two values of the caller that must survive the ring (the chain lane's two words).  They pass through the asm statement as operands, which pins
// them to registers the ring does not clobber; left to itself the register allocator parks loop-long values (the chain lane's two
// words) in v96..v127 and spills them to scratch around the ring.
__device__ __forceinline__ float chain_rows_sel(unsigned addr, int bytes, unsigned trips, unsigned &keep0, unsigned &keep1)
{
    float acc = 0.0f;
    if (trips == 0) return acc;
    unsigned sa, sb;
    asm volatile(
        LCCRF_ASM_SEL(0, 96, 99, sa) LCCRF_ASM_SEL(16, 100, 103, sb)
        LCCRF_ASM_SEL(32, 104, 107, sa) LCCRF_ASM_SEL(48, 108, 111, sb)
        LCCRF_ASM_SEL(64, 112, 115, sa) LCCRF_ASM_SEL(80, 116, 119, sb)
        "1:\n\t"
        LCCRF_ASM_SEL(96, 120, 123, sa) LCCRF_ASM_SEL(112, 124, 127, sb)
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v96, v97, v98, v99, v100, v101, v102, v103)
        "v_add_u32_e32 %[ad], 0x80, %[ad]\n\t"
        "v_subrev_u32_e32 %[rem], 0x80, %[rem]\n\t"
        LCCRF_ASM_SEL(0, 96, 99, sa) LCCRF_ASM_SEL(16, 100, 103, sb)
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v104, v105, v106, v107, v108, v109, v110, v111)
        LCCRF_ASM_SEL(32, 104, 107, sa) LCCRF_ASM_SEL(48, 108, 111, sb)
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v112, v113, v114, v115, v116, v117, v118, v119)
        LCCRF_ASM_SEL(64, 112, 115, sa) LCCRF_ASM_SEL(80, 116, 119, sb)
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v120, v121, v122, v123, v124, v125, v126, v127)
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [acc] "+v"(acc), [ad] "+v"(addr), [rem] "+v"(bytes), [n] "+s"(trips), [sa] "=&v"(sa), [sb] "=&v"(sb), "+v"(keep0), "+v"(keep1)
        :
        : "scc", "vcc", "memory", LCCRF_CHAIN_RING_CLOBBERS);
    return acc;
}

// chain_rows (fused_loop.h) with the same two pass-through operands
__device__ __forceinline__ float chain_rows_keep(unsigned addr, unsigned end, unsigned trips, unsigned &keep0, unsigned &keep1)
{
    float acc = 0.0f;
    if (trips == 0) return acc;
    const unsigned e1 = end - 32u, e2 = end - 64u, e3 = end - 96u;
    unsigned sel;
    asm volatile(LCCRF_CHAIN_ROWS_ASM
                 : [acc] "+v"(acc), [ad] "+v"(addr), [n] "+s"(trips), [sel] "=&v"(sel), "+v"(keep0), "+v"(keep1)
                 : [e0] "v"(end), [e1] "v"(e1), [e2] "v"(e2), [e3] "v"(e3)
                 : "scc", "memory", LCCRF_CHAIN_RING_CLOBBERS);
    return acc;
}

// LDS by absolute byte address (the plan starts at LDS address 0, as the rings assume): no `smem +` in the address arithmetic
__device__ __forceinline__ float2 lds_f2(unsigned addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float lean_f2 __attribute__((ext_vector_type(2)));
    const lean_f2 v = *reinterpret_cast<const __attribute__((address_space(3))) lean_f2 *>((unsigned long)addr);
    return make_float2(v.x, v.y);
#else
    (void)addr;                           // (host pass of the translation unit: never called)
    return make_float2(0.f, 0.f);
#endif
}

__device__ __forceinline__ void lds_store_f(unsigned addr, float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<__attribute__((address_space(3))) float *>((unsigned long)addr) = v;
#else
    (void)addr; (void)v;
#endif
}
__device__ __forceinline__ void lds_store_f2(unsigned addr, float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float lean_f2 __attribute__((ext_vector_type(2)));
    lean_f2 v;
    v.x = a;
    v.y = b;
    *reinterpret_cast<__attribute__((address_space(3))) lean_f2 *>((unsigned long)addr) = v;
#else
    (void)addr; (void)a; (void)b;
#endif
}

// point_products of fused_loop.h for the lean plan's slot words: a product's ABSOLUTE LDS byte address / 4 (16 bits: the plan is < 256 KB),
// so that a store's address is one shift away (the slot-index form costs an extract, a scale and a base add per store)
template <int PPT, int K, int CH>
__device__ __forceinline__ void point_products_lean(const FusedLayout &lay, const PointRegs<PPT, K> &pr, int s, int k)
{
    const unsigned a0 = (pr.ix[s][k][1] >> 16) << 2, a1 = (pr.ix[s][k][2] & 0xffffu) << 2, a2 = (pr.ix[s][k][2] >> 16) << 2;
    if (chain_k<CH>(lay, k)) {                             // chain kernel: one plane per label
        const unsigned pb = (unsigned)lay.Ecap[k] * 4u;
        lds_store_f(a0, pr.bary[s][k][0] * pr.q[s].x);
        lds_store_f(a0 + pb, pr.bary[s][k][0] * pr.q[s].y);
        lds_store_f(a1, pr.bary[s][k][1] * pr.q[s].x);
        lds_store_f(a1 + pb, pr.bary[s][k][1] * pr.q[s].y);
        lds_store_f(a2, pr.bary[s][k][2] * pr.q[s].x);
        lds_store_f(a2 + pb, pr.bary[s][k][2] * pr.q[s].y);
    } else {                                              // short rows: labels interleaved
        lds_store_f2(a0, pr.bary[s][k][0] * pr.q[s].x, pr.bary[s][k][0] * pr.q[s].y);
        lds_store_f2(a1, pr.bary[s][k][1] * pr.q[s].x, pr.bary[s][k][1] * pr.q[s].y);
        lds_store_f2(a2, pr.bary[s][k][2] * pr.q[s].x, pr.bary[s][k][2] * pr.q[s].y);
    }
}

// slice_point of fused_loop.h for the lean plan's vertex words (absolute addresses, place_products_lean)
template <int PPT, int K>
__device__ __forceinline__ float2 slice_point_lean(const PointRegs<PPT, K> &pr, int s, int k, float alpha)
{
    const float2 x0 = lds_f2(pr.ix[s][k][0] & 0xffffu), x1 = lds_f2(pr.ix[s][k][0] >> 16), x2 = lds_f2(pr.ix[s][k][1] & 0xffffu);
    const float w0 = pr.bary[s][k][0] * alpha, w1 = pr.bary[s][k][1] * alpha, w2 = pr.bary[s][k][2] * alpha;   // permutohedral_cpu.h:689
    float t0 = w0 * x0.x, t1 = w0 * x0.y;
    t0 += w1 * x1.x; t1 += w1 * x1.y;
    t0 += w2 * x2.x; t1 += w2 * x2.y;
    return make_float2(t0, t1);
}

// Chain lanes of the lean plan.  Ranking as chain_setup (counting sort on the rows' 16-product block count, longest first); then
// one wavefront places the rows by a scan in rank order -- ceil4(length) slots, + 8 zeros for the kChainTop first -- and leaves
// the starts in lay.pstart.  ChainLane: a = row address | wavefront max quads << 18;  b = quads of the padded row | pad slots << 13 |
// output index << 16.  Every lane of the workgroup calls it; ends with a barrier (the scratch becomes the product buffer).
template <int NT>
__device__ __forceinline__ ChainLane chain_setup_lean(unsigned char *smem, const FusedLayout &lay, int V0, int tid)
{
    constexpr int k = 0;
    ChainLane cl{0u, 0u};
    int *hist = reinterpret_cast<int *>(smem + lay.prod[k]);            // [64] counts, [64] bases
    if (tid < 128) hist[tid] = 0;
    __syncthreads();
    const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
    unsigned short *srt = reinterpret_cast<unsigned short *>(smem + lay.prod[k]) + 256;   // [V] vertex of rank r
    unsigned short *pstart = reinterpret_cast<unsigned short *>(smem + lay.pstart);
    int key = 0;
    if (tid < V0) {
        const int len = (int)row[tid + 1] - (int)row[tid];
        key = 63 - min((len + 3) >> 4, 63);
        atomicAdd(&hist[key], 1);
    }
    __syncthreads();
    if (tid < 64) {                                   // exclusive scan of the 64 bucket counts
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
    if (tid < 64) {                                   // row starts: a scan over the rows in rank order, four per lane (V0 <= 256)
        int sz[4], sum = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = tid * 4 + u;
            sz[u] = 0;
            if (r < V0) {
                const int v = srt[r];
                const int rl = (int)row[v + 1] - (int)row[v];
                sz[u] = ((rl + 3) & ~3) + (r < kChainTop ? 8 : 0);
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
            if (r < V0) pstart[srt[r]] = (unsigned short)base;
            base += sz[u];
        }
    }
    __syncthreads();
    // wavefront pair p = (2p, 2p+1) owns labels 0 and 1 of a rank range (chain_setup): pair 0 the kChainTop longest rows
    const int l = (tid >> 6) & 1, pr = tid >> 7, ln = tid & 63;
    const int r = pr == 0 ? (chain_top_rank(ln) >= 0 ? chain_top_rank(ln) : V0) : kChainTop + ((pr - 1) << 6) + ln;
    unsigned quads = 0, addr = 0;
    if (r < V0) {
        const int v = srt[r];
        const int rl = (int)row[v + 1] - (int)row[v], len4 = (rl + 3) & ~3;
        addr = (unsigned)(lay.prod[k] + 4 * (l * lay.Ecap[k] + (int)pstart[v]));               // < 2^18
        quads = (unsigned)(len4 >> 2);                                                          // < 2^13
        cl.b = quads | ((unsigned)(len4 - rl) << 13) | ((unsigned)((v + 1) * 2 + l) << 16);
    }
    unsigned m = quads;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    cl.a = addr | (m << 18);
    __syncthreads();                      // the ranking scratch becomes the product buffer
    return cl;
}

// behind a chain row: +0 up to a multiple of 4; the first wavefront pair's rows also get the eight +0 chain_rows reads past the end
__device__ __forceinline__ void chain_pads_lean(unsigned char *smem, const ChainLane &cl, int tid)
{
    if (cl.b >> 16) {
        float *e = reinterpret_cast<float *>(smem + (cl.a & 0x3ffffu)) + (cl.b & 0x1fffu) * 4u;
        const unsigned npad = (cl.b >> 13) & 3u;
        for (unsigned z = 1; z <= npad; ++z) e[-(int)z] = 0.0f;
        if ((tid >> 7) == 0) {
            reinterpret_cast<float4 *>(e)[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            reinterpret_cast<float4 *>(e)[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// place_products for the lean plan: a chain row starts at pstart[v] instead of pst(row[v], v)
template <int PPT, int K, int CH, int NT>
__device__ __forceinline__ void place_products_lean(unsigned char *smem, const FusedLayout &lay, int N, int tid,
                                                    const unsigned (&pk)[PPT][K][kD1], PointRegs<PPT, K> &pr)
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
        const unsigned short *pstart = reinterpret_cast<const unsigned short *>(smem + lay.pstart);
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            // the three vertices as ABSOLUTE LDS byte addresses of their values in the array the slice reads (val[k][d+1 & 1]; the plan
            // keeps the value arrays in its first 64 KB): a gather is then one bit-field extract away from its ds_read (the index
            // form costs a second instruction per gather, 24 per wavefront and iteration)
            const unsigned vb = (unsigned)lay.val[k][kD1 & 1];
            pr.ix[s][k][0] = (vb + 8u * (pk[s][k][0] & 0xffffu)) | ((vb + 8u * (pk[s][k][1] & 0xffffu)) << 16);
            pr.ix[s][k][1] = vb + 8u * (pk[s][k][2] & 0xffffu);
            pr.ix[s][k][2] = 0;
            if (tid + s * NT < N) {
                unsigned sl[kD1];
#pragma unroll
                for (int j = 0; j < kD1; ++j) {
                    sl[j] = pk[s][k][j] >> 16;
                    if (chain_k<CH>(lay, k)) {
                        const int v = (int)(pk[s][k][j] & 0xffffu) - 1;
                        sl[j] = (unsigned)((int)pstart[v] + ((int)(pk[s][k][j] >> 16) - (int)row[v]));
                        sl[j] = ((unsigned)lay.prod[k] >> 2) + sl[j];              // absolute byte address / 4 (label plane 0)
                    } else {
                        sl[j] = ((unsigned)lay.prod[k] >> 2) + 2u * sl[j];         // ... of the float2
                    }
                }
                pr.ix[s][k][1] |= sl[0] << 16;
                pr.ix[s][k][2] = sl[1] | (sl[2] << 16);
            }
        }
    }
}

// This frame's slices of the per-kernel arrays the loop reads from HBM / L2, as buffer resources: a load takes a 32-bit lane
// offset + a scalar offset (no 64-bit address registers to keep or spill), and a lane past the end of the slice reads 0.
typedef unsigned lean_u3 __attribute__((ext_vector_type(3)));
struct LeanSrc {
    __amdgpu_buffer_rsrc_t nbr[kMaxFusedK];   // KernelDev::nbr16 [D1][Epad]: (n1 + 1) | (n2 + 1) << 16
    __amdgpu_buffer_rsrc_t bary[kMaxFusedK];  // KernelDev::bary  [Epad]
    __amdgpu_buffer_rsrc_t norm[kMaxFusedK];  // KernelDev::norm  [maxN]
    __amdgpu_buffer_rsrc_t unary;             // CrfDev::unary    [maxN][2]
    int nbr_axis_bytes[kMaxFusedK];           // Epad * 4
    // byte offset of each array inside its resource: 0 when every array has a resource of its own (k_fused_lean); the one-launch frame
    // kernel keeps a frame's records in ONE block behind one resource, at compile-time offsets (kLeanRec*) -- four scalar registers
    // instead of twenty-four
    int off_nbr[kMaxFusedK], off_bary[kMaxFusedK], off_norm[kMaxFusedK], off_unary;
};
// A frame's record block of the one-launch kernel (frame_lean.hip; frames of at most 4 * 512 points):
constexpr int kLeanRecPoints = 4 * kNTSmall;
constexpr int kLeanRecUnary = 0;                                                    // float2 [points]
constexpr int kLeanRecBary = kLeanRecUnary + kLeanRecPoints * 8;                     // per kernel: float [points][3]
constexpr int kLeanRecNorm = kLeanRecBary + kMaxFusedK * kLeanRecPoints * kD1 * 4;   // per kernel: float [points]
constexpr int kLeanRecNbr = kLeanRecNorm + kMaxFusedK * kLeanRecPoints * 4;          // the LAST kernel's table: u32 [3][lean_max_v]
constexpr int kLeanRecNbrAxis = lean_max_v(kNTSmall) * 4;
constexpr int kLeanRecBytes = (kLeanRecNbr + kD1 * kLeanRecNbrAxis + 255) & ~255;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lean_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);   // raw buffer, 32-bit data format (gfx9)
}

template <int PPT, int K>
__device__ __forceinline__ void opaque_ids(PointRegs<PPT, K> &pr)
{
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("" : "+v"(pr.ix[s][k][0]), "+v"(pr.ix[s][k][1]), "+v"(pr.ix[s][k][2]));
    }
}

// ---- prepared launch records (round 6) -----------------------------------------------------------------------------------------
// What the prologue of k_fused_lean derives from a frame's lattice and the batch's LDS plan -- the ranking of the chain rows and
// their placement (chain_setup_lean: six barriers, LDS atomics, two scans), every point's vertex addresses and product slots
// (place_products_lean), the u16 row table and the chain kernel's neighbour table -- is the same in every inference on that lattice.
// It is computed ONCE behind a build (k_fused_lean / k_fused <.., MODE 1>, by the SECOND inference on the same lattices: a caller
// with one inference per lattice never pays for it) into one block per frame:
//     [ix   : (s, k, lane) x 3 words]   the final PointRegs::ix, 12 bytes per (point slot, kernel, lane), lane-contiguous
//     [cl   : lane x 2 words]           the lane's ChainLane
//     [row k, nbr k for every kernel]   the bytes of the LDS tables, each padded to 16
// and an inference (MODE 2) starts with coalesced loads that depend on nothing but the frame index: no load waits for the frame's
// point count, nothing is ranked, placed or converted.  Stamps of one steady-state C2 workgroup (profiles/r6_fused_c2/stamps.txt):
// first barrier 16.5k -> see there, ranking 4.9k -> 0, placement 3.0k -> 0 of a workgroup's ~135k cycles.
struct LeanPrepPlan {
    int cl_off;
    int row_off[kMaxFusedK], row_bytes[kMaxFusedK];
    int nbr_off[kMaxFusedK], nbr_bytes[kMaxFusedK];     // nbr_bytes = 0: the kernel's table is not in LDS
    int total;                                           // multiple of 256
};
__host__ __device__ inline LeanPrepPlan lean_prep_plan(const FusedLayout &lay, int K, const int *Vcap, int nt, int ppt)
{
    LeanPrepPlan p{};
    int o = K * ppt * nt * 12;
    p.cl_off = o;
    o += nt * 8;
    for (int k = 0; k < K; ++k) {
        p.row_off[k] = o;
        p.row_bytes[k] = ((Vcap[k] + 2) * 2 + 15) & ~15;
        o += p.row_bytes[k];
        p.nbr_off[k] = o;
        p.nbr_bytes[k] = lay.nbr[k] >= 0 ? ((kD1 * Vcap[k] * 4 + 15) & ~15) : 0;
        o += p.nbr_bytes[k];
    }
    p.total = (o + 255) & ~255;
    return p;
}

// n_iter x stepInference on the lean plan.  Per iteration, between workgroup barriers:
//   [X: slice + apply + softmax of every point, its products for kernel K-1 right behind]  |  S(K-1)  |  P(k) | S(k) for k = K-2 .. 0  |
//   blur pass 0 | 1 | 2 (every kernel)
// RELOAD: the unary energies, barycentric weights and norms are not kept in registers across the iteration -- every X re-reads this
// lane's (L2-resident) records; what a lane holds between iterations is Q and the packed ids / slots (8 registers per point).
// NORM (the one-launch frame kernel): ONE pass that leaves every kernel's norm = 1 / (K * 1 + 1e-20) (pairwise3d.h:20-28) in src.norm
// (RELOAD; else w * norm in pr.wn) instead of updating Q -- the caller sets Q = 1; the same splat / row sums / blur / slice as an iteration's, phase by phase.
// PRELOADED: the caller has the first iteration's records in pr already (k_fused_lean from prepared blocks requests them with everything else).
template <int PPT, int K, int CH, int NT, bool RELOAD, bool NORM = false, bool PRELOADED = false>
__device__ __forceinline__ void mean_field_lean(unsigned char *smem, const FusedLayout &lay, const int (&V)[K], int N, int &t,
                                                PointRegs<PPT, K> &pr, ChainLane &cl, const float (&alpha)[K],
                                                const float (&wk)[K], const LeanSrc &src, int n_iter, float relax, float omr, Instr &ins)
{
    constexpr int D1 = kD1;
    constexpr int KF = K - 1;                             // the kernel whose products follow the softmax
    // everything below indexes with `t`, the caller's lane id, RE-FORMED at the top of every iteration and phase from the wavefront's
    // base (a scalar register: it survives the rings, which clobber v96..v127) and the lane's position in the wavefront (mbcnt over
    // an opaque zero, so that it cannot be hoisted): the offsets and LDS addresses derived from it are then computed where they are
    // used instead of being carried through the loop in registers the loop does not have (cf. opaque()), and no copy of the id has
    // to live across a ring -- left to itself the allocator parks it in the clobbered range and spills it around the asm.
    const int wave_base = __builtin_amdgcn_readfirstlane(t & ~63);
    (void)wave_base;
#ifndef LCCRF_LEAN_T_MBCNT
#define LCCRF_LEAN_T_MBCNT 1              // A/B (scripts/gpu_ab_build.sh "" "-DLCCRF_LEAN_T_MBCNT=0"): 0 = one opaque copy kept through the loop
#endif
#if LCCRF_LEAN_T_MBCNT
#define LEAN_FRESH_T()                                                                                               \
    do {                                                                                                             \
        int z_ = 0;                                                                                                  \
        asm volatile("" : "+v"(z_));                                                                                 \
        t = wave_base + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z_));          \
    } while (0)
#else
#define LEAN_FRESH_T() asm volatile("" : "+v"(t))
#endif
    auto load_weights = [&]() {
        if (!RELOAD) return;
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = t + s * NT;                   // (a lane without a point in this slot reads the slice's spare rows or 0: unused)
            if (!NORM) {
                typedef unsigned lean_u2 __attribute__((ext_vector_type(2)));
                const lean_u2 u = __builtin_amdgcn_raw_buffer_load_b64(src.unary, i * 8, src.off_unary, LEAN_AUX(4));
                pr.un[s] = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const lean_u3 b = __builtin_amdgcn_raw_buffer_load_b96(src.bary[k], i * (D1 * 4), src.off_bary[k], LEAN_AUX(16));
                pr.bary[s][k][0] = __uint_as_float(b.x);
                pr.bary[s][k][1] = __uint_as_float(b.y);
                pr.bary[s][k][2] = __uint_as_float(b.z);
                if (!NORM) pr.wn[s][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(src.norm[k], i * 4, src.off_norm[k], LEAN_AUX(4)));
            }
        }
        if (NORM) return;
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) pr.wn[s][k] = wk[k] * pr.wn[s][k];                    // pairwise3d.h:77 (w_*norm_[i])
    };
    auto phase_P = [&](int k) {
#pragma unroll
        for (int s = 0; s < PPT; ++s)
            if (t + s * NT < N) point_products_lean<PPT, K, CH>(lay, pr, s, k);
        if (chain_k<CH>(lay, k)) chain_pads_lean(smem, cl, t);      // (the buffer held the other kernel's products)
    };
    auto phase_S = [&](int k) {
        float *val = reinterpret_cast<float *>(smem + lay.val[k][0]);
        if (chain_k<CH>(lay, k)) {
            const int npairs = 1 + ((max(V[k] - kChainTop, 0) + 63) >> 6);
            // (marked likely: the register allocator weighs spill code by block frequency, and values it parks in v96..v127 -- which
            // the rings clobber -- would be spilled around this block if it looked cold)
            if (__builtin_expect((t >> 7) < npairs, 1)) {                            // whole wavefronts
                FL_PSTAMP();
                if ((t >> 7) == 0) __builtin_amdgcn_s_setprio(3);
                // (opaque: the row's address, end and trip counts are formed HERE -- hoisted out of the loop they would sit in
                // registers, or scratch, for the whole launch; the two words themselves ride through the ring as operands)
                asm volatile("" : "+v"(cl.a), "+v"(cl.b));
                if (__builtin_expect((cl.b >> 16) != 0, 1) && !LEAN_SKIP(2)) {     // (LEAN_SKIP: timing experiments, see the top of the file)
                    const unsigned row_addr = cl.a & 0x3ffffu, row_end = row_addr + (cl.b & 0x1fffu) * 16u;
                    const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)(cl.a >> 18));       // the wavefront's longest row, quads
                    const float acc = (t >> 7) == 0 ? chain_rows_keep(row_addr, row_end, (((m + 1u) >> 1) + 3u) >> 2, cl.a, cl.b)
                                                      : chain_rows_sel(row_addr, (int)((cl.b & 0x1fffu) * 16u), (m + 7u) >> 3, cl.a, cl.b);
                    val[cl.b >> 16] = acc;
                }
                LEAN_FRESH_T();                               // (nothing of the lane id was kept across the ring)
                if ((t >> 7) == 0) __builtin_amdgcn_s_setprio(0);
                FL_PSTAMP();
            }
            return;
        }
        // short rows: one lane per vertex sums both labels (products are stored label-interleaved), 8 at a time; a lane past the
        // end of its row reads the zero block (x + 0 is exact, see chain_rows).  This lane's rows (vertices t, t + NT, ...) are walked
        // TOGETHER: every row's pointers first, then every row's first eight products -- most rows of a smoothness kernel end there --
        // so the phase waits for two rounds of LDS latency instead of two per row; each row is still added strictly left to right.
        if (LEAN_SKIP(32)) return;
        const float2 *pl = reinterpret_cast<const float2 *>(smem + lay.prod[k]);
        const float2 *zero = reinterpret_cast<const float2 *>(smem + lay.zero);
        const unsigned short *row = reinterpret_cast<const unsigned short *>(smem + lay.row[k]);
        // (two large short-row lattices -- K = 2 without a chain kernel, which rarely fits this plan's LDS anyway -- keep one row at a
        // time: 48 registers of products beside that variant's other state would spill)
        constexpr int RS = (K == 2 && CH == 0) ? 1 : lean_rounds(NT);
        for (int base = 0; base < lean_rounds(NT); base += RS) {
            int p0[RS], pe[RS];
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                const int v = t + (base + r) * NT;
                p0[r] = pe[r] = 0;
                if (v < V[k]) { p0[r] = row[v]; pe[r] = row[v + 1]; }
            }
            // (element u of a row: its product, or -- past the row's end -- LDS bytes [8u, 8u + 8) of the plan's all-zero first 128
            // bytes: one compare + one select between the row's base address and 0 per element, the element's offset in the instruction)
            float2 x[RS][8];
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                const unsigned base = (unsigned)lay.prod[k] + 8u * (unsigned)p0[r];
                const int n = pe[r] - p0[r];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[r][u] = lds_f2(((u < n) ? base : 0u) + 8u * u);
            }
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                const int v = t + (base + r) * NT;
                float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                for (int u = 0; u < 8; ++u) { a0 += x[r][u].x; a1 += x[r][u].y; }   // strictly left to right
                for (int p = p0[r] + 8; p < pe[r] && !LEAN_SKIP(16); p += 8) {          // (rows of more than eight products)
                    float2 y[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) y[u] = *((p + u < pe[r]) ? pl + p + u : zero);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { a0 += y[u].x; a1 += y[u].y; }
                }
                if (v < V[k]) reinterpret_cast<float2 *>(val)[v + 1] = make_float2(a0, a1);
            }
        }
        (void)V;
    };
    // neighbour words of a lattice whose table stays in HBM / L2: all d+1 passes' words of this lane's vertices (t, t + NT, ...) are
    // requested at the top of the iteration and land under the row sums (requested pass by pass they were what a pass waited for:
    // 6.6-10k cycles for the three passes of a C2 frame instead of ~4k); a table in LDS is read pass by pass
    constexpr int R = lean_rounds(NT);
    constexpr int OVL_K = (K == 2 && CH == 1) ? 1 : K;   // (the overlapped schedule below has its own word arrays)
    auto load_nbr = [&](unsigned (&w)[OVL_K][D1][R]) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int j = 0; j < D1; ++j) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    w[k][j][r] = 0;
                    if (!(CH == 1 && k == 0) && (r == 0 || r * NT < V[k]))      // (uniform; v >= V: an unused word of the table)
                        w[k][j][r] = __builtin_amdgcn_raw_buffer_load_b32(src.nbr[k], (t + r * NT) * 4, src.off_nbr[k] + j * src.nbr_axis_bytes[k], LEAN_AUX(2));
                }
            }
        }
    };
    auto blur_pass = [&](int j, const unsigned (&w)[OVL_K][D1][R]) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float2 *src_v = reinterpret_cast<const float2 *>(smem + lay.val[k][j & 1]);
            float2 *dst = reinterpret_cast<float2 *>(smem + lay.val[k][(j & 1) ^ 1]);
            const bool in_lds = CH == 1 && k == 0;         // (layout_lean: the chain kernel's table, and only that one; k is unrolled)
            const unsigned *lds = reinterpret_cast<const unsigned *>(smem + (in_lds ? lay.nbr[k] : 0)) + j * V[k];
#pragma unroll
            for (int r = 0; r < (in_lds ? 1 : R); ++r) {  // (a chain kernel has at most chain_max_v(NT) < NT vertices)
                const int v = t + r * NT;
                if (v < V[k]) {
                    const unsigned n = in_lds ? lds[v] : w[k][j][r];
                    const float2 o = src_v[v + 1], x = src_v[n & 0xffffu], y = src_v[n >> 16];
                    float2 u;
                    u.x = o.x + 0.5f * (x.x + y.x);
                    u.y = o.y + 0.5f * (x.y + y.y);
                    dst[v + 1] = u;
                }
            }
        }
    };
    // ---- K = 2 with a chain kernel (the SLAM configuration): the blur passes are dealt to the wavefronts so that they hide ------
    //   under the chain (S of kernel 0): the wavefronts behind the first pair, once their own chain rows are summed, run blur
    //       pass 0 of kernel 1 -- its row sums were complete two barriers ago -- beside the pair that adds the longest rows;
    //   behind the next barrier: the last wavefront alone runs the passes of kernel 0's small lattice (one wavefront's LDS operations
    //       execute in order, so its passes need no workgroup barrier between them) -- passes 0 and 1 while the others run pass 1 of
    //       kernel 1, pass 2 beside their pass 2.  Three barriers from the chain to X instead of four, kernel 0's passes and a third of kernel 1's
    //   off the critical path.  Same operations on the same values per vertex: only who executes them, and when, differs.
    constexpr bool OVL = K == 2 && CH == 1;
    constexpr int NA0 = NT - 128, RA = 4;                 // pass 0 of kernel 1: lanes 128 .. NT-1, vertex (t - 128) + r * NA0
    constexpr int NB = NT - 64, RB = 3;                   // passes 1, 2 of kernel 1: lanes 0 .. NT-65, vertex t + r * NB
    auto load_nbr_ovl = [&](unsigned (&wa)[RA], unsigned (&wb)[2][RB]) {
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            wa[r] = 0;
            if ((r == 0 || r * NA0 < V[K - 1]) && !LEAN_SKIP(512))
                wa[r] = __builtin_amdgcn_raw_buffer_load_b32(src.nbr[K - 1], (t - 128 + r * NA0) * 4, src.off_nbr[K - 1], LEAN_AUX(2));      // (lanes < 128: out of range reads 0, unused)
        }
#pragma unroll
        for (int j = 1; j < D1; ++j) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                wb[j - 1][r] = 0;
                if ((r == 0 || r * NB < V[K - 1]) && !LEAN_SKIP(512))
                    wb[j - 1][r] = __builtin_amdgcn_raw_buffer_load_b32(src.nbr[K - 1], (t + r * NB) * 4, src.off_nbr[K - 1] + j * src.nbr_axis_bytes[K - 1], LEAN_AUX(2));
            }
        }
    };
    auto blur_vertex = [&](const float2 *src_v, float2 *dst, int v, unsigned n) {
        if (LEAN_SKIP(64)) return;
        const float2 o = src_v[v + 1], x = src_v[n & 0xffffu], y = src_v[n >> 16];
        float2 u;
        u.x = o.x + 0.5f * (x.x + y.x);
        u.y = o.y + 0.5f * (x.y + y.y);
        dst[v + 1] = u;
    };
    auto blur_big = [&](int j, int lane, int stride, int rounds, const unsigned *w) {       // kernel K-1, pass j
        constexpr int k = K - 1;
        const float2 *src_v = reinterpret_cast<const float2 *>(smem + lay.val[k][j & 1]);
        float2 *dst = reinterpret_cast<float2 *>(smem + lay.val[k][(j & 1) ^ 1]);
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            if (r < rounds) {
                const int v = lane + r * stride;
                if (v < V[k]) blur_vertex(src_v, dst, v, w[r]);
            }
        }
    };
    auto blur_small = [&](int lane, int j_lo, int j_hi) {  // kernel 0, passes j_lo .. j_hi - 1 by ONE wavefront (lane = 0 .. 63)
        const unsigned *tbl = reinterpret_cast<const unsigned *>(smem + lay.nbr[0]);
#pragma unroll
        for (int j = 0; j < D1; ++j) {
            if (j < j_lo || j >= j_hi) continue;
            const float2 *src_v = reinterpret_cast<const float2 *>(smem + lay.val[0][j & 1]);
            float2 *dst = reinterpret_cast<float2 *>(smem + lay.val[0][(j & 1) ^ 1]);
            for (int v = lane; v < V[0]; v += 64) blur_vertex(src_v, dst, v, tbl[j * V[0] + v]);
            // the next pass reads what other lanes of this wavefront have just written: LDS operations of one wavefront execute in
            // order; this only keeps the compiler from moving loads across the stores
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    };
    auto point_update = [&](int s) {
        if constexpr (NORM) {                                     // pairwise3d.h:22-27: norm_[i] = 1 / (compute(ones)[i] + 1e-20)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float v = slice_point_lean(pr, s, k, alpha[k]).x;
                const float nrm = 1.0f / (v + 1e-20f);
                if (RELOAD) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(nrm), src.norm[k], (t + s * NT) * 4, src.off_norm[k], 0);
                else pr.wn[s][k] = wk[k] * nrm;                  // pairwise3d.h:77 (w_*norm_[i])
            }
            return;
        }
        float nx[2] = {-pr.un[s].x, -pr.un[s].y};                 // stepInit, densecrf3d.h:154-158
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float2 t = slice_point_lean(pr, s, k, alpha[k]);
            nx[0] += pr.wn[s][k] * t.x;                           // pairwise3d.h:77
            nx[1] += pr.wn[s][k] * t.y;
        }
        pr.q[s] = softmax2(nx[0], nx[1], pr.q[s], relax, omr);
    };

    if (n_iter <= 0) return;
    if (!PRELOADED) load_weights();
#pragma unroll
    for (int s = 0; s < PPT; ++s)
        if (t + s * NT < N) point_products_lean<PPT, K, CH>(lay, pr, s, KF);
    if (chain_k<CH>(lay, KF)) chain_pads_lean(smem, cl, t);
    for (int it = 0; it < n_iter; ++it) {
        LEAN_FRESH_T();
        if (RELOAD) opaque_ids(pr);
        else opaque(pr);
        unsigned w[OVL ? 1 : K][D1][R];
        unsigned wa[RA], wb[2][RB];
        if constexpr (OVL) load_nbr_ovl(wa, wb);          // (on their way during the row sums)
        else load_nbr(w);
        __syncthreads();
        FL_STAMP();
        phase_S(KF);
#pragma unroll
        for (int k = KF - 1; k >= 0; --k) {
            __syncthreads();
            opaque_ids(pr);                               // (slot addresses are formed here, not carried from the top of the iteration)
            phase_P(k);
            __syncthreads();
            phase_S(k);
        }
        if constexpr (OVL) {
            // ---- d+1 Jacobi blur passes, permutohedral_cpu.h:663-679 (D1 = 3), dealt to the wavefronts as described above -----------
            if (t >= 128) blur_big(0, t - 128, NA0, RA, wa);
            __syncthreads();
            FL_STAMP();
            LEAN_FRESH_T();
#ifndef LCCRF_LEAN_EARLY_WEIGHTS
#define LCCRF_LEAN_EARLY_WEIGHTS 0        // A/B (scripts/gpu_ab_build.sh): request the re-read records one blur phase earlier
#endif
            if (LCCRF_LEAN_EARLY_WEIGHTS && !NORM && !LEAN_SKIP(256)) load_weights();
            if (t >= NB) blur_small(t - NB, 0, 2);        // (two of its three dependent passes here, the third beside pass 2 below:
            else blur_big(1, t, NB, RB, wb[0]);           //  all three in one phase made that phase as long as this one wavefront)
            __syncthreads();
            if (!LCCRF_LEAN_EARLY_WEIGHTS && !NORM && !LEAN_SKIP(256)) load_weights();   // (requested before the last pass: they land under it; a NORM pass has no next iteration)
            if (t < NB) blur_big(2, t, NB, RB, wb[1]);
            else blur_small(t - NB, 2, D1);
            __syncthreads();
        } else {
            __syncthreads();
            FL_STAMP();
            LEAN_FRESH_T();
            // ---- d+1 Jacobi blur passes, permutohedral_cpu.h:663-679 (D1 = 3) -----------------------
            blur_pass(0, w);
            __syncthreads();
            blur_pass(1, w);
            __syncthreads();
            if (!NORM && !LEAN_SKIP(256)) load_weights();    // (requested before the last pass: they land under it)
            blur_pass(2, w);
            __syncthreads();
        }
        FL_STAMP();
        LEAN_FRESH_T();
        opaque_ids(pr);
        const bool more = it + 1 < n_iter;
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            if (t + s * NT < N && !LEAN_SKIP(128)) {
                point_update(s);
                if (more) point_products_lean<PPT, K, CH>(lay, pr, s, KF);
            }
        }
        FL_STAMP();
    }
}

#undef LEAN_FRESH_T

}  // namespace fl
}  // namespace lccrf
