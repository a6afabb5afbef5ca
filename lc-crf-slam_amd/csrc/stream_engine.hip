// stream_engine.hip -- the general ("streaming") engine: every array lives in HBM, one
// kernel launch per phase, frames of a batch in grid.y.  Handles any N / d / L.
//
// Compiled with -ffp-contract=off: the reference is an SSE2 build without FMA, so every
// a*b+c must stay two roundings.  No -ffast-math: fp32 division must
// be IEEE (V/=tt, 1/(norm+1e-20)), denormals are kept (gfx950 default).
//
// Reference being restated (paths under /root/reference/Thirdparty/DenseCRF/include/):
//   lattice construction   permutohedral_cpu.h:241-424   -> k_points / k_insert / k_offsets / k_neighbors
//   splat / blur / slice   permutohedral_cpu.h:634-699   -> k_splat / k_blur / k_slice
//   normalisation          pairwise3d.h:20-28            -> launch_norm
//   apply                  pairwise3d.h:73-78            -> k_slice (mode APPLY)
//   stepInit               densecrf3d.h:154-158          -> k_slice (first kernel)
//   expAndNormalize        densecrf3d.h:51-98            -> k_softmax
//   buildMap               densecrf3d.h:136-151          -> k_map
#include "engine.h"
#include "device_math.h"
#include "lattice_device.h"

#include <algorithm>
#include <cstdlib>

namespace lccrf {

namespace {

constexpr int kBlock = 256;
constexpr int kSmallFBlock = 64;          // lanes per workgroup of the iteration kernels with one or two frames in flight (iter_block)

// XCD-aware grids.  The chip's eight XCDs have private 4 MB L2s and workgroup L of a launch runs on XCD L % 8 (observed
// dispatch order; used for speed only, never for correctness).  With the frame in blockIdx.y every XCD touches every frame's
// lattice values.  Instead the launch's work -- F frames of nb blocks each, frame after frame -- is cut into EIGHT CONTIGUOUS
// parts of `per` blocks, one per XCD: workgroup L = 8 q + x handles block x per + q of that line.
//   F a multiple of 8      one XCD owns whole frames, and those frames' value arrays (4.7 MB each at C5) are what its L2 sees in
//                          the blur gathers;
//   F < 8                  an XCD handles a contiguous chunk of a frame's row-major vertex range (blur), row range (splat) or point
//                          range (slice): the neighbours a blur gather wants are mostly the centre lines of nearby blocks, i.e.
//                          lines the same L2 is fetching anyway (BASELINE config 5 as written is ONE frame: with the plain grid
//                          consecutive blocks went round-robin over the XCDs and every L2 saw the gathers of the whole array);
//   anything else          (3, 6, 12 frames ...) the same cut: every XCD gets F / 8 of a frame's worth -- frames pinned to
//                          XCDs whole, or to power-of-two groups of them, left a quarter of the chip idle at 3, 6 or 12 frames.
// `nb` = 0 selects the plain (x, frame) grid (empty launches only).
struct XcdMap { int nb, per; };
struct FrameBlock { int f, bx; };
__device__ __forceinline__ FrameBlock frame_block(XcdMap m)
{
    if (m.nb == 0) return FrameBlock{(int)blockIdx.y, (int)blockIdx.x};
    const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
    const int idx = xcd * m.per + q, f = idx / m.nb;
    return FrameBlock{f, idx - f * m.nb};                 // (the last XCD's tail lies beyond frame F - 1: the kernels' own range check)
}
// lanes per workgroup of the iteration kernels: 256; with one or two frames in flight a pass is a chain of latencies and smaller
// workgroups drain sooner (scripts/ubench/phasecost.hip: 256 -> 64 lanes 6.8 -> 6.4 us per pass of one C5 frame; in the engine,
// `FRAMES=1 WORKLOAD=c5 scripts/gpu_env_ab.sh LCCRF_SMALL_F_BLOCK=256 ""`: one frame 41.0 -> 40.2 us per iteration, two 32.8 -> 32.1, four +-0)
inline int iter_block(int F)
{
    static const char *e = ab_env("LCCRF_SMALL_F_BLOCK");             // A/B switch: same results
    static const int small = e ? std::min(std::max(atoi(e), 64), 256) & ~63 : kSmallFBlock;
    return F <= 2 ? small : 256;
}
inline dim3 grid_xcd(long work, int F, XcdMap *m, int block = 256)
{
    const long n = (work + block - 1) / block;
    if (n < 1) { *m = XcdMap{0, 1}; return dim3(1u, (unsigned)F); }
    static const bool no_chunk = ab_env("LCCRF_NO_XCD_CHUNK") != nullptr;   // A/B switch (same results): plain (x, frame) grid below 8 frames
    if (no_chunk && F < 8) { *m = XcdMap{0, 1}; return dim3((unsigned)n, (unsigned)F); }
    const long per = (n * F + 7) / 8;
    *m = XcdMap{(int)n, (int)per};
    return dim3((unsigned)(8L * per));
}

inline dim3 grid_for(long work, int F)
{
    const long nb = (work + kBlock - 1) / kBlock;
    return dim3((unsigned)(nb > 0 ? nb : 1), (unsigned)F);   // empty frames still get a (no-op) block
}

// ---------------------------------------------------------------------------------------
// lattice construction
// ---------------------------------------------------------------------------------------

// One thread per point (phantom lanes included): elevate, round, rank, barycentric.
// ref: permutohedral_cpu.h:294-366.
template <int D>
__device__ __forceinline__ void vertex_grid_coords(const int16_t (&key)[D], int (&c)[D]);

// VB: also leave, per workgroup, the bounds of the grid coordinates of every corner this workgroup's points touch (locality mode's
// vertex order, see launch_sort_vertices below) in vpartial[f][blockIdx.x][2 * kMaxD]
template <int D, bool VB>
__global__ void __launch_bounds__(kBlock) k_points(KernelDev kd, const int *__restrict__ n_points, int *__restrict__ vpartial, int *__restrict__ vbad)
{
    constexpr int D1 = D + 1;
    __shared__ int red[VB ? kBlock / 64 : 1][2 * D];
    const int f = blockIdx.y;
    const int N = n_points[f];
    const int Npad = (N + 3) & ~3;                       // blocks of four, :294
    const int n = blockIdx.x * kBlock + threadIdx.x;
    const bool live = n < Npad;
    if (!VB && !live) return;
    int lo[D], hi[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { lo[j] = 0x7fffffff; hi[j] = (int)0x80000000; }
    if (live) {
        float feat[D];
        const int src = (kd.perm && n < N) ? kd.perm[(size_t)f * kd.maxNpad + n] : n;   // locality mode: position n holds point perm[n]
        const float *fp = kd.feat + ((size_t)f * kd.maxN + src) * D;
#pragma unroll
        for (int j = 0; j < D; ++j) feat[j] = (n < N) ? fp[j] : 0.0f;   // phantom lanes, :299
        if (VB) {
            // Keys are int16 (the reference's `short`, permutohedral_cpu.h:304-366).  A feature thousands of cells wide wraps them, and a
            // wrapped key is no lattice point any more: the sorted build matches neighbours by code +- stride, the hash build by the
            // wrapped key itself -- they differ at the +-32768 seam (and when d + 1 divides 65536 the integrality check of k_ecode
            // cannot see the wrap: ADVICE r4).  Every |elevated coordinate| <= (d+1) sum |f_j scale_j|, a key is within 2 (d+1) of it:
            // a point that could leave the int16 range hands the frame to the hash build (pinned flag, conservative).
            float mag = 2.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) mag += fabsf(feat[j] * kd.scale[j]);
            if (!(mag * (float)D1 < 32000.0f) && vbad) *vbad = 1;
        }

        int16_t r0[D];
        uint8_t rk[D];
        float b[D1];
        point_record<D>(feat, kd.scale, kd.inv_dp1, r0, rk, b);

        int16_t *r0p = kd.rem0 + ((size_t)f * kd.maxNpad + n) * D;
        uint8_t *rkp = kd.rank + ((size_t)f * kd.maxNpad + n) * D;
        float *bp = kd.bary + (size_t)f * kd.Epad + (size_t)n * D1;
#pragma unroll
        for (int i = 0; i < D; ++i) { r0p[i] = r0[i]; rkp[i] = rk[i]; }
#pragma unroll
        for (int i = 0; i < D1; ++i) bp[i] = b[i];
        if (VB) {
#pragma unroll
            for (int rem = 0; rem < D1; ++rem) {
                int16_t key[D];
                int c[D];
#pragma unroll
                for (int i = 0; i < D; ++i) key[i] = vertex_coord<D>(r0[i], rk[i], rem);
                vertex_grid_coords<D>(key, c);
#pragma unroll
                for (int j = 0; j < D; ++j) { lo[j] = min(lo[j], c[j]); hi[j] = max(hi[j], c[j]); }
            }
        }
    }
    if (VB) {                                             // (every lane of the workgroup gets here: one barrier for all)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            int l = lo[j], h = hi[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                l = min(l, __shfl_xor(l, o, 64));
                h = max(h, __shfl_xor(h, o, 64));
            }
            if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][2 * j] = l; red[threadIdx.x >> 6][2 * j + 1] = h; }
        }
        __syncthreads();
        if (threadIdx.x < 2 * D) {
            int x = red[0][threadIdx.x];
            for (int w = 1; w < kBlock / 64; ++w) x = (threadIdx.x & 1) ? max(x, red[w][threadIdx.x]) : min(x, red[w][threadIdx.x]);
            vpartial[((size_t)f * gridDim.x + blockIdx.x) * 2 * kMaxD + threadIdx.x] = x;
        }
    }
}

// One thread per entry: insert its vertex key into the frame's hash table.  A slot ends up
// holding the LOWEST entry id carrying that key, i.e. the entry at which the reference's
// sequential HashTableCPU::find(create=true) would have created the vertex (:134-161,371-377).
template <int D>
__global__ void __launch_bounds__(kBlock) k_insert(KernelDev kd, const int *__restrict__ n_points, int F, XcdMap nb)
{
    constexpr int D1 = D + 1;
    const FrameBlock fb = frame_block(nb);               // many frames: one XCD per frame, its table stays in that L2
    const int f = fb.f;
    if (f >= F) return;
    const int Npad = (n_points[f] + 3) & ~3;
    const int e = fb.bx * kBlock + threadIdx.x;
    if (e >= Npad * D1) return;

    int16_t key[D];
    load_entry_key<D>(kd, f, e, key);
    const unsigned mask = (unsigned)kd.cap - 1u;
    unsigned h = hash_key<D>(key) & mask;
    int *slot = kd.slot + (size_t)f * kd.cap;
    for (;;) {
        const int prev = atomicCAS(&slot[h], kEmpty, e);
        if (prev == kEmpty || prev == e) break;
        int16_t other[D];
        load_entry_key<D>(kd, f, prev, other);
        bool same = true;
#pragma unroll
        for (int i = 0; i < D; ++i) same &= (other[i] == key[i]);
        if (same) { atomicMin(&slot[h], e); break; }
        h = (h + 1u) & mask;
    }
    kd.slot_of[(size_t)f * kd.Epad + e] = (int)h;
}

// flag[e] = 1 iff entry e is the first occurrence of its vertex.
__global__ void __launch_bounds__(kBlock) k_first_flag(KernelDev kd, const int *__restrict__ n_points)
{
    const int f = blockIdx.y;
    const int live = ((n_points[f] + 3) & ~3) * kd.D1;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e > kd.Epad) return;
    int v = 0;
    if (e < live) {
        const int s = kd.slot_of[(size_t)f * kd.Epad + e];
        v = (kd.slot[(size_t)f * kd.cap + s] == e);
    }
    kd.flag[(size_t)f * (kd.Epad + 1) + e] = v;
}

// Exclusive scan of n ints per frame, any n, by many workgroups: every workgroup scans a tile of kScanTile
// elements and leaves the tile's total (k_scan_tiles), one workgroup per frame scans the tile totals
// (k_scan_totals), every tile adds its prefix (k_scan_apply).  in/out strided by `stride` per frame; the grand
// total goes to total[f] if non-null.  (Round 1 scanned a frame with ONE workgroup: 0.75 ms per 700k entries.)
constexpr int kScanTile = 4096;           // 1024 lanes x 4 consecutive elements

__device__ __forceinline__ int wave_incl_scan_i32(int x)     // DPP: row_shr 1,2,4,8 then row_bcast:15 / :31
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}

// exclusive prefix of one value per lane over a 1024-lane workgroup; `total` = sum over the workgroup
__device__ __forceinline__ int block_excl_scan_1024(int x, int &total)
{
    __shared__ int wave_sum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_incl_scan_i32(x);
    __syncthreads();
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    const int ws = lane < 16 ? wave_sum[lane] : 0;
    const int wincl = wave_incl_scan_i32(ws);
    total = __builtin_amdgcn_readlane(wincl, 15);
    return __builtin_amdgcn_readlane(wincl - ws, __builtin_amdgcn_readfirstlane(wave)) + incl - x;
}

__global__ void __launch_bounds__(1024) k_scan_tiles(const int *__restrict__ in, int *__restrict__ out, int n, int stride,
                                                     int *__restrict__ tile_sum, int tiles)
{
    const int f = blockIdx.y, t = blockIdx.x;
    in += (size_t)f * stride;
    out += (size_t)f * stride;
    const int i0 = t * kScanTile + threadIdx.x * 4;
    int x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = (i0 + u < n) ? in[i0 + u] : 0;
    int total;
    int run = block_excl_scan_1024(x[0] + x[1] + x[2] + x[3], total);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (i0 + u < n) out[i0 + u] = run;                // prefix inside the tile; k_scan_apply adds the tiles before it
        run += x[u];
    }
    if (threadIdx.x == 0) tile_sum[(size_t)f * tiles + t] = total;
}

__global__ void __launch_bounds__(1024) k_scan_totals(int *__restrict__ tile_sum, int tiles, int *__restrict__ total)
{
    const int f = blockIdx.x;
    int *ts = tile_sum + (size_t)f * tiles;
    int carry = 0;
    for (int base = 0; base < tiles; base += 1024) {      // (one round up to 4M elements per frame)
        const int i = base + threadIdx.x;
        const int x = i < tiles ? ts[i] : 0;
        int tot;
        const int ex = block_excl_scan_1024(x, tot);
        if (i < tiles) ts[i] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (total && threadIdx.x == 0) total[f] = carry;
}

__global__ void __launch_bounds__(1024) k_scan_apply(int *__restrict__ out, int n, int stride, const int *__restrict__ tile_sum,
                                                     int tiles)
{
    const int f = blockIdx.y, t = blockIdx.x;
    const int add = tile_sum[(size_t)f * tiles + t];
    out += (size_t)f * stride;
    const int i0 = t * kScanTile + threadIdx.x * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (i0 + u < n) out[i0 + u] += add;
}

// tile_sum: scratch of F * ceil(n / kScanTile) ints
void scan_frames(const int *in, int *out, int n, int stride, int *total, int *tile_sum, int F, hipStream_t s)
{
    const int tiles = (n + kScanTile - 1) / kScanTile;
    k_scan_tiles<<<dim3(tiles, F), 1024, 0, s>>>(in, out, n, stride, tile_sum, tiles);
    k_scan_totals<<<F, 1024, 0, s>>>(tile_sum, tiles, total);
    if (tiles > 1) k_scan_apply<<<dim3(tiles, F), 1024, 0, s>>>(out, n, stride, tile_sum, tiles);
}

// ---------------------------------------------------------------------------------------
// locality mode: an internal ordering of a frame's points along a Z-order curve of their lattice cells
// ---------------------------------------------------------------------------------------
// The blur pass gathers two neighbour values per vertex; with the reference's vertex numbering (insertion order = point
// order) the neighbours that belong to OTHER points' simplices sit anywhere in the frame's value array, and every miss
// moves a 128-byte line for 8 useful bytes (profiles/r3_fetchcal).  Numbering follows the order in which points are
// processed, so processing them cell by cell along a space-filling curve puts the vertices of neighbouring cells next to
// each other.  Only locality depends on this order, never a result: it is a counting sort on a 2^bits-bucket Z-order code
// (one pass, atomics; buckets of up to 64 points are then put in index order so the order is reproducible).

// cell of every real point: the remainder-0 lattice point's first D coordinates / (D+1), permutohedral_cpu.h:304-319;
// per-dimension min / max of every workgroup's cells go to ss.partial (reduced per frame by k_sort_plan)
template <int D>
__global__ void __launch_bounds__(kBlock) k_sort_cells(KernelDev kd, const int *__restrict__ n_points, SortScratch ss)
{
    __shared__ int red[kBlock / 64][2 * D];
    const int f = blockIdx.y;
    const int N = n_points[f];
    const int n = blockIdx.x * kBlock + threadIdx.x;
    int cell[D];
    const bool live = n < N;
    {
        float feat[D];
        const float *fp = kd.feat + ((size_t)f * kd.maxN + (live ? n : 0)) * D;
#pragma unroll
        for (int j = 0; j < D; ++j) feat[j] = N > 0 ? fp[j] : 0.0f;
        float el[D + 1];
        float sm = 0.0f;
#pragma unroll
        for (int j = D; j > 0; --j) {
            const float cf = feat[j - 1] * kd.scale[j - 1];
            el[j] = sm - (float)j * cf;
            sm += cf;
        }
        el[0] = sm;
#pragma unroll
        for (int j = 0; j < D; ++j) cell[j] = __float2int_rn(kd.inv_dp1 * el[j]);
        if (ss.rm_points) {                               // the sorted build's companion: cells in the lattice's own basis, c_j ~ v_d - v_j
            const int vd = __float2int_rn(kd.inv_dp1 * el[D]);
#pragma unroll
            for (int j = 0; j < D; ++j) cell[j] = vd - cell[j];
        }
    }
    if (live) {
        int *cp = ss.cells + ((size_t)f * kd.maxNpad + n) * kMaxD;
#pragma unroll
        for (int j = 0; j < D; ++j) cp[j] = cell[j];
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        int lo = live ? cell[j] : 0x7fffffff, hi = live ? cell[j] : (int)0x80000000;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o, 64));
            hi = max(hi, __shfl_xor(hi, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6][2 * j] = lo;
            red[threadIdx.x >> 6][2 * j + 1] = hi;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * D) {
        int x = red[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) x = (threadIdx.x & 1) ? max(x, red[w][threadIdx.x]) : min(x, red[w][threadIdx.x]);
        ss.partial[((size_t)f * gridDim.x + blockIdx.x) * 2 * kMaxD + threadIdx.x] = x;
    }
}

// One workgroup per frame: bounds of the frame's cells, then the plan of the Z-order code -- the `bits` code bits are
// dealt to the dimensions by their spans (the widest remaining span gets the next bit).  plan[f] = {lo[kMaxD], span[kMaxD], nb[kMaxD]}.
__global__ void __launch_bounds__(kBlock) k_sort_plan(int D, int nblocks, SortScratch ss)
{
    __shared__ int red[kBlock / 64][2 * kMaxD];
    __shared__ int bnd[2 * kMaxD];
    const int f = blockIdx.x;
    for (int j = 0; j < D; ++j) {
        int lo = 0x7fffffff, hi = (int)0x80000000;
        for (int b = threadIdx.x; b < nblocks; b += kBlock) {
            const int *pp = ss.partial + ((size_t)f * nblocks + b) * 2 * kMaxD;
            lo = min(lo, pp[2 * j]);
            hi = max(hi, pp[2 * j + 1]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o, 64));
            hi = max(hi, __shfl_xor(hi, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6][2 * j] = lo;
            red[threadIdx.x >> 6][2 * j + 1] = hi;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * D) {
        int x = red[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) x = (threadIdx.x & 1) ? max(x, red[w][threadIdx.x]) : min(x, red[w][threadIdx.x]);
        bnd[threadIdx.x] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int *plan = ss.plan + (size_t)f * 3 * kMaxD;
        long rem[kMaxD];
        int nb[kMaxD];
        for (int j = 0; j < D; ++j) {
            const int lo = bnd[2 * j], hi = bnd[2 * j + 1];
            const int span = hi >= lo ? hi - lo + 1 : 1;   // (an empty frame has no cells)
            plan[j] = hi >= lo ? lo : 0;
            plan[kMaxD + j] = span;
            rem[j] = span;
            nb[j] = 0;
        }
        for (int b = 0; b < ss.bits; ++b) {
            int best = 0;
            for (int j = 1; j < D; ++j)
                if (rem[j] > rem[best]) best = j;
            ++nb[best];
            rem[best] = (rem[best] + 1) >> 1;
        }
        for (int j = 0; j < D; ++j) plan[2 * kMaxD + j] = nb[j];
    }
}

// Z-order bucket of every point: each coordinate quantised to its share of the code bits, bits interleaved level by level
template <int D>
__global__ void __launch_bounds__(kBlock) k_sort_code(KernelDev kd, const int *__restrict__ n_points, SortScratch ss)
{
    const int f = blockIdx.y;
    const int N = n_points[f];
    const int n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const int *plan = ss.plan + (size_t)f * 3 * kMaxD;
    const int *cp = ss.cells + ((size_t)f * kd.maxNpad + n) * kMaxD;
    if (ss.rm_points) {
        // points in (coarse) ROW-MAJOR order of their cell in the lattice's own basis -- the order the sorted build gives the vertices:
        // the points of consecutive vertices' rows are then close together too (splat), and so are the corners of consecutive points
        unsigned long long code = 0, range = 1;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            code += (unsigned long long)(unsigned)(cp[j] - plan[j]) * range;
            range *= (unsigned long long)(unsigned)plan[kMaxD + j];          // (spans of cells: a product beyond 2^64 only scrambles an order)
        }
        unsigned b = range >> ss.bits ? (unsigned)(code / ((range >> ss.bits) + 1)) : (unsigned)code;
        b = min(b, (1u << ss.bits) - 1u);                 // (a product of spans beyond 2^64 wraps: any bucket will do, inside the histogram)
        ss.code[(size_t)f * kd.maxNpad + n] = (int)b;
        atomicAdd(&ss.hist[(size_t)f * ((1 << ss.bits) + 1) + b], 1);
        return;
    }
    unsigned q[D];
    int nb[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        nb[j] = plan[2 * kMaxD + j];
        q[j] = (unsigned)((((unsigned long long)(unsigned)(cp[j] - plan[j])) << nb[j]) / (unsigned)plan[kMaxD + j]);
    }
    unsigned code = 0;
    int pos = 0;
    for (int level = 0; level < ss.bits && pos < ss.bits; ++level)
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (nb[j] > level) code |= ((q[j] >> level) & 1u) << pos++;
    ss.code[(size_t)f * kd.maxNpad + n] = (int)code;
    atomicAdd(&ss.hist[(size_t)f * ((1 << ss.bits) + 1) + code], 1);
}

// arrival order of the scatter is arbitrary: ss.cells (dead by now) takes the unordered buckets ...
__global__ void __launch_bounds__(kBlock) k_sort_scatter(int maxNpad, const int *__restrict__ n_points, SortScratch ss)
{
    const int f = blockIdx.y;
    const int n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_points[f]) return;
    const int code = ss.code[(size_t)f * maxNpad + n];
    const int pos = atomicAdd(&ss.start[(size_t)f * ((1 << ss.bits) + 1) + code], 1);   // start[c] ends up at the END of bucket c
    ss.cells[(size_t)f * maxNpad * kMaxD + pos] = n;
}

// ... and every point takes the place of its rank inside its bucket, so the order is reproducible (index order inside
// buckets of up to 64 points; larger buckets -- everything in one cell -- keep their arrival order: locality only)
__global__ void __launch_bounds__(kBlock) k_sort_place(int maxNpad, const int *__restrict__ n_points, SortScratch ss)
{
    const int f = blockIdx.y;
    const int pos = blockIdx.x * kBlock + threadIdx.x;
    int *p = ss.perm + (size_t)f * maxNpad, *ip = ss.iperm + (size_t)f * maxNpad;
    if (pos >= maxNpad) return;
    if (pos >= n_points[f]) {                            // phantom lanes and the unused tail: identity
        p[pos] = pos;
        ip[pos] = pos;
        return;
    }
    const int *tmp = ss.cells + (size_t)f * maxNpad * kMaxD;
    const int *end = ss.start + (size_t)f * ((1 << ss.bits) + 1);
    const int i = tmp[pos], c = ss.code[(size_t)f * maxNpad + i];
    const int lo = c ? end[c - 1] : 0, hi = end[c];
    int at = pos;
    if (hi - lo <= 64) {
        at = lo;
        for (int q = lo; q < hi; ++q) at += tmp[q] < i;
    }
    p[at] = i;
    ip[i] = at;
}

template <int D>
void sort_points_d(const KernelDev &kd, const CrfDev &c, const SortScratch &ss, hipStream_t s)
{
    const int F = c.F, nbk = (1 << ss.bits) + 1;
    const dim3 gp = grid_for(kd.maxNpad, F);
    (void)hipMemsetAsync(ss.hist, 0, (size_t)F * nbk * sizeof(int), s);
    k_sort_cells<D><<<gp, kBlock, 0, s>>>(kd, c.n_points, ss);
    k_sort_plan<<<F, kBlock, 0, s>>>(D, (int)gp.x, ss);
    k_sort_code<D><<<gp, kBlock, 0, s>>>(kd, c.n_points, ss);
    scan_frames(ss.hist, ss.start, nbk, nbk, nullptr, ss.tiles, F, s);
    k_sort_scatter<<<gp, kBlock, 0, s>>>(kd.maxNpad, c.n_points, ss);
    k_sort_place<<<gp, kBlock, 0, s>>>(kd.maxNpad, c.n_points, ss);
}

// ---- locality mode, part two: the VERTICES in row-major order of the lattice's own axes --------------------------------------
// A lattice vertex is x in Z^(d+1) with sum 0 and all coordinates congruent mod d+1 (its key = the first d of them).  In the basis
// of the blur directions u_j = (1, .., 1) - (d+1) e_j it is the integer point c with c_j = (x_d - x_j) / (d+1), j < d: blur axis
// j < d is the unit step along c_j and axis d the step along the main diagonal (permutohedral_cpu.h:408-421 in those
// coordinates).  Numbering the vertices in ROW-MAJOR order of c makes the neighbours of consecutive ids consecutive ids of
// another row: a wavefront's blur gather touches 9 distinct 128-byte lines instead of 19 (scripts/sim_vertex_order.py, the C5
// lattice) and the pass takes 19.5 instead of 23.2 us per 8 frames (scripts/ubench/blurorder.hip on the real tables).  Only
// locality depends on the numbering, never a result (a vertex's value is a sum over its own row, neighbours are matched by key).
// One-pass bucket sort on the top ~20 bits of the code (a histogram, its scan, a scatter); a bucket is ~200 consecutive codes -- a
// few lattice rows, one or two vertices -- and keeps its arrival order.

template <int D>
__device__ __forceinline__ void vertex_grid_coords(const int16_t (&key)[D], int (&c)[D])
{
    int xd = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) xd -= key[j];
#pragma unroll
    for (int j = 0; j < D; ++j) c[j] = (xd - key[j]) / (D + 1);          // exact: the coordinates are congruent mod d+1
}

// One workgroup per frame: bounds of the grid coordinates -> row-major strides (coordinate 0 fastest) and the shift that maps a
// code to its bucket.  vplan[f] = {lo[kMaxD], stride[kMaxD], bucket scale, usable, range}; a lattice whose bounding box overflows 62 bits
// (features spread over thousands of cells in every dimension) raises *vbad and is rebuilt with the hash table.
constexpr int kVPlan = 2 * kMaxD + 3;
__global__ void __launch_bounds__(kBlock) k_vsort_plan(int D, int nblocks, SortScratch ss)
{
    __shared__ int red[kBlock / 64][2 * kMaxD];
    __shared__ int bnd[2 * kMaxD];
    const int f = blockIdx.x;
    for (int j = 0; j < D; ++j) {
        int lo = 0x7fffffff, hi = (int)0x80000000;
        for (int b = threadIdx.x; b < nblocks; b += kBlock) {
            const int *pp = ss.vpartial + ((size_t)f * nblocks + b) * 2 * kMaxD;
            lo = min(lo, pp[2 * j]);
            hi = max(hi, pp[2 * j + 1]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o, 64));
            hi = max(hi, __shfl_xor(hi, o, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6][2 * j] = lo;
            red[threadIdx.x >> 6][2 * j + 1] = hi;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * D) {
        int x = red[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) x = (threadIdx.x & 1) ? max(x, red[w][threadIdx.x]) : min(x, red[w][threadIdx.x]);
        bnd[threadIdx.x] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long *plan = ss.vplan + (size_t)f * kVPlan;
        unsigned long long range = 1;
        bool ok = true;
        for (int j = 0; j < D; ++j) {
            const int lo = bnd[2 * j], hi = bnd[2 * j + 1];
            // one EMPTY guard column per dimension: a step of +-1 off the edge of the box lands in it (or beyond `range`) and finds no
            // vertex, so the neighbour search needs no coordinate arithmetic (k_eneighbors)
            const unsigned long long span = hi >= lo ? (unsigned long long)((long long)hi - lo + 2) : 2ull;   // (an empty frame has no vertices)
            plan[j] = hi >= lo ? lo : 0;
            plan[kMaxD + j] = (long long)range;
            if (span > (1ull << 62) / range) ok = false;
            else range *= span;
        }
        // bucket of a code = floor(code * buckets / range) as a multiply-high: the whole histogram is used whatever the range (a
        // shift would leave up to half of it empty and make the buckets twice as long).  0 = the code itself (range <= buckets).
        // An unusable plan "codes" the entries by their own index (k_ecode): the same formula over [0, vcap) keeps them inside.
        const unsigned long long reff = ok ? range : (unsigned long long)ss.vcap, nbuckets = 1ull << ss.vbits;
        plan[2 * kMaxD] = reff <= nbuckets ? 0ll : (long long)((~0ull / reff) << ss.vbits);
        plan[2 * kMaxD + 1] = ok ? 1 : 0;
        plan[2 * kMaxD + 2] = (long long)reff;             // (unusable plan: whatever k_eneighbors derives from the meaningless strides stays inside too)
        if (!ok && ss.vbad) *ss.vbad = 1;                 // pinned host word: the host rebuilds these lattices with the hash
    }
}

// ---- the SORTED build of locality mode (round 4): vertices found and numbered by sorting the entries ---------------------------
// Every entry e = (point, remainder) gets the row-major code of its vertex; a bucket sort on the code's top ~21 bits + an exact
// order inside each bucket by (code, entry) makes the entries of one vertex a RUN whose first element is the vertex's first
// occurrence: run starts -> scan -> ids in row-major order, with no hash table, no first-occurrence pass and no separate vertex
// sort; blur neighbours are then found by CODE (the neighbour along axis j has code +- stride_j) among the few vertices of the
// bucket that code falls in -- one contiguous read instead of a hash probe plus a key recomputation.  Everything downstream
// (offset, rep, V, nbr, the CSR kernels) sees the same arrays as after the hash build, with other ids.
constexpr int kLongBucket = 512;          // buckets beyond this are sorted by a workgroup of their own (identical or clustered features)
constexpr int kVPh = 64;                  // ints per frame of the phantom-entry list: count, then up to 3 (d+1) <= 27 positions
__device__ __forceinline__ void csr_emit(const KernelDev &kd, size_t fe, int pos, int e, int v);   // (with the CSR kernels below)

__device__ __forceinline__ int vsort_bucket(const SortScratch &ss, int f, unsigned long long code)
{
    const unsigned long long scale = (unsigned long long)ss.vplan[(size_t)f * kVPlan + 2 * kMaxD];
    return scale ? (int)__umul64hi(code, scale) : (int)code;
}

template <int D>
__global__ void __launch_bounds__(kBlock) k_ecode(KernelDev kd, const int *__restrict__ n_points, SortScratch ss)
{
    const int f = blockIdx.y;
    const int live = ((n_points[f] + 3) & ~3) * (D + 1);
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= live) return;
    const long long *plan = ss.vplan + (size_t)f * kVPlan;
    unsigned long long code = (unsigned long long)e;                      // (unusable plan: the host rebuilds with the hash; stay in bounds)
    if (plan[2 * kMaxD + 1]) {
        int16_t key[D];
        int c[D];
        load_entry_key<D>(kd, f, e, key);
        vertex_grid_coords<D>(key, c);
        code = 0;
        int xd = 0;
        bool exact = true;
#pragma unroll
        for (int j = 0; j < D; ++j) xd -= key[j];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            code += (unsigned long long)((long long)c[j] - plan[j]) * (unsigned long long)plan[kMaxD + j];
            exact &= (xd - key[j]) == c[j] * (D + 1);
        }
        // keys that wrapped around int16 (features thousands of cells wide: the reference's short arithmetic wraps the same way) are no
        // lattice points any more and the code would not tell them apart: such a frame is rebuilt with the hash, which compares keys
        if (!exact && ss.vbad) *ss.vbad = 1;
    }
    ss.vcode[(size_t)f * ss.vcap + e] = code;
    // the histogram's atomic hands back the entry's arrival number inside its bucket: the scatter below needs no second round of atomics
    kd.csr_pos[(size_t)f * kd.Epad + e] = atomicAdd(&ss.vhist[(size_t)f * ((1 << ss.vbits) + 1) + vsort_bucket(ss, f, code)], 1);
}

// bucket b = positions [vstart[b], vstart[b+1]).  slot_of takes the entries bucket by bucket (unordered inside), rep -- free until
// k_eoffsets -- the same entries under their ORIGINAL id: a vertex's row is ordered by original point index whatever the internal
// order of the points (quirk Q6), and that is the order the runs are given.
__global__ void __launch_bounds__(kBlock) k_escatter(KernelDev kd, const int *__restrict__ n_points, SortScratch ss)
{
    const int f = blockIdx.y;
    const int live = ((n_points[f] + 3) & ~3) * kd.D1;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= live) return;
    const size_t fe = (size_t)f * kd.Epad;
    const int b = vsort_bucket(ss, f, ss.vcode[(size_t)f * ss.vcap + e]);
    const int pos = ss.vstart[(size_t)f * ((1 << ss.vbits) + 1) + b] + kd.csr_pos[fe + e];
    const int pt = e / kd.D1;
    kd.slot_of[fe + pos] = e;
    kd.rep[fe + pos] = kd.perm[(size_t)f * kd.maxNpad + pt] * kd.D1 + (e - pt * kd.D1);
}

__device__ __forceinline__ bool entry_less(unsigned long long ca, int oa, unsigned long long cb, int ob) { return ca < cb || (ca == cb && oa < ob); }

// exact order inside every bucket: (code, original entry id).  kd.slot (the hash table of the other build) takes the sorted entries.
__global__ void __launch_bounds__(kBlock) k_eorder(KernelDev kd, const int *__restrict__ n_points, SortScratch ss)
{
    const int f = blockIdx.y;
    const int live = ((n_points[f] + 3) & ~3) * kd.D1;
    const int pos = blockIdx.x * kBlock + threadIdx.x;
    if (pos >= live) return;
    const int *tmp = kd.slot_of + (size_t)f * kd.Epad, *toe = kd.rep + (size_t)f * kd.Epad;
    const unsigned long long *code = ss.vcode + (size_t)f * ss.vcap;
    const int *start = ss.vstart + (size_t)f * ((1 << ss.vbits) + 1);
    const int e = tmp[pos], oe = toe[pos];
    const unsigned long long ce = code[e];
    const int b = vsort_bucket(ss, f, ce);
    const int lo = start[b], hi = start[b + 1];
    int *sorted = kd.slot + (size_t)f * kd.cap;
    if (hi - lo > kLongBucket) {
        if (pos == lo) sorted[kd.Epad + atomicAdd(&kd.rowmax[f], 1)] = b;                 // (second half of the table: the list of long buckets)
        return;
    }
    int r = 0;
    for (int q = lo; q < hi; ++q) r += entry_less(code[tmp[q]], toe[q], ce, oe);
    sorted[lo + r] = e;
}

// One workgroup per long bucket: bitonic sort of its entries by (code, original id) in place, then out (as k_csr_sort_long).
constexpr int kSortBitmapWords = 15872;  // 62 KB of LDS: original ids spanning up to ~508 000
__global__ void __launch_bounds__(kBlock) k_esort_long(KernelDev kd, SortScratch ss)
{
    __shared__ unsigned bm[kSortBitmapWords];
    __shared__ int chunk_base[kBlock];
    __shared__ int s_mixed, s_min, s_max;
    const int tid = threadIdx.x;
    const int f = blockIdx.y;
    const int nlong = kd.rowmax[f];
    int *tmp = kd.slot_of + (size_t)f * kd.Epad, *toe = kd.rep + (size_t)f * kd.Epad;
    int *sorted = kd.slot + (size_t)f * kd.cap;
    const unsigned long long *code = ss.vcode + (size_t)f * ss.vcap;
    const int *start = ss.vstart + (size_t)f * ((1 << ss.vbits) + 1);
    for (int li = blockIdx.x; li < nlong; li += gridDim.x) {
        const int b = sorted[kd.Epad + li];
        const int lo = start[b], n = start[b + 1] - lo;
        int *r = tmp + lo, *ro = toe + lo;
        // A giant bucket is usually ONE vertex (a whole uniformly coloured region of the image demo: 50 000 entries of the same code),
        // where the order wanted is the order of the original ids: a presence bitmap over the ids' range in LDS, prefix popcounts,
        // done -- O(n) instead of the 136 global-memory stages of the bitonic network below (0.8 ms per such bucket).
        if (tid == 0) { s_mixed = 0; s_min = INT_MAX; s_max = INT_MIN; }
        __syncthreads();
        {
            const unsigned long long c0 = code[r[0]];
            int mn = INT_MAX, mx = INT_MIN, mixed = 0;
            for (int i = tid; i < n; i += kBlock) {
                mixed |= code[r[i]] != c0;
                mn = min(mn, ro[i]);
                mx = max(mx, ro[i]);
            }
            if (mixed) s_mixed = 1;
            atomicMin(&s_min, mn);
            atomicMax(&s_max, mx);
        }
        __syncthreads();
        const int base_id = s_min, words = (s_max - s_min) / 32 + 1;
        if (!s_mixed && words <= kSortBitmapWords && n >= 1024) {       // (uniform; below ~1000 entries the network is cheaper than a walk over the span's words)
            for (int w = tid; w < words; w += kBlock) bm[w] = 0u;
            __syncthreads();
            for (int i = tid; i < n; i += kBlock) atomicOr(&bm[(ro[i] - base_id) >> 5], 1u << ((ro[i] - base_id) & 31));
            __syncthreads();
            const int chunk = (words + kBlock - 1) / kBlock;                 // words per lane
            int cnt = 0;
            for (int w = tid * chunk; w < min((tid + 1) * chunk, words); ++w) cnt += __popc(bm[w]);
            chunk_base[tid] = cnt;
            __syncthreads();
            if (tid == 0) {
                int run = 0;
                for (int t = 0; t < kBlock; ++t) { const int c = chunk_base[t]; chunk_base[t] = run; run += c; }
            }
            __syncthreads();
            for (int i = tid; i < n; i += kBlock) {
                const int b = ro[i] - base_id, w = b >> 5;
                int pos = chunk_base[w / chunk] + __popc(bm[w] & ((1u << (b & 31)) - 1u));
                for (int u = (w / chunk) * chunk; u < w; ++u) pos += __popc(bm[u]);
                sorted[lo + pos] = r[i];
            }
            __syncthreads();
            continue;
        }
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        for (int k = 2; k <= np2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < np2; i += kBlock) {
                    const int l = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);
                    if (l > i && l < n) {
                        const int a = r[i], c = r[l], oa = ro[i], oc = ro[l];
                        if (entry_less(code[c], oc, code[a], oa)) { r[i] = c; r[l] = a; ro[i] = oc; ro[l] = oa; }
                    }
                }
                __syncthreads();
            }
        }
        for (int i = threadIdx.x; i < n; i += kBlock) sorted[lo + i] = r[i];
        __syncthreads();
    }
}

// flag[pos] = 1 where a run (= a vertex) starts; the (at most 3 (d+1)) entries of the phantom points are listed by position --
// they make vertices but no products, and sit at the END of their runs (their original ids are the largest)
__global__ void __launch_bounds__(kBlock) k_eflag(KernelDev kd, const int *__restrict__ n_points, SortScratch ss)
{
    const int f = blockIdx.y;
    const int N = n_points[f], live = ((N + 3) & ~3) * kd.D1;
    const int pos = blockIdx.x * kBlock + threadIdx.x;
    if (pos > kd.Epad) return;
    int v = 0;
    if (pos < live) {
        const int *sorted = kd.slot + (size_t)f * kd.cap;
        const unsigned long long *code = ss.vcode + (size_t)f * ss.vcap;
        const int e = sorted[pos];
        v = pos == 0 || code[sorted[pos - 1]] != code[e];
        if (e >= N * kd.D1) ss.vph[(size_t)f * kVPh + 1 + atomicAdd(&ss.vph[(size_t)f * kVPh], 1)] = pos;
    }
    kd.flag[(size_t)f * (kd.Epad + 1) + pos] = v;
}

// ids, representatives, and the CSR of splat contributions straight from the sorted runs: a run minus its trailing phantom entries
// IS the vertex's row in ascending original point order (what k_csr_count / fill / order / sort_long produce for the hash build)
__global__ void __launch_bounds__(kBlock) k_eoffsets(KernelDev kd, const int *__restrict__ n_points, SortScratch ss)
{
    const int f = blockIdx.y;
    const int N = n_points[f], live = ((N + 3) & ~3) * kd.D1;
    const int pos = blockIdx.x * kBlock + threadIdx.x;
    if (pos >= live) return;
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const int e = kd.slot[(size_t)f * kd.cap + pos];
    const int first = kd.flag[f1 + pos];
    const int id = kd.prefix[f1 + pos] + first - 1;                       // (exclusive scan of the run starts)
    const int *ph = ss.vph + (size_t)f * kVPh;
    const int nph = ph[0];
    int before = 0;
    for (int u = 0; u < nph; ++u) before += ph[1 + u] < pos;               // phantom entries ahead of this position
    const int cpos = pos - before;
    kd.offset[fe + e] = id;
    if (first) {
        kd.rep[fe + id] = e;                                              // the run's first element = the vertex's first occurrence
        ss.vkey[(size_t)f * ss.vcap + id] = ss.vcode[(size_t)f * ss.vcap + e];
        kd.rowptr[f1 + id] = cpos;
    }
    if (pos == 0) kd.rowptr[f1 + kd.V[f]] = live - nph;
    if (e < N * kd.D1) csr_emit(kd, fe, cpos, e, id);
}

// vhist (free after its scan) takes, per bucket, the id of the bucket's first vertex: ids are bucket-major, so bucket b holds the
// vertices [bvert[b], bvert[b+1]) and the neighbour search reads one 8-byte pair instead of two positions and two prefixes
__global__ void __launch_bounds__(kBlock) k_ebucket_vertices(KernelDev kd, SortScratch ss)
{
    const int f = blockIdx.y, nbk = (1 << ss.vbits) + 1;
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= nbk) return;
    ss.vhist[(size_t)f * nbk + b] = kd.prefix[(size_t)f * (kd.Epad + 1) + ss.vstart[(size_t)f * nbk + b]];   // (prefix[live] = V)
}

// The window splat's per-vertex record (KernelDev::srec), packed once per build from the arrays the sorted build has just written.
__global__ void __launch_bounds__(kBlock) k_pack_srec(KernelDev kd, int F, XcdMap nb)
{
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int V = kd.V[f];
    const int v = fb.bx * kBlock + threadIdx.x;
    if (v >= V) return;
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const int s = kd.rowptr[f1 + v], t = kd.rowptr[f1 + v + 1];
    const char2 *off1 = reinterpret_cast<const char2 *>(kd.nearoff) + (size_t)f * 2 * kd.Epad, *off2 = off1 + kd.Epad;
    const char2 o1 = off1[v], o2 = off2[v];
    const unsigned len7 = (unsigned)min(t - s, 127), nx = kd.fastn[fe + v] ? 1u : 0u;
    uint4 r;
    r.x = (unsigned)s;
    r.y = (s < t ? ((unsigned)kd.csr_pt[fe + s] & 0xffffffu) : 0u) | (len7 << 24) | (nx << 31);
    r.z = s < t ? __float_as_uint(kd.csr_w[fe + s]) : 0u;
    r.w = (unsigned)(unsigned char)o1.x | ((unsigned)(unsigned char)o1.y << 8) | ((unsigned)(unsigned char)o2.x << 16) | ((unsigned)(unsigned char)o2.y << 24);
    kd.srec[fe + v] = r;
}

// One thread per vertex v, all d + 1 axes at once: n2_j(v) by code -- +1 along grid coordinate j (j < d), -1 along every coordinate
// (j = d); a step off the box lands in a guard column or beyond `range` and matches nothing -- among the vertices of the bucket the
// code falls in (ids are bucket-major: bucket b holds the vertices [bvert[b], bvert[b+1])).  The look-ups of the d + 1 axes are
// independent, so their loads are issued together (a thread per (axis, vertex) ran three dependent loads at a time: 0.46 ms per 8
// C5 frames).  The relation is mutual, so finding n2 fills both sides (the table was preset to -1).  ref: permutohedral_cpu.h:408-421.
template <int D>
__global__ void __launch_bounds__(kBlock) k_eneighbors(KernelDev kd, int F, XcdMap nb, SortScratch ss)
{
    constexpr int D1 = D + 1;
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int V = kd.V[f];
    const int v = fb.bx * kBlock + threadIdx.x;
    if (v >= V) return;
    const long long *plan = ss.vplan + (size_t)f * kVPlan;
    const unsigned long long *vkey = ss.vkey + (size_t)f * ss.vcap;
    const int *bvert = ss.vhist + (size_t)f * ((1 << ss.vbits) + 1);
    const unsigned long long code = vkey[v], range = (unsigned long long)plan[2 * kMaxD + 2];
    unsigned long long target[D1], sum = 0;
    bool ok[D1];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const unsigned long long st = (unsigned long long)plan[kMaxD + j];
        target[j] = code + st;                            // + 1 along c_j
        ok[j] = target[j] < range;
        sum += st;
    }
    target[D] = code - sum;                               // - 1 along every coordinate
    ok[D] = code >= sum;
    int v0[D1], v1[D1];
    // axis 0 needs no bucket: it is the code's fastest coordinate and the ids follow the codes, so code + 1 -- if it exists -- is the
    // next id (one coalesced read of the neighbouring lane's key instead of a bucket pair and a key)
    v0[0] = v + 1;
    v1[0] = min(v + 2, V);
#pragma unroll
    for (int j = 1; j < D1; ++j) {
        const int b = ok[j] ? vsort_bucket(ss, f, target[j]) : 0;
        v0[j] = bvert[b];
        v1[j] = bvert[b + 1];
    }
    unsigned long long first[D1];
#pragma unroll
    for (int j = 0; j < D1; ++j) first[j] = vkey[min(v0[j], V - 1)];
    bool next0 = false;
    int dist[kNdistAxes] = {};           // |neighbour id - id| along the first axes (the splat's window, KernelDev::ndist)
#pragma unroll
    for (int j = 0; j < D1; ++j) {
        if (!ok[j] || v0[j] >= v1[j]) continue;
        int found = first[j] == target[j] ? v0[j] : -1;
        for (int u = v0[j] + 1; found < 0 && u < v1[j]; ++u)
            if (vkey[u] == target[j]) found = u;
        if (found < 0) continue;
        if (j == 0) {                    // axis 0 is the code's fastest coordinate: code + 1 is the next id
            next0 = true;
            if (found != v + 1 && kd.tbl_bad) kd.tbl_bad[1] = 1;
        }
        if (j < kNdistAxes) dist[j] = abs(found - v);
        if ((j == 1 || j == 2) && kd.nearoff) {          // the window passes' table: signed byte offsets (unused unless all of them fit)
            const int o = found - v;
            int8_t *no = kd.nearoff + ((size_t)f * 2 + (j - 1)) * kd.Epad * 2;
            const bool fits = o >= -127 && o <= 127;     // (the main diagonal, axis d, steps DOWN the codes: o < 0)
            no[2 * (size_t)v + 1] = (int8_t)(fits ? o : 0);               // my n2
            no[2 * (size_t)found] = (int8_t)(fits ? -o : 0);              // its n1
        }
        int *nbp = kd.nbr + ((size_t)f * D1 + j) * kd.Epad * 2;
        nbp[2 * v + 1] = found;          // my n2
        nbp[2 * found] = v;              // its n1
    }
    if (kd.fastn) kd.fastn[(size_t)f * kd.Epad + v] = next0 ? 1 : 0;
    if (kd.ndist) {
#pragma unroll
        for (int j = 1; j < (D1 < kNdistAxes ? D1 : kNdistAxes); ++j) {
            int m = dist[j];
#pragma unroll
            for (int w = 32; w >= 1; w >>= 1) m = max(m, __shfl_xor(m, w, 64));
            // (a look first: the maximum settles after a few wavefronts, and 70 000 atomics on one address cost 2 ms)
            if ((threadIdx.x & 63) == 0 && m > __builtin_nontemporal_load(&kd.ndist[j])) atomicMax(&kd.ndist[j], m);
        }
    }
}

// offset[e] = dense id of e's vertex; the first entry of each vertex registers as its
// representative.  Ids come out in first-insertion order, exactly the reference's ids.
__global__ void __launch_bounds__(kBlock) k_offsets(KernelDev kd, const int *__restrict__ n_points)
{
    const int f = blockIdx.y;
    const int live = ((n_points[f] + 3) & ~3) * kd.D1;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= live) return;
    const size_t fe = (size_t)f * kd.Epad;
    const int *prefix = kd.prefix + (size_t)f * (kd.Epad + 1);
    const int r = kd.slot[(size_t)f * kd.cap + kd.slot_of[fe + e]];
    const int id = prefix[r];
    kd.offset[fe + e] = id;
    if (r == e) kd.rep[fe + id] = e;
}

template <int D>
__device__ __forceinline__ int find_vertex(const KernelDev &kd, int f, const int16_t (&key)[D])
{
    const unsigned mask = (unsigned)kd.cap - 1u;
    unsigned h = hash_key<D>(key) & mask;
    const int *slot = kd.slot + (size_t)f * kd.cap;
    for (;;) {
        const int r = slot[h];
        if (r == kEmpty) return -1;
        int16_t other[D];
        load_entry_key<D>(kd, f, r, other);
        bool same = true;
#pragma unroll
        for (int i = 0; i < D; ++i) same &= (other[i] == key[i]);
        if (same) return kd.prefix[(size_t)f * (kd.Epad + 1) + r];
        h = (h + 1u) & mask;
    }
}

// One thread per (axis j, vertex v): the blur neighbours, ref: :408-421.  The relation is mutual -- B = n2_j(A)
// iff A = n1_j(B) -- so one hash probe per (axis, vertex) finds n2 and fills both sides; the table was preset to
// -1 (absent) by the caller.
template <int D>
__global__ void __launch_bounds__(kBlock) k_neighbors(KernelDev kd, int F, XcdMap nb)
{
    constexpr int D1 = D + 1;
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int V = kd.V[f];
    const int idx = fb.bx * kBlock + threadIdx.x;
    if (idx >= V * D1) return;
    const int j = idx / V, v = idx - j * V;
    int16_t key[D], n2[D];
    load_entry_key<D>(kd, f, kd.rep[(size_t)f * kd.Epad + v], key);
#pragma unroll
    for (int t = 0; t < D; ++t) n2[t] = (int16_t)(key[t] + 1);
#pragma unroll
    for (int t = 0; t < D; ++t)          // axis d touches only the implied last coordinate
        if (t == j) n2[t] = (int16_t)(key[t] - D);
    const int b = find_vertex<D>(kd, f, n2);
    if (b < 0) return;
    int *nbp = kd.nbr + ((size_t)f * D1 + j) * kd.Epad * 2;
    nbp[2 * v + 1] = b;                  // my n2
    nbp[2 * b] = v;                      // its n1
}

// The compact neighbour table of the sorted build (KernelDev::nbrc): per block of kNbrcBlock = 64 vertices (one wavefront) and axis
// the smallest n1 and the smallest n2, and per vertex the two 16-bit offsets from them.  Ids follow the codes, so n1 / n2 grow with v
// and a block's neighbours span about a block; should a span not fit 16 bits the pinned word is raised and the blur keeps to the
// 32-bit table.  The host does not know V yet (a grid over the CAPACITY was 0.16 ms of empty workgroups at C5 x 8): kNbrcGrid
// workgroups per (frame, axis) walk the blocks below V, four per wavefront and round with their loads issued together (one block
// per round and a workgroup-wide minimum through LDS: 156 us per 4 C5 frames).
constexpr int kNbrcGrid = 64;
__device__ __forceinline__ int wave_min(int x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x = min(x, __shfl_xor(x, m, 64));
    return x;
}
__global__ void __launch_bounds__(kBlock) k_nbr_compact(KernelDev kd, int limit)          // limit: 0xffff (the instrumented library's test hook lowers it)
{
    static_assert(kNbrcBlock == 64, "one wavefront per base");
    constexpr int U = 4;
    const int f = blockIdx.z, j = blockIdx.y;
    const int V = kd.V[f];
    const size_t fj = (size_t)f * kd.D1 + j;
    const int2 *tab = reinterpret_cast<const int2 *>(kd.nbr) + fj * kd.Epad;
    unsigned *out = reinterpret_cast<unsigned *>(kd.nbrc) + fj * kd.Epad;
    int2 *base = reinterpret_cast<int2 *>(kd.nbrc_base) + fj * (kd.Epad / kNbrcBlock + 1);
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nwaves = kNbrcGrid * (kBlock / 64);
    for (int b0 = wave * U; b0 * kNbrcBlock < V; b0 += nwaves * U) {
        int2 n[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = (b0 + u) * kNbrcBlock + lane;
            n[u] = v < V ? tab[v] : make_int2(-1, -1);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u, v = b * kNbrcBlock + lane;
            if (b * kNbrcBlock >= V) break;
            const int m0 = wave_min(n[u].x < 0 ? INT_MAX : n[u].x), m1 = wave_min(n[u].y < 0 ? INT_MAX : n[u].y);
            const int lo0 = m0 == INT_MAX ? 0 : m0, lo1 = m1 == INT_MAX ? 0 : m1;
            if (lane == 0) base[b] = make_int2(lo0, lo1);
            if (v >= V) continue;
            const int o0 = n[u].x < 0 ? 0xffff : n[u].x - lo0, o1 = n[u].y < 0 ? 0xffff : n[u].y - lo1;
            if ((n[u].x >= 0 && o0 >= limit) || (n[u].y >= 0 && o1 >= limit)) kd.tbl_bad[0] = 1;
            out[v] = (unsigned)(o0 & 0xffff) | ((unsigned)(o1 & 0xffff) << 16);
        }
    }
}

// nbr16 = (n1 + 1) | (n2 + 1) << 16 per (axis, vertex) for frames whose ids fit 16 bits (the fused engine's table)
__global__ void __launch_bounds__(kBlock) k_neighbors16(KernelDev kd)
{
    const int f = blockIdx.y;
    const int V = kd.V[f];
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= V * kd.D1) return;
    const int j = idx / V, v = idx - j * V;
    const int2 r = reinterpret_cast<const int2 *>(kd.nbr)[((size_t)f * kd.D1 + j) * kd.Epad + v];
    kd.nbr16[((size_t)f * kd.D1 + j) * kd.Epad + v] = (unsigned)(r.x + 1) | ((unsigned)(r.y + 1) << 16);
}

// two-hop table of the pass pair (first + 2p, first + 2p + 1) for single-frame engines (KernelDev::nbr2): everything k_blur2x2 looks
// up on its way, looked up once at build time, so that a launch of two passes is a table read and ONE level of gathers
__global__ void __launch_bounds__(kBlock) k_neighbors_2hop(KernelDev kd, int first, int npairs)
{
    const int f = blockIdx.y;
    const int V = kd.V[f];
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= V * npairs) return;
    const int p = idx / V, v = idx - p * V;
    const int2 *nj = reinterpret_cast<const int2 *>(kd.nbr) + ((size_t)f * kd.D1 + first + 2 * p) * kd.Epad;
    const int2 *nj1 = nj + kd.Epad;
    const int2 ab = nj1[v], nv = nj[v];
    const int2 na = ab.x >= 0 ? nj[ab.x] : make_int2(-1, -1), nb = ab.y >= 0 ? nj[ab.y] : make_int2(-1, -1);
    int4 *out = reinterpret_cast<int4 *>(kd.nbr2) + (((size_t)f * npairs + p) * kd.Epad + v) * 2;
    out[0] = make_int4(nv.x, nv.y, ab.x, ab.y);
    out[1] = make_int4(na.x, na.y, nb.x, nb.y);
}

// ---- CSR of splat contributions: vertex -> (point, weight), points ascending ------------
// The reference splats sequentially over points (:653-661), so each vertex's sum is taken
// in ascending point order; an ordered CSR walk reproduces that sum bit for bit.

__global__ void __launch_bounds__(kBlock) k_csr_count(KernelDev kd, const int *__restrict__ n_points)
{
    const int f = blockIdx.y;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_points[f] * kd.D1) return;               // real points only
    atomicAdd(&kd.flag[(size_t)f * (kd.Epad + 1) + kd.offset[(size_t)f * kd.Epad + e]], 1);
}

// rowmax[f] = longest CSR row of the frame (lets the fused engine pick its splat strategy): one atomic per workgroup.
__global__ void __launch_bounds__(kBlock) k_row_max(KernelDev kd)
{
    __shared__ int wave_max[kBlock / 64];
    const int f = blockIdx.y;
    const int v = blockIdx.x * kBlock + threadIdx.x;
    const int *rp = kd.rowptr + (size_t)f * (kd.Epad + 1);
    int m = v < kd.V[f] ? rp[v + 1] - rp[v] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = max(m, wave_max[w]);
        if (m > 0) atomicMax(&kd.rowmax[f], m);
    }
}

__global__ void __launch_bounds__(kBlock) k_csr_fill(KernelDev kd, const int *__restrict__ n_points)
{
    const int f = blockIdx.y;
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_points[f] * kd.D1) return;
    const size_t f1 = (size_t)f * (kd.Epad + 1);
    const int v = kd.offset[(size_t)f * kd.Epad + e];
    const int k = atomicSub(&kd.flag[f1 + v], 1) - 1;   // counts run down to zero
    // rows are ordered by ORIGINAL point index whatever the internal order of the points: list the entry under its original id
    int oe = e;
    if (kd.perm) {
        const int pt = e / kd.D1;
        oe = kd.perm[(size_t)f * kd.maxNpad + pt] * kd.D1 + (e - pt * kd.D1);
    }
    kd.slot_of[(size_t)f * kd.Epad + kd.rowptr[f1 + v] + k] = oe;   // slot_of reused: unsorted rows
}

// entry id in the internal order of an entry listed under its original id
__device__ __forceinline__ int internal_entry(const KernelDev &kd, int f, int oe)
{
    if (!kd.iperm) return oe;
    const int pt = oe / kd.D1;
    return kd.iperm[(size_t)f * kd.maxNpad + pt] * kd.D1 + (oe - pt * kd.D1);
}

// Rows come out of k_csr_fill in arrival order; the splat needs them in ascending entry order.  Short rows (the
// common case: a vertex holds a handful of contributions): every entry counts the smaller entries of its row.
// That is quadratic in the row length, so rows longer than kLongRow (clustered or identical features put tens of
// thousands of entries on one vertex) are only listed here and sorted by k_csr_sort_long, one workgroup per row.
constexpr int kLongRow = 128;

__device__ __forceinline__ void csr_emit(const KernelDev &kd, size_t fe, int pos, int e, int v)
{
    kd.csr_pt[fe + pos] = e / kd.D1;
    kd.csr_w[fe + pos] = kd.bary[fe + e];
    kd.csr_pos[fe + e] = pos;
    if (kd.Epad < 65535) kd.pk[fe + e] = (unsigned)(v + 1) | ((unsigned)pos << 16);
}

__global__ void __launch_bounds__(kBlock) k_csr_order(KernelDev kd, const int *__restrict__ n_points)
{
    const int f = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_points[f] * kd.D1) return;
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const int *rows = kd.slot_of + fe;
    const int e = rows[p];                               // (original id: what the row is ordered by)
    const int ei = internal_entry(kd, f, e);
    const int v = kd.offset[fe + ei];
    const int s = kd.rowptr[f1 + v], t = kd.rowptr[f1 + v + 1];
    if (t - s > kLongRow) {                              // flag[] is all zero again after k_csr_fill: reuse it as the list of
        if (p == s) kd.flag[f1 + atomicAdd(&kd.flag[f1 + kd.Epad], 1)] = v;   // long rows, its last slot as their count
        return;
    }
    int rank = 0;
    for (int q = s; q < t; ++q) rank += (rows[q] < e);
    csr_emit(kd, fe, s + rank, ei, v);
}

// One workgroup per long row: bitonic sort of the row's entry ids in place (global memory, the row belongs to this
// workgroup alone), O(n log^2 n) compare-exchanges instead of n^2 compares.
__global__ void __launch_bounds__(kBlock) k_csr_sort_long(KernelDev kd)
{
    const int f = blockIdx.y;
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const int nlong = kd.flag[f1 + kd.Epad];
    int *rows = kd.slot_of + fe;
    for (int li = blockIdx.x; li < nlong; li += gridDim.x) {
        const int v = kd.flag[f1 + li];
        const int s = kd.rowptr[f1 + v], n = kd.rowptr[f1 + v + 1] - s;
        int *r = rows + s;
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        // every compare-exchange orders its pair ascending (first step of a merge: partner = mirror image inside the
        // block of k, then half-cleaners), so the virtual +infinity elements behind n never have to move
        for (int k = 2; k <= np2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < np2; i += kBlock) {
                    const int l = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);
                    if (l > i && l < n) {
                        const int a = r[i], b = r[l];
                        if (a > b) { r[i] = b; r[l] = a; }
                    }
                }
                __syncthreads();
            }
        }
        for (int i = threadIdx.x; i < n; i += kBlock) csr_emit(kd, fe, s + i, internal_entry(kd, f, r[i]), v);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// splat / blur / slice  (value width L at run time; one thread per (vertex|point, label))
// ---------------------------------------------------------------------------------------

// val0[v+1][l] = sum over the vertex's contributions, ascending point order.  in == nullptr
// means the all-ones input of the normalisation pass (pairwise3d.h:23-24).
constexpr int kSplatUnroll = 16;
typedef float lccrf_f4u __attribute__((ext_vector_type(4), aligned(4)));     // four labels of a row, wherever L puts them
__global__ void __launch_bounds__(kBlock) k_splat(KernelDev kd, const float *__restrict__ in,
                                                  int in_stride, int L)
{
    const int f = blockIdx.y;
    const int V = kd.V[f];
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= V * L) return;
    const int v = idx / L, l = idx - v * L;
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const int s = kd.rowptr[f1 + v], t = kd.rowptr[f1 + v + 1];
    if (kd.longrow_ok && t - s > kLongRowMin && kd.longcnt[f] <= kLongRowCap) return;   // k_splat_long's
    const float *x = in ? in + (size_t)f * in_stride : nullptr;
    float acc = 0.0f;
    int p = s;
    // long rows (a coarse kernel over many points: the reference's image demo has rows of ~900 entries): the adds must go one by one
    // in point order, the LOADS need not -- kSplatUnroll entries' indices, weights and inputs in flight per round trip instead of one
    // (the demo's splat 944 -> ~100 us per launch)
    for (; p + kSplatUnroll <= t; p += kSplatUnroll) {
        int pt[kSplatUnroll];
        float w[kSplatUnroll], xv[kSplatUnroll];
#pragma unroll
        for (int i = 0; i < kSplatUnroll; ++i) { pt[i] = kd.csr_pt[fe + p + i]; w[i] = kd.csr_w[fe + p + i]; }
#pragma unroll
        for (int i = 0; i < kSplatUnroll; ++i) xv[i] = x ? x[(size_t)pt[i] * L + l] : 1.0f;
#pragma unroll
        for (int i = 0; i < kSplatUnroll; ++i) acc += w[i] * xv[i];
    }
    for (; p < t; ++p) {
        const float xv = x ? x[(size_t)kd.csr_pt[fe + p] * L + l] : 1.0f;
        acc += kd.csr_w[fe + p] * xv;
    }
    kd.val0[(size_t)f * kd.vstride + kd.vbase + (long)v * L + l] = acc;
}

// ... four labels per thread from L = 4 on: a row's indices and weights are read once per four labels, the inputs as 16-byte loads
__global__ void __launch_bounds__(kBlock) k_splat4(KernelDev kd, const float *__restrict__ in, int in_stride, int L, int C)
{
    const int f = blockIdx.y;
    const int V = kd.V[f];
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= V * C) return;
    const int v = idx / C, l = (idx - v * C) * 4, nl = min(4, L - l);
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const int s = kd.rowptr[f1 + v], t = kd.rowptr[f1 + v + 1];
    if (kd.longrow_ok && t - s > kLongRowMin && kd.longcnt[f] <= kLongRowCap) return;   // k_splat_long's
    const float *x = in + (size_t)f * in_stride + l;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int U = 8;
    int p = s;
    if (nl == 4) {
        for (; p + U <= t; p += U) {
            int pt[U];
            float w[U];
            lccrf_f4u xv[U];
#pragma unroll
            for (int i = 0; i < U; ++i) { pt[i] = kd.csr_pt[fe + p + i]; w[i] = kd.csr_w[fe + p + i]; }
#pragma unroll
            for (int i = 0; i < U; ++i) xv[i] = *reinterpret_cast<const lccrf_f4u *>(x + (size_t)pt[i] * L);
#pragma unroll
            for (int i = 0; i < U; ++i) { acc[0] += w[i] * xv[i].x; acc[1] += w[i] * xv[i].y; acc[2] += w[i] * xv[i].z; acc[3] += w[i] * xv[i].w; }
        }
        for (; p < t; ++p) {
            const float w = kd.csr_w[fe + p];
            const lccrf_f4u xv = *reinterpret_cast<const lccrf_f4u *>(x + (size_t)kd.csr_pt[fe + p] * L);
            acc[0] += w * xv.x; acc[1] += w * xv.y; acc[2] += w * xv.z; acc[3] += w * xv.w;
        }
    } else {
        for (; p < t; ++p) {
            const float w = kd.csr_w[fe + p];
            const float *xp = x + (size_t)kd.csr_pt[fe + p] * L;
            for (int u = 0; u < nl; ++u) acc[u] += w * xp[u];
        }
    }
    float *d = kd.val0 + (size_t)f * kd.vstride + kd.vbase + (long)v * L + l;
    for (int u = 0; u < nl; ++u) d[u] = acc[u];
}

// Rows of thousands of entries (a coarse kernel over many points -- the appearance kernel of the reference's image demo puts whole
// uniformly coloured regions on one vertex): the adds of a row must still go one by one in point order (quirk Q6), but nothing says
// the LOADS must.  A workgroup per listed row (KernelDev::longrow, filled by the build): all lanes form the products
// w[p] * in[pt[p]][l] of a tile of entries in LDS, then lane l < L adds its label's column top to bottom -- the same products, the
// same order, the same bits as the in-line walk.
constexpr int kLongTile = 8192;          // products per tile (floats); two tiles in LDS
__global__ void __launch_bounds__(kBlock) k_splat_long(KernelDev kd, const float *__restrict__ in, int in_stride, int L)
{
    __shared__ __attribute__((aligned(16))) float prod[2][kLongTile];
    const int f = blockIdx.y;
    const int *lr = kd.longrow + (size_t)f * kLongRowCap;
    const int n = kd.longcnt[f];
    if (n > kLongRowCap) return;
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const float *x = in ? in + (size_t)f * in_stride : nullptr;
    // a tile holds ec entries of every label, label-major: prod[l * ecp + e] (ecp = ec + 4, a multiple of 4: the adder reads its
    // label's column four entries per 16-byte LDS load; the loaders' stores land ecp words apart -- a few ways of bank conflict)
    const int tid = threadIdx.x, ec = (kLongTile / L - 4) & ~3, ecp = ec + 4;
    constexpr int kLoaders = kBlock - 64;                 // wavefront 0 adds, the other three load: the tile being added and the tile
    for (int i = blockIdx.x; i < n; i += gridDim.x) {     // being loaded are different halves of `prod`, one barrier per tile
        const int v = lr[i];
        const int s = kd.rowptr[f1 + v], t = kd.rowptr[f1 + v + 1];
        const int ntiles = (t - s + ec - 1) / ec;
        float acc = 0.0f;
        for (int k = -1; k < ntiles; ++k) {
            if (tid >= 64) {                              // load tile k + 1
                const int p0 = s + (k + 1) * ec;
                const int m = k + 1 < ntiles ? min(ec, t - p0) * L : 0;
                float *dst = prod[(k + 1) & 1];
                int idx = tid - 64;
                for (; idx + 7 * kLoaders < m; idx += 8 * kLoaders) {        // eight products per lane and round trip
                    int pt[8], l[8], e[8];
                    float w[8], xv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        e[u] = (idx + u * kLoaders) / L;
                        l[u] = idx + u * kLoaders - e[u] * L;
                        pt[u] = kd.csr_pt[fe + p0 + e[u]];
                        w[u] = kd.csr_w[fe + p0 + e[u]];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) xv[u] = x ? x[(size_t)pt[u] * L + l[u]] : 1.0f;
#pragma unroll
                    for (int u = 0; u < 8; ++u) dst[l[u] * ecp + e[u]] = w[u] * xv[u];
                }
                for (; idx < m; idx += kLoaders) {
                    const int e = idx / L, l = idx - e * L;
                    const float xv = x ? x[(size_t)kd.csr_pt[fe + p0 + e] * L + l] : 1.0f;
                    dst[l * ecp + e] = kd.csr_w[fe + p0 + e] * xv;
                }
            } else if (k >= 0 && tid < L) {               // add tile k: label tid's column, top to bottom
                const int p0 = s + k * ec;
                const int m = min(ec, t - p0);
                const float *src = prod[k & 1] + tid * ecp;
                int e = 0;
                for (; e + 8 <= m; e += 8) {
                    const float4 q0 = *reinterpret_cast<const float4 *>(src + e), q1 = *reinterpret_cast<const float4 *>(src + e + 4);
                    acc += q0.x; acc += q0.y; acc += q0.z; acc += q0.w;
                    acc += q1.x; acc += q1.y; acc += q1.z; acc += q1.w;
                }
                for (; e < m; ++e) acc += src[e];
            }
            __syncthreads();
        }
        if (tid < L) kd.val0[(size_t)f * kd.vstride + kd.vbase + (long)v * L + tid] = acc;
    }
}

// the list of long rows, once per build
__global__ void __launch_bounds__(kBlock) k_long_rows(KernelDev kd)
{
    const int f = blockIdx.y;
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= kd.V[f]) return;
    const size_t f1 = (size_t)f * (kd.Epad + 1);
    if (kd.rowptr[f1 + v + 1] - kd.rowptr[f1 + v] <= kLongRowMin) return;
    const int i = atomicAdd(&kd.longcnt[f], 1);
    if (i < kLongRowCap) kd.longrow[(size_t)f * kLongRowCap + i] = v;
}

// the generic splat: rows in line, the long ones by a workgroup each
inline void launch_splat(const KernelDev &kd, const float *in, int in_stride, int L, int F, int maxV, hipStream_t s)
{
    // four labels per thread where the rows are short (a fine lattice: about one entry per vertex at d = 5 or 6); a coarse kernel's
    // rows of tens to hundreds of entries want every (vertex, label) walk in flight on its own (the image demo: 344 vs 554 us)
    const bool short_rows = (long)kd.maxN * kd.D1 <= 4L * std::max(maxV, 1);
    if (L >= 4 && in && short_rows) k_splat4<<<grid_for((long)maxV * ((L + 3) / 4), F), kBlock, 0, s>>>(kd, in, in_stride, L, (L + 3) / 4);
    else k_splat<<<grid_for((long)maxV * L, F), kBlock, 0, s>>>(kd, in, in_stride, L);
    if (kd.longrow_ok) k_splat_long<<<dim3((unsigned)std::max(256 / std::max(F, 1), 8), (unsigned)F), kBlock, 0, s>>>(kd, in, in_stride, L);
}

// One Jacobi blur pass along axis j.  ref: :663-679.
__global__ void __launch_bounds__(kBlock) k_blur(KernelDev kd, const float *__restrict__ src,
                                                 float *__restrict__ dst, int j, int L)
{
    const int f = blockIdx.y;
    const int V = kd.V[f];
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= V * L) return;
    const int v = idx / L, l = idx - v * L;
    const size_t fv = (size_t)f * kd.vstride;
    const int2 nb = reinterpret_cast<const int2 *>(kd.nbr)[((size_t)f * kd.D1 + j) * kd.Epad + v];
    const float *o = src + fv + kd.vbase;      // o[v*L+l], v = -1 is the all-zero "absent" vertex
    const float a = o[(long)nb.x * L + l];
    const float c = o[(long)nb.y * L + l];
    dst[fv + kd.vbase + (long)v * L + l] = o[(long)v * L + l] + 0.5f * (a + c);
}

// ... four labels per thread from L = 4 on (16-byte accesses on 4-byte alignment: a vertex's row starts wherever L puts it): the
// pass at L = 21 is bound by instructions per byte, not by bytes (55 us per 558 000 vertices against 18 at the streaming rate).
__global__ void __launch_bounds__(kBlock) k_blur4(KernelDev kd, const float *__restrict__ src, float *__restrict__ dst, int j, int L, int C)
{
    const int f = blockIdx.y;
    const int V = kd.V[f];
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= V * C) return;
    const int v = idx / C, l = (idx - v * C) * 4;         // C = ceil(L / 4) chunks per vertex
    const size_t fv = (size_t)f * kd.vstride;
    const int2 nb = reinterpret_cast<const int2 *>(kd.nbr)[((size_t)f * kd.D1 + j) * kd.Epad + v];
    const float *o = src + fv + kd.vbase;                 // o[v*L+l], v = -1 is the all-zero "absent" vertex
    float *d = dst + fv + kd.vbase + (long)v * L + l;
    const float *pa = o + (long)nb.x * L + l, *pc = o + (long)nb.y * L + l, *po = o + (long)v * L + l;
    if (l + 4 <= L) {
        const lccrf_f4u a = *reinterpret_cast<const lccrf_f4u *>(pa), c = *reinterpret_cast<const lccrf_f4u *>(pc),
                        m = *reinterpret_cast<const lccrf_f4u *>(po);
        lccrf_f4u r;
        r.x = m.x + 0.5f * (a.x + c.x);
        r.y = m.y + 0.5f * (a.y + c.y);
        r.z = m.z + 0.5f * (a.z + c.z);
        r.w = m.w + 0.5f * (a.w + c.w);
        *reinterpret_cast<lccrf_f4u *>(d) = r;
    } else {
        for (int u = 0; u < L - l; ++u) d[u] = po[u] + 0.5f * (pa[u] + pc[u]);
    }
}

// ... and ONE label (the normalisation's filter of all-ones, pairwise3d.h:22-27: seven passes per build): four vertices per thread --
// two 16-byte table loads, one 16-byte centre load, eight 4-byte gathers, one 16-byte store instead of four times (8 + 4 + 2 x 4 + 4).
typedef int lccrf_i4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ void __launch_bounds__(kBlock) k_blur1x4(KernelDev kd, const float *__restrict__ src, float *__restrict__ dst, int j)
{
    const int f = blockIdx.y;
    const int V = kd.V[f];
    const int v = 4 * (blockIdx.x * kBlock + threadIdx.x);
    if (v >= V) return;
    const size_t fv = (size_t)f * kd.vstride;
    const float *o = src + fv + kd.vbase;                 // o[-1] = the all-zero "absent" vertex
    float *d = dst + fv + kd.vbase;
    const int *nbp = kd.nbr + (((size_t)f * kd.D1 + j) * kd.Epad + v) * 2;
    if (v + 4 <= V) {
        const lccrf_i4u n0 = *reinterpret_cast<const lccrf_i4u *>(nbp), n1 = *reinterpret_cast<const lccrf_i4u *>(nbp + 4);
        const lccrf_f4u c = *reinterpret_cast<const lccrf_f4u *>(o + v);
        const float a0 = o[n0.x], b0 = o[n0.y], a1 = o[n0.z], b1 = o[n0.w], a2 = o[n1.x], b2 = o[n1.y], a3 = o[n1.z], b3 = o[n1.w];
        lccrf_f4u r;
        r.x = c.x + 0.5f * (a0 + b0);
        r.y = c.y + 0.5f * (a1 + b1);
        r.z = c.z + 0.5f * (a2 + b2);
        r.w = c.w + 0.5f * (a3 + b3);
        *reinterpret_cast<lccrf_f4u *>(d + v) = r;
    } else {
        for (int u = v; u < V; ++u) d[u] = o[u] + 0.5f * (o[nbp[2 * (u - v)]] + o[nbp[2 * (u - v) + 1]]);
    }
}

enum SliceMode { SLICE_NORM = 0, SLICE_APPLY_FIRST = 1, SLICE_APPLY = 2, SLICE_PLAIN = 3 };

// slice (+ what the caller does with it).  ref: :684-694, pairwise3d.h:25-27,73-78,
// densecrf3d.h:154-158.
__global__ void __launch_bounds__(kBlock) k_slice(KernelDev kd, CrfDev c, const float *__restrict__ val,
                                                  int L, int mode)
{
    const int f = blockIdx.y;
    const int N = c.n_points[f];
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= N * L) return;
    const int i = idx / L, l = idx - i * L;
    const size_t fe = (size_t)f * kd.Epad;
    const float *vf = val + (size_t)f * kd.vstride + kd.vbase;
    float t = 0.0f;
    // (every corner's id and weight, then every gather, before the first use: one round trip per level instead of one per corner)
    int o[kMaxD + 1];
    float wgt[kMaxD + 1], x[kMaxD + 1];
#pragma unroll
    for (int j = 0; j <= kMaxD; ++j)
        if (j < kd.D1) { o[j] = kd.offset[fe + (size_t)i * kd.D1 + j]; wgt[j] = kd.bary[fe + (size_t)i * kd.D1 + j] * kd.alpha; }
#pragma unroll
    for (int j = 0; j <= kMaxD; ++j)
        if (j < kd.D1) x[j] = vf[(long)o[j] * L + l];
#pragma unroll
    for (int j = 0; j <= kMaxD; ++j)
        if (j < kd.D1) t += wgt[j] * x[j];
    if (mode == SLICE_NORM) {
        kd.norm[(size_t)f * kd.maxN + i] = 1.0f / (t + 1e-20f);
    } else if (mode == SLICE_PLAIN) {                      // the bare filter: out = compute(in), permutohedral_cpu.h:634-699
        c.next[((size_t)f * c.maxN + i) * L + l] = t;
    } else {
        const size_t q = ((size_t)f * c.maxN + i) * L + l;
        const float base = (mode == SLICE_APPLY_FIRST) ? -c.unary[q] : c.next[q];
        c.next[q] = base + kd.w * kd.norm[(size_t)f * kd.maxN + i] * t;
    }
}

// the normalisation's slice with d + 1 known at compile time: every load of a point issued before the first use (the generic kernel
// above walks its corners one dependent gather at a time: 92 -> 35 us per 8 C5 frames).  Same operations, same order.
template <int D1>
__global__ void __launch_bounds__(kBlock) k_slice_norm(KernelDev kd, CrfDev c, const float *__restrict__ val)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.n_points[f]) return;
    const size_t fe = (size_t)f * kd.Epad;
    const float *vf = val + (size_t)f * kd.vstride + kd.vbase;
    int o[D1];
    float w[D1], x[D1];
#pragma unroll
    for (int j = 0; j < D1; ++j) { o[j] = kd.offset[fe + (size_t)i * D1 + j]; w[j] = kd.bary[fe + (size_t)i * D1 + j]; }
#pragma unroll
    for (int j = 0; j < D1; ++j) x[j] = vf[o[j]];
    float t = 0.0f;
#pragma unroll
    for (int j = 0; j < D1; ++j) t += (w[j] * kd.alpha) * x[j];
    kd.norm[(size_t)f * kd.maxN + i] = 1.0f / (t + 1e-20f);
}

// ---- two-label specialisations (the SLAM configuration, L = 2): one thread per vertex / point,
// both labels in a float2.  Same operations per label as the generic kernels above.
// (by value: `ok ? lds[i] : zero` selects between two ADDRESSES and parks the zero in scratch memory)
__device__ __forceinline__ float2 lds_or_zero(bool ok, const float2 *p)
{
    float2 r = make_float2(0.0f, 0.0f);                   // what the absent vertex's slot holds
    if (ok) r = *p;
    return r;
}

// The two-label splat of a COARSE kernel over many points (KernelDev::long_mode: a 2-D smoothness kernel on 100 000 points has 2000
// vertices and rows of 150 entries on average): eight entries' loads per round trip, the adds in order; the listed rows are left to
// k_splat_long (L = 2: the same value layout).  8 frames of C5 never come here (1.2 entries per row).
__global__ void __launch_bounds__(kBlock) k_splat2l(KernelDev kd, const float2 *__restrict__ in, int in_stride, int F, XcdMap nb)
{
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int v = fb.bx * (int)blockDim.x + threadIdx.x;
    if (v >= kd.V[f]) return;
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const int s = kd.rowptr[f1 + v], t = kd.rowptr[f1 + v + 1];
    if (kd.longrow_ok && t - s > kLongRowMin && kd.longcnt[f] <= kLongRowCap) return;   // k_splat_long's
    const float2 *x = in + (size_t)f * in_stride;
    float a0 = 0.0f, a1 = 0.0f;
    int p = s;
    for (; p + 8 <= t; p += 8) {
        int pt[8];
        float w[8];
        float2 q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { pt[i] = kd.csr_pt[fe + p + i]; w[i] = kd.csr_w[fe + p + i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = x[pt[i]];
#pragma unroll
        for (int i = 0; i < 8; ++i) { a0 += w[i] * q[i].x; a1 += w[i] * q[i].y; }
    }
    for (; p < t; ++p) {
        const float w = kd.csr_w[fe + p];
        const float2 q = x[kd.csr_pt[fe + p]];
        a0 += w * q.x;
        a1 += w * q.y;
    }
    reinterpret_cast<float2 *>(kd.val0 + (size_t)f * kd.vstride + kd.vbase)[v] = make_float2(a0, a1);
}

// ... and with rows of tens of entries on average (long_mode 2) a WAVEFRONT per vertex: its 64 lanes load 64 entries' products in one
// round trip (coalesced index / weight reads, one gather), then every lane adds them in order off the others' registers (a uniform
// lane index: v_readlane) -- two interleaved chains, one per label.  2000 vertices x 150 entries: 45 -> ~6 us.
__global__ void __launch_bounds__(kBlock) k_splat2v(KernelDev kd, const float2 *__restrict__ in, int in_stride, int F, XcdMap nb)
{
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int lane = threadIdx.x & 63;
    const int v = fb.bx * ((int)blockDim.x / 64) + (threadIdx.x >> 6);      // (grid: one wavefront per vertex)
    if (v >= kd.V[f]) return;
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const int s = kd.rowptr[f1 + v], t = kd.rowptr[f1 + v + 1];
    if (kd.longrow_ok && t - s > kLongRowMin && kd.longcnt[f] <= kLongRowCap) return;   // k_splat_long's
    const float2 *x = in + (size_t)f * in_stride;
    float a0 = 0.0f, a1 = 0.0f;
    for (int base = s; base < t; base += 64) {
        const int p = base + lane, m = min(64, t - base);
        float p0 = 0.0f, p1 = 0.0f;
        if (p < t) {
            const float w = kd.csr_w[fe + p];
            const float2 q = x[kd.csr_pt[fe + p]];
            p0 = w * q.x;
            p1 = w * q.y;
        }
        for (int u = 0; u < m; ++u) {                     // (u is uniform: v_readlane, no trip through the LDS crossbar)
            a0 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0), u));
            a1 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1), u));
        }
    }
    if (lane == 0) reinterpret_cast<float2 *>(kd.val0 + (size_t)f * kd.vstride + kd.vbase)[v] = make_float2(a0, a1);
}

// BLUR0 (sorted build, KernelDev::fast0_ok): the FIRST blur pass rides along.  Axis 0 is the fastest coordinate of the row-major
// vertex code, so a vertex's axis-0 neighbours are v - 1 and v + 1 (or absent): the workgroup's row sums go to LDS -- every thread
// sums one row, the first and the last only for their neighbours' sake (blockDim - 2 results per workgroup) -- and
// t[v] = s[v] + 0.5 (s[n1] + s[n2]) is formed from there: the operations of k_blur2 on the stored sums, in the same order, hence the
// same bits, without the pass's launch, its table-to-gather round trip and its 16 bytes per vertex of reads and writes.
template <bool BLUR0>
__global__ void __launch_bounds__(kBlock) k_splat2(KernelDev kd, const float2 *__restrict__ in, int in_stride, int F, XcdMap nb)
{
    __shared__ float2 tile[BLUR0 ? kBlock : 1];
    __shared__ uint8_t next[BLUR0 ? kBlock : 1];
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int V = kd.V[f];
    const int v0 = BLUR0 ? fb.bx * ((int)blockDim.x - 2) - 1 : fb.bx * (int)blockDim.x;      // the vertex of thread 0
    const int v = v0 + threadIdx.x;
    if (v0 + (BLUR0 ? 1 : 0) >= V) return;                // (the whole workgroup)
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const float2 *x = in + (size_t)f * in_stride;
    float a0 = 0.0f, a1 = 0.0f;
    uint8_t nx = 0;
    if (v >= 0 && v < V) {
        const int s = kd.rowptr[f1 + v], t = kd.rowptr[f1 + v + 1];
        if (BLUR0) nx = kd.fastn[fe + v];
        for (int p = s; p < t; ++p) {
            const float w = kd.csr_w[fe + p];
            const float2 q = x[kd.csr_pt[fe + p]];
            a0 += w * q.x;
            a1 += w * q.y;
        }
    }
    float2 *out = reinterpret_cast<float2 *>(kd.val0 + (size_t)f * kd.vstride + kd.vbase);
    if (!BLUR0) {
        if (v < V) out[v] = make_float2(a0, a1);
        return;
    }
    tile[threadIdx.x] = make_float2(a0, a1);
    next[threadIdx.x] = nx;
    __syncthreads();
    if (threadIdx.x == 0 || threadIdx.x == blockDim.x - 1 || v >= V) return;
    const float2 p = lds_or_zero(next[threadIdx.x - 1], &tile[threadIdx.x - 1]), q = lds_or_zero(nx, &tile[threadIdx.x + 1]);   // n1 = v - 1, n2 = v + 1
    out[v] = make_float2(a0 + 0.5f * (p.x + q.x), a1 + 0.5f * (p.y + q.y));
}

// ... and the passes along axes 1 (and 2) too, when the sorted build found their neighbours within a few ids (KernelDev::ndist): an
// OVERLAPPED window.  A workgroup sums the rows of B consecutive vertices into LDS and runs the passes 0 .. P-1 there; a neighbour
// outside the window reads as zero, which spoils its neighbours' values pass by pass -- by at most `halo` = 1 + dist_1 (+ dist_2)
// positions from either end, so the inner B - 2 halo results are exactly what P launches of k_blur2 would have stored (the same
// operations on the same values in the same order) and only those are written.  One launch, one table read per extra pass.
template <int LANES, int U, bool REC = false, bool RNT = true>
__global__ void __launch_bounds__(LANES) k_splat2w(KernelDev kd, const float2 *__restrict__ in, int in_stride, int F, XcdMap nb, int P, int halo)
{
    constexpr int B = LANES * U;                          // the window: U vertices per lane, at stride LANES (coalesced)
    __shared__ float2 buf[2][B];
    __shared__ uint8_t next[B];
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int V = kd.V[f];
    const int v0 = fb.bx * (B - 2 * halo) - halo;         // the window's first vertex
    if (v0 + halo >= V) return;                           // (the whole workgroup)
    const size_t fe = (size_t)f * kd.Epad, f1 = (size_t)f * (kd.Epad + 1);
    const float2 *x = in + (size_t)f * in_stride;
    const char2 *off1 = reinterpret_cast<const char2 *>(kd.nearoff) + (size_t)f * 2 * kd.Epad, *off2 = off1 + kd.Epad;   // axes 1, 2
    const int tid = threadIdx.x;
    float a0[U], a1[U];
    char2 o1[U], o2[U];
    uint8_t nx[U];
    int s[U], t[U];
    int pt0[U];
    float w0[U];
    float2 q0[U];
    if (REC) {
        // REC: one 16-byte record per vertex (KernelDev::srec) says what seven loads said -- row start and length, the first entry's
        // point and weight, fastn and the four byte offsets; a row of 127 entries or more reads its end from the row pointers
        uint4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = v0 + tid + u * LANES;
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            r[u] = make_uint4(0u, 0u, 0u, 0u);
            if (v >= 0 && v < V) {                        // (read once per iteration: non-temporal, out of the value arrays' way in L2)
                const u4v *rp = reinterpret_cast<const u4v *>(kd.srec + fe + v);
                const u4v q = RNT ? __builtin_nontemporal_load(rp) : *rp;
                r[u] = make_uint4(q.x, q.y, q.z, q.w);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int len7 = (int)((r[u].y >> 24) & 127u);
            s[u] = (int)r[u].x;
            t[u] = s[u] + len7;
            if (len7 == 127) t[u] = kd.rowptr[f1 + v0 + tid + u * LANES + 1];
            nx[u] = (uint8_t)(r[u].y >> 31);
            pt0[u] = (int)(r[u].y & 0xffffffu);
            w0[u] = __uint_as_float(r[u].z);
            o1[u] = make_char2((signed char)(r[u].w & 0xffu), (signed char)((r[u].w >> 8) & 0xffu));
            o2[u] = make_char2((signed char)((r[u].w >> 16) & 0xffu), (signed char)(r[u].w >> 24));
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = v0 + tid + u * LANES;
            const bool live = v >= 0 && v < V;
            s[u] = t[u] = 0;
            nx[u] = 0;
            o1[u] = o2[u] = make_char2(0, 0);
            if (live) {
                s[u] = kd.rowptr[f1 + v];
                t[u] = kd.rowptr[f1 + v + 1];
                nx[u] = kd.fastn[fe + v];
                o1[u] = off1[v];
                if (P > 2) o2[u] = off2[v];
            }
        }
        // the first entry of each of the lane's U rows together (rows hold 1.2 entries on average: most are done after this), then
        // whatever is left of each row in order
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool any = s[u] < t[u];
            pt0[u] = any ? kd.csr_pt[fe + s[u]] : 0;
            w0[u] = any ? kd.csr_w[fe + s[u]] : 0.0f;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) q0[u] = x[pt0[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float b0 = 0.0f, b1 = 0.0f;
        if (s[u] < t[u]) {
            b0 += w0[u] * q0[u].x;
            b1 += w0[u] * q0[u].y;
        }
        for (int p = s[u] + 1; p < t[u]; ++p) {
            const float w = kd.csr_w[fe + p];
            const float2 q = x[kd.csr_pt[fe + p]];
            b0 += w * q.x;
            b1 += w * q.y;
        }
        a0[u] = b0;
        a1[u] = b1;
        buf[0][tid + u * LANES] = make_float2(b0, b1);
        next[tid + u * LANES] = nx[u];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {                         // pass 0: n1 = v - 1, n2 = v + 1
        const int i = tid + u * LANES;
        const float2 p = lds_or_zero(i > 0 && next[i > 0 ? i - 1 : 0], &buf[0][i > 0 ? i - 1 : 0]);
        const float2 q = lds_or_zero(nx[u] && i + 1 < B, &buf[0][i + 1 < B ? i + 1 : i]);
        a0[u] = a0[u] + 0.5f * (p.x + q.x);
        a1[u] = a1[u] + 0.5f * (p.y + q.y);
        buf[1][i] = make_float2(a0[u], a1[u]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {                         // pass 1 (P >= 2)
        const int i = tid + u * LANES;
        const unsigned i1 = (unsigned)(i + o1[u].x), i2 = (unsigned)(i + o1[u].y);
        const float2 p = lds_or_zero(o1[u].x != 0 && i1 < (unsigned)B, &buf[1][i1 < (unsigned)B ? i1 : 0]);
        const float2 q = lds_or_zero(o1[u].y != 0 && i2 < (unsigned)B, &buf[1][i2 < (unsigned)B ? i2 : 0]);
        a0[u] = a0[u] + 0.5f * (p.x + q.x);
        a1[u] = a1[u] + 0.5f * (p.y + q.y);
    }
    if (P > 2) {                                          // pass 2 (buf[0] was last read before the barrier above)
#pragma unroll
        for (int u = 0; u < U; ++u) buf[0][tid + u * LANES] = make_float2(a0[u], a1[u]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * LANES;
            const unsigned i1 = (unsigned)(i + o2[u].x), i2 = (unsigned)(i + o2[u].y);
            const float2 p = lds_or_zero(o2[u].x != 0 && i1 < (unsigned)B, &buf[0][i1 < (unsigned)B ? i1 : 0]);
            const float2 q = lds_or_zero(o2[u].y != 0 && i2 < (unsigned)B, &buf[0][i2 < (unsigned)B ? i2 : 0]);
            a0[u] = a0[u] + 0.5f * (p.x + q.x);
            a1[u] = a1[u] + 0.5f * (p.y + q.y);
        }
    }
    float2 *out = reinterpret_cast<float2 *>(kd.val0 + (size_t)f * kd.vstride + kd.vbase);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = tid + u * LANES, v = v0 + i;
        if (i >= halo && i < B - halo && v < V) out[v] = make_float2(a0[u], a1[u]);
    }
}

typedef int lccrf_v4i __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ int4 load_nbr_pair(const int *p)
{
    const lccrf_v4i x = NT ? __builtin_nontemporal_load(reinterpret_cast<const lccrf_v4i *>(p)) : *reinterpret_cast<const lccrf_v4i *>(p);
    return make_int4(x.x, x.y, x.z, x.w);
}
// Two vertices per thread: the neighbour pairs (int4), the centres (float4) and the results (float4) move as 16-byte
// accesses (the frame's value array is laid out so that vertex 2t is 16-byte aligned, see Engine::add_kernel).
// What bounds the pass is the CU's vector-memory path, not HBM bytes: an 8-byte gather costs ~16 + 2 cycles per distinct
// 128-byte line it touches (scripts/ubench/tacost.hip), which is why locality mode -- fewer distinct lines per gather --
// helps and why everything tried on top of it lost (notes/r3_experiments.md: 2-8 pairs per lane with all loads issued
// first, a presence-bit + id-list neighbour table, a 4096/8192-vertex LDS tile serving the in-tile neighbours).
// NT: the neighbour table is read once per pass -- with many frames in flight (a working set beyond every cache) it is
// loaded non-temporally, out of the value array's way in L2; with a few frames everything lives in L2 / the Infinity Cache
// and the plain load is the faster one (scripts/ubench/phasecost.hip: 7.2 -> 6.8 us per pass of one C5 frame).
template <bool NT>
__global__ void __launch_bounds__(kBlock) k_blur2(KernelDev kd, const float *__restrict__ src,
                                                  float *__restrict__ dst, int j, int F, XcdMap nb)
{
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int V = kd.V[f];
    const int v = 2 * (fb.bx * (int)blockDim.x + threadIdx.x);
    if (v >= V) return;
    const float2 *o = reinterpret_cast<const float2 *>(src + (size_t)f * kd.vstride + kd.vbase);   // o[-1] = absent
    float2 *d = reinterpret_cast<float2 *>(dst + (size_t)f * kd.vstride + kd.vbase);
    const int *nbp = kd.nbr + (((size_t)f * kd.D1 + j) * kd.Epad + v) * 2;
    if (v + 1 < V) {
        const int4 nb4 = load_nbr_pair<NT>(nbp);
        const float4 c = *reinterpret_cast<const float4 *>(o + v);
        const float2 x0 = o[nb4.x], y0 = o[nb4.y], x1 = o[nb4.z], y1 = o[nb4.w];
        *reinterpret_cast<float4 *>(d + v) = make_float4(c.x + 0.5f * (x0.x + y0.x), c.y + 0.5f * (x0.y + y0.y),
                                                         c.z + 0.5f * (x1.x + y1.x), c.w + 0.5f * (x1.y + y1.y));
    } else {
        const int2 n2 = *reinterpret_cast<const int2 *>(nbp);
        const float2 c = o[v], x = o[n2.x], y = o[n2.y];
        d[v] = make_float2(c.x + 0.5f * (x.x + y.x), c.y + 0.5f * (x.y + y.y));
    }
}

// The same pass off the COMPACT neighbour table of the sorted build (KernelDev::nbrc): 4 instead of 8 table bytes per vertex, the
// ids rebuilt as base-of-the-block + 16-bit offset (0xffff = absent -> -1).  Same neighbours, same operations, same bits.
template <bool NT>
__global__ void __launch_bounds__(kBlock) k_blur2c(KernelDev kd, const float *__restrict__ src,
                                                   float *__restrict__ dst, int j, int F, XcdMap nb)
{
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int V = kd.V[f];
    const int v = 2 * (fb.bx * (int)blockDim.x + threadIdx.x);
    if (v >= V) return;
    const float2 *o = reinterpret_cast<const float2 *>(src + (size_t)f * kd.vstride + kd.vbase);   // o[-1] = absent
    float2 *d = reinterpret_cast<float2 *>(dst + (size_t)f * kd.vstride + kd.vbase);
    const size_t fj = (size_t)f * kd.D1 + j;
    const int2 base = reinterpret_cast<const int2 *>(kd.nbrc_base)[fj * (kd.Epad / kNbrcBlock + 1) + v / kNbrcBlock];
    const unsigned *tp = reinterpret_cast<const unsigned *>(kd.nbrc) + fj * kd.Epad + v;            // (v even: 8-byte aligned)
    if (v + 1 < V) {
        typedef unsigned lccrf_v2u __attribute__((ext_vector_type(2)));
        const lccrf_v2u t = NT ? __builtin_nontemporal_load(reinterpret_cast<const lccrf_v2u *>(tp)) : *reinterpret_cast<const lccrf_v2u *>(tp);
        const unsigned a0 = t.x & 0xffffu, b0 = t.x >> 16, a1 = t.y & 0xffffu, b1 = t.y >> 16;
        const int n0 = a0 == 0xffffu ? -1 : base.x + (int)a0, m0 = b0 == 0xffffu ? -1 : base.y + (int)b0;
        const int n1 = a1 == 0xffffu ? -1 : base.x + (int)a1, m1 = b1 == 0xffffu ? -1 : base.y + (int)b1;
        const float4 c = *reinterpret_cast<const float4 *>(o + v);
        const float2 x0 = o[n0], y0 = o[m0], x1 = o[n1], y1 = o[m1];
        *reinterpret_cast<float4 *>(d + v) = make_float4(c.x + 0.5f * (x0.x + y0.x), c.y + 0.5f * (x0.y + y0.y),
                                                         c.z + 0.5f * (x1.x + y1.x), c.w + 0.5f * (x1.y + y1.y));
    } else {
        const unsigned t = *tp, a0 = t & 0xffffu, b0 = t >> 16;
        const int n0 = a0 == 0xffffu ? -1 : base.x + (int)a0, m0 = b0 == 0xffffu ? -1 : base.y + (int)b0;
        const float2 c = o[v], x = o[n0], y = o[m0];
        d[v] = make_float2(c.x + 0.5f * (x.x + y.x), c.y + 0.5f * (x.y + y.y));
    }
}

// TWO passes (axes j, j + 1) in one launch, no extra tables -- for one or two frames in flight, where a pass is a chain of
// latencies (launch ~2.5 us, table load, gather: ~7 us per pass of one C5 frame against ~0.5 us of streaming) and every launch
// saved counts: out[v] = t[v] + 0.5 (t[a] + t[b]) with {a, b} = the axis-(j+1) neighbours of v and
// t[x] = s[x] + 0.5 (s[n1_j(x)] + s[n2_j(x)]) recomputed for x = v, a, b.  The same operations in the same order as two
// launches of k_blur2 (permutohedral_cpu.h:663-679), hence the same bits; the absent vertex (-1) has no neighbours and
// t[-1] = 0 + 0.5 (0 + 0) = 0 exactly, which is what the separate passes leave in its slot.  9 gathers instead of 4, a
// three-level chain instead of twice two levels + a launch: 13.6 -> 10.3 us per pair of passes (scripts/ubench/phasecost.hip).
__global__ void __launch_bounds__(kBlock) k_blur2x2(KernelDev kd, const float *__restrict__ src, float *__restrict__ dst, int j, int F, XcdMap nb)
{
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int v = fb.bx * (int)blockDim.x + threadIdx.x;
    if (v >= kd.V[f]) return;
    const float2 *o = reinterpret_cast<const float2 *>(src + (size_t)f * kd.vstride + kd.vbase);   // o[-1] = absent
    float2 *d = reinterpret_cast<float2 *>(dst + (size_t)f * kd.vstride + kd.vbase);
    const int2 *nj = reinterpret_cast<const int2 *>(kd.nbr) + ((size_t)f * kd.D1 + j) * kd.Epad;
    const int2 *nj1 = nj + kd.Epad;
    const int2 ab = nj1[v], nv = nj[v];
    const float2 sv = o[v];
    const int2 na = ab.x >= 0 ? nj[ab.x] : make_int2(-1, -1), nbb = ab.y >= 0 ? nj[ab.y] : make_int2(-1, -1);
    const float2 sa = o[ab.x], sb = o[ab.y], v1 = o[nv.x], v2 = o[nv.y];
    const float2 a1 = o[na.x], a2 = o[na.y], b1 = o[nbb.x], b2 = o[nbb.y];
    const float2 tv = make_float2(sv.x + 0.5f * (v1.x + v2.x), sv.y + 0.5f * (v1.y + v2.y));
    const float2 ta = make_float2(sa.x + 0.5f * (a1.x + a2.x), sa.y + 0.5f * (a1.y + a2.y));
    const float2 tb = make_float2(sb.x + 0.5f * (b1.x + b2.x), sb.y + 0.5f * (b1.y + b2.y));
    d[v] = make_float2(tv.x + 0.5f * (ta.x + tb.x), tv.y + 0.5f * (ta.y + tb.y));
}

// ... and with the two-hop table of the pair (KernelDev::nbr2, filled by the streaming build of single-frame engines) the launch is a
// table read and one level of gathers: same operations, same order, same bits
__global__ void __launch_bounds__(kBlock) k_blur2x2t(KernelDev kd, const float *__restrict__ src, float *__restrict__ dst, int pair, int npairs, int F,
                                                     XcdMap nb)
{
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= F) return;
    const int v = fb.bx * (int)blockDim.x + threadIdx.x;
    if (v >= kd.V[f]) return;
    const float2 *o = reinterpret_cast<const float2 *>(src + (size_t)f * kd.vstride + kd.vbase);   // o[-1] = absent
    float2 *d = reinterpret_cast<float2 *>(dst + (size_t)f * kd.vstride + kd.vbase);
    const int4 *tb = reinterpret_cast<const int4 *>(kd.nbr2) + (((size_t)f * npairs + pair) * kd.Epad + v) * 2;
    const int4 h0 = tb[0], h1 = tb[1];                   // {v1, v2, a, b}, {a1, a2, b1, b2}
    const float2 sv = o[v];
    const float2 v1 = o[h0.x], v2 = o[h0.y], sa = o[h0.z], sb = o[h0.w];
    const float2 a1 = o[h1.x], a2 = o[h1.y], b1 = o[h1.z], b2 = o[h1.w];
    const float2 tv = make_float2(sv.x + 0.5f * (v1.x + v2.x), sv.y + 0.5f * (v1.y + v2.y));
    const float2 ta = make_float2(sa.x + 0.5f * (a1.x + a2.x), sa.y + 0.5f * (a1.y + a2.y));
    const float2 tbv = make_float2(sb.x + 0.5f * (b1.x + b2.x), sb.y + 0.5f * (b1.y + b2.y));
    d[v] = make_float2(tv.x + 0.5f * (ta.x + tbv.x), tv.y + 0.5f * (ta.y + tbv.y));
}

constexpr int kSliceBlurMaxFrames = 1;       // the last blur pass inside the slice (k_slice2<D1, true>) when passes go one per launch, up to this many frames in flight (with the sorted build: two frames +2 % without it, four and eight +-0: `FRAMES=2 WORKLOAD=c5 scripts/gpu_env_ab.sh LCCRF_SLICE_BLUR_MAX=8 ""`)
constexpr int kPairFuseMaxFrames = 1;        // (measured, `FRAMES=1 WORKLOAD=c5 scripts/gpu_env_ab.sh LCCRF_NO_PAIR_FUSE=1 ""`: one C5 frame 52.5 -> 45.2 us per iteration; two or four frames in flight: +-0)
// ... or, whatever the number of frames, when the launch is SMALL: up to ~0.7 M vertices over all frames (one C5 frame: 0.59 M; two: +-0)
// the passes are launch- and latency-bound, e.g. 8 frames of 5000 points (30 000 vertices each): 9 launches of ~3.8 us per iteration
constexpr long kPairFuseMaxVertices = 700000;
inline bool pair_fuse(int F, int maxV)
{
    static const bool off = ab_env("LCCRF_NO_PAIR_FUSE") != nullptr;      // A/B switch: same results either way
    static const char *force = ab_env("LCCRF_PAIR_FUSE_MAX");             //  (A/B: frames-in-flight threshold)
    if (off) return false;
    if (force) return F <= atoi(force);
    return F <= kPairFuseMaxFrames || (long)F * maxV <= kPairFuseMaxVertices;
}

// (the table non-temporally: C5 with 5 / 6 / 7 frames in flight 23.7 -> 26.1 / 26.9 -> 25.7 / 27.0 -> 24.8 us per frame and iteration)
constexpr int kBlurNtMinFrames = 6;
inline void launch_blur2(const KernelDev &kd, const float *src, float *dst, int j, int F, int maxV, hipStream_t s)
{
    XcdMap nb;
    const int blk = iter_block(F);
    const dim3 g = grid_xcd((maxV + 1) / 2, F, &nb, blk);
    static const bool no_compact = ab_env("LCCRF_NO_COMPACT_NBR") != nullptr;      // A/B switch: same results either way
    static const char *env_nt = ab_env("LCCRF_BLUR_NT");                            // A/B: 0 plain loads, 1 non-temporal
    const bool nt = env_nt ? atoi(env_nt) != 0 : F >= kBlurNtMinFrames;
    if (kd.nbrc && kd.nbrc_ok && !no_compact) {
        if (nt) k_blur2c<true><<<g, blk, 0, s>>>(kd, src, dst, j, F, nb);
        else k_blur2c<false><<<g, blk, 0, s>>>(kd, src, dst, j, F, nb);
    } else if (nt) k_blur2<true><<<g, blk, 0, s>>>(kd, src, dst, j, F, nb);
    else k_blur2<false><<<g, blk, 0, s>>>(kd, src, dst, j, F, nb);
}

// slice + apply for L = 2; the LAST kernel of the step also does the softmax (saves a pass over next).
// BLUR (one frame in flight, odd d + 1: the pass that is left over when passes go two per launch): `val` holds the values BEFORE
// the last blur pass and every point blurs its own d + 1 vertices on the way -- t = s[o] + 0.5 (s[n1(o)] + s[n2(o)]) along the last
// axis, the operations of k_blur2 in the same order, so the same bits -- instead of a launch of its own for that pass (a vertex
// shared by several points is blurred once per point: 1.2 x the work at C5, one launch and one chain of latencies less).
template <int D1, bool BLUR = false>
__global__ void __launch_bounds__(kBlock) k_slice2(KernelDev kd, CrfDev c, const float *__restrict__ val,
                                                   int first, int last, float relax, XcdMap nb)
{
    const FrameBlock fb = frame_block(nb);
    const int f = fb.f;
    if (f >= c.F) return;
    const int i = fb.bx * (int)blockDim.x + threadIdx.x;
    if (i >= c.n_points[f]) return;
    const size_t fe = (size_t)f * kd.Epad;
    const float2 *vf = reinterpret_cast<const float2 *>(val + (size_t)f * kd.vstride + kd.vbase);
    float t0 = 0.0f, t1 = 0.0f;
    if (BLUR) {
        const int2 *nl = reinterpret_cast<const int2 *>(kd.nbr) + ((size_t)f * D1 + (D1 - 1)) * kd.Epad;
        int o[D1];
        int2 n[D1];
        float2 x[D1], a[D1], b[D1];
#pragma unroll
        for (int j = 0; j < D1; ++j) o[j] = kd.offset[fe + (size_t)i * D1 + j];
#pragma unroll
        for (int j = 0; j < D1; ++j) { n[j] = nl[o[j]]; x[j] = vf[o[j]]; }
#pragma unroll
        for (int j = 0; j < D1; ++j) { a[j] = vf[n[j].x]; b[j] = vf[n[j].y]; }
#pragma unroll
        for (int j = 0; j < D1; ++j) {
            const float wgt = kd.bary[fe + (size_t)i * D1 + j] * kd.alpha;
            const float bx = x[j].x + 0.5f * (a[j].x + b[j].x), by = x[j].y + 0.5f * (a[j].y + b[j].y);
            t0 += wgt * bx;
            t1 += wgt * by;
        }
    } else {
#pragma unroll
        for (int j = 0; j < D1; ++j) {
            const float wgt = kd.bary[fe + (size_t)i * D1 + j] * kd.alpha;
            const float2 x = vf[kd.offset[fe + (size_t)i * D1 + j]];
            t0 += wgt * x.x;
            t1 += wgt * x.y;
        }
    }
    const size_t q = (size_t)f * c.maxN + i;
    float2 base;
    if (first) {
        const float2 u = reinterpret_cast<const float2 *>(c.unary)[q];
        base = make_float2(-u.x, -u.y);
    } else {
        base = reinterpret_cast<const float2 *>(c.next)[q];
    }
    const float wn = kd.w * kd.norm[q];
    const float2 nx = make_float2(base.x + wn * t0, base.y + wn * t1);
    if (last) {
        float2 *Q = reinterpret_cast<float2 *>(c.Q);
        Q[q] = softmax2(1.0f * nx.x, 1.0f * nx.y, Q[q], relax);
    } else {
        reinterpret_cast<float2 *>(c.next)[q] = nx;
    }
}

// ---------------------------------------------------------------------------------------
// CRF-level kernels
// ---------------------------------------------------------------------------------------

// unary[i][:] from a label and the three energy tables {u, n[L], p[L]}.  densecrf3d.h:116-129.
__global__ void __launch_bounds__(kBlock) k_unary_from_label_tbl(CrfDev c, const int16_t *__restrict__ label, UnaryTable tbl)
{
    const int f = blockIdx.y;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= c.n_points[f] * c.L) return;
    const int i = idx / c.L, m = idx - i * c.L;
    const int t = label[(size_t)f * c.maxN + (c.perm ? c.perm[(size_t)f * c.perm_stride + i] : i)];
    float u;
    if (t < 0 || t >= c.L) u = tbl.v[0];      // -1 = unknown; out-of-range labels (UB in the reference) likewise
    else u = (m == t) ? tbl.v[1 + c.L + t] : tbl.v[1 + t];
    c.unary[((size_t)f * c.maxN + i) * c.L + m] = u;
}

// out = softmax_fe(scale * in) (blended with the old out when relax != 1).
__global__ void __launch_bounds__(kBlock) k_softmax(CrfDev c, const float *__restrict__ in,
                                                    float *__restrict__ out, float scale, float relax)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= c.n_points[f]) return;
    const size_t q = ((size_t)f * c.maxN + i) * c.L;
    exp_and_normalize_row(in + q, out + q, c.L, scale, relax);
}

// ... for 3 to 32 labels with a LANE PER (point, label): a wavefront takes 64 / L consecutive points, whose rows are one contiguous
// block (coalesced loads and stores, no staging); the row maximum and the row sum are formed by every lane of the row from the
// others' values (__shfl) -- the sum in label order 0 .. L-1, one add at a time, as densecrf3d.h:80-84 forms it; each lane's
// exponential is computed once instead of twice (the same argument gives the same bits).  The lane-per-point kernel above reads
// rows L floats apart (64 lines per load) and runs 2 L exponentials per lane: L = 21 x 76 800 points 34 us, through LDS 26, this 24
// (L = 8: 7 us) -- 2 L lane-indexed reads per wavefront through the LDS crossbar are what is left.
constexpr int kSoftmaxMaxL = 32;
__global__ void __launch_bounds__(kBlock) k_softmax_rows(CrfDev c, const float *__restrict__ in, float *__restrict__ out, float scale, float relax)
{
    const int f = blockIdx.y, L = c.L;
    const int N = c.n_points[f];
    const int rpw = 64 / L;                               // rows per wavefront
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int row = lane / L, j = lane - row * L, i = wave * rpw + row;
    const bool live = row < rpw && i < N;
    const size_t q = ((size_t)f * c.maxN + (live ? i : 0)) * L + (live ? j : 0);
    const float s = live ? scale * in[q] : 0.0f;
    const int first = row * L;                            // the row's first lane
    // (the others' values eight at a time: a lane-indexed read is an LDS-crossbar round trip of ~100 cycles)
    float mx = __shfl(s, first, 64);
    for (int t0 = 1; t0 < L; t0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __shfl(s, first + min(t0 + u, L - 1), 64);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (t0 + u < L && mx < v[u]) mx = v[u];
    }
    const float e = fast_exp_nonpos(s - mx);              // (value - row maximum <= 0: the branch-free form, same bits)
    float tt = 0;
    for (int t0 = 0; t0 < L; t0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __shfl(e, first + min(t0 + u, L - 1), 64);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (t0 + u < L) tt += v[u];
    }
    if (!live) return;
    const float v = e / tt;
    if (relax == 1) out[q] = v;
    else out[q] = (1 - relax) * out[q] + relax * v;
}

inline void launch_softmax(const CrfDev &c, const float *in, float *out, float scale, float relax, hipStream_t s)
{
    if (c.L >= 3 && c.L <= kSoftmaxMaxL) {
        const int per_block = (kBlock / 64) * (64 / c.L);   // points per workgroup
        k_softmax_rows<<<dim3((unsigned)((c.maxN + per_block - 1) / per_block), (unsigned)c.F), kBlock, 0, s>>>(c, in, out, scale, relax);
    } else {
        k_softmax<<<grid_for(c.maxN, c.F), kBlock, 0, s>>>(c, in, out, scale, relax);
    }
}

// next = -unary (DenseCRF3D::stepInit, densecrf3d.h:154-158) into an arbitrary buffer
__global__ void __launch_bounds__(kBlock) k_step_init(CrfDev c, float *__restrict__ out)
{
    const int f = blockIdx.y;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= c.n_points[f] * c.L) return;
    const size_t q = (size_t)f * c.maxN * c.L + idx;
    out[q] = -c.unary[q];
}

__global__ void __launch_bounds__(kBlock) k_map(CrfDev c, const float *__restrict__ Q, int16_t *__restrict__ map)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int N = c.n_points[f];
    int lab = 0;
    if (i < N) {
        const float *p = Q + ((size_t)f * c.maxN + i) * c.L;
        lab = argmax_row(p, c.L);
        map[(size_t)f * c.maxN + i] = (int16_t)lab;
    }
    if (c.map_bits && map == c.map && c.L == 2 && (i >> 6) < c.bits_stride) {   // one bit per label: the label gather's wire format
        const unsigned long long m = __ballot(lab == 1);                           // (words beyond the frame's points: 0)
        if ((threadIdx.x & 63) == 0) c.map_bits[(size_t)f * c.bits_stride + (i >> 6)] = m;
    }
}

__global__ void __launch_bounds__(kBlock) k_validate_npoints(const int *__restrict__ in, int *__restrict__ out, int F,
                                                             int maxN, int *bad)
{
    const int f = blockIdx.x * kBlock + threadIdx.x;
    if (f >= F) return;
    const int n = in[f], c = min(max(n, 0), maxN);
    out[f] = c;
    if (c != n) *bad = 1;
}

template <int D>
void build_kernel_d(const KernelDev &kd, const CrfDev &c, hipStream_t s, const SortScratch *vsort)
{
    const int F = c.F, D1 = D + 1;
    const dim3 ge = grid_for(kd.Epad, F);
    if (vsort && kd.vorder) {
        // the sorted build (locality mode): entries sorted by the row-major code of their vertex
        const SortScratch &ss = *vsort;
        const int nbk = (1 << ss.vbits) + 1;
        k_points<D, true><<<grid_for(kd.maxNpad, F), kBlock, 0, s>>>(kd, c.n_points, ss.vpartial, ss.vbad);
        k_vsort_plan<<<F, kBlock, 0, s>>>(D, (int)grid_for(kd.maxNpad, F).x, ss);
        (void)hipMemsetAsync(ss.vhist, 0, (size_t)F * nbk * sizeof(int), s);
        (void)hipMemsetAsync(kd.rowmax, 0, (size_t)F * sizeof(int), s);                 // (the count of long buckets, until k_row_max)
        k_ecode<D><<<ge, kBlock, 0, s>>>(kd, c.n_points, ss);
        scan_frames(ss.vhist, ss.vstart, nbk, nbk, nullptr, ss.vtiles, F, s);
        k_escatter<<<ge, kBlock, 0, s>>>(kd, c.n_points, ss);
        k_eorder<<<ge, kBlock, 0, s>>>(kd, c.n_points, ss);
        k_esort_long<<<dim3(64, F), kBlock, 0, s>>>(kd, ss);
        (void)hipMemsetAsync(ss.vph, 0, (size_t)F * kVPh * sizeof(int), s);
        k_eflag<<<grid_for(kd.Epad + 1, F), kBlock, 0, s>>>(kd, c.n_points, ss);
        scan_frames(kd.flag, kd.prefix, kd.Epad + 1, kd.Epad + 1, kd.V, kd.rep, F, s); // (rep: the original ids are no longer needed, k_eoffsets rewrites it)
        k_eoffsets<<<ge, kBlock, 0, s>>>(kd, c.n_points, ss);
        (void)hipMemsetAsync(kd.nbr, 0xff, (size_t)F * D1 * kd.Epad * 2 * sizeof(int), s);      // every neighbour absent (-1)
        k_ebucket_vertices<<<grid_for(nbk, F), kBlock, 0, s>>>(kd, ss);
        XcdMap nb;
        const dim3 g = grid_xcd((long)kd.Epad, F, &nb);
        if (kd.ndist) (void)hipMemsetAsync(kd.ndist, 0, kNdistAxes * sizeof(int), s);
        if (kd.nearoff) (void)hipMemsetAsync(kd.nearoff, 0, (size_t)F * 2 * kd.Epad * 2, s);
        k_eneighbors<D><<<g, kBlock, 0, s>>>(kd, F, nb, ss);
        if (kd.srec && kd.nearoff && kd.fastn) k_pack_srec<<<g, kBlock, 0, s>>>(kd, F, nb);
        if (kd.nbrc && F >= kNbrcMinFrames && F <= kNbrcMaxFrames) {
            int limit = 0xffff;
#if LCCRF_INSTRUMENT
            if (const char *e = ab_env("LCCRF_NBRC_SPAN")) limit = std::min(std::max(atoi(e), 1), 0xffff);   // test hook: spans the table "cannot" hold
#endif
            k_nbr_compact<<<dim3(kNbrcGrid, D1, F), kBlock, 0, s>>>(kd, limit);
        }
    } else {
        (void)hipMemsetAsync(kd.slot, 0xff, (size_t)F * kd.cap * sizeof(int), s);
        k_points<D, false><<<grid_for(kd.maxNpad, F), kBlock, 0, s>>>(kd, c.n_points, nullptr, nullptr);
        {
            XcdMap nb;
            const dim3 g = grid_xcd(kd.Epad, F, &nb);
            k_insert<D><<<g, kBlock, 0, s>>>(kd, c.n_points, F, nb);
        }
        k_first_flag<<<grid_for(kd.Epad + 1, F), kBlock, 0, s>>>(kd, c.n_points);
        scan_frames(kd.flag, kd.prefix, kd.Epad + 1, kd.Epad + 1, kd.V, kd.rep, F, s); // (rep is written later, by k_offsets: free scratch)
        k_offsets<<<ge, kBlock, 0, s>>>(kd, c.n_points);
        (void)hipMemsetAsync(kd.nbr, 0xff, (size_t)F * D1 * kd.Epad * 2 * sizeof(int), s);      // every neighbour absent (-1)
        XcdMap nb;
        const dim3 g = grid_xcd((long)kd.Epad * D1, F, &nb);
        k_neighbors<D><<<g, kBlock, 0, s>>>(kd, F, nb);
    }
    if (kd.Epad < 65535) k_neighbors16<<<grid_for((long)kd.Epad * D1, F), kBlock, 0, s>>>(kd);
    if (kd.nbr2) {
        const int first = (vsort && kd.vorder && kd.tbl_bad) ? 1 : 0;     // the sorted build's splat takes pass 0 along (KernelDev::nbr2_first)
        const int npairs = (D1 - first) / 2;
        if (npairs > 0) k_neighbors_2hop<<<grid_for((long)kd.Epad * npairs, F), kBlock, 0, s>>>(kd, first, npairs);
    }
    if (!(vsort && kd.vorder)) {
        // CSR (the sorted build's runs are the rows already)
        (void)hipMemsetAsync(kd.flag, 0, (size_t)F * (kd.Epad + 1) * sizeof(int), s);
        k_csr_count<<<grid_for(kd.Epad, F), kBlock, 0, s>>>(kd, c.n_points);
        scan_frames(kd.flag, kd.rowptr, kd.Epad + 1, kd.Epad + 1, nullptr, kd.csr_pos, F, s);   // (csr_pos is written later, by k_csr_order)
        k_csr_fill<<<grid_for(kd.Epad, F), kBlock, 0, s>>>(kd, c.n_points);
        k_csr_order<<<grid_for(kd.Epad, F), kBlock, 0, s>>>(kd, c.n_points);
        k_csr_sort_long<<<dim3(64, F), kBlock, 0, s>>>(kd);                     // rows of more than kLongRow entries, if any
    }
    (void)hipMemsetAsync(kd.rowmax, 0, (size_t)F * sizeof(int), s);
    if (kd.Epad < 65535) k_row_max<<<grid_for(kd.Epad, F), kBlock, 0, s>>>(kd);   // only the one-workgroup engines ask (u16 ids)
    if (kd.longrow) {
        (void)hipMemsetAsync(kd.longcnt, 0, (size_t)F * sizeof(int), s);
        k_long_rows<<<grid_for(kd.Epad, F), kBlock, 0, s>>>(kd);
    }
}

}  // namespace

void launch_build_kernel(const KernelDev &kd, const CrfDev &c, int, hipStream_t s, const SortScratch *vsort)
{
    switch (kd.d) {
    case 1: build_kernel_d<1>(kd, c, s, vsort); break;
    case 2: build_kernel_d<2>(kd, c, s, vsort); break;
    case 3: build_kernel_d<3>(kd, c, s, vsort); break;
    case 4: build_kernel_d<4>(kd, c, s, vsort); break;
    case 5: build_kernel_d<5>(kd, c, s, vsort); break;
    case 6: build_kernel_d<6>(kd, c, s, vsort); break;
    case 7: build_kernel_d<7>(kd, c, s, vsort); break;
    case 8: build_kernel_d<8>(kd, c, s, vsort); break;
    default: break;
    }
}

static void filter_passes(const KernelDev &kd, int F, int maxV, int L, hipStream_t s, const float **result)
{
    const float *src = kd.val0;
    float *dst = kd.val1;
    for (int j = 0; j < kd.D1; ++j) {
        if (L >= 4) k_blur4<<<grid_for((long)maxV * ((L + 3) / 4), F), kBlock, 0, s>>>(kd, src, dst, j, L, (L + 3) / 4);
        else if (L == 1) k_blur1x4<<<grid_for(((long)maxV + 3) / 4, F), kBlock, 0, s>>>(kd, src, dst, j);
        else k_blur<<<grid_for((long)maxV * L, F), kBlock, 0, s>>>(kd, src, dst, j, L);
        const float *t = src;
        src = dst;
        dst = const_cast<float *>(t);
    }
    *result = src;
}

// norm = 1 / (compute(ones) + 1e-20), value width 1.  pairwise3d.h:22-27.
void launch_norm(const KernelDev &kd, const CrfDev &c, int maxV, hipStream_t s)
{
    launch_splat(kd, nullptr, 0, 1, c.F, maxV, s);
    const float *res;
    filter_passes(kd, c.F, maxV, 1, s, &res);
    const dim3 g = grid_for(c.maxN, c.F);
    switch (kd.D1) {
    case 2: k_slice_norm<2><<<g, kBlock, 0, s>>>(kd, c, res); break;
    case 3: k_slice_norm<3><<<g, kBlock, 0, s>>>(kd, c, res); break;
    case 4: k_slice_norm<4><<<g, kBlock, 0, s>>>(kd, c, res); break;
    case 5: k_slice_norm<5><<<g, kBlock, 0, s>>>(kd, c, res); break;
    case 6: k_slice_norm<6><<<g, kBlock, 0, s>>>(kd, c, res); break;
    case 7: k_slice_norm<7><<<g, kBlock, 0, s>>>(kd, c, res); break;
    case 8: k_slice_norm<8><<<g, kBlock, 0, s>>>(kd, c, res); break;
    case 9: k_slice_norm<9><<<g, kBlock, 0, s>>>(kd, c, res); break;
    default: k_slice<<<g, kBlock, 0, s>>>(kd, c, res, 1, SLICE_NORM); break;
    }
}

void launch_unary_from_label_tbl(const CrfDev &c, const int16_t *label, const UnaryTable &tbl, hipStream_t s)
{
    k_unary_from_label_tbl<<<grid_for((long)c.maxN * c.L, c.F), kBlock, 0, s>>>(c, label, tbl);
}

void launch_start(const CrfDev &c, hipStream_t s)   // densecrf_base.h:78-80
{
    launch_softmax(c, c.unary, c.Q, -1.0f, 1.0f, s);
}

void launch_step_stream(const CrfDev &c, const KernelDev *kds, const int *maxV, float relax, hipStream_t s)
{                                                    // densecrf_base.h:82-91
    const int L = c.L;
    if (c.K == 0) {
        // stepInit only: next = -unary, then softmax.  Done by the softmax with scale -1.
        launch_softmax(c, c.unary, c.Q, -1.0f, relax, s);
        return;
    }
    if (L == 2) {
        for (int k = 0; k < c.K; ++k) {
            const KernelDev &kd = kds[k];
            XcdMap nb;
            const int blk = iter_block(c.F);
            const bool pairs = pair_fuse(c.F, maxV[k]);
            // sorted build, one pass per launch: the first pass (axis 0 = the code's fastest coordinate) rides in the splat
            static const bool no_sb = ab_env("LCCRF_NO_SPLAT_BLUR") != nullptr;               // A/B switch: same results either way
            const int j0 = (kd.vorder && kd.fast0_ok && !no_sb && !kd.long_mode) ? std::max(kd.splat_passes, 1) : 0;   // passes the splat takes along
            if (j0 >= 2) {
                const int B = kd.splat_block, core = B - 2 * kd.splat_halo;
                const float2 *q2 = reinterpret_cast<const float2 *>(c.Q);
                // many frames in flight: 256 lanes x 1 / 2 / 4 vertices (C5 x 8, window 1024: 20.6 -> 18.9 us per frame-iteration against
                // 1024 lanes x 1: workgroups of four wavefronts wait less at the barriers); one or two frames: one or two vertices per lane
                // (a lane's four row walks in a row cost a single frame 33.3 -> 36.1)
                static const char *env_w = ab_env("LCCRF_SPLAT_WIDE_MAX");                  // (A/B: frames-in-flight threshold)
                const bool wide = c.F <= (env_w ? atoi(env_w) : 2);
                const int lanes = wide ? B : kBlock;
                const dim3 g = grid_xcd(((long)maxV[k] + core - 1) / core * lanes, c.F, &nb, lanes);
                // (REC: the per-vertex records of the sorted build, one load instead of seven -- KernelDev::srec; allocated by the
                // instrumented library under LCCRF_SPLAT_REC=1 only: +-1 % on two boxes)
                const bool rec = kd.srec && kd.srec_ok;
                static const bool rec_plain = ab_env("LCCRF_SPLAT_REC_PLAIN") != nullptr;      // A/B switch: records through plain (cached) loads
#define LCCRF_SPLAT2W(LN, UU, GRID) do { if (rec && rec_plain) k_splat2w<LN, UU, true, false><<<GRID, LN, 0, s>>>(kd, q2, c.maxN, c.F, nb, j0, kd.splat_halo); \
                                         else if (rec) k_splat2w<LN, UU, true><<<GRID, LN, 0, s>>>(kd, q2, c.maxN, c.F, nb, j0, kd.splat_halo); \
                                         else k_splat2w<LN, UU, false><<<GRID, LN, 0, s>>>(kd, q2, c.maxN, c.F, nb, j0, kd.splat_halo); } while (0)
                if (B == 256) LCCRF_SPLAT2W(256, 1, g);
                else if (B == 512 && wide) LCCRF_SPLAT2W(512, 1, g);
                else if (B == 512) LCCRF_SPLAT2W(256, 2, g);
                else if (c.F == 1) {                      // (one frame: 663 workgroups of 1024 lanes are 1.3 rounds of the chip's 512 slots;
                    const dim3 g5 = grid_xcd(((long)maxV[k] + core - 1) / core * 512, c.F, &nb, 512);    //  512 lanes x 2 vertices all run at once: 33.6 -> 32.5 us)
                    LCCRF_SPLAT2W(512, 2, g5);
                }
                else if (wide) LCCRF_SPLAT2W(1024, 1, g);
                else {
                    static const bool w512 = ab_env("LCCRF_SPLAT_512X2") != nullptr;            // A/B switch: 512 lanes x 2 vertices with many frames in flight
                    if (w512) {
                        const dim3 g5 = grid_xcd(((long)maxV[k] + core - 1) / core * 512, c.F, &nb, 512);
                        LCCRF_SPLAT2W(512, 2, g5);
                    } else LCCRF_SPLAT2W(256, 4, g);
                }
#undef LCCRF_SPLAT2W
            } else if (j0 == 1) {
                const dim3 g = grid_xcd(((long)maxV[k] + blk - 3) / (blk - 2) * blk, c.F, &nb, blk);
                k_splat2<true><<<g, blk, 0, s>>>(kd, reinterpret_cast<const float2 *>(c.Q), c.maxN, c.F, nb);
            } else if (kd.long_mode) {                    // a coarse kernel: long rows
                if (kd.long_mode == 2) {                  // (a wavefront per vertex: grid_xcd counts workgroups of kBlock / 64 vertices)
                    const dim3 g = grid_xcd((long)maxV[k] * 64, c.F, &nb, kBlock);
                    k_splat2v<<<g, kBlock, 0, s>>>(kd, reinterpret_cast<const float2 *>(c.Q), c.maxN, c.F, nb);
                } else {
                    const dim3 g = grid_xcd(maxV[k], c.F, &nb, kBlock);
                    k_splat2l<<<g, kBlock, 0, s>>>(kd, reinterpret_cast<const float2 *>(c.Q), c.maxN, c.F, nb);
                }
                if (kd.longrow_ok) k_splat_long<<<dim3((unsigned)std::max(256 / std::max(c.F, 1), 8), (unsigned)c.F), kBlock, 0, s>>>(kd, c.Q, c.maxN * 2, 2);
            } else {
                const dim3 g = grid_xcd(maxV[k], c.F, &nb, blk);
                k_splat2<false><<<g, blk, 0, s>>>(kd, reinterpret_cast<const float2 *>(c.Q), c.maxN, c.F, nb);
            }
            const float *src = kd.val0;
            float *dst = kd.val1;
            // the pass left over by the pairs rides in the slice; with a few frames in flight (one pass per launch) the last pass does
            static const char *sf = ab_env("LCCRF_SLICE_BLUR_MAX");                          // (A/B: frames-in-flight threshold)
            const bool blur_in_slice = kd.D1 <= 9 && j0 < kd.D1 && (pairs ? ((kd.D1 - j0) & 1) && kd.D1 >= 3 : c.F <= (sf ? atoi(sf) : kSliceBlurMaxFrames));   // (j0 == d + 1: a 2-D lattice's three passes can all ride in the splat)
            const int n_own = blur_in_slice ? kd.D1 - 1 : kd.D1;                           // blur passes with a launch of their own
            for (int j = j0; j < n_own;) {
                if (pairs && j + 1 < n_own) {             // one frame in flight: two passes per launch
                    const dim3 gp = grid_xcd(maxV[k], c.F, &nb, blk);
                    static const bool no_tbl = ab_env("LCCRF_NO_2HOP_TABLE") != nullptr;      // A/B switch: same results either way
                    const int jt = j - kd.nbr2_first;     // the table holds the pairs (first, first + 1), (first + 2, first + 3) ...
                    if (kd.nbr2 && kd.nbr2_ok && !no_tbl && jt >= 0 && !(jt & 1))
                        k_blur2x2t<<<gp, blk, 0, s>>>(kd, src, dst, jt / 2, (kd.D1 - kd.nbr2_first) / 2, c.F, nb);
                    else k_blur2x2<<<gp, blk, 0, s>>>(kd, src, dst, j, c.F, nb);
                    j += 2;
                } else {
                    launch_blur2(kd, src, dst, j, c.F, maxV[k], s);
                    j += 1;
                }
                const float *t = src;
                src = dst;
                dst = const_cast<float *>(t);
            }
            const int first = k == 0, last = k == c.K - 1;
            const dim3 g = grid_xcd(c.maxN, c.F, &nb, blk);
#define LCCRF_SLICE_CASE(D)                                                                       \
    case D:                                                                                      \
        if (blur_in_slice) k_slice2<D, true><<<g, blk, 0, s>>>(kd, c, src, first, last, relax, nb); \
        else k_slice2<D><<<g, blk, 0, s>>>(kd, c, src, first, last, relax, nb);                  \
        break;
            switch (kd.D1) {
            LCCRF_SLICE_CASE(2)
            LCCRF_SLICE_CASE(3)
            LCCRF_SLICE_CASE(4)
            LCCRF_SLICE_CASE(5)
            LCCRF_SLICE_CASE(6)
            LCCRF_SLICE_CASE(7)
            LCCRF_SLICE_CASE(8)
            LCCRF_SLICE_CASE(9)
            default: break;
            }
#undef LCCRF_SLICE_CASE
        }
        return;
    }
    for (int k = 0; k < c.K; ++k) {
        const KernelDev &kd = kds[k];
        launch_splat(kd, c.Q, c.maxN * L, L, c.F, maxV[k], s);
        const float *res;
        filter_passes(kd, c.F, maxV[k], L, s, &res);
        k_slice<<<grid_for((long)c.maxN * L, c.F), kBlock, 0, s>>>(kd, c, res, L,
                                                                 k == 0 ? SLICE_APPLY_FIRST : SLICE_APPLY);
    }
    launch_softmax(c, c.next, c.Q, 1.0f, relax, s);
}

// Measurement support (bench.py's roofline object): `reps` launches of the streaming engine's dominant kernel --
// one blur pass of kernel kd over all F frames -- bracketed by HIP events on stream s.  The lattice values it
// scribbles over are recomputed from Q by every mean-field step.
hipError_t time_blur_pass(const KernelDev &kd, int F, int maxV, int L, int reps, hipStream_t s, float *ms_per_launch)
{
    hipEvent_t e0, e1;
    hipError_t rc = hipEventCreate(&e0);
    if (rc != hipSuccess) return rc;
    if ((rc = hipEventCreate(&e1)) != hipSuccess) { (void)hipEventDestroy(e0); return rc; }
    auto pass = [&](int i) {
        const float *src = (i & 1) ? kd.val1 : kd.val0;
        float *dst = (i & 1) ? kd.val0 : kd.val1;
        if (L == 2) launch_blur2(kd, src, dst, i % kd.D1, F, maxV, s);
        else k_blur<<<grid_for((long)maxV * L, F), kBlock, 0, s>>>(kd, src, dst, i % kd.D1, L);
    };
    for (int i = 0; i < 3; ++i) pass(i);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) pass(i);
    (void)hipEventRecord(e1, s);
    rc = hipEventSynchronize(e1);
    if (rc == hipSuccess) rc = hipEventElapsedTime(ms_per_launch, e0, e1);
    *ms_per_launch /= (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// dst[i][:] = src[list[i]][:] (gather) or dst[list[i]][:] = src[i][:] (scatter), rows of `units` elements of type T
template <typename T>
__global__ void __launch_bounds__(kBlock) k_copy_frames(T *__restrict__ dst, size_t dst_stride, const T *__restrict__ src,
                                                        size_t src_stride, const int *__restrict__ list, size_t units, int gather)
{
    const int i = blockIdx.y, f = list[i];
    const T *sp = src + (gather ? (size_t)f : (size_t)i) * src_stride;
    T *dp = dst + (gather ? (size_t)i : (size_t)f) * dst_stride;
    for (size_t w = (size_t)blockIdx.x * kBlock + threadIdx.x; w < units; w += (size_t)gridDim.x * kBlock) dp[w] = sp[w];
}

void launch_copy_frames(void *dst, size_t dst_stride, const void *src, size_t src_stride, const int *list, int n_list,
                        size_t bytes, int gather, hipStream_t s)
{
    if (n_list <= 0 || bytes == 0) return;
    const bool w4 = ((bytes | dst_stride | src_stride | (size_t)(uintptr_t)dst | (size_t)(uintptr_t)src) & 3) == 0;
    const size_t units = bytes / (w4 ? 4 : 2);                                 // (every per-frame array is at least int16-aligned)
    const unsigned gx = (unsigned)std::min<size_t>((units + kBlock - 1) / kBlock, 64);
    const dim3 g(gx, (unsigned)n_list);
    if (w4)
        k_copy_frames<unsigned><<<g, kBlock, 0, s>>>(static_cast<unsigned *>(dst), dst_stride / 4, static_cast<const unsigned *>(src),
                                                     src_stride / 4, list, units, gather);
    else
        k_copy_frames<unsigned short><<<g, kBlock, 0, s>>>(static_cast<unsigned short *>(dst), dst_stride / 2,
                                                           static_cast<const unsigned short *>(src), src_stride / 2, list, units, gather);
}

__global__ void __launch_bounds__(kBlock) k_permute_rows(CrfDev c, float *__restrict__ dst, const float *__restrict__ src, int width,
                                                         int gather)
{
    const int f = blockIdx.y;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= c.n_points[f] * width) return;
    const int i = idx / width, m = idx - i * width;
    const int o = c.perm[(size_t)f * c.perm_stride + i];
    const size_t a = ((size_t)f * c.maxN + i) * width + m, b = ((size_t)f * c.maxN + o) * width + m;
    if (gather) dst[a] = src[b];
    else dst[b] = src[a];
}

void launch_permute_rows(const CrfDev &c, float *dst, const float *src, int width, int gather, hipStream_t s)
{
    k_permute_rows<<<grid_for((long)c.maxN * width, c.F), kBlock, 0, s>>>(c, dst, src, width, gather);
}

void launch_sort_points(const KernelDev &kd, const CrfDev &c, const SortScratch &ss, hipStream_t s)
{
    switch (kd.d) {
    case 1: sort_points_d<1>(kd, c, ss, s); break;
    case 2: sort_points_d<2>(kd, c, ss, s); break;
    case 3: sort_points_d<3>(kd, c, ss, s); break;
    case 4: sort_points_d<4>(kd, c, ss, s); break;
    case 5: sort_points_d<5>(kd, c, ss, s); break;
    case 6: sort_points_d<6>(kd, c, ss, s); break;
    case 7: sort_points_d<7>(kd, c, ss, s); break;
    case 8: sort_points_d<8>(kd, c, ss, s); break;
    default: break;
    }
}

void launch_validate_npoints(const int *in, int *out, int F, int maxN, int *bad, hipStream_t s)
{
    k_validate_npoints<<<(F + kBlock - 1) / kBlock, kBlock, 0, s>>>(in, out, F, maxN, bad);
}

void launch_map(const CrfDev &c, hipStream_t s)
{
    k_map<<<grid_for(c.maxN, c.F), kBlock, 0, s>>>(c, c.Q, c.map);
}

// ---- the protected virtuals of DenseCRF / the pure virtual of PairwisePotential on caller-chosen device buffers ------
void launch_map_of(const CrfDev &c, const float *prob, int16_t *map, hipStream_t s)
{
    k_map<<<grid_for(c.maxN, c.F), kBlock, 0, s>>>(c, prob, map);
}

void launch_exp_and_normalize(const CrfDev &c, const float *in, float *out, float scale, float relax, hipStream_t s)
{
    launch_softmax(c, in, out, scale, relax, s);
}

void launch_step_init(const CrfDev &c, float *out, hipStream_t s)
{
    k_step_init<<<grid_for((long)c.maxN * c.L, c.F), kBlock, 0, s>>>(c, out);
}

// out (+)= [w * norm *] compute(in) with value width c.L: PairwisePotential::apply (accumulate = 1, pairwise3d.h:73-78)
// or the bare PermutohedralLatticeCPU::compute (accumulate = 0, permutohedral_cpu.h:634-699)
void launch_filter(const KernelDev &kd, const CrfDev &c, int maxV, const float *in, float *out, int accumulate, hipStream_t s)
{
    const int L = c.L;
    launch_splat(kd, in, c.maxN * L, L, c.F, maxV, s);
    const float *res;
    filter_passes(kd, c.F, maxV, L, s, &res);
    CrfDev c2 = c;
    c2.next = out;
    k_slice<<<grid_for((long)c.maxN * L, c.F), kBlock, 0, s>>>(kd, c2, res, L, accumulate ? SLICE_APPLY : SLICE_PLAIN);
}

}  // namespace lccrf
