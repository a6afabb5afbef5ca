// frame_lean.hip -- one frame's WHOLE CRF as one kernel launch (frame_engine.hip) for FULL-SIZE SLAM frames on HALF a CU:
// 512 lanes, 3-4 points per lane, 80 KB of LDS per frame, so that two frames share a CU in every phase -- one frame's hash
// inserts, scans and ordered row sums (latency, barriers) run under the other frame's point phases.  Per frame
//     both PottsPotential3D constructors (lattice + norm)  +  DenseCRF3D::inference(n, true)      (src/Tracking.cc:1920-1929)
// as in k_frame; what differs is where things live:
//   build     k_frame's algorithms (keys in an LDS hash table, ids by slot order, neighbours by probing, row places by bitmap or
//             entry lists; permutohedral_cpu.h:66-167,241-424), with the LDS scratch laid out by LIFETIME instead of carved in
//             line: the hash table (48 KB at 2000 points) dies with phase D, and the row bitmap / the entry lists are only born
//             there -- 78 KB at the peak instead of 96.  Each kernel's persistent tables (row starts; the appearance kernel's
//             neighbour table) are built straight into their places in the loop's plan (fused_lean.h: lean_tables);
//   records   what fused_lean.h's loop re-reads every iteration -- barycentric weights, norms, unary energies, the smoothness
//             lattice's neighbour table -- goes to the batch's own arrays in HBM (KernelDev::bary / norm / nbr16, CrfDev::unary)
//             as it is produced and comes back through L2; a lane keeps Q and its packed vertex / slot words;
//   norm      one pass of the loop's own phases with Q = 1 (mean_field_lean<.., NORM>), pairwise3d.h:20-28;
//   loop      mean_field_lean, unchanged.
// Two 2-D kernels, L = 2, the appearance kernel with long rows (chain lanes), 1025 .. 2048 points, batches of >= 256 frames -- C2 / C3.
// Anything else, and any frame whose lattices do not fit (it flags itself), runs on the other paths with identical results.
// The vertex numbering is this kernel's own (ids by hash slot order), as in k_frame: the batch's lattice arrays are NOT a build a
// later lccrf_batch_inference could iterate on (the engine marks them unbuilt).
#include "frame_build.h"
#include "fused_lean.h"

#include <algorithm>
#include <type_traits>

namespace lccrf {
namespace fb {

namespace {

struct LeanHdr {                          // the build's scan scratch and flags: the last bytes of the launch's LDS (dead before the loop)
    int wave_sum[16];
    int fail;
    int rowmax;
};
constexpr int kLeanHdrBytes = 128;

template <int PPT>
__global__ void __launch_bounds__(kNTSmall, 4) k_frame_lean(CrfDev c, FrameArgs a)
{
    constexpr int NT = kNTSmall, K = 2, D1 = kD1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int N = c.n_points[f];
    Instr ins{a.timing, a.timing_block, 8, 0, a.timing_lane};
    FL_STAMP();
    // (the lane id far from here -- the exits, which the compiler lays out as guarded blocks at the END of the kernel, and everything
    // behind the loops -- as scalar wave base + mbcnt: a vector register holding `tid` for those would be live across the loops, and
    // the allocator parks such values in the rings' v96..v127 and spills them around the asm; see k_fused_lean)
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    auto lane_id = [&]() {
        int z = 0;
        asm volatile("" : "+v"(z));
        return wave_base + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
    };
    auto flag_unfit = [&]() {             // left to the fallback path (frame_engine.hip: flag_unfit)
        if (lane_id() == 0) {
            if (a.status) *a.status = 1;
            if (a.frame_status) a.frame_status[f] = 1;
        }
    };
    if (N <= 0) {                                         // an empty frame has no lattice (V = 0) and nothing to infer
        const int t = lane_id();
        if (a.with_map) clear_label_bits<NT>(c, f, 0, t);
        if (t < K && a.V_out[t]) a.V_out[t][f] = 0;
        if (t == 0 && a.frame_status) a.frame_status[f] = 0;
        return;
    }
    const int Npad = (N + 3) & ~3;                       // blocks of four, permutohedral_cpu.h:294 (quirk Q1)
    const int hdr_off = a.lds_total - kLeanHdrBytes;
    LeanHdr *hdr = reinterpret_cast<LeanHdr *>(smem + hdr_off);
    const int hcap = a.hcap;
    const unsigned mask = (unsigned)hcap - 1u;
    const int hk_off = hdr_off - hcap * 4, ido_off = hk_off - hcap * 2;
    unsigned *hk = reinterpret_cast<unsigned *>(smem + hk_off);                  // hash table: the keys themselves
    unsigned short *ido = reinterpret_cast<unsigned short *>(smem + ido_off);    // slot -> vertex id

    // unary energies from the labels (densecrf3d.h:116-129 with L = 2) go straight to the batch's array: the loop re-reads them
    if (a.label) {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * NT;
            if (i < N) {
                const int t = a.label[(size_t)f * a.maxN + i];
                const bool known = t >= 0 && t < 2;
                float2 un;
                un.x = !known ? a.tbl[0] : (t == 0 ? a.tbl[3] : a.tbl[1 + t]);
                un.y = !known ? a.tbl[0] : (t == 1 ? a.tbl[4] : a.tbl[1 + t]);
                reinterpret_cast<float2 *>(c.unary)[(size_t)f * c.maxN + i] = un;
            }
        }
    }
    if (tid < 32) reinterpret_cast<float *>(smem)[tid] = 0.0f;       // LDS bytes [0, 128) of the loop's plan are zeros
    if (tid == 0) { hdr->fail = 0; hdr->rowmax = 0; }

    FusedLayout lay{};
    size_t plan = 128;                    // the loop's plan, kernel after kernel (fused_lean.h: layout_lean)
    int V[K], row0max = 0;
    unsigned pk[PPT][K][D1];              // (vertex id + 1) | place in the row << 16, as the HBM records of k_fused

    // ---- one kernel's lattice: false = the frame does not fit (uniform) ---------------------------------------------------
    auto build = [&](auto kc) -> bool {
        constexpr int k = decltype(kc)::value;
        // A: point records (elevate, round, rank, barycentric) and the keys of their three corners
        unsigned key[PPT][D1];
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * NT;
            const float2 ft = reinterpret_cast<const float2 *>(a.feat[k])[(size_t)f * a.maxN + min(i, N - 1)];
            float feat[2] = {i < N ? ft.x : 0.0f, i < N ? ft.y : 0.0f};   // phantom lanes, :299
            int16_t r0[2];
            uint8_t rk[2];
            float b[D1];
            point_record<2>(feat, a.scale, a.inv_dp1, r0, rk, b);
            if (i < N) {
                float *bo = a.bary_out[k] + (size_t)f * a.Epad + (size_t)i * D1;
#pragma unroll
                for (int j = 0; j < D1; ++j) bo[j] = b[j];
            }
#pragma unroll
            for (int j = 0; j < D1; ++j) key[s][j] = corner_key(r0, rk, j);
        }
        for (int u = tid; u < hcap; u += NT) hk[u] = kEmptyKey;
        __syncthreads();
        FL_PSTAMP();

        // B: insert (ds_cmpst claims an empty slot for the key or returns the key that lives there; frame_engine.hip phase B)
        unsigned slot[PPT][D1], got[PPT][D1];
        bool bad = false;
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                slot[s][j] = hash32(key[s][j]) & mask;
                got[s][j] = key[s][j];
                bad |= key[s][j] == kEmptyKey;
                if (tid + s * NT < Npad) got[s][j] = atomicCAS(&hk[slot[s][j]], kEmptyKey, key[s][j]);
            }
        int ncreated = 0;
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                const unsigned kk = key[s][j];
                if (got[s][j] != kEmptyKey && got[s][j] != kk) {           // somebody else's key lives there: linear probing
                    unsigned h = slot[s][j];
                    for (int probes = 0;; ++probes) {
                        h = (h + 1u) & mask;
                        got[s][j] = atomicCAS(&hk[h], kEmptyKey, kk);
                        if (got[s][j] == kEmptyKey || got[s][j] == kk) break;
                        if (probes >= hcap) { bad = true; break; }
                    }
                    slot[s][j] = h;
                }
                ncreated += got[s][j] == kEmptyKey;
            }
        if (bad) hdr->fail = 1;
        FL_PSTAMP();

        // C: dense vertex ids for the entries that created their vertex; this kernel's tables in the loop's plan, the scratch by lifetime:
        //      up to the end of D   hash table (end of LDS), vertex keys, counters, [kernel 1: neighbour table -- exported at the end of D]
        //      from the end of D    counters, row starts, row bitmap + prefixes or entry lists (over the dead hash table)
        int Vk;
        int id = block_excl_scan<NT>(ncreated, tid, hdr->wave_sum, Vk);       // (its barriers also close phase B)
        V[k] = Vk;
        FL_PSTAMP();
        if (Vk > lean_max_v(NT) || (k == 0 && (Vk > chain_max_v(NT) || Vk > 256))) return false;
        const int plan_before = (int)plan;
        const LeanTables lt = lean_tables(plan, Vk, k == 0);
        lay.val[k][0] = lt.val0;
        lay.val[k][1] = lt.val1;
        lay.nbr[k] = lt.nbr;
        lay.row[k] = lt.row;
        auto take = [&](int &o, int bytes) { const int r = o; o += (bytes + 15) & ~15; return r; };
        // (kernel 1's scratch starts at the end of kernel 0's tables: its value arrays are not in use before the loop, and its row starts --
        //  inside or right behind the scratch table -- are written in E, after the export)
        int vs = k == 0 ? (int)plan : plan_before;
        const int vkey_off = take(vs, Vk * 4);
        const int cnt_off = take(vs, (Vk + 1) * 4);      // arrival counters (short rows), then start | length of every row list
        const int nbs_off = k == 0 ? lt.nbr : take(vs, D1 * Vk * 4);
        const int vs_d = vs;                              // end of what lives beside the hash table
        if (k != 0) vs = max(vs, (int)plan);
        const int E = N * D1;
        const int W = (((Npad + 31) >> 5) + 3) & ~3;      // bitmap words per vertex, a multiple of 4
        const bool bitmap = (long)E >= 16L * Vk && vs + 5 * Vk * W + 64 <= hdr_off;   // long rows: rank by bitmap, if it fits
        const int bm_off = vs, pre_off = bm_off + Vk * W * 4;             // pre: entries before every 4-word group (u16)
        const int list_cap = (E + 7 * Vk + 8) & ~7;       // short rows: u16 entry lists, rows padded to 8
        const int list_off = vs;
        const int vs_end = bitmap ? pre_off + Vk * W / 2 : list_off + list_cap * 2;
        if (vs_d > ido_off || vs_end > hdr_off || Vk >= 32767 || E + 7 * Vk >= 65535 || hdr->fail) return false;
        unsigned *vkey = reinterpret_cast<unsigned *>(smem + vkey_off);
        unsigned *cnt = reinterpret_cast<unsigned *>(smem + cnt_off);
        unsigned *bm = reinterpret_cast<unsigned *>(smem + bm_off);
        unsigned short *pre = reinterpret_cast<unsigned short *>(smem + pre_off);
        unsigned short *list = reinterpret_cast<unsigned short *>(smem + list_off);
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int j = 0; j < D1; ++j)
                if (got[s][j] == kEmptyKey) {
                    ido[slot[s][j]] = (unsigned short)id;
                    vkey[id] = key[s][j];
                    ++id;
                }
        for (int v = tid; v <= Vk; v += NT) cnt[v] = 0u;
        {
            unsigned *nbz = reinterpret_cast<unsigned *>(smem + nbs_off);   // absent neighbours stay 0
            for (int u = tid; u < D1 * Vk; u += NT) nbz[u] = 0u;
        }
        __syncthreads();
        FL_PSTAMP();

        // D: every entry learns its vertex (short rows: and joins the vertex's row); blur neighbours per vertex
        unsigned vid[PPT][D1], arr[PPT][D1];              // vertex id, arrival index inside the row (short rows)
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int j = 0; j < D1; ++j) vid[s][j] = (tid + s * NT < Npad) ? (unsigned)ido[slot[s][j]] : 0u;
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                arr[s][j] = 0u;
                if (!bitmap && tid + s * NT < N) arr[s][j] = atomicAdd(&cnt[vid[s][j]], 1u);   // real points only: phantoms add vertices, not products
            }
        }
        {
            // permutohedral_cpu.h:408-421 with d = 2, one probe per (axis, vertex) fills both halves (frame_engine.hip phase D)
            unsigned short *nb16 = reinterpret_cast<unsigned short *>(smem + nbs_off);   // [axis][vertex][n1+1, n2+1]
            for (int t = tid; t < D1 * Vk; t += NT) {
                const int j = t >= 2 * Vk ? 2 : (t >= Vk ? 1 : 0), v = t - j * Vk;
                const unsigned kk = vkey[v];
                const unsigned qx = ((kk & 0xffffu) + (j == 0 ? 0xfffeu : 1u)) & 0xffffu, qy = ((kk >> 16) + (j == 1 ? 0xfffeu : 1u)) & 0xffffu;
                const unsigned q = qx | (qy << 16);
                unsigned h = hash32(q) & mask;
                unsigned o = hk[h];
                for (int probes = 0; o != kEmptyKey && o != q && probes < hcap; ++probes) {
                    h = (h + 1u) & mask;
                    o = hk[h];
                }
                if (o == q) {
                    const unsigned b = ido[h];
                    nb16[2 * t + 1] = (unsigned short)(b + 1u);                       // my n2
                    nb16[2 * (j * Vk + (int)b)] = (unsigned short)(v + 1);          // its n1
                }
            }
        }
        __syncthreads();                                  // the hash table, the slot ids and the vertex keys are dead
        FL_PSTAMP();
        if (k != 0) {                                     // the smoothness lattice's table: to HBM (the loop reads it pass by pass)
            const unsigned *nbs = reinterpret_cast<const unsigned *>(smem + nbs_off);
            unsigned *out = a.nbr_out[k] + (size_t)f * D1 * a.Epad;
            for (int u = tid; u < D1 * Vk; u += NT) {
                const int j = u >= 2 * Vk ? 2 : (u >= Vk ? 1 : 0);
                out[(size_t)j * a.Epad + (u - j * Vk)] = nbs[u];
            }
        }
        if (bitmap) {
            uint4 *b4 = reinterpret_cast<uint4 *>(bm);
            for (int u = tid; u < Vk * W / 4; u += NT) b4[u] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const int i = tid + s * NT;
                if (i < N) {
#pragma unroll
                    for (int j = 0; j < D1; ++j) atomicOr(&bm[vid[s][j] * W + (i >> 5)], 1u << (i & 31));
                }
            }
            __syncthreads();
        } else {
            uint4 *l4 = reinterpret_cast<uint4 *>(list);   // pads compare greater than every entry (entries are < 0x7fff); E's scans separate this from F's writes
            for (int u = tid; u < list_cap / 8; u += NT) l4[u] = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
        }
        FL_PSTAMP();

        // E: row lengths -> row starts (and, short rows, the start of every padded entry list)
        if (bitmap) {
            // prefix popcounts of every vertex's bitmap, one 16-lane group per vertex (frame_engine.hip phase E)
            const int lane16 = tid & 15, ng = W >> 2, gpl = (ng + 15) >> 4;
            for (int v = tid >> 4; v < Vk; v += NT / 16) {
                int pc[2] = {0, 0}, sum = 0;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int g = lane16 * gpl + u;
                    if (u < gpl && g < ng) {
                        const uint4 b = *reinterpret_cast<const uint4 *>(bm + v * W + 4 * g);
                        pc[u] = __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
                    }
                    sum += pc[u];
                }
                int incl = sum;                            // scan inside the 16-lane row only
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);
                int run = incl - sum;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int g = lane16 * gpl + u;
                    if (u < gpl && g < ng) pre[v * ng + g] = (unsigned short)run;
                    run += pc[u];
                }
                if (lane16 == 15) cnt[v] = (unsigned)incl;
            }
            __syncthreads();
        }
        unsigned short *row = reinterpret_cast<unsigned short *>(smem + lt.row);
        {
            // packed scan: low half = products before the row, high half = padded list entries before it
            const int vper = (Vk + 1 + NT - 1) / NT, v0 = tid * vper;
            unsigned sum = 0u;
            int mx = 0;
            for (int u = 0; u < vper; ++u) {
                const int v = v0 + u;
                if (v < Vk) {
                    const unsigned n = cnt[v];
                    sum += n | (((n + 7u) & ~7u) << 16);
                    mx = max(mx, (int)n);
                }
            }
            int tot;
            unsigned run = (unsigned)block_excl_scan<NT>((int)sum, tid, hdr->wave_sum, tot);
            for (int u = 0; u < vper; ++u) {
                const int v = v0 + u;
                if (v <= Vk) {
                    const unsigned n = v < Vk ? cnt[v] : 0u;
                    row[v] = (unsigned short)(run & 0xffffu);
                    cnt[v] = (run >> 16) | (n << 16);      // list start | row length
                    run += n | (((n + 7u) & ~7u) << 16);
                }
            }
            if (k == 0) {                                 // longest row of kernel 0 decides the chain path
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
                if ((tid & 63) == 0 && mx > 0) atomicMax(&hdr->rowmax, mx);
            }
        }
        __syncthreads();
        FL_PSTAMP();

        // F/G: the place of every entry in its row = number of smaller entries of the same vertex; one point slot at a time
        if (!bitmap) {
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const int i = tid + s * NT;
                if (i < N) {
#pragma unroll
                    for (int j = 0; j < D1; ++j) list[(cnt[vid[s][j]] & 0xffffu) + arr[s][j]] = (unsigned short)(i * D1 + j);
                }
            }
            __syncthreads();
        }
        FL_PSTAMP();
        // eight 16-bit entries against e at once: entries and e are < 0x8000, so (x | 0x8000) - e keeps bit 15 of a half exactly
        // when that half is >= e, and no half ever borrows from its neighbour
        auto below = [](const uint4 &x, unsigned e) {
            const unsigned e2 = e | (e << 16), hi = 0x80008000u;
            return 8u - (unsigned)(__popc(((x.x | hi) - e2) & hi) + __popc(((x.y | hi) - e2) & hi) + __popc(((x.z | hi) - e2) & hi) +
                                   __popc(((x.w | hi) - e2) & hi));
        };
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int i = tid + s * NT, ic = min(i, N - 1);
            const bool real = i < N;
#pragma unroll
            for (int j = 0; j < D1; ++j) {
                const unsigned v = vid[s][j];
                unsigned r;
                if (bitmap) {
                    const uint4 bq = *reinterpret_cast<const uint4 *>(bm + v * W + ((ic >> 5) & ~3));
                    const int wq = (ic >> 5) & 3;
                    const unsigned low = (1u << (ic & 31)) - 1u;
                    r = pre[v * (W >> 2) + (ic >> 7)] + (wq > 0 ? __popc(bq.x) : 0) + (wq > 1 ? __popc(bq.y) : 0) + (wq > 2 ? __popc(bq.z) : 0) +
                        __popc((wq == 0 ? bq.x : wq == 1 ? bq.y : wq == 2 ? bq.z : bq.w) & low);
                } else {
                    const unsigned lcv = cnt[v], e = (unsigned)(i * D1 + j);
                    const uint4 *lp = reinterpret_cast<const uint4 *>(list + (lcv & 0xffffu));
                    r = below(lp[0], e);
                    const int n8 = (int)(((lcv >> 16) + 7u) >> 3);
                    for (int u = 1; u < n8; ++u) r += below(lp[u], e);
                }
                pk[s][k][j] = (v + 1u) | (((unsigned)row[v] + (real ? r : 0u)) << 16);
            }
        }
        if (k == 0) row0max = hdr->rowmax;
        FL_PSTAMP();
        __syncthreads();                                  // the next kernel's build (or the loop's product buffer) reuses the scratch
        return true;
    };
    if (!build(std::integral_constant<int, 0>{}) || !build(std::integral_constant<int, 1>{})) {
        flag_unfit();
        return;
    }
    FL_STAMP();

    // ---- the loop's plan for THIS frame's sizes; the tables built above are already in their places ----------------------
    {
        // (kernel 0's rows go to the chain lanes however short they are -- a frame of a few points among full-size ones is not worth a
        //  second instance of the loop; its lattice is small enough for them, or the build has already left)
        FusedLayout L;
        if (!layout_lean(N, K, V, max(row0max, kChainMinRow), &L, NT, (size_t)a.lds_total) || !L.chain0 || L.row[0] != lay.row[0] ||
            L.row[1] != lay.row[1] || L.nbr[0] != lay.nbr[0]) {
            flag_unfit();
            return;
        }
        lay = L;
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            reinterpret_cast<float2 *>(smem + lay.val[k][0])[0] = make_float2(0.f, 0.f);
            reinterpret_cast<float2 *>(smem + lay.val[k][1])[0] = make_float2(0.f, 0.f);
        }
    }
    float wk[K], alpha[K];
    LeanSrc src;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        wk[k] = a.w[k];
        alpha[k] = a.alpha;
        src.nbr[k] = lean_rsrc(a.nbr_out[k] + (size_t)f * D1 * a.Epad, (size_t)D1 * a.Epad * 4);
        src.bary[k] = lean_rsrc(a.bary_out[k] + (size_t)f * a.Epad, (size_t)a.Epad * 4);
        src.norm[k] = lean_rsrc(a.norm_out[k] + (size_t)f * a.maxN, (size_t)a.maxN * 4);
        src.nbr_axis_bytes[k] = a.Epad * 4;
    }
    src.unary = lean_rsrc(c.unary + (size_t)f * c.maxN * 2, (size_t)c.maxN * 8);
    __syncthreads();                      // (also orders this workgroup's record stores before its loads: one CU, one path to L2)

    ChainLane cl = chain_setup_lean<NT>(smem, lay, V[0], tid);
    PointRegs<PPT, K> pr;
    place_products_lean<PPT, K, 1, NT>(smem, lay, N, tid, pk, pr);
    FL_STAMP();

    // norm = 1 / (compute(ones) + 1e-20), pairwise3d.h:22-27: one pass of the loop's phases with Q = 1
#pragma unroll
    for (int s = 0; s < PPT; ++s) pr.q[s] = make_float2(1.0f, 1.0f);
    int t = lane_id();
    mean_field_lean<PPT, K, 1, NT, true, true>(smem, lay, V, N, t, pr, cl, alpha, wk, src, 1, 0.0f, 0.0f, ins);
    __syncthreads();
    FL_STAMP();

    t = lane_id();
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        typedef unsigned lean_u2 __attribute__((ext_vector_type(2)));
        const lean_u2 u = __builtin_amdgcn_raw_buffer_load_b64(src.unary, (t + s * NT) * 8, 0, 0);
        pr.un[s] = make_float2(__uint_as_float(u.x), __uint_as_float(u.y));
    }
    start_inference<PPT, K, NT>(pr, N, t);
    mean_field_lean<PPT, K, 1, NT, true, false>(smem, lay, V, N, t, pr, cl, alpha, wk, src, a.n_iter, a.relax, a.omr, ins);
    t = lane_id();
    store_results<PPT, K, NT>(c, f, N, t, pr, a.with_map);
    if (t < K && a.V_out[t]) a.V_out[t][f] = t == 0 ? V[0] : V[K - 1];
    if (t == 0 && a.frame_status) a.frame_status[f] = 0;
    FL_STAMP();
    if (kInstr && a.timing && (int)blockIdx.x == a.timing_block && t == a.timing_lane) a.timing[63] = ins.n;
}

template <int PPT>
void launch_lean_ppt(const CrfDev &c, const FrameArgs &a, hipStream_t s)
{
    auto fn = k_frame_lean<PPT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
    fn<<<dim3(c.F), dim3(kNTSmall), a.lds_total, s>>>(c, a);
}

}  // namespace

// Is the half-CU form worth asking for?  Frames of NA points with lattices of the usual SLAM proportions (frame_engine.hip: an appearance
// kernel of ~112 vertices, a smoothness kernel of min(NA + 350, 1150)) must fit the loop's plan; frames with larger lattices flag
// themselves and are re-run, and an engine that sees more than 1/8 of a batch flagged stops asking.
bool frame_lean_plausible(int NA, int K, int F)
{
    if (K != 2 || F < kSmallMinFrames || NA <= 2 * kNTSmall || NA > 4 * kNTSmall) return false;
    const int vest[2] = {112, std::min(NA + 350, 1150)};
    FusedLayout est;
    return layout_lean(NA, K, vest, kChainMinRow, &est, kNTSmall, kLdsHalf);
}

void launch_frame_lean(const CrfDev &c, const FrameArgs &a, int NA, hipStream_t s)
{
    if (NA <= 3 * kNTSmall) launch_lean_ppt<3>(c, a, s);
    else launch_lean_ppt<4>(c, a, s);
}

}  // namespace fb
}  // namespace lccrf
