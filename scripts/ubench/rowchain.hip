// Micro-benchmark: strictly ordered fp32 row sums fed from LDS, the S phase of the fused engine.
// Compares the compiler-scheduled loop with a hand-scheduled one (explicit s_waitcnt so that the
// next block's ds_read_b128 are in flight while the current block's 16 dependent adds retire).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int kNT = 1024;

// rows: start[r] (in floats, multiple of 4), len[r] (entries), padded with zeros to a multiple of 16
__global__ void __launch_bounds__(kNT) k_rows_cxx(const float *src, int total, const int *start, const int *len, int R,
                                                   float *out, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < total; i += kNT) lds[i] = src[i];
    __syncthreads();
    long long t0 = clock64();
    const int tid = threadIdx.x;
    if (tid < R) {
        const float *pl = lds + start[tid];
        const int t = len[tid];
        float acc = 0.f;
        int p = 0;
        for (; p + 8 <= t; p += 8) {
            const float x0 = pl[p], x1 = pl[p + 1], x2 = pl[p + 2], x3 = pl[p + 3];
            const float x4 = pl[p + 4], x5 = pl[p + 5], x6 = pl[p + 6], x7 = pl[p + 7];
            acc += x0; acc += x1; acc += x2; acc += x3;
            acc += x4; acc += x5; acc += x6; acc += x7;
        }
        for (; p < t; ++p) acc += pl[p];
        out[tid] = acc;
    }
    __syncthreads();
    long long t1 = clock64();
    if (tid == 0) cyc[0] = t1 - t0;
}

// hand-scheduled: blocks of 16 entries, two blocks per trip, A = v[96:111], B = v[112:127]
__device__ __forceinline__ float chain_asm(unsigned addr, unsigned nblk, unsigned wave_max_blk)
{
    float acc = 0.f;
    unsigned it = 0;
    unsigned long long save;
    if (wave_max_blk == 0) return acc;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "ds_read_b128 v[96:99], %[ad]\n\t"
        "ds_read_b128 v[100:103], %[ad] offset:16\n\t"
        "ds_read_b128 v[104:107], %[ad] offset:32\n\t"
        "ds_read_b128 v[108:111], %[ad] offset:48\n\t"
        "1:\n\t"
        "ds_read_b128 v[112:115], %[ad] offset:64\n\t"
        "ds_read_b128 v[116:119], %[ad] offset:80\n\t"
        "ds_read_b128 v[120:123], %[ad] offset:96\n\t"
        "ds_read_b128 v[124:127], %[ad] offset:112\n\t"
        "v_cmpx_lt_u32_e32 vcc, %[it], %[nb]\n\t"
        "v_add_u32_e32 %[ad], 0x80, %[ad]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_add_f32_e32 %[acc], %[acc], v96\n\t"
        "v_add_f32_e32 %[acc], %[acc], v97\n\t"
        "v_add_f32_e32 %[acc], %[acc], v98\n\t"
        "v_add_f32_e32 %[acc], %[acc], v99\n\t"
        "v_add_f32_e32 %[acc], %[acc], v100\n\t"
        "v_add_f32_e32 %[acc], %[acc], v101\n\t"
        "v_add_f32_e32 %[acc], %[acc], v102\n\t"
        "v_add_f32_e32 %[acc], %[acc], v103\n\t"
        "v_add_f32_e32 %[acc], %[acc], v104\n\t"
        "v_add_f32_e32 %[acc], %[acc], v105\n\t"
        "v_add_f32_e32 %[acc], %[acc], v106\n\t"
        "v_add_f32_e32 %[acc], %[acc], v107\n\t"
        "v_add_f32_e32 %[acc], %[acc], v108\n\t"
        "v_add_f32_e32 %[acc], %[acc], v109\n\t"
        "v_add_f32_e32 %[acc], %[acc], v110\n\t"
        "v_add_f32_e32 %[acc], %[acc], v111\n\t"
        "ds_read_b128 v[96:99], %[ad]\n\t"
        "ds_read_b128 v[100:103], %[ad] offset:16\n\t"
        "ds_read_b128 v[104:107], %[ad] offset:32\n\t"
        "ds_read_b128 v[108:111], %[ad] offset:48\n\t"
        "s_add_u32 %[it], %[it], 1\n\t"
        "v_cmpx_lt_u32_e32 vcc, %[it], %[nb]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_add_f32_e32 %[acc], %[acc], v112\n\t"
        "v_add_f32_e32 %[acc], %[acc], v113\n\t"
        "v_add_f32_e32 %[acc], %[acc], v114\n\t"
        "v_add_f32_e32 %[acc], %[acc], v115\n\t"
        "v_add_f32_e32 %[acc], %[acc], v116\n\t"
        "v_add_f32_e32 %[acc], %[acc], v117\n\t"
        "v_add_f32_e32 %[acc], %[acc], v118\n\t"
        "v_add_f32_e32 %[acc], %[acc], v119\n\t"
        "v_add_f32_e32 %[acc], %[acc], v120\n\t"
        "v_add_f32_e32 %[acc], %[acc], v121\n\t"
        "v_add_f32_e32 %[acc], %[acc], v122\n\t"
        "v_add_f32_e32 %[acc], %[acc], v123\n\t"
        "v_add_f32_e32 %[acc], %[acc], v124\n\t"
        "v_add_f32_e32 %[acc], %[acc], v125\n\t"
        "v_add_f32_e32 %[acc], %[acc], v126\n\t"
        "v_add_f32_e32 %[acc], %[acc], v127\n\t"
        "s_add_u32 %[it], %[it], 1\n\t"
        "s_cmp_lt_u32 %[it], %[mx]\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [acc] "+v"(acc), [ad] "+v"(addr), [it] "+s"(it), [sv] "=&s"(save)
        : [nb] "v"(nblk), [mx] "s"(wave_max_blk)
        : "vcc", "scc", "memory", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106",
          "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
          "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    return acc;
}

__global__ void __launch_bounds__(kNT) k_rows_asm(const float *src, int total, const int *start, const int *len, int R,
                                                   float *out, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < total; i += kNT) lds[i] = src[i];
    __syncthreads();
    long long t0 = clock64();
    const int tid = threadIdx.x;
    if (tid < ((R + 63) & ~63)) {                         // whole waves
        const bool live = tid < R;
        const unsigned addr = (unsigned)(size_t)(lds) + (live ? start[tid] : 0) * 4u;   // LDS byte address
        const unsigned nblk = live ? (unsigned)((len[tid] + 15) >> 4) : 0u;
        unsigned m = nblk;
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        const unsigned wmax = __builtin_amdgcn_readfirstlane(m);
        const float acc = chain_asm(addr, nblk, wmax);
        if (live) out[tid] = acc;
    }
    __syncthreads();
    long long t1 = clock64();
    if (tid == 0) cyc[0] = t1 - t0;
}

#define LCCRF_ASM_ADD8(a, b, c, d, e, f, g, h)                                                        \
    "v_add_f32_e32 %[acc], %[acc], " #a "\n\tv_add_f32_e32 %[acc], %[acc], " #b "\n\t"               \
    "v_add_f32_e32 %[acc], %[acc], " #c "\n\tv_add_f32_e32 %[acc], %[acc], " #d "\n\t"               \
    "v_add_f32_e32 %[acc], %[acc], " #e "\n\tv_add_f32_e32 %[acc], %[acc], " #f "\n\t"               \
    "v_add_f32_e32 %[acc], %[acc], " #g "\n\tv_add_f32_e32 %[acc], %[acc], " #h "\n\t"

__device__ __forceinline__ float chain_rows(unsigned addr, unsigned end, unsigned trips)
{
    float acc = 0.0f;
    if (trips == 0) return acc;
    const unsigned e1 = end - 32u, e2 = end - 64u, e3 = end - 96u;      // min(addr, end - k) + k == min(addr + k, end)
    unsigned sel;
    asm volatile(
        "v_min_u32_e32 %[sel], %[ad], %[e0]\n\t"
        "ds_read_b128 v[96:99], %[sel]\n\tds_read_b128 v[100:103], %[sel] offset:16\n\t"
        "v_min_u32_e32 %[sel], %[ad], %[e1]\n\t"
        "ds_read_b128 v[104:107], %[sel] offset:32\n\tds_read_b128 v[108:111], %[sel] offset:48\n\t"
        "v_min_u32_e32 %[sel], %[ad], %[e2]\n\t"
        "ds_read_b128 v[112:115], %[sel] offset:64\n\tds_read_b128 v[116:119], %[sel] offset:80\n\t"
        "1:\n\t"
        "v_min_u32_e32 %[sel], %[ad], %[e3]\n\t"
        "ds_read_b128 v[120:123], %[sel] offset:96\n\tds_read_b128 v[124:127], %[sel] offset:112\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v96, v97, v98, v99, v100, v101, v102, v103)
        "v_add_u32_e32 %[ad], 0x80, %[ad]\n\t"
        "v_min_u32_e32 %[sel], %[ad], %[e0]\n\t"
        "ds_read_b128 v[96:99], %[sel]\n\tds_read_b128 v[100:103], %[sel] offset:16\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v104, v105, v106, v107, v108, v109, v110, v111)
        "v_min_u32_e32 %[sel], %[ad], %[e1]\n\t"
        "ds_read_b128 v[104:107], %[sel] offset:32\n\tds_read_b128 v[108:111], %[sel] offset:48\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v112, v113, v114, v115, v116, v117, v118, v119)
        "v_min_u32_e32 %[sel], %[ad], %[e2]\n\t"
        "ds_read_b128 v[112:115], %[sel] offset:64\n\tds_read_b128 v[116:119], %[sel] offset:80\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        LCCRF_ASM_ADD8(v120, v121, v122, v123, v124, v125, v126, v127)
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [acc] "+v"(acc), [ad] "+v"(addr), [n] "+s"(trips), [sel] "=&v"(sel)
        : [e0] "v"(end), [e1] "v"(e1), [e2] "v"(e2), [e3] "v"(e3)
        : "scc", "memory", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",
          "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120",
          "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    return acc;
}

// the shipped chain_rows (copied from csrc/fused_engine.hip): rows sorted
// longest-first, wavefront w owns label (w & 1) of rows 64*(w >> 1)..+63, `waves` limits who runs
__global__ void __launch_bounds__(kNT) k_rows_ring(const float *src, int total, const int *start, const int *len, int V,
                                                    int plane, int waves, int lanes, float *out, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 2 * plane + 64; i += kNT) lds[i] = i < 48 ? 0.0f : src[(i - 48) % total];
    __syncthreads();
    const int tid = threadIdx.x;
    const int Vr = (V + 63) & ~63;
    const int l = (tid >> 6) & 1, r = ((tid >> 7) << 6) | (tid & 63);
    const bool run = tid < 2 * Vr && (tid >> 6) < waves;
    const bool live = run && r < V && (tid & 63) < lanes;
    const int len4 = live ? (len[r] + 3) & ~3 : 0;
    const unsigned addr = (unsigned)(size_t)(lds) + (48 + l * plane + (live ? start[r] : 0)) * 4u;
    const unsigned nh = (unsigned)((len4 + 7) >> 3);
    unsigned m = nh;
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    m = __builtin_amdgcn_readfirstlane(m);
    float acc = 0.f;
    __syncthreads();
    long long t0 = clock64();
    if (run) acc = chain_rows(addr, addr + len4 * 4u, (m + 3u) >> 2);
    long long t1 = clock64();
    if (live) out[l * V + r] = acc;
    if (tid == 0) cyc[0] = t1 - t0;
}

int main()
{
    srand(7);
    const int R = 236;                                    // 118 vertices x 2 labels
    std::vector<int> len(R), start(R);
    // measured C2 appearance-kernel row lengths: a few very long rows, many short ones
    const int top[] = {437, 328, 320, 305, 272, 265, 240, 230, 210, 200, 190, 170, 160, 150, 140, 130};
    for (int v = 0; v < R / 2; ++v) {
        int l = v < 16 ? top[v] : (v < 44 ? 32 + rand() % 90 : 1 + rand() % 32);
        len[2 * v] = len[2 * v + 1] = l;
    }
    // shuffle vertices (rows are in vertex-id order in the kernel, not sorted)
    for (int v = R / 2 - 1; v > 0; --v) { int u = rand() % (v + 1); std::swap(len[2 * v], len[2 * u]); std::swap(len[2 * v + 1], len[2 * u + 1]); }
    int total = 0, sum = 0, mx = 0;
    for (int r = 0; r < R; ++r) { start[r] = total; total += (len[r] + 15) & ~15; sum += len[r]; mx = std::max(mx, len[r]); }
    std::vector<float> src(total, 0.f), ref(R);
    for (int r = 0; r < R; ++r) {
        float acc = 0.f;
        for (int p = 0; p < len[r]; ++p) { src[start[r] + p] = (float)rand() / RAND_MAX * 0.37f; acc += src[start[r] + p]; }
        ref[r] = acc;
    }
    float *dsrc, *dout; int *dstart, *dlen; long long *dc;
    hipMalloc(&dsrc, total * 4); hipMalloc(&dout, R * 4); hipMalloc(&dstart, R * 4); hipMalloc(&dlen, R * 4); hipMalloc(&dc, 8);
    hipMemcpy(dsrc, src.data(), total * 4, hipMemcpyHostToDevice);
    hipMemcpy(dstart, start.data(), R * 4, hipMemcpyHostToDevice);
    hipMemcpy(dlen, len.data(), R * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_rows_cxx, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k_rows_asm, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("rows %d entries %d padded %d longest %d\n", R, sum, total, mx);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(dout, 0, R * 4);
            if (mode == 0) k_rows_cxx<<<1, kNT, total * 4>>>(dsrc, total, dstart, dlen, R, dout, dc);
            else k_rows_asm<<<1, kNT, total * 4>>>(dsrc, total, dstart, dlen, R, dout, dc);
            hipDeviceSynchronize();
            long long c; std::vector<float> o(R);
            hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(o.data(), dout, R * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int r = 0; r < R; ++r) bad += (o[r] != ref[r]);
            printf("%s: %lld cycles, %.2f cycles per entry of the longest row, mismatches %d\n",
                   mode ? "asm pipelined b128" : "compiler 8x b32   ", c, (double)c / mx, bad);
        }
    }
    {
        // sorted rows, padded to 4, one label plane after the other (plane is a multiple of 64 floats)
        const int V = R / 2;
        std::vector<int> slen(V), sstart(V);
        for (int v = 0; v < V; ++v) slen[v] = len[2 * v];
        std::sort(slen.begin(), slen.end(), [](int a, int b) { return a > b; });
        int tot = 0;
        const bool aligned = getenv("ROWCHAIN_CONFLICT_FREE") != nullptr;   // give lane r the quad slot r % 16
        for (int v = 0; v < V; ++v) {
            if (aligned) while (((tot >> 2) & 15) != (v & 15)) tot += 4;
            sstart[v] = tot; tot += ((slen[v] + 3) & ~3) + 8;     // the row, +0 up to a multiple of 4, eight +0
        }
        printf("placement: %s, %d floats per plane\n", aligned ? "conflict-free quad slots" : "packed", tot);
        const int plane = (tot + 63) & ~63;
        std::vector<float> s2(2 * plane, 0.f), ref2(2 * V);
        for (int l = 0; l < 2; ++l)
            for (int v = 0; v < V; ++v) {
                float acc = 0.f;
                for (int p = 0; p < slen[v]; ++p) { float x = (float)rand() / RAND_MAX * 0.41f; s2[l * plane + sstart[v] + p] = x; acc += x; }
                ref2[l * V + v] = acc;
            }
        float *d2; int *ds, *dl; hipMalloc(&d2, s2.size() * 4); hipMalloc(&ds, V * 4); hipMalloc(&dl, V * 4);
        hipMemcpy(d2, s2.data(), s2.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(ds, sstart.data(), V * 4, hipMemcpyHostToDevice); hipMemcpy(dl, slen.data(), V * 4, hipMemcpyHostToDevice);
        hipFuncSetAttribute((const void *)k_rows_ring, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        // NOTE: the kernel copies src linearly behind 48 floats of zeros: lds[48 + i] = s2[i]
        for (int cfg = 0; cfg < 5; ++cfg) {
            const int waves = cfg < 3 ? (1 << cfg) : 1, lanes = cfg == 3 ? 16 : (cfg == 4 ? 1 : 64);
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(dout, 0, R * 4);
                k_rows_ring<<<1, kNT, (2 * plane + 64) * 4>>>(d2, (int)s2.size(), ds, dl, V, plane, waves, lanes, dout, dc);
                hipDeviceSynchronize();
                long long c; std::vector<float> o(R);
                hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(o.data(), dout, R * 4, hipMemcpyDeviceToHost);
                int bad = 0, chk = 0;
                for (int l = 0; l < 2; ++l) for (int v = 0; v < V; ++v) {
                    const int w = 2 * (v >> 6) + l;
                    if (w < waves && (v & 63) < lanes) { ++chk; bad += (o[l * V + v] != ref2[l * V + v]); }
                }
                printf("shipped chain_rows, %d wavefront(s) x %d lanes: %lld cycles, %.2f per entry of the longest row (%d), mismatches %d of %d\n",
                       waves, lanes, c, (double)c / slen[0], slen[0], bad, chk);
            }
        }
    }
    return 0;
}
