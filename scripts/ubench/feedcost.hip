// What does it cost one wavefront to feed a dependent v_add_f32 chain from LDS?
// Each variant runs `units` times: 8 dependent adds + the loads for a later unit (4-deep ring).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ADD8(a,b,c,d,e,f,g,h) \
    "v_add_f32 %[acc], %[acc], " #a "\n\tv_add_f32 %[acc], %[acc], " #b "\n\tv_add_f32 %[acc], %[acc], " #c "\n\t" \
    "v_add_f32 %[acc], %[acc], " #d "\n\tv_add_f32 %[acc], %[acc], " #e "\n\tv_add_f32 %[acc], %[acc], " #f "\n\t" \
    "v_add_f32 %[acc], %[acc], " #g "\n\tv_add_f32 %[acc], %[acc], " #h "\n\t"
#define CLOB "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
             "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"

template <int MODE>
__global__ void __launch_bounds__(64) k_feed(float *out, long long *cyc, int units, int stride)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = 1.0f + i * 1e-7f;
    __syncthreads();
    unsigned ad = (unsigned)(size_t)lds + ((threadIdx.x * stride * 4) & 0x7fff);
    float acc = 0.f;
    unsigned n = units / 4;
    long long t0 = clock64();
    if (MODE == 0) {          // adds only
        asm volatile("1:\n\t" ADD8(v96,v97,v98,v99,v100,v101,v102,v103) ADD8(v104,v105,v106,v107,v108,v109,v110,v111)
                     ADD8(v112,v113,v114,v115,v116,v117,v118,v119) ADD8(v120,v121,v122,v123,v124,v125,v126,v127)
                     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\t"
                     : [acc] "+v"(acc), [n] "+s"(n) : : "scc", CLOB);
    } else if (MODE == 1) {   // + 2 ds_read_b128 per unit, waited with lgkmcnt(6)
        asm volatile("ds_read_b128 v[96:99], %[ad]\n\tds_read_b128 v[100:103], %[ad] offset:16\n\t"
                     "ds_read_b128 v[104:107], %[ad] offset:32\n\tds_read_b128 v[108:111], %[ad] offset:48\n\t"
                     "ds_read_b128 v[112:115], %[ad] offset:64\n\tds_read_b128 v[116:119], %[ad] offset:80\n\t"
                     "1:\n\t"
                     "ds_read_b128 v[120:123], %[ad] offset:96\n\tds_read_b128 v[124:127], %[ad] offset:112\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v96,v97,v98,v99,v100,v101,v102,v103)
                     "ds_read_b128 v[96:99], %[ad] offset:128\n\tds_read_b128 v[100:103], %[ad] offset:144\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v104,v105,v106,v107,v108,v109,v110,v111)
                     "ds_read_b128 v[104:107], %[ad] offset:160\n\tds_read_b128 v[108:111], %[ad] offset:176\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v112,v113,v114,v115,v116,v117,v118,v119)
                     "ds_read_b128 v[112:115], %[ad] offset:192\n\tds_read_b128 v[116:119], %[ad] offset:208\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v120,v121,v122,v123,v124,v125,v126,v127)
                     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [acc] "+v"(acc), [n] "+s"(n) : [ad] "v"(ad) : "scc", "memory", CLOB);
    } else if (MODE == 2) {   // loads issued but the adds use other registers (no data dependence, wait only at the end)
        asm volatile("1:\n\t"
                     "ds_read_b128 v[120:123], %[ad] offset:96\n\tds_read_b128 v[124:127], %[ad] offset:112\n\t"
                     ADD8(v96,v97,v98,v99,v100,v101,v102,v103)
                     "ds_read_b128 v[120:123], %[ad] offset:128\n\tds_read_b128 v[124:127], %[ad] offset:144\n\t"
                     ADD8(v104,v105,v106,v107,v108,v109,v110,v111)
                     "ds_read_b128 v[120:123], %[ad] offset:160\n\tds_read_b128 v[124:127], %[ad] offset:176\n\t"
                     ADD8(v112,v113,v114,v115,v116,v117,v118,v119)
                     "ds_read_b128 v[120:123], %[ad] offset:192\n\tds_read_b128 v[124:127], %[ad] offset:208\n\t"
                     ADD8(v96,v97,v98,v99,v100,v101,v102,v103)
                     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [acc] "+v"(acc), [n] "+s"(n) : [ad] "v"(ad) : "scc", "memory", CLOB);
    } else if (MODE == 3) {   // as 1 with 4 ds_read_b64 per unit
        asm volatile("1:\n\t"
                     "ds_read_b64 v[120:121], %[ad] offset:96\n\tds_read_b64 v[122:123], %[ad] offset:104\n\tds_read_b64 v[124:125], %[ad] offset:112\n\tds_read_b64 v[126:127], %[ad] offset:120\n\t"
                     ADD8(v96,v97,v98,v99,v100,v101,v102,v103)
                     "ds_read_b64 v[120:121], %[ad] offset:96\n\tds_read_b64 v[122:123], %[ad] offset:104\n\tds_read_b64 v[124:125], %[ad] offset:112\n\tds_read_b64 v[126:127], %[ad] offset:120\n\t"
                     ADD8(v104,v105,v106,v107,v108,v109,v110,v111)
                     "ds_read_b64 v[120:121], %[ad] offset:96\n\tds_read_b64 v[122:123], %[ad] offset:104\n\tds_read_b64 v[124:125], %[ad] offset:112\n\tds_read_b64 v[126:127], %[ad] offset:120\n\t"
                     ADD8(v112,v113,v114,v115,v116,v117,v118,v119)
                     "ds_read_b64 v[120:121], %[ad] offset:96\n\tds_read_b64 v[122:123], %[ad] offset:104\n\tds_read_b64 v[124:125], %[ad] offset:112\n\tds_read_b64 v[126:127], %[ad] offset:120\n\t"
                     ADD8(v96,v97,v98,v99,v100,v101,v102,v103)
                     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [acc] "+v"(acc), [n] "+s"(n) : [ad] "v"(ad) : "scc", "memory", CLOB);
    } else if (MODE == 4) {   // 8 ds_read_b32 per unit, no dependence
        asm volatile("1:\n\t"
                     "ds_read_b32 v120, %[ad] offset:96\n\tds_read_b32 v121, %[ad] offset:100\n\tds_read_b32 v122, %[ad] offset:104\n\tds_read_b32 v123, %[ad] offset:108\n\t"
                     "ds_read_b32 v124, %[ad] offset:112\n\tds_read_b32 v125, %[ad] offset:116\n\tds_read_b32 v126, %[ad] offset:120\n\tds_read_b32 v127, %[ad] offset:124\n\t"
                     ADD8(v96,v97,v98,v99,v100,v101,v102,v103)
                     "ds_read_b32 v120, %[ad] offset:96\n\tds_read_b32 v121, %[ad] offset:100\n\tds_read_b32 v122, %[ad] offset:104\n\tds_read_b32 v123, %[ad] offset:108\n\t"
                     "ds_read_b32 v124, %[ad] offset:112\n\tds_read_b32 v125, %[ad] offset:116\n\tds_read_b32 v126, %[ad] offset:120\n\tds_read_b32 v127, %[ad] offset:124\n\t"
                     ADD8(v104,v105,v106,v107,v108,v109,v110,v111)
                     "ds_read_b32 v120, %[ad] offset:96\n\tds_read_b32 v121, %[ad] offset:100\n\tds_read_b32 v122, %[ad] offset:104\n\tds_read_b32 v123, %[ad] offset:108\n\t"
                     "ds_read_b32 v124, %[ad] offset:112\n\tds_read_b32 v125, %[ad] offset:116\n\tds_read_b32 v126, %[ad] offset:120\n\tds_read_b32 v127, %[ad] offset:124\n\t"
                     ADD8(v112,v113,v114,v115,v116,v117,v118,v119)
                     "ds_read_b32 v120, %[ad] offset:96\n\tds_read_b32 v121, %[ad] offset:100\n\tds_read_b32 v122, %[ad] offset:104\n\tds_read_b32 v123, %[ad] offset:108\n\t"
                     "ds_read_b32 v124, %[ad] offset:112\n\tds_read_b32 v125, %[ad] offset:116\n\tds_read_b32 v126, %[ad] offset:120\n\tds_read_b32 v127, %[ad] offset:124\n\t"
                     ADD8(v96,v97,v98,v99,v100,v101,v102,v103)
                     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [acc] "+v"(acc), [n] "+s"(n) : [ad] "v"(ad) : "scc", "memory", CLOB);
    }
    else if (MODE == 5) {   // mode 1 + the compiler's per-unit address select (v_mov, v_cmp, v_cndmask)
        unsigned nh = 100000u, sel, cst;
        asm volatile("ds_read_b128 v[96:99], %[ad]\n\tds_read_b128 v[100:103], %[ad] offset:16\n\t"
                     "ds_read_b128 v[104:107], %[ad] offset:32\n\tds_read_b128 v[108:111], %[ad] offset:48\n\t"
                     "ds_read_b128 v[112:115], %[ad] offset:64\n\tds_read_b128 v[116:119], %[ad] offset:80\n\t"
                     "1:\n\t"
                     "v_mov_b32 %[cst], 0x80\n\tv_cmp_lt_u32_e32 vcc, %[n], %[nh]\n\tv_cndmask_b32_e32 %[sel], %[cst], %[ad], vcc\n\t"
                     "ds_read_b128 v[120:123], %[sel] offset:96\n\tds_read_b128 v[124:127], %[sel] offset:112\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v96,v97,v98,v99,v100,v101,v102,v103)
                     "v_mov_b32 %[cst], 0x80\n\tv_cmp_lt_u32_e32 vcc, %[n], %[nh]\n\tv_cndmask_b32_e32 %[sel], %[cst], %[ad], vcc\n\t"
                     "ds_read_b128 v[96:99], %[sel] offset:128\n\tds_read_b128 v[100:103], %[sel] offset:144\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v104,v105,v106,v107,v108,v109,v110,v111)
                     "v_mov_b32 %[cst], 0x80\n\tv_cmp_lt_u32_e32 vcc, %[n], %[nh]\n\tv_cndmask_b32_e32 %[sel], %[cst], %[ad], vcc\n\t"
                     "ds_read_b128 v[104:107], %[sel] offset:160\n\tds_read_b128 v[108:111], %[sel] offset:176\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v112,v113,v114,v115,v116,v117,v118,v119)
                     "v_mov_b32 %[cst], 0x80\n\tv_cmp_lt_u32_e32 vcc, %[n], %[nh]\n\tv_cndmask_b32_e32 %[sel], %[cst], %[ad], vcc\n\t"
                     "ds_read_b128 v[112:115], %[sel] offset:192\n\tds_read_b128 v[116:119], %[sel] offset:208\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v120,v121,v122,v123,v124,v125,v126,v127)
                     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [acc] "+v"(acc), [n] "+s"(n), [sel] "=&v"(sel), [cst] "=&v"(cst) : [ad] "v"(ad), [nh] "v"(nh) : "scc", "vcc", "memory", CLOB);
    } else if (MODE == 6) {   // mode 1, but the running sum hops through the loaded registers as the compiler's code does
        asm volatile("ds_read_b128 v[96:99], %[ad]\n\tds_read_b128 v[100:103], %[ad] offset:16\n\t"
                     "ds_read_b128 v[104:107], %[ad] offset:32\n\tds_read_b128 v[108:111], %[ad] offset:48\n\t"
                     "ds_read_b128 v[112:115], %[ad] offset:64\n\tds_read_b128 v[116:119], %[ad] offset:80\n\t"
                     "1:\n\t"
                     "ds_read_b128 v[120:123], %[ad] offset:96\n\tds_read_b128 v[124:127], %[ad] offset:112\n\ts_waitcnt lgkmcnt(6)\n\t"
                     "v_add_f32 v96, %[acc], v96\n\tv_add_f32 v96, v97, v96\n\tv_add_f32 v96, v98, v96\n\tv_add_f32 v96, v99, v96\n\t"
                     "v_add_f32 v100, v100, v96\n\tv_add_f32 v100, v101, v100\n\tv_add_f32 v100, v102, v100\n\tv_add_f32 %[acc], v103, v100\n\t"
                     "ds_read_b128 v[96:99], %[ad] offset:128\n\tds_read_b128 v[100:103], %[ad] offset:144\n\ts_waitcnt lgkmcnt(6)\n\t"
                     "v_add_f32 v104, %[acc], v104\n\tv_add_f32 v104, v105, v104\n\tv_add_f32 v104, v106, v104\n\tv_add_f32 v104, v107, v104\n\t"
                     "v_add_f32 v108, v108, v104\n\tv_add_f32 v108, v109, v108\n\tv_add_f32 v108, v110, v108\n\tv_add_f32 %[acc], v111, v108\n\t"
                     "ds_read_b128 v[104:107], %[ad] offset:160\n\tds_read_b128 v[108:111], %[ad] offset:176\n\ts_waitcnt lgkmcnt(6)\n\t"
                     "v_add_f32 v112, %[acc], v112\n\tv_add_f32 v112, v113, v112\n\tv_add_f32 v112, v114, v112\n\tv_add_f32 v112, v115, v112\n\t"
                     "v_add_f32 v116, v116, v112\n\tv_add_f32 v116, v117, v116\n\tv_add_f32 v116, v118, v116\n\tv_add_f32 %[acc], v119, v116\n\t"
                     "ds_read_b128 v[112:115], %[ad] offset:192\n\tds_read_b128 v[116:119], %[ad] offset:208\n\ts_waitcnt lgkmcnt(6)\n\t"
                     "v_add_f32 v120, %[acc], v120\n\tv_add_f32 v120, v121, v120\n\tv_add_f32 v120, v122, v120\n\tv_add_f32 v120, v123, v120\n\t"
                     "v_add_f32 v124, v124, v120\n\tv_add_f32 v124, v125, v124\n\tv_add_f32 v124, v126, v124\n\tv_add_f32 %[acc], v127, v124\n\t"
                     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [acc] "+v"(acc), [n] "+s"(n) : [ad] "v"(ad) : "scc", "memory", CLOB);
    }
    else if (MODE == 7) {   // ring + one v_min_u32 per unit (address clamp) + one address add per trip
        unsigned e0 = ad + 0x7ffffu, e1 = e0 - 32u, e2 = e0 - 64u, e3 = e0 - 96u, sel;
        asm volatile("ds_read_b128 v[96:99], %[ad]\n\tds_read_b128 v[100:103], %[ad] offset:16\n\t"
                     "ds_read_b128 v[104:107], %[ad] offset:32\n\tds_read_b128 v[108:111], %[ad] offset:48\n\t"
                     "ds_read_b128 v[112:115], %[ad] offset:64\n\tds_read_b128 v[116:119], %[ad] offset:80\n\t"
                     "1:\n\t"
                     "v_min_u32 %[sel], %[ad], %[e3]\n\t"
                     "ds_read_b128 v[120:123], %[sel] offset:96\n\tds_read_b128 v[124:127], %[sel] offset:112\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v96,v97,v98,v99,v100,v101,v102,v103)
                     "v_add_u32 %[ad], 0, %[ad]\n\t"
                     "v_min_u32 %[sel], %[ad], %[e0]\n\t"
                     "ds_read_b128 v[96:99], %[sel] offset:128\n\tds_read_b128 v[100:103], %[sel] offset:144\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v104,v105,v106,v107,v108,v109,v110,v111)
                     "v_min_u32 %[sel], %[ad], %[e1]\n\t"
                     "ds_read_b128 v[104:107], %[sel] offset:160\n\tds_read_b128 v[108:111], %[sel] offset:176\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v112,v113,v114,v115,v116,v117,v118,v119)
                     "v_min_u32 %[sel], %[ad], %[e2]\n\t"
                     "ds_read_b128 v[112:115], %[sel] offset:192\n\tds_read_b128 v[116:119], %[sel] offset:208\n\ts_waitcnt lgkmcnt(6)\n\t"
                     ADD8(v120,v121,v122,v123,v124,v125,v126,v127)
                     "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : [acc] "+v"(acc), [n] "+s"(n), [sel] "=&v"(sel), [ad] "+v"(ad) : [e0] "v"(e0), [e1] "v"(e1), [e2] "v"(e2), [e3] "v"(e3) : "scc", "memory", CLOB);
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    float *d; long long *c; hipMalloc(&d, 4096); hipMalloc(&c, 8);
    const int units = 400;
    const char *names[8] = {"adds only", "2 x ds_read_b128 / unit, ring + lgkmcnt(6)", "2 x ds_read_b128 / unit, independent",
                           "4 x ds_read_b64 / unit, independent", "8 x ds_read_b32 / unit, independent",
                           "ring + per-unit address select (mov, cmp, cndmask)", "ring, sum hops through the loaded registers", "ring + v_min address clamp per unit + address add per trip"};
    for (int stride : {4}) {
        for (int m = 0; m < 8; ++m) {
            long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                switch (m) {
                case 0: k_feed<0><<<1, 64>>>(d, c, units, stride); break;
                case 1: k_feed<1><<<1, 64>>>(d, c, units, stride); break;
                case 2: k_feed<2><<<1, 64>>>(d, c, units, stride); break;
                case 3: k_feed<3><<<1, 64>>>(d, c, units, stride); break;
                case 4: k_feed<4><<<1, 64>>>(d, c, units, stride); break;
                case 5: k_feed<5><<<1, 64>>>(d, c, units, stride); break;
                case 6: k_feed<6><<<1, 64>>>(d, c, units, stride); break;
                default: k_feed<7><<<1, 64>>>(d, c, units, stride); break;
                }
                hipDeviceSynchronize(); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            }
            printf("lane stride %2d floats  %-46s %7lld cycles  %.2f per add\n", stride, names[m], h, (double)h / (units * 8));
        }
    }
    return 0;
}
