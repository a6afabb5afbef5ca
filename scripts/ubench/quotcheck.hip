// Exhaustive check (GPU box): the two quotients of the two-label softmax (device_math.h: softmax2_fresh) -- 1 / tt and e / tt with
// tt = fl(1 + e), e in {0} U [2^-60, 1] (the kernels' e lie in {0} U [2^-29, 1]) -- computed with ONE residual correction instead of two, against IEEE division, for EVERY e.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o quotcheck quotcheck.hip && ./quotcheck
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void k_check(unsigned lo, unsigned n, unsigned long long *bad /* [4] */, unsigned *first /* [4] */)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float e = __uint_as_float(lo + i);
    const float tt = 1.0f + e;
    const float ref_m = 1.0f / tt, ref_e = e / tt;                     // IEEE (hipcc's expansion)
    const float r0 = __builtin_amdgcn_rcpf(tt);
    const float r = __builtin_fmaf(__builtin_fmaf(-tt, r0, 1.0f), r0, r0);
    auto quot1 = [&](float nn) { const float q = nn * r; return __builtin_fmaf(__builtin_fmaf(-tt, q, nn), r, q); };
    auto quot2 = [&](float nn) { const float q2 = quot1(nn); return __builtin_fmaf(__builtin_fmaf(-tt, q2, nn), r, q2); };
    const float c[4] = {quot1(1.0f), quot1(e), quot2(1.0f), quot2(e)};
    const float w[4] = {ref_m, ref_e, ref_m, ref_e};
    for (int k = 0; k < 4; ++k)
        if (__float_as_uint(c[k]) != __float_as_uint(w[k])) {
            atomicAdd(&bad[k], 1ull);
            atomicMin(&first[k], lo + i);
        }
}

int main()
{
    unsigned long long *bad, hb[4];
    unsigned *first, hf[4];
    hipMalloc(&bad, 32); hipMalloc(&first, 16);
    hipMemset(bad, 0, 32); hipMemset(first, 0xff, 16);
    const unsigned lo = 0x21800000u, hi = 0x3f800000u;   // [2^-60, 1]: fast_exp's smallest non-zero value is exp(-20) = 2^-28.9
    const unsigned n = hi - lo + 1;
    const unsigned chunk = 1u << 28;
    for (unsigned long long s = 0; s < n; s += chunk) {
        const unsigned m = (unsigned)((n - s) < chunk ? (n - s) : chunk);
        k_check<<<(m + 255) / 256, 256>>>((unsigned)(lo + s), m, bad, first);
    }
    k_check<<<1, 256>>>(0u, 1u, bad, first);                            // e = 0
    hipDeviceSynchronize();
    hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 16, hipMemcpyDeviceToHost);
    const char *nm[4] = {"1/tt, one correction", "e/tt, one correction", "1/tt, two corrections (shipped)", "e/tt, two corrections (shipped)"};
    printf("%u values of e\n", n + 1);
    for (int k = 0; k < 4; ++k) printf("%-34s %llu differ from IEEE division (first e bits 0x%08x)\n", nm[k], hb[k], hf[k]);
    return 0;
}
