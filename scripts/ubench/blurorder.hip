// blurorder.hip -- the shipped blur pass (k_blur2 shape: two vertices per lane, 16-byte table / centre / store accesses, 8-byte
// gathers, non-temporal table loads, one XCD per frame) over 8 copies of a REAL C5 lattice whose vertices are numbered in
// different orders.  The neighbour tables come from scripts/sim_vertex_order.py (`python scripts/sim_vertex_order.py 100000
// zpoints,rowmajor,...` writes scripts/ubench/data/nbr_<order>.bin).  Prices VERDICT r3 item 2 before the build is touched:
//   ./blurorder data/nbr_zpoints.bin data/nbr_rowmajor.bin ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kBlock = 256, kF = 8;
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(kBlock) k_blur(const float2 *__restrict__ src, float2 *__restrict__ dst, const int *__restrict__ nbr, int V, size_t vstride,
                                                 size_t nstride, int nb)
{
    const int L = blockIdx.x, f = L & 7, bx = L >> 3;           // workgroup L runs on XCD L % 8: one XCD per frame
    if (bx >= nb) return;
    const int v = 2 * (bx * kBlock + threadIdx.x);
    if (v + 1 >= V) return;
    const float2 *o = src + f * vstride;
    float2 *d = dst + f * vstride;
    const v4i n = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(nbr + f * nstride + 2 * (size_t)v));
    const float4 c = *reinterpret_cast<const float4 *>(o + v);
    const float2 x0 = o[n.x], y0 = o[n.y], x1 = o[n.z], y1 = o[n.w];
    *reinterpret_cast<float4 *>(d + v) = make_float4(c.x + 0.5f * (x0.x + y0.x), c.y + 0.5f * (x0.y + y0.y), c.z + 0.5f * (x1.x + y1.x),
                                                     c.w + 0.5f * (x1.y + y1.y));
}

int main(int argc, char **argv)
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int a = 1; a < argc; ++a) {
        FILE *fh = fopen(argv[a], "rb");
        if (!fh) { fprintf(stderr, "cannot open %s\n", argv[a]); continue; }
        int hdr[2];
        if (fread(hdr, 4, 2, fh) != 2) return 1;
        const int V = hdr[0] & ~1, Vfile = hdr[0], axes = hdr[1];
        std::vector<int> tab((size_t)axes * Vfile * 2);
        if (fread(tab.data(), 4, tab.size(), fh) != tab.size()) return 1;
        fclose(fh);
        const size_t vstride = ((size_t)Vfile + 4 + 1) & ~(size_t)1, nstride = (size_t)axes * Vfile * 2;
        float2 *A, *B;
        int *nbr;
        CK(hipMalloc(&A, kF * vstride * sizeof(float2)));
        CK(hipMalloc(&B, kF * vstride * sizeof(float2)));
        CK(hipMalloc(&nbr, kF * nstride * 4));
        CK(hipMemset(A, 0, kF * vstride * sizeof(float2)));
        CK(hipMemset(B, 0, kF * vstride * sizeof(float2)));
        for (int f = 0; f < kF; ++f) CK(hipMemcpy(nbr + f * nstride, tab.data(), nstride * 4, hipMemcpyHostToDevice));
        const int nb = ((V + 1) / 2 + kBlock - 1) / kBlock;
        auto pass = [&](int p) {
            const float2 *s = ((p & 1) ? B : A) + 2;         // [-1] = the absent vertex, vertex 0 16-byte aligned
            float2 *d = ((p & 1) ? A : B) + 2;
            k_blur<<<8 * nb, kBlock, 0, st>>>(s, d, nbr + (size_t)(p % axes) * Vfile * 2, V, vstride, nstride, nb);
        };
        for (int p = 0; p < 14; ++p) pass(p);
        CK(hipStreamSynchronize(st));
        const int reps = 20 * axes;
        CK(hipEventRecord(e0, st));
        for (int p = 0; p < reps; ++p) pass(p);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%-44s V %d  %7.2f us per pass over %d frames  = %.2f TB/s at 24 B per vertex (%.3f of 8 TB/s)\n", argv[a], Vfile, us, kF,
               24.0 * Vfile * kF / us * 1e-6, 24.0 * Vfile * kF / us * 1e-6 / 8.0);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(nbr));
    }
    return 0;
}
