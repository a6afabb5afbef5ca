// Micro-benchmarks behind the fused engine's design (run on the GPU box):
//   dependent v_add_f32 chain, dependent v_pk_add_f32 chain, LDS-fed chains, ds_read latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_add_chain(float *out, long long *cyc, int n, float x)
{
    float acc = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc += x;
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_pk_chain(float2 *out, long long *cyc, int n, float2 x)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc = {out[threadIdx.x].x, out[threadIdx.x].y};
    f2 xx = {x.x, x.y};
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc += xx;
    }
    long long t1 = clock64();
    out[threadIdx.x] = make_float2(acc.x, acc.y);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// each lane walks its own contiguous run of floats in LDS (stride between lanes = `stride` floats)
template <int DEPTH>
__global__ void k_lds_chain(float *out, long long *cyc, int nblk, int stride)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1.0f + i * 1e-7f;
    __syncthreads();
    const float4 *p = reinterpret_cast<const float4 *>(lds + (threadIdx.x * stride) % 8192);
    float acc = 0.f;
    long long t0 = clock64();
    float4 A[DEPTH], B[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) A[u] = p[u];
    for (int b = 0; b < nblk; b += 2) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) B[u] = p[((b + 1) * DEPTH + u) & 255];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) { acc += A[u].x; acc += A[u].y; acc += A[u].z; acc += A[u].w; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) A[u] = p[((b + 2) * DEPTH + u) & 255];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) { acc += B[u].x; acc += B[u].y; acc += B[u].z; acc += B[u].w; }
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_lds_latency(int *out, long long *cyc, int n)
{
    __shared__ int lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (i * 97 + 13) & 4095;
    __syncthreads();
    int j = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) j = lds[j];
    long long t1 = clock64();
    out[threadIdx.x] = j;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *d; long long *c; hipMalloc(&d, 1 << 20); hipMalloc(&c, 1024);
    hipMemset(d, 0, 1 << 20);
    long long h;
    auto rd = [&](const char *name, double ops) {
        hipDeviceSynchronize(); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-44s %8lld cycles  %.2f cycles/op\n", name, h, h / ops);
    };
    for (int threads : {64, 256, 512, 1024}) {
        char nm[96];
        k_add_chain<<<1, threads>>>(d, c, 100, 1.0f); snprintf(nm, 96, "v_add_f32 dependent chain, %d thr", threads); rd(nm, 3200);
        k_pk_chain<<<1, threads>>>((float2 *)d, c, 100, make_float2(1.f, 2.f)); snprintf(nm, 96, "v_pk_add_f32 dependent chain, %d thr", threads); rd(nm, 3200);
    }
    for (int threads : {64, 256}) {
        for (int stride : {4, 36, 132}) {
            char nm[96];
            k_lds_chain<4><<<1, threads, 65536>>>(d, c, 32, stride); snprintf(nm, 96, "LDS chain depth 4x b128, %d thr stride %d", threads, stride); rd(nm, 32 * 16 + 0.0);
            k_lds_chain<8><<<1, threads, 65536>>>(d, c, 16, stride); snprintf(nm, 96, "LDS chain depth 8x b128, %d thr stride %d", threads, stride); rd(nm, 16 * 32 + 0.0);
        }
    }
    k_lds_latency<<<1, 64>>>((int *)d, c, 1000); rd("ds_read_b32 dependent latency, 64 thr", 1000);
    k_lds_latency<<<1, 1024>>>((int *)d, c, 1000); rd("ds_read_b32 dependent latency, 1024 thr", 1000);
    return 0;
}
