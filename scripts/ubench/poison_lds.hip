// Fill the LDS of every CU with a pattern (default: quiet NaNs) so that a later kernel that reads
// LDS it never wrote shows up as a wrong result instead of passing on stale, plausible data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(1024) k_poison(unsigned pat, unsigned *sink)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds[i] = pat ? pat : (unsigned)(i * 2654435761u);
    __syncthreads();
    if (threadIdx.x == 0 && lds[17] == 12345u) sink[0] = 1;
    // stay resident long enough that every CU gets a workgroup
    long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
}
int main(int argc, char **argv)
{
    unsigned pat = argc > 1 ? (unsigned)strtoul(argv[1], nullptr, 16) : 0x7fc00000u;
    unsigned *sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void *)k_poison, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int r = 0; r < 3; ++r) k_poison<<<1024, 1024, 160 * 1024>>>(pat, sink);
    hipError_t e = hipDeviceSynchronize();
    printf("poisoned LDS with %08x: %s\n", pat, hipGetErrorString(e));
    return e != hipSuccess;
}
