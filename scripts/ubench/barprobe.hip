// Can the host write device memory directly (large PCIe BAR)?  Probe only: fine-grained device allocations, a CPU store,
// a kernel that reads it back.  A machine without host-visible VRAM faults in the CPU store (the process dies; nothing else).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_sum(const unsigned *p, int n, unsigned *out)
{
    unsigned s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    atomicAdd(out, s);
}

int main()
{
    const int n = 9216;                                   // 36 KB
    unsigned *dev = nullptr, *out = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&dev), n * 4, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    (void)hipHostMalloc(reinterpret_cast<void **>(&out), 4, hipHostMallocCoherent);
    hipPointerAttribute_t at{};
    (void)hipPointerGetAttributes(&at, dev);
    printf("type %d device %d hostPointer %p devicePointer %p\n", (int)at.type, at.device, at.hostPointer, at.devicePointer);
    fflush(stdout);
    std::vector<unsigned> src(n);
    for (int i = 0; i < n; ++i) src[i] = i * 2654435761u;
    unsigned want = 0;
    for (unsigned v : src) want += v;
    printf("CPU store into device memory...\n");
    fflush(stdout);
    for (int rep = 0; rep < 5; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        memcpy(dev, src.data(), n * 4);                   // faults here without a host-visible mapping
        __builtin_ia32_sfence();
        const auto t1 = std::chrono::steady_clock::now();
        *out = 0;
        k_sum<<<1, 256>>>(dev, n, out);
        (void)hipDeviceSynchronize();
        printf("rep %d: memcpy %.2f us, kernel saw %s\n", rep, std::chrono::duration<double, std::micro>(t1 - t0).count(), *out == want ? "the data" : "something else");
    }
    return 0;
}
