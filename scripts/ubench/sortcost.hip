// How long does one global hipcub radix sort of (frame | Z-order code, index) pairs take?  
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main()
{
    for (int n : {100000, 800000, 4700000}) {
        for (int bits : {40, 64}) {
            std::vector<unsigned long long> h(n);
            std::mt19937_64 rng(1);
            for (auto &x : h) x = rng() >> (64 - bits);
            unsigned long long *ki, *ko; int *vi, *vo; void *tmp = nullptr; size_t tb = 0;
            hipMalloc(&ki, n * 8); hipMalloc(&ko, n * 8); hipMalloc(&vi, n * 4); hipMalloc(&vo, n * 4);
            hipMemcpy(ki, h.data(), n * 8, hipMemcpyHostToDevice);
            hipcub::DeviceRadixSort::SortPairs(tmp, tb, ki, ko, vi, vo, n, 0, bits);
            hipMalloc(&tmp, tb);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int r = 0; r < 3; ++r) hipcub::DeviceRadixSort::SortPairs(tmp, tb, ki, ko, vi, vo, n, 0, bits);
            hipEventRecord(a);
            for (int r = 0; r < 10; ++r) hipcub::DeviceRadixSort::SortPairs(tmp, tb, ki, ko, vi, vo, n, 0, bits);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("n=%d bits=%d: %.1f us per sort (temp %zu KB)\n", n, bits, ms * 100.0f, tb >> 10);
            hipFree(ki); hipFree(ko); hipFree(vi); hipFree(vo); hipFree(tmp);
        }
    }
    return 0;
}
