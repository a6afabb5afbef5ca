// fetchcal.hip -- known-byte access patterns for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// ("calibrate FETCH_SIZE for 8-byte gathers on a known-byte micro-benchmark").
//
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -o run -- ./fetchcal
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -o run -- ./fetchcal
//
// Every kernel touches a buffer far beyond the 256 MB Infinity Cache exactly once, so the bytes that must cross the
// L2's memory side are known: printed per kernel as `expect_read` / `expect_write` (bytes requested by the lanes) and,
// for the gathers, `lines64` / `lines128` (bytes if every gather moves one 64-B / 128-B line).  Compare with the
// counters' per-dispatch values (KiB) to get the factor for each access shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kB = 256;

// streaming reads of one element per lane per step: 16 B, 8 B, 4 B
template <typename T>
__global__ void __launch_bounds__(kB) k_stream_read(const T *__restrict__ in, size_t n, float *sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * kB + threadIdx.x; i < n; i += (size_t)gridDim.x * kB) {
        const T v = in[i];
        acc += reinterpret_cast<const float *>(&v)[0];
    }
    if (acc == 12345.678f) *sink = acc;
}

template <typename T>
__global__ void __launch_bounds__(kB) k_stream_write(T *__restrict__ out, size_t n, T v)
{
    for (size_t i = (size_t)blockIdx.x * kB + threadIdx.x; i < n; i += (size_t)gridDim.x * kB) out[i] = v;
}

// 8-byte gathers at pseudo-random positions of a table of `tn` float2 (every lane its own line with high probability)
__global__ void __launch_bounds__(kB) k_gather8(const float2 *__restrict__ table, size_t tn, size_t count, float *sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * kB + threadIdx.x; i < count; i += (size_t)gridDim.x * kB) {
        size_t h = i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        acc += table[h % tn].x;
    }
    if (acc == 12345.678f) *sink = acc;
}

// the blur pass's shape: stream an 8-byte centre + an 8-byte neighbour pair, two 8-byte gathers, one 8-byte store
__global__ void __launch_bounds__(kB) k_blur_shape(const float2 *__restrict__ src, float2 *__restrict__ dst, const int2 *__restrict__ nbr,
                                                   size_t n)
{
    const size_t v = (size_t)blockIdx.x * kB + threadIdx.x;
    if (v >= n) return;
    const int2 nb = nbr[v];
    const float2 c = src[v], a = src[nb.x], b = src[nb.y];
    dst[v] = make_float2(c.x + 0.5f * (a.x + b.x), c.y + 0.5f * (a.y + b.y));
}

int main()
{
    const size_t bytes = (size_t)2 << 30;                 // 2 GiB per buffer: 8x the Infinity Cache
    char *a, *b;
    float *sink;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, bytes));
    CK(hipMemset(b, 0, bytes));
    const int grid = 256 * 16;
    const size_t gathers = (size_t)64 << 20;              // 64 Mi gathers
    printf("kernel,expect_read,expect_write,lines64,lines128\n");
    k_stream_read<float4><<<grid, kB>>>(reinterpret_cast<float4 *>(a), bytes / 16, sink);
    printf("k_stream_read<float4>,%zu,0,,\n", bytes);
    k_stream_read<float2><<<grid, kB>>>(reinterpret_cast<float2 *>(a), bytes / 8, sink);
    printf("k_stream_read<float2>,%zu,0,,\n", bytes);
    k_stream_read<float><<<grid, kB>>>(reinterpret_cast<float *>(a), bytes / 4, sink);
    printf("k_stream_read<float>,%zu,0,,\n", bytes);
    k_stream_write<float4><<<grid, kB>>>(reinterpret_cast<float4 *>(b), bytes / 16, make_float4(1, 2, 3, 4));
    printf("k_stream_write<float4>,0,%zu,,\n", bytes);
    k_stream_write<float2><<<grid, kB>>>(reinterpret_cast<float2 *>(b), bytes / 8, make_float2(1, 2));
    printf("k_stream_write<float2>,0,%zu,,\n", bytes);
    k_gather8<<<grid, kB>>>(reinterpret_cast<float2 *>(a), bytes / 8, gathers, sink);
    printf("k_gather8,%zu,0,%zu,%zu\n", gathers * 8, gathers * 64, gathers * 128);
    // blur shape over 128 Mi vertices (1 GiB of values, 1 GiB of neighbour pairs): neighbours random over the whole array
    {
        const size_t n = (size_t)128 << 20;
        std::vector<int2> h(n);
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < n; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            h[i].x = (int)(s % n);
            h[i].y = (int)((s >> 32) % n);
        }
        int2 *nbr;
        CK(hipMalloc(&nbr, n * sizeof(int2)));
        CK(hipMemcpy(nbr, h.data(), n * sizeof(int2), hipMemcpyHostToDevice));
        k_blur_shape<<<(unsigned)((n + kB - 1) / kB), kB>>>(reinterpret_cast<float2 *>(a), reinterpret_cast<float2 *>(b), nbr, n);
        printf("k_blur_shape(random),%zu,%zu,%zu,%zu\n", n * 32, n * 8, n * 16 + 2 * n * 64, n * 16 + 2 * n * 128);
        for (size_t i = 0; i < n; ++i) { h[i].x = (int)(i > 0 ? i - 1 : 0); h[i].y = (int)(i + 1 < n ? i + 1 : i); }
        CK(hipMemcpy(nbr, h.data(), n * sizeof(int2), hipMemcpyHostToDevice));
        k_blur_shape<<<(unsigned)((n + kB - 1) / kB), kB>>>(reinterpret_cast<float2 *>(a), reinterpret_cast<float2 *>(b), nbr, n);
        printf("k_blur_shape(sequential),%zu,%zu,%zu,%zu\n", n * 16, n * 8, n * 16, n * 16);
        CK(hipFree(nbr));
    }
    CK(hipDeviceSynchronize());
    return 0;
}
