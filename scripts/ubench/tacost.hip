// tacost.hip -- what a vector-memory instruction costs the CU's texture-address / L1 path on gfx950, by access shape.
//
// Every workgroup hammers a small L2-resident table (1 MB per XCD slice: no HBM traffic after the first touch), so the
// time per instruction is what the TA / TCP pipeline charges, not bandwidth.  One block of 256 lanes per CU x 8 waves of
// residency (2048 blocks), each lane issuing `kIters` loads of the given shape with 4 independent accumulators.
//   ./tacost            prints cycles per wave-instruction per CU (= time * clock / (waves per CU * instructions))
// Used for the design of the C5 iteration kernels: gathers cost ~1 cycle per distinct 128-byte
// line and lane-group, not per byte.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kB = 256, kIters = 256;
constexpr size_t kTableBytes = 1 << 20;

// mode: how lane l of a wave picks its address in iteration it (all inside the 1 MB table)
//  0 same address for all lanes            1 consecutive 8 B (coalesced float2)      2 consecutive 16 B (float4, one load)
//  3 stride 128 B (64 distinct lines)      4 stride 32 B (16 lines)                  5 random 8 B over the table
//  6 like 3 but only 16 lanes active       7 like 3 but only 4 lanes active          8 dword at stride 28 B (AoS of 7 floats)
//  9 random 8 B, 22 of 64 lanes active
template <int MODE>
__global__ void __launch_bounds__(kB) k_ta(const char *__restrict__ table, float *sink, unsigned seed)
{
    const int lane = threadIdx.x & 63;
    const unsigned wid = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    unsigned h = wid * 2654435761u + seed;
    bool active = true;
    if (MODE == 6) active = lane < 16;
    if (MODE == 7) active = lane < 4;
    if (MODE == 9) active = (lane % 3) == 0;
    if (!active) return;
#pragma unroll 4
    for (int it = 0; it < kIters; ++it) {
        h = h * 1664525u + 1013904223u;
        const size_t base = (size_t)(h >> 8) % (kTableBytes / 2);       // wave-uniform base, multiple of nothing in particular
        size_t a;
        if (MODE == 0) a = (base & ~127ull);
        else if (MODE == 1) a = (base & ~511ull) + lane * 8;
        else if (MODE == 2) a = (base & ~1023ull) + lane * 16;
        else if (MODE == 3 || MODE == 6 || MODE == 7) a = (base & ~127ull) % (kTableBytes / 2 - 8192) + lane * 128;
        else if (MODE == 4) a = (base & ~127ull) % (kTableBytes / 2 - 2048) + lane * 32;
        else if (MODE == 5 || MODE == 9) { unsigned g = (h ^ (lane * 0x9E3779B9u)) * 2246822519u; a = (size_t)(g >> 8) % (kTableBytes / 8) * 8; }
        else a = (base & ~127ull) % (kTableBytes / 2 - 2048) + lane * 28;
        float v;
        if (MODE == 2) { const float4 t = *reinterpret_cast<const float4 *>(table + a); v = t.x + t.w; }
        else if (MODE == 8) v = *reinterpret_cast<const float *>(table + a);
        else { const float2 t = *reinterpret_cast<const float2 *>(table + a); v = t.x + t.y; }
        if ((it & 3) == 0) acc0 += v; else if ((it & 3) == 1) acc1 += v; else if ((it & 3) == 2) acc2 += v; else acc3 += v;
    }
    const float acc = acc0 + acc1 + acc2 + acc3;
    if (acc == 12345.678f) *sink = acc;
}

template <int MODE>
void run(const char *name, const char *table, float *sink)
{
    const int blocks = 256 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_ta<MODE><<<blocks, kB>>>(table, sink, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k_ta<MODE><<<blocks, kB>>>(table, sink, 7 + r);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    const double waves_per_cu = blocks * (kB / 64) / 256.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (waves_per_cu * kIters);
    printf("%-52s %8.3f ms  %7.1f cycles per wave-instruction per CU\n", name, ms, cyc);
}

int main()
{
    char *table;
    float *sink;
    CK(hipMalloc(&table, kTableBytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(table, 0, kTableBytes));
    run<0>("0 all lanes one address (8 B)", table, sink);
    run<1>("1 consecutive 8 B per lane (512 B, 4 lines)", table, sink);
    run<2>("2 consecutive 16 B per lane (1 KB, 8 lines)", table, sink);
    run<4>("4 8 B at stride 32 B (16 lines)", table, sink);
    run<3>("3 8 B at stride 128 B (64 lines)", table, sink);
    run<6>("6 8 B at stride 128 B, 16 lanes active (16 lines)", table, sink);
    run<7>("7 8 B at stride 128 B, 4 lanes active (4 lines)", table, sink);
    run<5>("5 8 B random over 1 MB (~64 lines)", table, sink);
    run<9>("9 8 B random over 1 MB, 22 lanes active", table, sink);
    run<8>("8 4 B at stride 28 B (14 lines)", table, sink);
    return 0;
}
