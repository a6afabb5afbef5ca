// bf_match.hip -- Tracking::BfMatch (src/Tracking.cc:1747-1766): brute-force Hamming 2-nearest-
// neighbour matching of two frames' ORB descriptors with the 0.6 ratio test.
//
// The reference calls cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, k = 2) and keeps query q
// when `match[0].distance < match[1].distance * 0.6` (float distance against a double product).
// OpenCV is absent from /root/reference (third-party, 3.x per CMakeLists.txt:31); its published
// behaviour is restated: cv::batchDistance keeps, per query, the K smallest distances in ascending
// order and inserts a candidate only where it is strictly smaller, scanning the train set in index
// order -- i.e. the two smallest (distance, train index) pairs in lexicographic order.  Integer-only
// apart from the one double comparison, so results are exact.  PARITY UNPINNED against a real OpenCV
// build; checked against the oracle's scalar restatement (oracle/lccrf_oracle.c: orc_bf_match).
//
// Layout: 16 lanes per query (2000 queries fill 125 workgroups instead of 32), each scanning a sixteenth of the train
// descriptors staged in LDS in tiles; candidates are (distance << 22 | train index) keys, so "lexicographically
// smaller" is one unsigned compare and the sixteen partial top-2 lists merge with four shuffles.
#include "engine.h"

#include <cstring>
#include <mutex>

namespace lccrf {

namespace {

constexpr int kLPQ = 16;                  // lanes per query
constexpr int kQPB = 16;                  // queries per block (x kLPQ lanes)
constexpr int kTile = 1024;               // train descriptors per LDS tile (32 KB)
constexpr unsigned kNone = 0xffffffffu;

__device__ __forceinline__ void top2_insert(unsigned key, unsigned &k0, unsigned &k1)
{
    if (key < k0) { k1 = k0; k0 = key; }
    else if (key < k1) k1 = key;
}

__global__ void __launch_bounds__(kQPB * kLPQ) k_bf_match(const uint4 *__restrict__ query, int n_query,
                                                        const uint4 *__restrict__ train, int n_train, double ratio,
                                                        int *__restrict__ out, int *__restrict__ n_matches)
{
    __shared__ uint4 tile[kTile * 2];
    const int tid = threadIdx.x, sub = tid & (kLPQ - 1);
    const int q = blockIdx.x * kQPB + tid / kLPQ;
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    if (q < n_query) { qa = query[2 * (size_t)q]; qb = query[2 * (size_t)q + 1]; }
    unsigned k0 = kNone, k1 = kNone;
    for (int t0 = 0; t0 < n_train; t0 += kTile) {
        const int nt = min(kTile, n_train - t0);
        __syncthreads();
        for (int i = tid; i < 2 * nt; i += kQPB * kLPQ) tile[i] = train[2 * (size_t)t0 + i];
        __syncthreads();
        for (int t = sub; t < nt; t += kLPQ) {
            const uint4 a = tile[2 * t], b = tile[2 * t + 1];
            const int d = __popc(qa.x ^ a.x) + __popc(qa.y ^ a.y) + __popc(qa.z ^ a.z) + __popc(qa.w ^ a.w) +
                          __popc(qb.x ^ b.x) + __popc(qb.y ^ b.y) + __popc(qb.z ^ b.z) + __popc(qb.w ^ b.w);
            top2_insert(((unsigned)d << 22) | (unsigned)(t0 + t), k0, k1);
        }
    }
    // merge the lanes of a query (xor 1, 2, 4, 8)
#pragma unroll
    for (int o = 1; o < kLPQ; o <<= 1) {
        const unsigned p0 = (unsigned)__shfl_xor((int)k0, o), p1 = (unsigned)__shfl_xor((int)k1, o);
        top2_insert(p0, k0, k1);
        top2_insert(p1, k0, k1);
    }
    if (q < n_query && sub == 0) {
        int m = -1;
        if (k1 != kNone) {                                // match.size() == 2
            const double d0 = (double)(float)(k0 >> 22), d1 = (double)(float)(k1 >> 22);
            if (d0 < d1 * ratio) m = (int)(k0 & 0x3fffffu);   // Tracking.cc:1755
        }
        out[q] = m;
        if (m >= 0) atomicAdd(n_matches, 1);
    }
}

// one pinned host arena (query | train | results | count) and one device buffer for the descriptors:
// one upload command per call, results written by the kernel straight into pinned host memory
struct Scratch {
    std::mutex m;
    unsigned char *host = nullptr, *dev = nullptr;
    size_t host_cap = 0, dev_cap = 0;
    int device = -1;
    hipStream_t stream = nullptr;
} g_bf;

}  // namespace

hipError_t run_bf_match(int device_id, int n_query, const uint8_t *desc_query, int n_train, const uint8_t *desc_train,
                        double ratio, int32_t *train_of_query_out, int32_t *n_matches_out)
{
    std::lock_guard<std::mutex> g(g_bf.m);
    hipError_t e;
    if (g_bf.device != device_id) {                       // the staging belongs to one device at a time
        if (g_bf.host) (void)hipHostFree(g_bf.host);
        if (g_bf.dev) (void)hipFree(g_bf.dev);
        if (g_bf.stream) (void)hipStreamDestroy(g_bf.stream);
        g_bf.host = g_bf.dev = nullptr;
        g_bf.host_cap = g_bf.dev_cap = 0;
        g_bf.stream = nullptr;
        g_bf.device = device_id;
    }
    if (!g_bf.stream && (e = hipStreamCreateWithFlags(&g_bf.stream, hipStreamNonBlocking)) != hipSuccess) return e;
    hipStream_t s = g_bf.stream;
    if (n_matches_out) *n_matches_out = 0;
    if (n_query <= 0) return hipSuccess;
    const size_t bq = (size_t)n_query * 32, bt = (size_t)n_train * 32, bo = ((size_t)n_query + 2) * sizeof(int);
    const size_t total_in = bq + bt, total = total_in + bo;
    if (g_bf.host_cap < total) {
        if (g_bf.host) (void)hipHostFree(g_bf.host);
        g_bf.host = nullptr;
        g_bf.host_cap = 0;
        if ((e = hipHostMalloc(reinterpret_cast<void **>(&g_bf.host), total + total / 2 + 4096, hipHostMallocDefault)) != hipSuccess) return e;
        g_bf.host_cap = total + total / 2 + 4096;
    }
    if (g_bf.dev_cap < total_in + 64) {
        if (g_bf.dev) (void)hipFree(g_bf.dev);
        g_bf.dev = nullptr;
        g_bf.dev_cap = 0;
        if ((e = hipMalloc(reinterpret_cast<void **>(&g_bf.dev), total_in + total_in / 2 + 4096)) != hipSuccess) return e;
        g_bf.dev_cap = total_in + total_in / 2 + 4096;
    }
    memcpy(g_bf.host, desc_query, bq);
    if (bt) memcpy(g_bf.host + bq, desc_train, bt);
    int *out = reinterpret_cast<int *>(g_bf.host + total_in);
    out[n_query] = 0;                                     // the match counter (host write, visible to the kernel launched below)
    if ((e = hipMemcpyAsync(g_bf.dev, g_bf.host, total_in, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    k_bf_match<<<dim3((n_query + kQPB - 1) / kQPB), dim3(kQPB * kLPQ), 0, s>>>(
        reinterpret_cast<const uint4 *>(g_bf.dev), n_query, reinterpret_cast<const uint4 *>(g_bf.dev + bq), n_train, ratio,
        out, out + n_query);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    memcpy(train_of_query_out, out, (size_t)n_query * sizeof(int));
    if (n_matches_out) *n_matches_out = out[n_query];
    return hipSuccess;
}

}  // namespace lccrf
