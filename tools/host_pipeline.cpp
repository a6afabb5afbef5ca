// host_pipeline.cpp -- SURVEY 8(d)'s end to end from a C++ host: batches of frames from HOST arrays to HOST label bits, pipelined.
// The reference pays its per-frame cost host to host (src/Tracking.cc:1919-1930); a replay with many frames in flight has to keep
// the PCIe link busy under the kernels.  Per batch, on `handles` batch handles used round-robin:
//     lccrf_batch_set_inputs_host_async  ->  lccrf_batch_run  ->  lccrf_batch_download_async(label bits)
// and, `handles` batches later, lccrf_batch_wait_download of that handle.  Product code only (no oracle): bench.py compiles and runs
// it for the `end_to_end.host_to_host` record and compares the label bits it dumps with the synchronous path's.
//
//   host_pipeline <inputs.bin> <frames_per_batch> <batches> <handles> <pageable|pinned|serial> [bits_out.bin|-] [copy_threads]
//       inputs: int32 n_distinct, int32 N, int32 n_iter, float w1, float w2, float conf, then per distinct frame
//               float app[N][2], float smooth[N][2], int16 label[N]   (features as the reference's factories leave them)
//       a batch is the distinct frames tiled to frames_per_batch; `serial` = one handle, lccrf_batch_set_inputs_host + run + map
//   prints one JSON object: frames/s, upload GB/s, seconds; bits_out receives the last batch's label bits (uint64 [B][words])
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lccrf.h"

#define CHECK(expr)                                                                  \
    do {                                                                             \
        const int rc_ = (expr);                                                      \
        if (rc_ != LCCRF_OK) {                                                       \
            printf("{\"error\": \"%s: %s\"}\n", #expr, lccrf_last_error());         \
            return 4;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) return 2;
    int D = 0, N = 0, n_iter = 0;
    float w1 = 0, w2 = 0, conf1 = 0;
    if (fread(&D, 4, 1, fp) != 1 || fread(&N, 4, 1, fp) != 1 || fread(&n_iter, 4, 1, fp) != 1 || fread(&w1, 4, 1, fp) != 1 ||
        fread(&w2, 4, 1, fp) != 1 || fread(&conf1, 4, 1, fp) != 1 || D < 1 || N < 1) return 2;
    const int B = atoi(argv[2]), nb = atoi(argv[3]), H = atoi(argv[4]);
    const std::string mode = argv[5];
    const bool pinned = mode == "pinned", serial = mode == "serial";
    if (B < 1 || nb < 1 || H < 1 || H > 16) return 2;
    std::vector<float> dapp((size_t)D * N * 2), dsm((size_t)D * N * 2);
    std::vector<int16_t> dlab((size_t)D * N);
    for (int f = 0; f < D; ++f)
        if (fread(&dapp[(size_t)f * N * 2], 8, N, fp) != (size_t)N || fread(&dsm[(size_t)f * N * 2], 8, N, fp) != (size_t)N ||
            fread(&dlab[(size_t)f * N], 2, N, fp) != (size_t)N) return 2;
    fclose(fp);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { printf("{\"error\": \"no HIP device\"}\n"); return 3; }

    // the caller's arrays: ordinary (pageable) memory, or pinned
    const size_t nf = (size_t)B * N * 2, nl = (size_t)B * N;
    float *app = nullptr, *sm = nullptr;
    int16_t *lab = nullptr;
    if (pinned) {
        if (hipHostMalloc((void **)&app, nf * 4, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&sm, nf * 4, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&lab, nl * 2, hipHostMallocDefault) != hipSuccess) return 4;
    } else {
        app = (float *)malloc(nf * 4); sm = (float *)malloc(nf * 4); lab = (int16_t *)malloc(nl * 2);
    }
    std::vector<int32_t> npts(B, N);
    for (int f = 0; f < B; ++f) {
        memcpy(app + (size_t)f * N * 2, &dapp[(size_t)(f % D) * N * 2], (size_t)N * 8);
        memcpy(sm + (size_t)f * N * 2, &dsm[(size_t)(f % D) * N * 2], (size_t)N * 8);
        memcpy(lab + (size_t)f * N, &dlab[(size_t)(f % D) * N], (size_t)N * 2);
    }
    lccrf_batch_desc desc{};
    desc.max_frames = B; desc.max_points = N; desc.n_labels = 2; desc.n_kernels = 2;
    desc.feat_dims[0] = desc.feat_dims[1] = 2;
    desc.weights[0] = w1; desc.weights[1] = w2;
    const int nh = serial ? 1 : H;
    std::vector<lccrf_batch_handle> hs(nh, nullptr);
    for (auto &h : hs) CHECK(lccrf_batch_create(&h, 0, &desc));
    if (argc > 7)
        for (auto &h : hs) CHECK(lccrf_batch_set_option(h, LCCRF_OPT_COPY_THREADS, atoi(argv[7])));
    const float conf[2] = {conf1, conf1};
    const float *feats[2] = {app, sm};
    const int words = (N + 63) / 64;
    std::vector<uint64_t> last((size_t)B * words);
    std::vector<int16_t> map_host(serial ? nl : 0);
    using Clock = std::chrono::steady_clock;

    auto pump = [&](int n) -> int {
        if (serial) {
            for (int i = 0; i < n; ++i) {
                CHECK(lccrf_batch_set_inputs_host(hs[0], B, npts.data(), nullptr, lab, conf, feats));
                CHECK(lccrf_batch_run(hs[0], n_iter, 1, 1.0f, nullptr));
                CHECK(lccrf_batch_get_map_host(hs[0], map_host.data()));
            }
            return 0;
        }
        for (int i = 0; i < n + nh; ++i) {
            lccrf_batch_handle h = hs[i % nh];
            if (i >= nh) {
                const uint64_t *bits = nullptr;
                int w = 0;
                CHECK(lccrf_batch_wait_download(h, &bits, &w, nullptr, nullptr));
                if (i - nh == n - 1) memcpy(last.data(), bits, last.size() * 8);
            }
            if (i < n) {
                CHECK(lccrf_batch_set_inputs_host_async(h, B, npts.data(), nullptr, lab, conf, feats, pinned ? LCCRF_HOST_PINNED : 0));
                CHECK(lccrf_batch_run(h, n_iter, 1, 1.0f, nullptr));
                CHECK(lccrf_batch_download_async(h, LCCRF_DOWNLOAD_LABEL_BITS));
            }
        }
        return 0;
    };
    if (int rc = pump(std::max(2 * nh, 3))) return rc;       // staging allocated, kernels loaded
    const auto t0 = Clock::now();
    if (int rc = pump(nb)) return rc;
    const double secs = std::chrono::duration<double>(Clock::now() - t0).count();
    if (serial)                                              // the same bits from the int16 labels
        for (int f = 0; f < B; ++f)
            for (int w = 0; w < words; ++w) {
                uint64_t m = 0;
                for (int i = 0; i < 64 && w * 64 + i < N; ++i) m |= (uint64_t)(map_host[(size_t)f * N + w * 64 + i] == 1) << i;
                last[(size_t)f * words + w] = m;
            }
    if (argc > 6 && strcmp(argv[6], "-") != 0) {
        FILE *fo = fopen(argv[6], "wb");
        if (fo) { fwrite(last.data(), 8, last.size(), fo); fclose(fo); }
    }
    const double up_bytes = (double)nb * ((double)nf * 8 + (double)nl * 2 + (double)B * 4);
    printf("{\"mode\": \"%s\", \"frames_per_batch\": %d, \"batches\": %d, \"handles\": %d, \"seconds\": %.6f, \"frames_per_s\": %.6g, "
           "\"upload_GBs\": %.4f, \"download_bytes_per_frame\": %d}\n",
           mode.c_str(), B, nb, nh, secs, (double)nb * B / secs, up_bytes / secs / 1e9, serial ? N * 2 : words * 8);
    for (auto h : hs) lccrf_batch_destroy(h);
    return 0;
}
