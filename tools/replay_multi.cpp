// replay_multi.cpp -- replay capture records (include/lccrf_record.h) across the GPUs of ONE node from a C++ host, on librccl
// directly: the C++ twin of `tools/replay.py` + `lc-crf-slam_amd/sharding.py` (BASELINE north_star:
// "Host stays C++ ... sharded over independent sequence frames ... RCCL only for the final label gather").
//
//   replay_multi frames.lccrfrec [--gpus G] [--batch B] [--single-workgroup] [--serial]
//
// Every frame is run exactly as the call site does (reference src/Tracking.cc:1919-1930): unary from the recorded initial
// labels and confidence, appearance kernel (vobservs / stdev_beta, verrors / stdev_alpha), smoothness kernel
// (coord2d / point2d_stdev), n_iterations mean-field iterations, MAP.
//   sharding     frame f -> GPU f mod G; one host thread per GPU (ncclCommInitAll); a frame is never split.  G defaults to
//                hipGetDeviceCount(), so the first 8-GPU box needs no flag.
//   per batch    B frames per GPU: lccrf_batch_set_inputs_host_async (pinned staging, upload on the batch's copy stream) ->
//                lccrf_batch_run (lattices + inference, one launch per frame) -> ONE ncclAllGather of
//                lccrf_batch_device_label_bits (the MAP labels one bit per point, written by the inference kernel itself) -- the
//                path's only collective; no other inter-GPU traffic.  Three batches are in flight per GPU (rank_main), so a
//                batch's kernels run under the previous batch's gather and the next batch's upload; --serial runs them one at a
//                time for the A/B.  The index arithmetic (frame -> rank, slot, round, gathered word) is include/lccrf_sharding.h.
//   check        thread 0 compares EVERY frame's gathered labels (its own and the other GPUs') with the labels the reference
//                recorded (ref_label); every thread compares its own frames' probabilities with ref_prob bit for bit.
// Prints one JSON line with the keys of tools/replay.py (+ gpus, batch, label_gathers); exit status 1 on any mismatch.
//
// Build:
//   hipcc -std=c++17 -O2 -Iinclude tools/replay_multi.cpp -o replay_multi lc-crf-slam_amd/liblccrf_hip.so -lrccl -lpthread \
//         -Wl,-rpath,$PWD/lc-crf-slam_amd -Wl,-rpath,/opt/rocm/lib
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "lccrf.h"
#include "lccrf_record.h"
#include "lccrf_sharding.h"

namespace {

struct Frame {
    lccrf_rec_frame_header h{};
    std::vector<float> vobservs, verrors, coord2d, ref_prob;
    std::vector<int16_t> init_label, ref_label;
    bool has_ref_label = false, has_ref_prob = false;
};

bool read_exact(FILE *f, void *p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }

// the reader of lc-crf-slam_amd/records.py in C++: version 1 and 2, sections skipped by their byte count; every size field
// is checked against what is left of the file before anything is allocated
bool read_records(const char *path, std::vector<Frame> &out, uint32_t &origin, std::string &err)
{
    FILE *f = fopen(path, "rb");
    if (!f) { err = "cannot open file"; return false; }
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    fseek(f, 0, SEEK_SET);
    lccrf_rec_file_header fh{};
    if (!read_exact(f, &fh, sizeof(fh)) || memcmp(fh.magic, LCCRF_REC_MAGIC, 8) != 0) { err = "not a record file"; return false; }
    if (fh.version != 1 && fh.version != LCCRF_REC_VERSION) { err = "unsupported version"; return false; }
    if (fh.header_bytes < sizeof(fh) || fh.frame_header_bytes < sizeof(lccrf_rec_frame_header) || (long)fh.header_bytes > fsize ||
        (long)fh.frame_header_bytes > fsize) { err = "bad header sizes"; return false; }
    origin = fh.version >= 2 ? fh.origin : 0u;
    fseek(f, fh.header_bytes, SEEK_SET);
    for (;;) {
        const long at = ftell(f);
        if (at >= fsize) break;
        Frame fr;
        std::vector<unsigned char> raw(fh.frame_header_bytes);
        if (!read_exact(f, raw.data(), raw.size())) { err = "truncated frame header"; return false; }
        memcpy(&fr.h, raw.data(), sizeof(fr.h));
        const size_t n = fr.h.n_points;
        const uint32_t fl = fr.h.flags;
        size_t bytes = n * (4 * 3 + 8 + 2) + ((fl & LCCRF_REC_HAS_MATCH_PROB) ? 8 * n : 0) + ((fl & LCCRF_REC_HAS_REF_LABEL) ? 2 * n : 0) +
                       ((fl & LCCRF_REC_HAS_REF_PROB) ? 8 * n : 0);
        if ((long)bytes > fsize - ftell(f)) { err = "truncated frame"; return false; }
        std::vector<float> vdepths(n);
        std::vector<double> match_prob;
        fr.vobservs.resize(n); fr.verrors.resize(n); fr.coord2d.resize(2 * n); fr.init_label.resize(n);
        bool ok = read_exact(f, fr.vobservs.data(), 4 * n) && read_exact(f, fr.verrors.data(), 4 * n) && read_exact(f, vdepths.data(), 4 * n) &&
                  read_exact(f, fr.coord2d.data(), 8 * n) && read_exact(f, fr.init_label.data(), 2 * n);
        if (ok && (fl & LCCRF_REC_HAS_MATCH_PROB)) { match_prob.resize(n); ok = read_exact(f, match_prob.data(), 8 * n); }
        if (ok && (fl & LCCRF_REC_HAS_REF_LABEL)) { fr.ref_label.resize(n); fr.has_ref_label = true; ok = read_exact(f, fr.ref_label.data(), 2 * n); }
        if (ok && (fl & LCCRF_REC_HAS_REF_PROB)) { fr.ref_prob.resize(2 * n); fr.has_ref_prob = true; ok = read_exact(f, fr.ref_prob.data(), 8 * n); }
        if (!ok) { err = "truncated frame"; return false; }
        const size_t size = fh.frame_header_bytes + bytes;
        fseek(f, (long)((8 - size % 8) % 8), SEEK_CUR);
        const uint32_t nsec = (fh.version >= 2 && (fl & LCCRF_REC_HAS_SECTIONS)) ? fr.h.n_sections : 0u;
        for (uint32_t s = 0; s < nsec; ++s) {                 // the neighbouring steps' sections: tools/replay.py replays those
            lccrf_rec_section_header sh{};
            if (!read_exact(f, &sh, sizeof(sh))) { err = "truncated section header"; return false; }
            if (sh.payload_bytes > (uint64_t)(fsize - ftell(f))) { err = "section larger than the file"; return false; }
            fseek(f, (long)(sh.payload_bytes + (8 - sh.payload_bytes % 8) % 8), SEEK_CUR);
        }
        out.push_back(std::move(fr));
    }
    return true;
}

struct Group { size_t first, count; int max_n; };      // consecutive frames with the same weights, confidence and iteration count

struct Totals {
    std::atomic<long> points{0}, checked_frames{0}, label_mismatches{0}, prob_mismatches{0}, dynamic_points{0}, gathers{0}, errors{0};
    std::atomic<int> max_dq_bits{0};                    // max |dQ| as float bits (non-negative floats order like their bit patterns)
    std::string error;
    std::atomic_flag error_lock = ATOMIC_FLAG_INIT;
    std::vector<ncclComm_t> *comms = nullptr;
    std::vector<double> rank_seconds;                   // wall clock of every rank's share
    void fail(const std::string &what)
    {
        if (errors.fetch_add(1) == 0) {
            while (error_lock.test_and_set()) {}
            error = what;
            error_lock.clear();
            if (comms)                                      // the other ranks may be waiting in a collective this rank will never join
                for (ncclComm_t c : *comms) (void)ncclCommAbort(c);
        }
    }
};

#define TRY_LCCRF(expr)                                                                        \
    do {                                                                                       \
        const int rc_ = (expr);                                                                \
        if (rc_ != LCCRF_OK) { tot.fail(std::string(#expr) + ": " + lccrf_last_error()); return; } \
    } while (0)
#define TRY_HIP(expr)                                                                          \
    do {                                                                                       \
        const hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) { tot.fail(std::string(#expr) + ": " + hipGetErrorString(e_)); return; } \
    } while (0)
#define TRY_NCCL(expr)                                                                         \
    do {                                                                                       \
        const ncclResult_t r_ = (expr);                                                        \
        if (r_ != ncclSuccess) { tot.fail(std::string(#expr) + ": " + ncclGetErrorString(r_)); return; } \
    } while (0)

// One GPU's share of the replay.  Every rank walks the same groups and the same number of rounds per group (a rank whose
// share of the last round is short pads with empty frames), so the collectives line up.
//
// Three rounds are in flight per rank, each on a batch handle and that handle's own stream (slot = round mod 3):
//     iteration t:   stage + upload + launch round t        lccrf_batch_set_inputs_host_async -> lccrf_batch_run
//                    settle + gather round t-1              lccrf_batch_synchronize: the host waits for round t-1's OWN streams (its kernel
//                                                           has been running under the staging of round t) and re-runs the frames the
//                                                           one-launch kernel could not take; then ncclAllGather of the label bits, its
//                                                           copy to the host and the probabilities' download are queued, not waited for
//                    check round t-2                        wait for ITS copies only, compare on the host
// so round t's kernels run under round t-1's gather and round t+1's staging and upload: what an 8-GPU run then measures is the
// split, not a serialised upload -> launch -> synchronise -> gather -> download per batch.
struct Slot {
    lccrf_batch_handle b = nullptr;
    hipStream_t stream = nullptr;
    const uint64_t *d_bits = nullptr;
    uint64_t *d_all = nullptr;
    uint64_t *h_all = nullptr;                          // pinned
    bool need_prob = false;
};

void rank_main(int rank, int G, int B, bool single_wg, bool serial, const std::vector<Frame> &frames, const std::vector<Group> &groups,
               ncclComm_t comm, Totals &tot)
{
    TRY_HIP(hipSetDevice(rank));
    constexpr int kSlots = 3;
    const auto t_rank0 = std::chrono::steady_clock::now();
    for (const Group &g : groups) {
        const Frame &f0 = frames[g.first];
        const int maxn = std::max(g.max_n, 1);
        lccrf_batch_desc desc{};
        desc.max_frames = B;
        desc.max_points = maxn;
        desc.n_labels = 2;
        desc.n_kernels = 2;
        desc.feat_dims[0] = desc.feat_dims[1] = 2;
        desc.weights[0] = f0.h.w1;                          // Tracking.cc:1923-1927: appearance kernel first, then smoothness
        desc.weights[1] = f0.h.w2;
        Slot slots[kSlots];
        int words = 0;
        for (Slot &sl : slots) {
            TRY_LCCRF(lccrf_batch_create(&sl.b, rank, &desc));
            // the batch's OWN stream carries its kernels, the gather and the copy: streams created with the handles, one after the other,
            // land on different hardware queues; caller-made ones may share one and then run the slots' kernels one after the other
            // (include/lccrf.h: lccrf_batch_get_stream)
            void *own = nullptr;
            TRY_LCCRF(lccrf_batch_get_stream(sl.b, &own));
            sl.stream = static_cast<hipStream_t>(own);
            if (single_wg) TRY_LCCRF(lccrf_batch_set_option(sl.b, LCCRF_OPT_SINGLE_WORKGROUP, 1));
            TRY_LCCRF(lccrf_batch_device_label_bits(sl.b, &sl.d_bits, &words));
        }
        const size_t per_rank = (size_t)B * words;
        for (Slot &sl : slots) {
            TRY_HIP(hipMalloc(reinterpret_cast<void **>(&sl.d_all), per_rank * G * sizeof(uint64_t)));
            TRY_HIP(hipHostMalloc(reinterpret_cast<void **>(&sl.h_all), per_rank * G * sizeof(uint64_t), hipHostMallocDefault));
        }
        // (one set of staging arrays: lccrf_batch_set_inputs_host_async has copied them out when it returns)
        std::vector<float> app((size_t)B * maxn * 2), smooth((size_t)B * maxn * 2);
        std::vector<int16_t> label((size_t)B * maxn);
        std::vector<int32_t> npts(B);
        const size_t rounds = lccrf_shard_rounds(g.count, G, B);
        const float conf[2] = {f0.h.confidence, f0.h.confidence};      // setUnaryEnergyFromLabel(label, mConf), Tracking.cc:1921
        auto frame_of = [&](size_t t, int r, int i) -> long {
            const long k = lccrf_shard_frame(t, r, i, G, B, g.count);
            return k < 0 ? -1 : (long)g.first + k;
        };

        auto launch = [&](size_t t) {                       // stage + upload + launch round t
            Slot &sl = slots[t % kSlots];
            std::fill(app.begin(), app.end(), 0.0f);
            std::fill(smooth.begin(), smooth.end(), 0.0f);
            std::fill(label.begin(), label.end(), (int16_t)-1);
            sl.need_prob = false;
            for (int i = 0; i < B; ++i) {
                const long fi = frame_of(t, rank, i);
                npts[i] = 0;
                if (fi < 0) continue;
                const Frame &fr = frames[fi];
                const int n = (int)fr.h.n_points;
                npts[i] = n;
                sl.need_prob |= fr.has_ref_prob;
                float *a = &app[(size_t)i * maxn * 2], *s = &smooth[(size_t)i * maxn * 2];
                for (int p = 0; p < n; ++p) {               // pairwise3d.h:41-44 and :64-66: fp32 divisions on the host, as the reference
                    a[2 * p + 0] = fr.vobservs[p] / fr.h.stdev_beta;
                    a[2 * p + 1] = fr.verrors[p] / fr.h.stdev_alpha;
                    s[2 * p + 0] = fr.coord2d[2 * p + 0] / fr.h.point2d_stdev;
                    s[2 * p + 1] = fr.coord2d[2 * p + 1] / fr.h.point2d_stdev;
                }
                memcpy(&label[(size_t)i * maxn], fr.init_label.data(), (size_t)n * sizeof(int16_t));
            }
            const float *feats[2] = {app.data(), smooth.data()};
            TRY_LCCRF(lccrf_batch_set_inputs_host_async(sl.b, B, npts.data(), nullptr, label.data(), conf, feats, 0));
            TRY_LCCRF(lccrf_batch_run(sl.b, (int)f0.h.n_iterations, 1, 1.0f, sl.stream));
        };
        auto gather = [&](size_t t) {                       // settle round t (a wait on ITS streams only), then gather: queued, not waited for
            Slot &sl = slots[t % kSlots];
            TRY_LCCRF(lccrf_batch_synchronize(sl.b));       // (settles frames the one-launch kernel could not take: the bits are complete behind this)
            // the path's one collective: every rank's bit-packed labels to every rank
            TRY_NCCL(ncclAllGather(sl.d_bits, sl.d_all, per_rank, ncclUint64, comm, sl.stream));
            TRY_HIP(hipMemcpyAsync(sl.h_all, sl.d_all, per_rank * G * sizeof(uint64_t), hipMemcpyDeviceToHost, sl.stream));
            if (sl.need_prob) TRY_LCCRF(lccrf_batch_download_async(sl.b, LCCRF_DOWNLOAD_PROBABILITY));
            if (rank == 0) tot.gathers.fetch_add(1);
        };
        auto check = [&](size_t t) {                        // wait for round t's copies and compare
            Slot &sl = slots[t % kSlots];
            TRY_HIP(hipStreamSynchronize(sl.stream));
            const float *prob = nullptr;
            if (sl.need_prob) TRY_LCCRF(lccrf_batch_wait_download(sl.b, nullptr, nullptr, nullptr, &prob));
            // every rank checks the probabilities of its own frames ...
            for (int i = 0; i < B; ++i) {
                const long fi = frame_of(t, rank, i);
                if (fi < 0 || !frames[fi].has_ref_prob) continue;
                const Frame &fr = frames[fi];
                const float *q = &prob[(size_t)i * maxn * 2];
                long bad = 0;
                for (size_t p = 0; p < fr.h.n_points; ++p) {
                    bad += memcmp(q + 2 * p, &fr.ref_prob[2 * p], 8) != 0;
                    for (int l = 0; l < 2; ++l) {
                        float d = q[2 * p + l] - fr.ref_prob[2 * p + l];
                        d = d < 0 ? -d : d;
                        int bits;
                        memcpy(&bits, &d, 4);
                        int cur = tot.max_dq_bits.load();
                        while (bits > cur && !tot.max_dq_bits.compare_exchange_weak(cur, bits)) {}
                    }
                }
                tot.prob_mismatches.fetch_add(bad);
            }
            // ... and rank 0 the gathered labels of EVERY frame of the round, whichever GPU produced them
            if (rank == 0) {
                for (int r = 0; r < G; ++r)
                    for (int i = 0; i < B; ++i) {
                        const long fi = frame_of(t, r, i);
                        if (fi < 0) continue;
                        const Frame &fr = frames[fi];
                        const uint64_t *w = &sl.h_all[lccrf_gather_word(r, i, B, words)];
                        long dyn = 0, bad = 0;
                        for (size_t p = 0; p < fr.h.n_points; ++p) {
                            const int lab = (int)((w[p >> 6] >> (p & 63)) & 1u);
                            dyn += lab == 0;
                            if (fr.has_ref_label) bad += lab != fr.ref_label[p];
                        }
                        tot.points.fetch_add((long)fr.h.n_points);
                        tot.dynamic_points.fetch_add(dyn);
                        if (fr.has_ref_label) {
                            tot.checked_frames.fetch_add(1);
                            tot.label_mismatches.fetch_add(bad);
                        }
                    }
            }
        };
        if (serial) {                                       // --serial: one round at a time (the A/B of the pipeline)
            for (size_t t = 0; t < rounds && !tot.errors.load(); ++t) { launch(t); gather(t); check(t); }
        } else {
            for (size_t t = 0; t < rounds + 2 && !tot.errors.load(); ++t) {
                if (t < rounds) launch(t);
                if (t >= 1 && t - 1 < rounds) gather(t - 1);
                if (t >= 2) check(t - 2);
            }
        }
        for (Slot &sl : slots) {
            if (sl.stream) (void)hipStreamSynchronize(sl.stream);
            (void)hipFree(sl.d_all);
            (void)hipHostFree(sl.h_all);
            lccrf_batch_destroy(sl.b);                   // (takes its stream with it)
        }
    }
    tot.rank_seconds[rank] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_rank0).count();
}

}  // namespace

int main(int argc, char **argv)
{
    const char *path = nullptr;
    int G = 0, B = 256;                   // (from 256 frames per batch the one-launch kernel runs two frames per CU: csrc/frame_lean.hip)
    bool single_wg = false, serial = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) G = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--batch") && i + 1 < argc) B = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--single-workgroup")) single_wg = true;
        else if (!strcmp(argv[i], "--serial")) serial = true;
        else if (argv[i][0] != '-' && !path) path = argv[i];
        else { fprintf(stderr, "usage: %s frames.lccrfrec [--gpus G] [--batch B] [--single-workgroup] [--serial]\n", argv[0]); return 2; }
    }
    if (!path || B < 1) { fprintf(stderr, "usage: %s frames.lccrfrec [--gpus G] [--batch B] [--single-workgroup] [--serial]\n", argv[0]); return 2; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { fprintf(stderr, "no HIP device: this tool has no CPU path\n"); return 2; }
    if (G <= 0) G = ndev;                                   // every GPU of the node by default
    if (G > ndev) { fprintf(stderr, "--gpus %d but the node has %d device(s)\n", G, ndev); return 2; }

    std::vector<Frame> frames;
    uint32_t origin = 0;
    std::string err;
    if (!read_records(path, frames, origin, err)) { fprintf(stderr, "%s: %s\n", path, err.c_str()); return 2; }
    std::vector<Group> groups;
    for (size_t i = 0; i < frames.size();) {
        Group g{i, 0, 0};
        const lccrf_rec_frame_header &h0 = frames[i].h;
        while (i < frames.size() && frames[i].h.w1 == h0.w1 && frames[i].h.w2 == h0.w2 && frames[i].h.confidence == h0.confidence &&
               frames[i].h.n_iterations == h0.n_iterations) {
            g.max_n = std::max(g.max_n, (int)frames[i].h.n_points);
            ++g.count;
            ++i;
        }
        groups.push_back(g);
    }

    std::vector<int> devs(G);
    for (int r = 0; r < G; ++r) devs[r] = r;
    std::vector<ncclComm_t> comms(G);
    const ncclResult_t nr = ncclCommInitAll(comms.data(), G, devs.data());
    if (nr != ncclSuccess) { fprintf(stderr, "ncclCommInitAll: %s\n", ncclGetErrorString(nr)); return 2; }

    Totals tot;
    tot.comms = &comms;
    tot.rank_seconds.assign(G, 0.0);
    int comm_ranks = 0;                                  // what RCCL itself says the communicator spans
    (void)ncclCommCount(comms[0], &comm_ranks);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> threads;
    for (int r = 0; r < G; ++r) threads.emplace_back(rank_main, r, G, B, single_wg, serial, std::cref(frames), std::cref(groups), comms[r], std::ref(tot));
    for (auto &t : threads) t.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!tot.errors.load())
        for (int r = 0; r < G; ++r) (void)ncclCommDestroy(comms[r]);

    if (tot.errors.load()) {
        fprintf(stderr, "replay failed: %s\n", tot.error.c_str());
        return 2;
    }
    float max_dq;
    const int mb = tot.max_dq_bits.load();
    memcpy(&max_dq, &mb, 4);
    std::string rank_list;
    for (int r = 0; r < G; ++r) {
        char buf[32];
        snprintf(buf, sizeof(buf), "%s%.6f", r ? ", " : "", tot.rank_seconds[r]);
        rank_list += buf;
    }
    const char *base = strrchr(path, '/');
    printf("{\"file\": \"%s\", \"frames\": %zu, \"points\": %ld, \"checked_frames\": %ld, \"label_mismatches\": %ld, \"prob_mismatches\": %ld, "
           "\"max_abs_dQ\": %.9g, \"dynamic_points\": %ld, \"origin\": \"%s\", \"gpus\": %d, \"batch\": %d, \"label_gathers\": %ld, "
           "\"label_gather\": \"ncclAllGather of the bit-packed labels (uint64 words), one per batch, on the batch's stream\", "
           "\"sharding\": \"frame f -> GPU f mod G, one host thread per GPU\", \"pipeline\": \"%s\", \"rccl_comm_ranks\": %d, "
           "\"rank_seconds\": [%s], \"frames_per_s_host_to_host\": %.6g}\n",
           base ? base + 1 : path, frames.size(), tot.points.load(), tot.checked_frames.load(), tot.label_mismatches.load(),
           tot.prob_mismatches.load(), (double)max_dq, tot.dynamic_points.load(),
           origin == LCCRF_REC_ORIGIN_SYNTHETIC ? "synthetic (outputs from this repository's restatements: pins nothing)" : "reference", G, B,
           tot.gathers.load(),
           serial ? "serial: upload, launch, settle, gather, download and check one round at a time"
                  : "three rounds in flight per rank: round t uploads and launches while round t-1 gathers and round t-2 is checked",
           comm_ranks, rank_list.c_str(), secs > 0 ? frames.size() / secs : 0.0);
    return (tot.label_mismatches.load() || tot.prob_mismatches.load()) ? 1 : 0;
}
