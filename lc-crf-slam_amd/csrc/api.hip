// api.hip -- the C-ABI of include/lccrf.h on top of the two HIP engines.
//
// Host-side glue only: argument checks, device memory, pinned staging, launch order.
// All arithmetic of the path runs in the kernels (stream_engine.hip / fused_engine.hip);
// the only numbers computed here are the per-kernel constants of
// permutohedral_cpu.h:249,282-285,681 and the 2L+1 unary energies of densecrf3d.h:109-115,
// which the reference also computes once, outside its loops.
#include "engine.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <vector>

using namespace lccrf;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? LCCRF_E_NOMEM : LCCRF_E_HIP, "%s: %s",  \
                        #expr, hipGetErrorString(e_));                                       \
    } while (0)

// spin-wait hint of the host polls below (the library must build on non-x86 ROCm hosts too)
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#endif
}

int next_pow2(long v)
{
    long p = 16;
    while (p < v) p <<= 1;
    return (int)p;
}

int use_device(int device_id)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(LCCRF_E_NO_DEVICE, "no HIP device (hipGetDeviceCount: %s); this library has no CPU fallback",
                    hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(LCCRF_E_INVALID, "device_id %d out of range [0,%d)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    return LCCRF_OK;
}

// Owns every device / pinned allocation of one engine.
struct Arena {
    std::vector<void *> dev, pinned;
    // Allocations are zeroed ON THE STREAM THE OWNER'S WORK IS ISSUED ON (StreamScope retargets this together with
    // Engine::stream while a batch call runs on a caller's stream: a lazily allocated buffer zeroed on the engine's
    // own stream would be unordered with the kernels of that call -- ADVICE r3).  hipMemset on the null stream is
    // asynchronous and the engine streams are non-blocking, i.e. unordered with the null stream: a null-stream
    // memset could land after the first copy into the new buffer (seen as a one-in-hundreds lattice
    // of all-zero features).
    hipStream_t stream = nullptr;
    template <typename T>
    int alloc(T **out, size_t count, bool zero = true)
    {
        void *p = nullptr;
        const size_t bytes = (count ? count : 1) * sizeof(T);
        HIP_TRY(hipMalloc(&p, bytes));
        dev.push_back(p);
        if (zero) {
            HIP_TRY(hipMemsetAsync(p, 0, bytes, stream));
            if (!stream) HIP_TRY(hipStreamSynchronize(nullptr));
        }
        *out = static_cast<T *>(p);
        return LCCRF_OK;
    }
    template <typename T>
    int alloc_pinned(T **out, size_t count)
    {
        void *p = nullptr;
        // fine-grained (coherent) on purpose, not left to HIP_HOST_COHERENT: kernels publish status words, done words and labels
        // here and the host reads them while the kernel is still running
        HIP_TRY(hipHostMalloc(&p, (count ? count : 1) * sizeof(T), hipHostMallocCoherent));
        pinned.push_back(p);
        *out = static_cast<T *>(p);
        return LCCRF_OK;
    }
    void release()
    {
        for (void *p : dev) (void)hipFree(p);
        for (void *p : pinned) (void)hipHostFree(p);
        dev.clear();
        pinned.clear();
    }
};

struct KernelState {
    KernelDev dev{};
    float *feat_own = nullptr;    // device copy of host-provided features
    float *feat_stage = nullptr;  // pinned staging for the object API
    int maxV = 0;                 // max over frames of V once known, else Epad
    int maxRow = 0;               // max over frames of the longest CSR row
};

// One CRF problem set: F frames x maxN points x L labels, K kernels.
struct Engine {
    int device = 0;
    hipStream_t stream = nullptr;
    Arena mem;
    int Fcap = 0, F = 0, maxN = 0, maxNpad = 0, L = 0;
    int activeN = 0;                   // largest frame of the current inputs if known on the host, else 0
    CrfDev crf{};
    int *npoints_own = nullptr;
    float *unary_own = nullptr;
    int16_t *label_own = nullptr;
    int *V_host = nullptr;             // pinned [K][Fcap]
    int *row_host = nullptr;           // pinned [K][Fcap]
    int *long_host = nullptr;          // pinned [K][Fcap]: rows beyond kLongRowMin entries per frame (KernelDev::longcnt)
    std::vector<KernelState> kernels;
    std::vector<KernelState> spare;    // allocated lattices of earlier uses of this (recycled) engine
    std::vector<KernelDev> kdevs;      // contiguous copy handed to the launchers
    std::vector<int> maxV, maxRow;
    bool unary_set = false, built = false, sizes_known = false, started = false;
    int built_upto = 0;                // kernels [0, built_upto) have their lattice
    int engine_pref = 0, engine_used = 1;   // engine_used is REPORT-only state (lccrf_batch_get_engine): 1 streaming, 2 fused, 3 one launch per frame
    int sized_engine = 1;              // what learn_sizes() chose for inference on lattices that sit in HBM (1 or 2)
    int last_with_map = 0;             // did the last inference produce MAP labels?
    size_t fused_lds = 0;
    int fused_shape = 0;               // report only: what the last fused inference launched (launch_inference_fused)
    LeanPrep lean_prep;                // prepared launch records of the two-frames-per-CU inference kernel (engine.h)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // build begin/end, inference begin/end
    hipEvent_t ev_order = nullptr;     // orders a caller-supplied stream against the engine's own (StreamScope)
    int *npoints_bad = nullptr;        // pinned: set by the validation kernel when a bound n_points[f] is outside [0, maxN]
    float *io_a = nullptr, *io_b = nullptr;   // device [Fcap][maxN][L] each: caller buffers of apply / expAndNormalize / ... (lazy)
    int16_t *io_map = nullptr;

    int need_io()
    {
        if (io_a) return LCCRF_OK;
        const size_t nl = (size_t)Fcap * maxN * L;
        int rc;
        if ((rc = mem.alloc(&io_a, nl))) return rc;
        if ((rc = mem.alloc(&io_b, nl))) return rc;
        if ((rc = mem.alloc(&io_map, (size_t)Fcap * maxN))) return rc;
        HIP_TRY(hipStreamSynchronize(stream));
        return LCCRF_OK;
    }
    // Late-bound fused inference (object API): queued right behind the build, before the host has seen
    // the lattice sizes; `late_status` (pinned) tells afterwards whether the frame fitted.
    int *late_status = nullptr;
    // setUnaryEnergyFromLabel is deferred: the labels wait in pinned memory and either the late-bound
    // fused kernel derives the energies itself or ensure_unary() launches the small kernel.
    bool unary_deferred = false;
    const int16_t *deferred_label = nullptr;
    UnaryTable deferred_tbl{};
    bool late_ok = false;              // set by the object API: single frame, automatic engine choice
    bool late_pending = false;
    // Called by resolve_late() right behind a re-run of flagged frames (before its final synchronisation): the owner of host-side
    // copies of the results (the batch's download pipe) re-queues them, whichever entry point happened to settle the run.
    int (*after_rerun)(void *) = nullptr;
    void *after_rerun_ctx = nullptr;
    // Per-frame fallback of the one-launch kernel: k_frame leaves 1 in frame_status[f] for a frame whose lattices do not
    // fit its LDS plan; resolve_late() gathers exactly those frames into the sub-engine `fb`, runs the two-kernel path
    // there and scatters Q / labels / V back.
    int *frame_status = nullptr;       // device [Fcap]
    int *frame_status_host = nullptr;  // pinned [Fcap]
    int *fb_list = nullptr;            // device [Fcap]: frames being re-run
    int *fb_list_host = nullptr;       // pinned [Fcap]
    Engine *fb = nullptr;
    int fallback_frames = 0;           // frames the last one-launch run had to re-run (report only)
    bool frame_small_ok = true;        // small frames may run as 512-lane workgroups (two per CU); turned off for this engine when
    bool frame_small_used = false;     //   more than 1/8 of a batch's frames did not fit that plan
    bool frame_lean_ok = true;         // ... and so may batches of full-size frames (frame_lean.hip), under the same rule
    bool frame_lean_used = false;
    unsigned char *lean_rec = nullptr; //   [lean_rec_frames][kLeanRecBytes] per-point records of that kernel (lazy, sized for the frames bound)
    int lean_rec_frames = 0;
    static constexpr int kDualMaxFrames = 64;    // (measured, `scripts/small_batch_latency.py`: 4 frames 98 -> 86 us, 32 frames 93 -> 83 us, 128 frames 112 -> 114 us)
    bool opt_single_wg = false;        // LCCRF_OPT_SINGLE_WORKGROUP: never the two-workgroup form (lccrf_set_option)
    int opt_vertex_order = 0;          // LCCRF_OPT_VERTEX_ORDER: 0 automatic (= on), 1 on, 2 off -- locality mode's sorted build
    bool vorder_on = false;            // ... as decided by the last build of every kernel
    bool vorder_broken = false;        // a frame's code space overflowed (*sort.vbad): this engine keeps to the hash build
    int *ndist_host = nullptr;         // pinned [K][kNdistAxes]: the sorted build's largest neighbour distances (KernelDev::ndist)
    int *tbl_bad = nullptr;            // pinned [2]: what the sorted build found wrong with its tables (KernelDev::tbl_bad)
    unsigned *dual_area = nullptr;     // hand-off area of the two-workgroup form of the frame kernel (batches of up to kDualMaxFrames frames)
    unsigned dual_epoch = 0;
    // object API: the frame kernel's last act is a store of `done_epoch` into this pinned word, behind its results; the host
    // polls it instead of waiting for the runtime's completion signal (which arrives a few microseconds later)
    unsigned *done_word = nullptr;
    unsigned done_epoch = 0;
    bool done_armed = false;
    // ... and with_map launches of the object API do not even wait for that word: the host pre-fills the pinned label array
    // with -1 and takes each label as it arrives (the kernel's last stores; no fence, no acknowledgement round trip)
    int16_t *map_host = nullptr;       // the handle's pinned label array (= crf.map), or null
    bool labels_armed = false;
    bool idle_by_done = false;         // the last call on the handle was a getMap() that saw the done word: nothing is in flight
    bool idle_needs_done = false;      //   ... or every label (the stores right before the done word): the word itself is still to come
    bool park_check = false;           // parked with that word still unseen: ensure_parked_idle() settles it for the next user
    int late_iter = 0, late_map = 0;
    float late_relax = 1.0f;
    bool timed_build = false, timed_inf = false;
    bool event_timing = true;          // LCCRF_OPT_EVENT_TIMING: HIP events around every build / inference / run of the batch API
    // Locality mode (batch engines, frames of >= kPermMinPoints points on the streaming engine): the lattices are built
    // with the points in an internal Z-order of their lattice cells (stream_engine.hip: launch_sort_points); Q, next and
    // the unaries of the iteration live in that order (Qp, unary_p / unary_own) and Q is un-permuted on the way out.
    static constexpr int kPermMinPointsDefault = 8192;
    static int perm_min_points()
    {
        static const char *e = ab_env("LCCRF_PERM_MIN");                         // A/B switch (same results): locality mode from this many points
        static const int v = e ? std::max(atoi(e), 64) : kPermMinPointsDefault;
        return v;
    }
    bool allow_perm = false;           // set by lccrf_batch_create, and by lccrf_create for handles of >= kPermMinPoints points ...
    bool perm_scoped = false;          // ... where it only applies to lattices that are first asked for by inference() (perm_scope): every
    bool perm_scope = false;           //   other entry point of the object API (stepwise inference, PairwisePotential::apply, the lattice
    bool perm_banned = false;          //   probes) works on the caller's point order -- such a handle is built, or re-built once, the plain way
    bool perm_on = false;              // the lattices now in HBM were built in locality mode
    bool unary_is_label = false;       // the unaries come from labels (re-derivable in any order) rather than from a raw array
    bool unary_p_valid = false;        // unary_p holds the raw unaries in the current internal order
    SortScratch sort{};
    float *Qp = nullptr, *unary_p = nullptr;   // [Fcap][maxN][L]

    bool sort_scratch_complete = false;
    int ensure_sort_scratch()
    {
        // (complete or nothing: an allocation that fails half way -- E_NOMEM -- must not leave a later build with some of the
        // arrays null; what a failed attempt did allocate stays with the arena until the engine goes -- ADVICE r4)
        if (sort_scratch_complete) return LCCRF_OK;
        int bits = 8;
        while (bits < 16 && (1 << bits) < 2 * maxN) ++bits;
        sort.bits = bits;
        const size_t nbk = ((size_t)1 << bits) + 1, Fz = (size_t)Fcap;
        int rc;
        if ((rc = mem.alloc(&sort.cells, Fz * maxNpad * kMaxD))) return rc;
        if ((rc = mem.alloc(&sort.partial, Fz * ((maxNpad + 255) / 256 + 1) * 2 * kMaxD))) return rc;
        if ((rc = mem.alloc(&sort.plan, Fz * 3 * kMaxD))) return rc;
        if ((rc = mem.alloc(&sort.code, Fz * maxNpad))) return rc;
        if ((rc = mem.alloc(&sort.hist, Fz * nbk))) return rc;
        if ((rc = mem.alloc(&sort.start, Fz * nbk))) return rc;
        if ((rc = mem.alloc(&sort.tiles, Fz * ((nbk + 4095) / 4096 + 1)))) return rc;
        if ((rc = mem.alloc(&sort.perm, Fz * maxNpad))) return rc;
        if ((rc = mem.alloc(&sort.iperm, Fz * maxNpad))) return rc;
        if ((rc = mem.alloc(&Qp, Fz * maxN * L))) return rc;
        if ((rc = mem.alloc(&unary_p, Fz * maxN * L))) return rc;
        // the vertex order of every kernel (one kernel at a time on the engine's stream: one set of arrays sized for the largest)
        int vcap = 0;
        for (auto &ks : kernels) vcap = std::max(vcap, ks.dev.Epad);
        for (auto &ks : spare) vcap = std::max(vcap, ks.dev.Epad);
        sort.vcap = vcap;
        int vbits = 10;
        while (vbits < 21 && (1 << vbits) < 4 * vcap) ++vbits;
        if (const char *e = ab_env("LCCRF_VBITS")) vbits = std::min(std::max(atoi(e), 10), 21);   // A/B switch (same results): buckets of the vertex sort
        sort.vbits = vbits;
        const size_t vnbk = ((size_t)1 << vbits) + 1;
        if ((rc = mem.alloc(&sort.vcode, Fz * vcap))) return rc;
        if ((rc = mem.alloc(&sort.vkey, Fz * vcap))) return rc;
        if ((rc = mem.alloc(&sort.vph, Fz * 64))) return rc;
        if ((rc = mem.alloc_pinned(&sort.vbad, 1))) return rc;
        *sort.vbad = 0;
        if ((rc = mem.alloc(&sort.vhist, Fz * vnbk))) return rc;
        if ((rc = mem.alloc(&sort.vstart, Fz * vnbk))) return rc;
        if ((rc = mem.alloc(&sort.vtiles, Fz * ((vnbk + 4095) / 4096 + 1)))) return rc;
        if ((rc = mem.alloc(&sort.vpartial, Fz * ((maxNpad + 255) / 256 + 1) * 2 * kMaxD))) return rc;
        if ((rc = mem.alloc(&sort.vplan, Fz * (2 * kMaxD + 3)))) return rc;
        sort_scratch_complete = true;
        return LCCRF_OK;
    }

    int init(int device_id, int frames, int max_points, int n_labels)
    {
        device = device_id;
        Fcap = F = frames;
        maxN = max_points;
        maxNpad = (max_points + 3) & ~3;
        L = n_labels;
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        mem.stream = stream;
        for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventCreateWithFlags(&ev_order, hipEventDisableTiming));
        const size_t nl = (size_t)Fcap * maxN * L;
        int rc;
        if ((rc = mem.alloc(&npoints_own, Fcap))) return rc;
        if ((rc = mem.alloc(&unary_own, nl))) return rc;
        if ((rc = mem.alloc(&crf.Q, nl))) return rc;
        if ((rc = mem.alloc(&crf.next, nl))) return rc;
        if ((rc = mem.alloc(&crf.map, (size_t)Fcap * maxN))) return rc;
        crf.bits_stride = (maxN + 63) / 64;
        if (n_labels == 2 && (rc = mem.alloc(&crf.map_bits, (size_t)Fcap * std::max(crf.bits_stride, 1)))) return rc;
        if ((rc = mem.alloc(&label_own, (size_t)Fcap * maxN))) return rc;
        if ((rc = mem.alloc_pinned(&V_host, (size_t)LCCRF_MAX_KERNELS * Fcap))) return rc;
        if ((rc = mem.alloc_pinned(&row_host, (size_t)LCCRF_MAX_KERNELS * Fcap))) return rc;
        if ((rc = mem.alloc_pinned(&long_host, (size_t)LCCRF_MAX_KERNELS * Fcap))) return rc;
        if ((rc = mem.alloc_pinned(&late_status, 1))) return rc;
        *late_status = 0;
        if ((rc = mem.alloc_pinned(&done_word, 1))) return rc;
        *done_word = 0;
        if ((rc = mem.alloc(&frame_status, Fcap))) return rc;
        if ((rc = mem.alloc(&fb_list, Fcap))) return rc;
        if ((rc = mem.alloc_pinned(&frame_status_host, Fcap))) return rc;
        if ((rc = mem.alloc_pinned(&fb_list_host, Fcap))) return rc;
        if ((rc = mem.alloc_pinned(&npoints_bad, 1))) return rc;
        *npoints_bad = 0;
        crf.F = F;
        crf.maxN = maxN;
        crf.L = L;
        crf.K = 0;
        crf.n_points = npoints_own;
        crf.unary = unary_own;
        return LCCRF_OK;
    }

    void destroy()
    {
        park_check = false;
        if (stream) (void)hipStreamSynchronize(stream);
        if (fb) {
            fb->destroy();
            delete fb;
            fb = nullptr;
        }
        mem.release();
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
        if (ev_order) (void)hipEventDestroy(ev_order);
        if (lean_prep.ev0) (void)hipEventDestroy(lean_prep.ev0);
        if (lean_prep.ev1) (void)hipEventDestroy(lean_prep.ev1);
        lean_prep = LeanPrep{};
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }

    // Allocate the lattice of one more pairwise kernel (all frames).
    int add_kernel(int d, float w, bool own_features, bool stage)
    {
        if ((int)kernels.size() >= LCCRF_MAX_KERNELS)
            return fail(LCCRF_E_CAPACITY, "at most %d pairwise kernels", LCCRF_MAX_KERNELS);
        if (d < 1 || d > LCCRF_MAX_DIMS) return fail(LCCRF_E_INVALID, "feature dims %d not in [1,%d]", d, LCCRF_MAX_DIMS);
        for (size_t i = 0; i < spare.size(); ++i) {     // a recycled engine already owns a lattice of this shape
            if (spare[i].dev.d == d && (!stage || spare[i].feat_stage) && (!own_features || spare[i].feat_own)) {
                KernelState ks = spare[i];
                spare.erase(spare.begin() + i);
                ks.dev.w = w;
                ks.dev.feat = ks.feat_own;
                ks.maxV = ks.dev.Epad;
                ks.maxRow = 0;
                kernels.push_back(ks);
                sync_views();
                return LCCRF_OK;
            }
        }
        KernelState ks;
        KernelDev &k = ks.dev;
        k.d = d;
        k.D1 = d + 1;
        k.maxN = maxN;
        k.maxNpad = maxNpad;
        k.Epad = maxNpad * k.D1;
        k.cap = next_pow2(2L * k.Epad);
        // lattice values of a frame: [0, L) unused, [L, 2L) the all-zero "absent neighbour" (vertex -1), vertex v at
        // vbase + v*L; with L = 2 and an even stride, vertex pairs (2t, 2t+1) are 16-byte aligned (k_blur2)
        k.vstride = (k.Epad + 2) * L;
        k.vbase = 2 * L;
        k.w = w;
        // permutohedral_cpu.h:681 / :249 / :282-285 (quirk Q4): same expressions, same types
        k.alpha = 1.0f / (1 + powf(2, -d));
        k.inv_dp1 = 1.0f / (d + 1);
        const float inv_std_dev = sqrt(2.0 / 3.0) * (d + 1);
        for (int i = 0; i < d; ++i) k.scale[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * inv_std_dev);
        const size_t Fz = (size_t)Fcap, E = (size_t)k.Epad;
        int rc;
        if ((rc = mem.alloc(&k.rem0, Fz * maxNpad * d))) return rc;
        if ((rc = mem.alloc(&k.rank, Fz * maxNpad * d))) return rc;
        if ((rc = mem.alloc(&k.bary, Fz * E))) return rc;
        if ((rc = mem.alloc(&k.offset, Fz * E))) return rc;
        if ((rc = mem.alloc(&k.slot, Fz * k.cap))) return rc;
        if ((rc = mem.alloc(&k.slot_of, Fz * E))) return rc;
        if ((rc = mem.alloc(&k.flag, Fz * (E + 1)))) return rc;
        if ((rc = mem.alloc(&k.prefix, Fz * (E + 1)))) return rc;
        if ((rc = mem.alloc(&k.rep, Fz * E))) return rc;
        if ((rc = mem.alloc(&k.V, Fz))) return rc;
        if ((rc = mem.alloc(&k.rowmax, Fz))) return rc;
        if ((rc = mem.alloc(&k.nbr, Fz * k.D1 * E * 2))) return rc;
        // one frame in flight, large frames (the object API's handles are sized for SLAM frames, which run on the one-workgroup engines):
        // the blur passes go two per launch and read a two-hop neighbour table
        if (Fcap == 1 && L == 2 && allow_perm && (rc = mem.alloc(&k.nbr2, (size_t)(k.D1 / 2) * E * 8))) return rc;
        if (L == 2 && allow_perm && maxN >= perm_min_points()) {
            if (!tbl_bad) {
                if ((rc = mem.alloc_pinned(&tbl_bad, 2))) return rc;
                tbl_bad[0] = tbl_bad[1] = 0;
            }
            k.tbl_bad = tbl_bad;
            if ((rc = mem.alloc(&k.fastn, Fz * E))) return rc;
            if ((rc = mem.alloc(&k.ndist, (size_t)kNdistAxes))) return rc;
            if ((rc = mem.alloc(&k.nearoff, Fz * 2 * E * 2))) return rc;
            // (the window splat's 16-byte vertex records: measured +-1 % on C5 x 8, round 5 -- notes/r5_experiments.md; kept as an
            // experiment of the instrumented library, LCCRF_SPLAT_REC=1)
            if (maxN < (1 << 24) && ab_env("LCCRF_SPLAT_REC") && (rc = mem.alloc(&k.srec, Fz * E))) return rc;
            if (!ndist_host && (rc = mem.alloc_pinned(&ndist_host, (size_t)LCCRF_MAX_KERNELS * kNdistAxes))) return rc;
        }
        // large frames, a few in flight: the sorted build also leaves the neighbour table in its compact form (16-bit offsets), for
        // the batches of kNbrcMinFrames..kNbrcMaxFrames frames this engine may see
        if (k.tbl_bad && Fcap >= kNbrcMinFrames) {
            const size_t Fc = (size_t)std::min(Fcap, kNbrcMaxFrames);
            if ((rc = mem.alloc(&k.nbrc, Fc * k.D1 * E * 2))) return rc;
            if ((rc = mem.alloc(&k.nbrc_base, Fc * k.D1 * (E / kNbrcBlock + 1) * 2))) return rc;
        }
        if ((rc = mem.alloc(&k.rowptr, Fz * (E + 1)))) return rc;
        // (frames of the one-workgroup engines' range keep their rows in LDS there; on the streaming engine they are summed in line: a
        // batch engine for 16 384 SLAM frames would carry 270 MB of lists per kernel)
        if (maxN > kLongRowListMinPoints) {
            if ((rc = mem.alloc(&k.longrow, Fz * kLongRowCap))) return rc;
            if ((rc = mem.alloc(&k.longcnt, Fz))) return rc;
        }
        if ((rc = mem.alloc(&k.csr_pt, Fz * E))) return rc;
        if ((rc = mem.alloc(&k.csr_w, Fz * E))) return rc;
        if ((rc = mem.alloc(&k.csr_pos, Fz * E))) return rc;
        if ((rc = mem.alloc(&k.pk, Fz * E))) return rc;
        if ((rc = mem.alloc(&k.nbr16, Fz * k.D1 * E))) return rc;
        if ((rc = mem.alloc(&k.norm, Fz * maxN))) return rc;
        if ((rc = mem.alloc(&k.val0, Fz * k.vstride))) return rc;
        if ((rc = mem.alloc(&k.val1, Fz * k.vstride))) return rc;
        if (own_features) {
            if ((rc = mem.alloc(&ks.feat_own, Fz * maxN * d))) return rc;
            k.feat = ks.feat_own;
        }
        if (stage)
            if ((rc = mem.alloc_pinned(&ks.feat_stage, Fz * maxN * d))) return rc;
        ks.maxV = k.Epad;
        kernels.push_back(ks);
        sync_views();
        return LCCRF_OK;
    }

    // A parked engine whose last frame was taken label by label: its kernel is over once the done word shows this frame's epoch
    // (the kernel's very last store); normally that happened long ago, otherwise wait for the stream the ordinary way.
    void ensure_parked_idle()
    {
        if (!park_check) return;
        park_check = false;
        const volatile unsigned *w = done_word;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (unsigned spins = 1; !(seen = (*w == done_epoch)); ++spins) {
            if ((spins & 0x3f) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(50)) break;
            cpu_relax();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!seen && stream) (void)hipStreamSynchronize(stream);
    }

    // Park everything for the next user of this engine (object API handle cache).
    void recycle()
    {
        // (a synchronisation that starts after the kernel has ended costs ~10 us: the runtime queues a barrier packet and
        // waits for its round trip; the done word has already told us that nothing is in flight)
        // getMap() saw every label land: the kernel's last stores (V_out, frame_status, the done word) follow within a microsecond or
        // two.  Nobody waits for them here (the tracker's next frame is 33 ms away); the next user of this handle settles it before it
        // touches anything the kernel could still read or write (ensure_parked_idle, called by lccrf_create).
        if (stream && idle_by_done && idle_needs_done) park_check = true;
        else if (stream && !idle_by_done) (void)hipStreamSynchronize(stream);
        idle_by_done = idle_needs_done = false;
        labels_armed = done_armed = false;
        for (auto &ks : kernels) spare.push_back(ks);
        kernels.clear();
        F = Fcap;
        crf.unary = unary_own;
        crf.n_points = npoints_own;
        unary_set = built = sizes_known = started = false;
        late_pending = false;
        unary_deferred = false;
        perm_on = vorder_on = perm_banned = perm_scope = false;      // (the next user's lattices decide afresh)
        unary_is_label = unary_p_valid = false;
        built_upto = 0;
        engine_pref = 0;
        engine_used = 1;
        sync_views();
    }

    void sync_views()
    {
        kdevs.resize(kernels.size());
        maxV.resize(kernels.size());
        maxRow.resize(kernels.size());
        for (size_t i = 0; i < kernels.size(); ++i) {
            kernels[i].dev.V_host = V_host + i * Fcap;
            kernels[i].dev.rowmax_host = row_host + i * Fcap;
            kernels[i].dev.perm = perm_on ? sort.perm : nullptr;
            kernels[i].dev.iperm = perm_on ? sort.iperm : nullptr;
            kernels[i].dev.vorder = (perm_on && vorder_on && sort.vkey && kernels[i].dev.Epad <= sort.vcap) ? 1 : 0;
            kdevs[i] = kernels[i].dev;
            maxV[i] = kernels[i].maxV;
            maxRow[i] = kernels[i].maxRow;
        }
        crf.K = (int)kernels.size();
        crf.F = F;
        crf.activeN = activeN;
        crf.perm = perm_on ? sort.perm : nullptr;
        crf.perm_stride = maxNpad;
    }

    // Lattice + normalisation of kernels [k0, k0+n) for every frame (PottsPotential3D ctor).
    // SLAM-size frames take the fused build (one launch for all of them when they share d);
    // anything else the streaming build (19 launches per kernel).
    int build_kernels(int k0, int n)
    {
        for (int k = k0; k < k0 + n; ++k) kernels[k].maxV = kernels[k].dev.Epad;
        if (k0 == 0) {                                     // a build of every kernel decides the internal point order afresh
            static const bool no_perm = ab_env("LCCRF_NO_PERM") != nullptr;   // A/B and cross-check switch: same results either way
            const int NAp = activeN > 0 ? activeN : maxN;
            const bool want = allow_perm && !no_perm && !perm_banned && (!perm_scoped || perm_scope) && n > 0 && n == (int)kernels.size() && NAp >= perm_min_points();
            if (want || perm_on) {                         // whatever was derived in the old order is stale
                if (unary_is_label) unary_deferred = true;
                unary_p_valid = false;
            }
            perm_on = false;
            // ... and, with the points permuted, whether the lattices are built by SORTING the entries on the row-major code of their
            // vertex (ids along the lattice's axes: the blur pass touches half as many lines) -- yes unless switched off: with the
            // points in row-major order too it wins at every number of frames in flight (notes/r4_experiments.md section 2)
            static const char *env_vo = ab_env("LCCRF_VERTEX_ORDER");               // A/B switch: 1 on, 0 off (same results)
            const int vo = env_vo ? (atoi(env_vo) ? 1 : 2) : opt_vertex_order;
            vorder_on = want && !vorder_broken && vo != 2;
            sync_views();
            if (want) {
                int rcs = ensure_sort_scratch();
                if (rcs) return rcs;
                int src = 0;                               // one order for the whole CRF (Q is shared): the kernel with most dimensions decides
                for (int k = 1; k < n; ++k)
                    if (kernels[k].dev.d > kernels[src].dev.d) src = k;
                // with the sorted build the points follow the vertices: (coarse) row-major order of their cells in the lattice's own basis
                // instead of the Z-order curve (C5 x 8: splat 30.5 -> 27.7, slice 24.2 -> 19.5 us, build 1.93 -> 1.76 ms)
                static const bool env_z = ab_env("LCCRF_POINTS_ZORDER") != nullptr;        // A/B switch (same results)
                sort.rm_points = (vorder_on && !env_z) ? 1 : 0;
                launch_sort_points(kdevs[src], crf, sort, stream);
                HIP_TRY(hipGetLastError());
                perm_on = true;
            }
        }
        sync_views();
        const bool no_small = ab_env("LCCRF_NO_FUSED_BUILD") != nullptr;   // debug / cross-check switch
        int k = k0;
        while (k < k0 + n) {
            int m = 1;
            const int NA = activeN > 0 ? activeN : maxN;
            // (locality mode: the one-workgroup build knows nothing of the internal point order -- ADVICE r4: with the threshold lowered
            // below its range it would have built in the caller's order under an engine that iterates in the permuted one)
            const bool small_ok = !no_small && !perm_on;
            if (small_ok && k + 1 < k0 + n && build_small_supported(&kdevs[k], 2, NA)) m = 2;
            if (small_ok && build_small_supported(&kdevs[k], m, NA)) {
                for (int u = 0; u < m; ++u) kernels[k + u].dev.nbr2_ok = kernels[k + u].dev.nbrc_ok = kernels[k + u].dev.fast0_ok = kernels[k + u].dev.longrow_ok = kernels[k + u].dev.srec_ok = 0;
                launch_build_small(&kdevs[k], m, NA, crf, stream);   // writes V / rowmax to the pinned mirrors itself
            } else {
                m = 1;
                kernels[k].dev.nbr2_ok = kernels[k].dev.nbr2 != nullptr;      // (the streaming build fills the two-hop table when there is one)
                kernels[k].dev.longrow_ok = kdevs[k].longrow_ok = kernels[k].dev.longrow != nullptr;   // (... and lists the long rows: the normalisation below reads the list)
                kernels[k].dev.nbrc_ok = kernels[k].dev.nbrc != nullptr && kernels[k].dev.vorder && F >= kNbrcMinFrames && F <= kNbrcMaxFrames;   // (... and the sorted build the compact one)
                kernels[k].dev.fast0_ok = kernels[k].dev.tbl_bad != nullptr && kernels[k].dev.vorder;
                kernels[k].dev.srec_ok = kernels[k].dev.srec != nullptr && kernels[k].dev.vorder;          // (the sorted build packs the splat records)
                kernels[k].dev.nbr2_first = kernels[k].dev.fast0_ok;              // (what build_kernel_d derives from the same two fields)
                launch_build_kernel(kdevs[k], crf, kernels[k].maxV, stream, perm_on ? &sort : nullptr);
                launch_norm(kdevs[k], crf, kernels[k].maxV, stream);
                HIP_TRY(hipMemcpyAsync(V_host + (size_t)k * Fcap, kernels[k].dev.V, sizeof(int) * F, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipMemcpyAsync(row_host + (size_t)k * Fcap, kernels[k].dev.rowmax, sizeof(int) * F, hipMemcpyDeviceToHost, stream));
                if (kernels[k].dev.longrow)
                    HIP_TRY(hipMemcpyAsync(long_host + (size_t)k * Fcap, kernels[k].dev.longcnt, sizeof(int) * F, hipMemcpyDeviceToHost, stream));
                if (kernels[k].dev.fast0_ok)
                    HIP_TRY(hipMemcpyAsync(ndist_host + (size_t)k * kNdistAxes, kernels[k].dev.ndist, sizeof(int) * kNdistAxes, hipMemcpyDeviceToHost, stream));
            }
            k += m;
        }
        HIP_TRY(hipGetLastError());
        sync_views();                                       // (nbr2_ok)
        sizes_known = false;
        return LCCRF_OK;
    }
    int build_kernel(int k) { return build_kernels(k, 1); }

    // After the builds: learn max V per kernel so inference grids are sized to the lattice.
    // Build every kernel added since the last build (object API: add_pairwise only stages features).
    int flush_builds()
    {
        const int n = (int)kernels.size();
        if (built_upto >= n) return LCCRF_OK;
        const int k0 = built_upto;
        built_upto = n;
        return build_kernels(k0, n - k0);
    }

    int learn_sizes()
    {
        int rc0 = flush_builds();
        if (rc0) return rc0;
        if (sizes_known) return LCCRF_OK;
        HIP_TRY(hipStreamSynchronize(stream));
        if (*npoints_bad) {
            *npoints_bad = 0;
            return fail(LCCRF_E_CAPACITY, "a bound n_points[f] lies outside [0, max_points=%d] (the kernels clamped it)", maxN);
        }
        if (sort.vbad && *sort.vbad) {                     // a frame's vertices do not fit a 62-bit row-major code: hash build from now on
            *sort.vbad = 0;
            vorder_broken = true;
            int rcb = build_kernels(0, (int)kernels.size());
            if (rcb) return rcb;
            HIP_TRY(hipStreamSynchronize(stream));
        }
#if LCCRF_INSTRUMENT
        if (tbl_bad && ab_env("LCCRF_FAST0_BREAK")) tbl_bad[1] = 1;      // test hook: as if the build had found an axis-0 neighbour elsewhere
#endif
        if (tbl_bad && (tbl_bad[0] || tbl_bad[1])) {
            // [0] a block's neighbours span more than 16 bits: the blur reads the 32-bit table; [1] an axis-0 neighbour that is not the
            // next / previous id (cannot happen with exact codes -- checked all the same): the first blur pass keeps its own launch
            for (auto &ks : kernels) {
                if (tbl_bad[0]) ks.dev.nbrc_ok = 0;
                if (tbl_bad[1]) ks.dev.fast0_ok = 0;
            }
#if LCCRF_INSTRUMENT
            if (tbl_bad[0]) fprintf(stderr, "lccrf: compact neighbour table abandoned (a block's neighbours span more than its offsets hold)\n");
            if (tbl_bad[1]) fprintf(stderr, "lccrf: first blur pass keeps its own launch (an axis-0 neighbour is not an adjacent id)\n");
#endif
            tbl_bad[0] = tbl_bad[1] = 0;
        }
        for (size_t k = 0; k < kernels.size(); ++k) {
            // how many blur passes ride in the splat (sorted build): pass 0 always (adjacent ids); passes 1 and 2 when the largest id
            // distances of their neighbours fit the halo of a window of up to 1024 vertices that keeps >= 3/4 of its lanes productive
            KernelDev &kd = kernels[k].dev;
            kd.splat_passes = kd.fast0_ok ? 1 : 0;
            kd.splat_halo = 1;
            kd.splat_block = 0;
            static const char *env_sp = ab_env("LCCRF_SPLAT_PASSES");               // A/B switch (same results): at most this many
            const int cap = env_sp ? atoi(env_sp) : 3;
            if (kd.fast0_ok && cap >= 2) {
                const int *nd = ndist_host + k * kNdistAxes;
                int halo = 1;
                for (int j = 1; j < std::min(kd.D1, std::min(cap, 3)); ++j) {
                    if (nd[j] < 1 || nd[j] > 127 || halo + nd[j] > 1024 / 8) break;
                    halo += nd[j];
                    kd.splat_passes = j + 1;
                    kd.splat_halo = halo;
                }
                // single frames pair the passes behind the splat off a two-hop table built for the pairs (1,2), (3,4) ...: 1 or 3 passes
                if (kd.splat_passes == 2 && kd.nbr2 && kd.D1 > 3) { kd.splat_passes = 1; kd.splat_halo = 1; }
                if (kd.splat_passes >= 2) kd.splat_block = kd.splat_halo * 8 <= 256 ? 256 : kd.splat_halo * 8 <= 512 ? 512 : 1024;
            }
            int m = 0, r = 0, nlong = 0;
            for (int f = 0; f < F; ++f) {
                m = std::max(m, V_host[k * Fcap + f]);
                r = std::max(r, row_host[k * Fcap + f]);
                if (kd.longrow_ok) nlong = std::max(nlong, long_host[k * Fcap + f]);
            }
            // a coarse kernel over many points (listed rows, or more than ~4 entries per row on average): the row-walking splat with
            // its loads up front + a workgroup per listed row, no blur pass in the splat
            const int NAr = activeN > 0 ? activeN : maxN;
            kd.long_mode = kd.longrow_ok && (nlong > 0 || (long)NAr * kd.D1 > 4L * std::max(m, 1));
            if (kd.long_mode && (long)NAr * kd.D1 >= 16L * std::max(m, 1)) kd.long_mode = 2;       // ... a wavefront per vertex from ~16 entries per row
            kernels[k].maxV = m;
            kernels[k].maxRow = r;
        }
        sync_views();
        sizes_known = true;
        lean_prep.valid = false;                           // (every path that changes a lattice clears sizes_known and so comes through here)
        lean_prep.seen_key = 0;
        sized_engine = 1;
        if (engine_pref != 1 && !perm_on && fused_supported(crf, kdevs.data(), maxV.data(), maxRow.data(), &fused_lds)) sized_engine = 2;
        if (engine_pref == 2 && sized_engine != 2)
            return fail(LCCRF_E_CAPACITY, "fused engine requested but the problem does not fit one workgroup's LDS");
        return LCCRF_OK;
    }

    // setUnaryEnergyFromLabel (densecrf3d.h:100-130) is deferred: the 2L+1 energies travel as a kernel
    // argument and either the one-launch-per-frame kernel derives the unaries itself or ensure_unary()
    // launches the small kernel when somebody needs the array.  `label` must stay readable until then.
    void defer_unary_from_label(const int16_t *label, const float *conf)
    {
        // densecrf3d.h:109-115.  log(float) binds to the float overload at the reference's call
        // site (src/Tracking.cc:21-43 sees `using namespace std` from include/Tracking.h:55).
        UnaryTable tb;
        tb.v[0] = -logf(1.0f / L);
        for (int i = 0; i < L; ++i) {
            tb.v[1 + i] = -logf((1.0f - conf[i]) / (L - 1));
            tb.v[1 + L + i] = -logf(conf[i]);
        }
        crf.unary = unary_own;
        unary_deferred = true;
        unary_is_label = true;
        deferred_label = label;
        deferred_tbl = tb;
        unary_set = true;
    }

    int ensure_unary()
    {
        if (!unary_deferred) return LCCRF_OK;
        unary_deferred = false;
        launch_unary_from_label_tbl(crf, deferred_label, deferred_tbl, stream);
        HIP_TRY(hipGetLastError());
        return LCCRF_OK;
    }

    int start()
    {
        if (!unary_set) return fail(LCCRF_E_STATE, "unary energies not set");
        { int rl = resolve_late(); if (rl) return rl; }
        { int rp = ensure_plain(); if (rp) return rp; }   // (object API after a locality-mode inference(): label-derived energies sit in that order)
        { int ru = ensure_unary(); if (ru) return ru; }
        launch_start(crf, stream);
        started = true;
        return LCCRF_OK;
    }

    int step(float relax)
    {
        if (!started) return fail(LCCRF_E_STATE, "stepInference before startInference");
        int rc = resolve_late();
        if (rc) return rc;
        if ((rc = ensure_plain())) return rc;
        rc = learn_sizes();
        if (rc) return rc;
        if ((rc = ensure_unary())) return rc;             // (label-derived energies follow the lattices' point order: re-derived after a re-build)
        launch_step_stream(crf, kdevs.data(), maxV.data(), relax, stream);
        return LCCRF_OK;
    }

    // Can the whole frame (lattices + normalisation + inference) run as ONE launch (frame_engine.hip)?
    bool frame_ok() const
    {
        static const bool no_frame = ab_env("LCCRF_NO_FRAME") != nullptr;   // cross-check switch: two-kernel path, same results
        return !no_frame && engine_pref == 0 && !kernels.empty() && frame_supported(crf, kdevs.data());
    }

    // One launch per frame; whether every frame fitted the kernel's LDS plan is known at the next
    // synchronisation point (resolve_late), which re-runs the batch on the two-kernel path if not.
    int run_frame(int n_iter, int with_map, float relax)
    {
        *late_status = 0;
        // (label-derived energies already written for lattices in locality mode sit in the INTERNAL point order: the frame kernel, which
        //  works in the caller's order, derives them from the labels again)
        const bool from_label = (unary_deferred || (unary_is_label && perm_on)) && L == 2;
        unsigned *dual = nullptr;
        // The tracker's case (one frame, 255 idle CUs) and small batches (a quarter as many frames as CUs, or fewer): every
        // frame gets two workgroups, one per lattice build.
        if (crf.K == 2 && F <= kDualMaxFrames && !opt_single_wg) {
            if (!dual_area) {
                int rc = mem.alloc(reinterpret_cast<char **>(&dual_area), frame_dual_bytes(std::min(Fcap, kDualMaxFrames)));
                if (rc) return rc;
            }
            dual = dual_area;
            if (++dual_epoch == 0) dual_epoch = 1;
        }
        static const bool no_done_word = ab_env("LCCRF_NO_DONE_WORD") != nullptr;   // A/B switch: wait on the stream instead
        done_armed = late_ok && F == 1 && !no_done_word;
        if (done_armed && ++done_epoch == 0) done_epoch = 1;
        labels_armed = done_armed && with_map && L == 2 && map_host && map_host == crf.map && activeN > 0;
        if (labels_armed) memset(map_host, 0xff, (size_t)activeN * sizeof(int16_t));
        // full-size frames, two per CU (frame_lean.hip): the record area its loop re-reads is allocated when a batch first wants it
        // (sized for the frames BOUND, not for the handle's capacity -- 96 KB per frame: a handle created for 16384 frames that runs 1024 takes
        //  0.1 GB, not 1.6; a later, larger batch gets a new area, the old one stays with the handle until it is destroyed -- ADVICE r5)
        if (frame_lean_ok && lean_rec_frames < F && frame_lean_wanted(crf)) {
            // (no memory for the area: the batch keeps the one-frame-per-CU kernel, which needs none)
            lean_rec = nullptr;
            lean_rec_frames = 0;
            const int want = std::min(Fcap, std::max(F, 256));
            if (mem.alloc(reinterpret_cast<char **>(&lean_rec), frame_lean_rec_bytes(want), false)) {   // (written before it is read: no zeroing)
                lean_rec = nullptr;
                frame_lean_ok = false;
                (void)hipGetLastError();
            } else {
                lean_rec_frames = want;
            }
        }
        const int shape = launch_frame(crf, kdevs.data(), n_iter, with_map, relax, late_status, frame_status,
                                       from_label ? deferred_label : nullptr, deferred_tbl.v, stream, frame_small_ok, dual, dual_epoch,
                                       done_armed ? done_word : nullptr, done_epoch, frame_lean_ok ? lean_rec : nullptr);
        HIP_TRY(hipGetLastError());
        frame_small_used = shape == 1;
        frame_lean_used = shape == 2;
        fused_shape = shape == 0 ? (1024 | 1 << 16) : (512 | 2 << 16);       // lanes per frame | frames per CU

        late_pending = true;
        late_iter = n_iter;
        late_map = with_map;
        late_relax = relax;
        started = true;
        engine_used = 3;
        last_with_map = with_map;
        fallback_frames = 0;
        return LCCRF_OK;
    }

    int inference(int n_iter, int with_map, float relax)
    {
        if (n_iter < 0) return fail(LCCRF_E_INVALID, "n_iterations < 0");
        if (!unary_set) return fail(LCCRF_E_STATE, "unary energies not set");
        int rc = resolve_late();
        if (rc) return rc;
        // Object API: nothing has been built yet (add_pairwise only stages features) -> do not build, run the
        // frame in one launch.  If a probe already forced the lattices into HBM, iterate on those instead.
        if (late_ok && frame_ok() && !(built_upto == (int)kernels.size() && sizes_known)) return run_frame(n_iter, with_map, relax);
        perm_scope = true;                                // (large frames: lattices built from here may use locality mode)
        rc = inference_sized(n_iter, with_map, relax);
        perm_scope = false;
        return rc;
    }

    // Object API, large frames: DenseCRF::inference() runs in locality mode (internal point order, sorted build); the entry points
    // that expose or continue from per-point lattice state in the CALLER's order -- stepInference, PairwisePotential::apply, the
    // lattice / norm / unary probes -- get the plain build instead: built that way if they come first, re-built once if not.
    int ensure_plain()
    {
        if (!perm_on || !perm_scoped) return LCCRF_OK;
        perm_banned = true;
        built_upto = 0;
        sizes_known = false;
        return flush_builds();
    }

    // lccrf_batch_run: per frame the PottsPotential ctors + inference(n, with_map)
    int run(int n_iter, int with_map, float relax)
    {
        if (n_iter < 0) return fail(LCCRF_E_INVALID, "n_iterations < 0");
        if (!unary_set) return fail(LCCRF_E_STATE, "unary energies not set");
        int rc = resolve_late();
        if (rc) return rc;
        if (frame_ok()) return run_frame(n_iter, with_map, relax);
        built_upto = 0;                                   // two-kernel path: rebuild for the current inputs, then infer
        sizes_known = false;
        return inference_sized(n_iter, with_map, relax);
    }

    int inference_sized(int n_iter, int with_map, float relax)
    {
        int rc = learn_sizes();                            // (builds what is pending: the internal point order is known after this)
        if (rc) return rc;
        rc = ensure_unary();
        if (rc) return rc;
        engine_used = sized_engine;
        last_with_map = with_map;
        if (perm_on) {
            // locality mode: iterate on the internal-order view, un-permute Q on the way out (densecrf_base.h:65-73 otherwise)
            CrfDev cp = crf;
            cp.Q = Qp;
            if (!unary_is_label) {
                // a BOUND unary array ("bound, not copied") may have been updated in place since the last call: permute it
                // afresh every time, as the other engines read it live; only the engine's own copy is cached
                if (!unary_p_valid || crf.unary != unary_own) launch_permute_rows(crf, unary_p, crf.unary, L, 1, stream);
                unary_p_valid = true;
                cp.unary = unary_p;
            }
            launch_start(cp, stream);
            for (int it = 0; it < n_iter; ++it) launch_step_stream(cp, kdevs.data(), maxV.data(), relax, stream);
            launch_permute_rows(crf, crf.Q, Qp, L, 0, stream);
            if (with_map) launch_map(crf, stream);
            started = true;
        } else if (sized_engine == 2) {
            // the two-frames-per-CU kernel runs from prepared launch records (fused_lean.h: LeanPrepPlan): one block per frame, owned
            // here, (re)written by the first inference behind a build -- learn_sizes() clears `valid` whenever a lattice has changed
            const size_t prep_need = lean_prep_bytes(crf, kdevs.data(), maxV.data(), maxRow.data());
            if (prep_need > lean_prep.bytes) {
                lean_prep.valid = false;
                lean_prep.buf = nullptr;                       // (an outgrown block stays with the arena until the handle goes)
                lean_prep.bytes = 0;
                unsigned char *p = nullptr;
                if (mem.alloc(&p, prep_need, false) == LCCRF_OK) {
                    lean_prep.buf = p;
                    lean_prep.bytes = prep_need;
                } else {
                    (void)hipGetLastError();                   // no room: the self-contained kernel runs instead, same results
                }
                if (!lean_prep.ev0) {
                    (void)hipEventCreate(&lean_prep.ev0);
                    (void)hipEventCreate(&lean_prep.ev1);
                }
            }
            fused_shape = launch_inference_fused(crf, kdevs.data(), maxV.data(), maxRow.data(), n_iter, with_map, relax, stream,
                                                 prep_need ? &lean_prep : nullptr);
            started = true;
        } else {
            if ((rc = start())) return rc;
            for (int it = 0; it < n_iter; ++it) launch_step_stream(crf, kdevs.data(), maxV.data(), relax, stream);
            if (with_map) launch_map(crf, stream);
        }
        HIP_TRY(hipGetLastError());
        return LCCRF_OK;
    }

    // Before anything looks at (or continues from) the results of a late-bound inference.
    // *seen_done (optional): the frame's results were observed through the done word -- the caller may read the pinned
    // outputs without a stream synchronisation of its own.
    int resolve_late(bool *seen_done = nullptr)
    {
        if (seen_done) *seen_done = false;
        if (!late_pending) return LCCRF_OK;
        late_pending = false;
        labels_armed = false;
        bool seen = false;
        if (done_armed) {                                  // bounded poll; a word that never comes is waited for the ordinary way
            done_armed = false;
            const volatile unsigned *w = done_word;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 1; !(seen = (*w == done_epoch)); ++spins) {
                if ((spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
                cpu_relax();
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) HIP_TRY(hipStreamSynchronize(stream));
        if (*npoints_bad) {                                // (the one-launch path never passes through learn_sizes())
            *npoints_bad = 0;
            return fail(LCCRF_E_CAPACITY, "a bound n_points[f] lies outside [0, max_points=%d] (the kernels clamped it)", maxN);
        }
        if (*late_status == 0) {
            if (seen_done) *seen_done = seen;
            return LCCRF_OK;
        }
        *late_status = 0;                                  // some frame did not fit the one-launch kernel
        if (seen) HIP_TRY(hipStreamSynchronize(stream));   // (the kernel has nothing left to do; the stream formally still holds it)
        HIP_TRY(hipMemcpy(frame_status_host, frame_status, sizeof(int) * F, hipMemcpyDeviceToHost));
        int n = 0;
        for (int f = 0; f < F; ++f)
            if (frame_status_host[f]) fb_list_host[n++] = f;
        fallback_frames = n;
        if (frame_small_used && 8 * n > F) frame_small_ok = false;   // these lattices want the whole CU: 1024 lanes from now on
        if (frame_lean_used && 8 * n > F) frame_lean_ok = false;
        int rc = LCCRF_OK;
        if (n >= F || n == 0) {                            // every frame (always so for the object API): two-kernel path in place
            built_upto = 0;
            sizes_known = false;
            rc = inference_sized(late_iter, late_map, late_relax);
        } else {
            rc = rerun_frames(n);                          // the batch stays a one-launch batch; only the flagged frames pay twice
            engine_used = 3;
        }
        if (rc) return rc;
        if (timed_inf) HIP_TRY(hipEventRecord(ev[3], stream));   // the timed region now ends behind the re-run
        if (after_rerun && (rc = after_rerun(after_rerun_ctx))) return rc;
        HIP_TRY(hipStreamSynchronize(stream));             // callers read borrowed device buffers right behind a synchronisation point
        return LCCRF_OK;
    }

    // Two-kernel path for the frames fb_list_host[0, n) only, in a compact sub-engine; results scattered back.
    int rerun_frames(int n)
    {
        if (fb && fb->Fcap < n) {
            fb->destroy();
            delete fb;
            fb = nullptr;
        }
        if (!fb) {
            fb = new (std::nothrow) Engine;
            if (!fb) return fail(LCCRF_E_NOMEM, "host allocation failed");
            int rc = fb->init(device, std::min(Fcap, std::max(8, next_pow2(n))), maxN, L);
            for (size_t k = 0; k < kernels.size() && !rc; ++k) rc = fb->add_kernel(kernels[k].dev.d, kernels[k].dev.w, true, false);
            if (!rc && hipStreamSynchronize(fb->stream) != hipSuccess) rc = fail(LCCRF_E_HIP, "hipStreamSynchronize after allocation failed");
            if (rc) {
                fb->destroy();
                delete fb;
                fb = nullptr;
                return rc;
            }
        }
        Engine &g = *fb;
        const hipStream_t g_own = g.stream;
        g.stream = g.mem.stream = stream;                  // everything below (lazy allocations' zeroing included) is ordered on this engine's stream
        struct Restore { Engine &g; hipStream_t s; ~Restore() { g.stream = g.mem.stream = s; } } restore{g, g_own};
        g.F = n;
        g.activeN = activeN;
        g.engine_pref = engine_pref == 1 ? 1 : 0;
        HIP_TRY(hipMemcpyAsync(fb_list, fb_list_host, sizeof(int) * n, hipMemcpyHostToDevice, stream));
        launch_copy_frames(g.npoints_own, 4, crf.n_points, 4, fb_list, n, 4, 1, stream);
        g.crf.n_points = g.npoints_own;
        for (size_t k = 0; k < kernels.size(); ++k) {
            const size_t row = (size_t)maxN * kernels[k].dev.d * sizeof(float);
            launch_copy_frames(g.kernels[k].feat_own, row, kernels[k].dev.feat, row, fb_list, n, row, 1, stream);
            g.kernels[k].dev.feat = g.kernels[k].feat_own;
            g.kernels[k].dev.w = kernels[k].dev.w;
        }
        g.crf.unary = g.unary_own;
        if (unary_deferred || (unary_is_label && perm_on)) {   // (see run_frame: energies written in the internal point order are of no use here)
            launch_copy_frames(g.label_own, (size_t)maxN * 2, deferred_label, (size_t)maxN * 2, fb_list, n, (size_t)maxN * 2, 1, stream);
            g.unary_deferred = true;
            g.deferred_label = g.label_own;
            g.deferred_tbl = deferred_tbl;
        } else {
            const size_t row = (size_t)maxN * L * sizeof(float);
            launch_copy_frames(g.unary_own, row, crf.unary, row, fb_list, n, row, 1, stream);
            g.unary_deferred = false;
        }
        g.unary_set = true;
        g.built_upto = 0;
        g.sizes_known = false;
        g.sync_views();
        HIP_TRY(hipGetLastError());
        int rc = g.inference_sized(late_iter, late_map, late_relax);
        if (rc) return rc;
        const size_t qrow = (size_t)maxN * L * sizeof(float);
        launch_copy_frames(crf.Q, qrow, g.crf.Q, qrow, fb_list, n, qrow, 0, stream);
        if (late_map) {
            launch_copy_frames(crf.map, (size_t)maxN * 2, g.crf.map, (size_t)maxN * 2, fb_list, n, (size_t)maxN * 2, 0, stream);
            if (crf.map_bits && g.crf.map_bits)
                launch_copy_frames(crf.map_bits, (size_t)crf.bits_stride * 8, g.crf.map_bits, (size_t)crf.bits_stride * 8, fb_list, n,
                                   (size_t)crf.bits_stride * 8, 0, stream);
        }
        for (size_t k = 0; k < kernels.size(); ++k)
            launch_copy_frames(kernels[k].dev.V, 4, g.kernels[k].dev.V, 4, fb_list, n, 4, 0, stream);
        HIP_TRY(hipGetLastError());
        return LCCRF_OK;
    }
};

}  // namespace

// object API: up to this many points a handle's inputs (count, labels, features) are read by the kernels straight from pinned host
// memory -- the one-launch frame kernel's range (frame_supported: 4 x 1024 points), where an upload command costs more than the reads
constexpr int kObjectPinnedMaxPoints = 4096;

struct lccrf_crf {
    Engine eng;
    int N = 0;                      // points of the CRF this handle currently represents
    int cap = 0;                    // capacity it was allocated for (eng.maxN)
    int16_t *stage_i16 = nullptr;   // pinned [cap]
    float *stage_f32 = nullptr;     // pinned [cap*L]
    int *stage_n = nullptr;         // pinned [1]: the point count where the kernels of a SLAM frame read it (no upload command)
    bool label_stage_busy = false;  // a kernel that reads stage_i16 may still be pending
    int16_t *map_pin = nullptr;     // pinned [cap]: the kernels write the MAP labels straight into host memory
};

namespace {
// The reference builds and destroys a DenseCRF3D per frame (src/Tracking.cc:1920); doing that with
// device memory, a stream and pinned staging costs milliseconds.  Destroyed handles are therefore
// parked here and handed out again by lccrf_create when device, label count and capacity fit.
struct HandleCache {
    std::mutex m;
    std::vector<lccrf_crf *> parked;
    static constexpr size_t kMaxParked = 8;
} g_cache;

int capacity_for(int n) { return std::max(1024, ((n + n / 4 + 511) / 512) * 512); }

// lccrf_pose_optimization's staging area (device + pinned host + stream), one per device, kept between calls: the
// reference calls Optimizer::PoseOptimization once per frame (src/Tracking.cc:1002) and an allocation costs more than the solve.
struct PoseStage {
    std::mutex m;
    char *d = nullptr, *h = nullptr;
    size_t cap = 0;
    hipStream_t stream = nullptr;
};
constexpr int kMaxDevices = 64;
PoseStage g_pose_stage[kMaxDevices];
}  // namespace

// The asynchronous host path of a batch (lccrf_batch_set_inputs_host_async / _download_async / _wait_download): pinned staging for
// the inputs and the results, one stream per copy direction (the GPU has DMA engines for both), events instead of host waits.
// One batch is in flight per handle; a caller that wants batch i+1 uploaded under batch i's kernels alternates between handles.
struct HostPipe {
    hipStream_t up = nullptr, down = nullptr;
    hipEvent_t ev_up = nullptr, ev_down = nullptr, ev_q = nullptr;   // upload landed / download landed / "everything queued so far"
    int *npoints = nullptr;                  // pinned staging, sized for the batch's capacities
    int16_t *label = nullptr;
    float *unary = nullptr;
    std::vector<float *> feat;
    uint64_t *bits = nullptr;                // pinned results
    int16_t *map = nullptr;
    float *prob = nullptr;
    bool up_pending = false, down_pending = false;
    int down_what = 0, down_frames = 0;
    int copy_threads = 8;                    // host threads of the staging copy (LCCRF_OPT_COPY_THREADS)
    void destroy()
    {
        if (up) (void)hipStreamDestroy(up);
        if (down) (void)hipStreamDestroy(down);
        for (hipEvent_t ev : {ev_up, ev_down, ev_q})
            if (ev) (void)hipEventDestroy(ev);
        up = down = nullptr;
        ev_up = ev_down = ev_q = nullptr;
    }
};

struct lccrf_batch {
    Engine eng;
    lccrf_batch_desc desc{};
    bool inputs_set = false, labels_bound = false;
    const int16_t *d_label = nullptr;
    const int32_t *d_pose_total = nullptr;   // lccrf_batch_pose_set_crf_counts
    HostPipe pipe;
};

namespace {
std::atomic<int> g_default_single_wg{0};

int apply_option(Engine &e, int option, int value)
{
    switch (option) {
    case LCCRF_OPT_SINGLE_WORKGROUP: e.opt_single_wg = value != 0; return LCCRF_OK;
    case LCCRF_OPT_VERTEX_ORDER:                                                         // (takes effect with the next build)
        if (value < 0 || value > 2) return fail(LCCRF_E_INVALID, "LCCRF_OPT_VERTEX_ORDER takes 0 (automatic), 1 (on) or 2 (off)");
        e.opt_vertex_order = value;
        return LCCRF_OK;
    default: return fail(LCCRF_E_INVALID, "unknown option %d", option);
    }
}

// caller's arrays -> pinned staging by a few host threads (one core copies ~10 GB/s, the link takes ~55): a small process-wide pool
// of workers that sleep between batches; the calling thread takes its share of the chunks too.
class CopyPool {
public:
    struct Job { char *dst; const char *src; size_t bytes; };
    // Never throws (it sits under an extern "C" entry point): if a worker thread or the chunk list cannot be created
    // (std::system_error, std::bad_alloc) the calling thread copies the arrays itself.
    void run(const std::vector<Job> &arrays, int max_threads) noexcept
    {
        try {
            run_pooled(arrays, max_threads);
        } catch (...) {
            {
                std::lock_guard<std::mutex> g(m_);         // (nothing was published: the throwing statements come before jobs_ is set)
                jobs_ = nullptr;
            }
            for (const Job &j : arrays) memcpy(j.dst, j.src, j.bytes);
        }
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }

private:
    void run_pooled(const std::vector<Job> &arrays, int max_threads)
    {
        constexpr size_t kChunk = (size_t)512 << 10;
        std::vector<Job> chunks;
        size_t total = 0;
        for (const Job &j : arrays)
            for (size_t lo = 0; lo < j.bytes; lo += kChunk) {
                chunks.push_back({j.dst + lo, j.src + lo, std::min(kChunk, j.bytes - lo)});
                total += chunks.back().bytes;
            }
        const int want = (int)std::min<size_t>((size_t)std::max(max_threads, 1), (total + 4 * kChunk - 1) / (4 * kChunk));
        if (want <= 1 || chunks.size() < 2) {
            for (const Job &c : chunks) memcpy(c.dst, c.src, c.bytes);
            return;
        }
        std::unique_lock<std::mutex> lk(m_);
        while ((int)workers_.size() < want - 1) workers_.emplace_back([this] { worker(); });
        jobs_ = &chunks;
        next_ = 0;
        left_ = chunks.size();
        helpers_ = want - 1;
        ++epoch_;
        lk.unlock();
        cv_.notify_all();
        take();                                            // the caller works too
        lk.lock();
        done_.wait(lk, [this] { return left_ == 0 && busy_ == 0; });
        jobs_ = nullptr;
    }
    void take()
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(m_);
            if (!jobs_ || next_ >= jobs_->size()) return;
            const Job c = (*jobs_)[next_++];
            lk.unlock();
            memcpy(c.dst, c.src, c.bytes);
            lk.lock();
            if (--left_ == 0) done_.notify_all();
        }
    }
    void worker()
    {
        unsigned long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && helpers_ > 0); });
            if (stop_) return;
            seen = epoch_;
            --helpers_;
            ++busy_;
            lk.unlock();
            take();
            lk.lock();
            if (--busy_ == 0 && left_ == 0) done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::vector<Job> *jobs_ = nullptr;
    size_t next_ = 0, left_ = 0;
    int helpers_ = 0, busy_ = 0;
    unsigned long epoch_ = 0;
    bool stop_ = false;
};
// One pool (and one staging lock) per DEVICE: a host with one thread per GPU (tools/replay_multi.cpp) stages its ranks' batches side by
// side instead of queueing all of them behind one pool's workers; handles of one device still stage one at a time.
constexpr int kCopyPools = 16;
CopyPool g_copy_pool[kCopyPools];
std::mutex g_copy_pool_user[kCopyPools];

template <typename T>
int pinned_plain(Arena &mem, T **out, size_t count)      // ordinary (cached, DMA-friendly) pinned memory owned by the arena
{
    void *p = nullptr;
    HIP_TRY(hipHostMalloc(&p, (count ? count : 1) * sizeof(T), hipHostMallocDefault));
    mem.pinned.push_back(p);
    *out = static_cast<T *>(p);
    return LCCRF_OK;
}
}  // namespace

extern "C" {

int lccrf_abi_version(void) { return LCCRF_ABI_VERSION; }
const char *lccrf_last_error(void) { return g_err.c_str(); }

int lccrf_device_count(int *count)
{
    if (!count) return fail(LCCRF_E_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(LCCRF_E_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return LCCRF_OK;
}

// --------------------------------------------------------------------------------------
// object API
// --------------------------------------------------------------------------------------
int lccrf_create(lccrf_handle *out, int device_id, int n_points, int n_labels)
{
    if (!out) return fail(LCCRF_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n_points < 0) return fail(LCCRF_E_INVALID, "n_points < 0");
    if (n_labels < 1 || n_labels > LCCRF_MAX_LABELS) return fail(LCCRF_E_INVALID, "n_labels %d not in [1,%d]", n_labels, LCCRF_MAX_LABELS);
    int rc = use_device(device_id);
    if (rc) return rc;
    lccrf_crf *h = nullptr;
    {
        std::lock_guard<std::mutex> g(g_cache.m);
        for (size_t i = 0; i < g_cache.parked.size(); ++i) {
            lccrf_crf *c = g_cache.parked[i];
            if (c->eng.device == device_id && c->eng.L == n_labels && c->cap >= n_points &&
                c->cap <= std::max(4 * n_points, 4096)) {
                h = c;
                g_cache.parked.erase(g_cache.parked.begin() + i);
                break;
            }
        }
    }
    if (!h) {
        h = new (std::nothrow) lccrf_crf;
        if (!h) return fail(LCCRF_E_NOMEM, "host allocation failed");
        h->cap = capacity_for(n_points);
        rc = h->eng.init(device_id, 1, h->cap, n_labels);
        // frames far beyond SLAM's (BASELINE config 5 through the reference's own interface): inference() in locality mode
        h->eng.allow_perm = h->eng.perm_scoped = h->cap >= Engine::perm_min_points();
        if (!rc) rc = h->eng.mem.alloc_pinned(&h->stage_i16, h->cap);
        if (!rc) rc = h->eng.mem.alloc_pinned(&h->stage_f32, (size_t)h->cap * n_labels);
        if (!rc) rc = h->eng.mem.alloc_pinned(&h->stage_n, 1);
        if (!rc) rc = h->eng.mem.alloc_pinned(&h->map_pin, h->cap);
        if (rc) {
            h->eng.destroy();
            delete h;
            return rc;
        }
    }
    h->eng.ensure_parked_idle();                          // (a reused handle: its last kernel has stored its done word, or we wait for it)
    h->N = n_points;
    h->eng.opt_single_wg = g_default_single_wg.load(std::memory_order_relaxed) != 0;   // (a recycled handle does not inherit its last user's options)
    h->eng.opt_vertex_order = 0;
    h->eng.activeN = n_points;
    h->eng.crf.map = h->map_pin;
    h->eng.map_host = h->map_pin;
    h->eng.crf.map_bits = nullptr;                       // the packed copy is the batch API's gather payload only
    static const bool no_late = ab_env("LCCRF_NO_LATE") != nullptr;   // debugging aid: always size the fused kernel on the host
    h->eng.late_ok = !no_late;
    h->label_stage_busy = false;    // a parked engine's stream is idle (recycle() synchronised it or saw the frame kernel's done word, its last store)
    // the kernels read the point count where the host wrote it (pinned, device-visible): no upload command on a path
    // whose every DMA packet costs microseconds of stream time (a parked engine's stream is idle, nothing reads the old value)
    *h->stage_n = n_points;
    h->eng.crf.n_points = h->stage_n;
    if (n_points > kObjectPinnedMaxPoints) {             // (large frames: every wavefront of every kernel reads the count)
        const hipError_t ec = hipMemcpyAsync(h->eng.npoints_own, h->stage_n, sizeof(int), hipMemcpyHostToDevice, h->eng.stream);
        if (ec != hipSuccess) {                          // (the handle is neither parked nor handed out yet: do not leak it -- ADVICE r4)
            h->eng.destroy();
            delete h;
            return fail(LCCRF_E_HIP, "hipMemcpyAsync(n_points): %s", hipGetErrorString(ec));
        }
        h->eng.crf.n_points = h->eng.npoints_own;
    }
    h->eng.sync_views();
    *out = h;
    return LCCRF_OK;
}

void lccrf_destroy(lccrf_handle h)
{
    if (!h) return;
    (void)hipSetDevice(h->eng.device);
    h->eng.recycle();
    {
        std::lock_guard<std::mutex> g(g_cache.m);
        if (g_cache.parked.size() < HandleCache::kMaxParked) {
            g_cache.parked.push_back(h);
            return;
        }
    }
    h->eng.destroy();
    delete h;
}

int lccrf_set_option(lccrf_handle h, int option, int value)
{
    if (!h) return fail(LCCRF_E_INVALID, "handle is NULL");
    return apply_option(h->eng, option, value);
}

int lccrf_batch_set_option(lccrf_batch_handle b, int option, int value)
{
    if (!b) return fail(LCCRF_E_INVALID, "handle is NULL");
    if (option == LCCRF_OPT_COPY_THREADS) {
        if (value < 1 || value > 64) return fail(LCCRF_E_INVALID, "LCCRF_OPT_COPY_THREADS takes 1 .. 64");
        b->pipe.copy_threads = value;
        return LCCRF_OK;
    }
    if (option == LCCRF_OPT_EVENT_TIMING) {
        b->eng.event_timing = value != 0;
        if (!value) b->eng.timed_build = b->eng.timed_inf = false;
        return LCCRF_OK;
    }
    return apply_option(b->eng, option, value);
}

int lccrf_set_default_option(int option, int value)
{
    switch (option) {
    case LCCRF_OPT_SINGLE_WORKGROUP: g_default_single_wg.store(value != 0, std::memory_order_relaxed); return LCCRF_OK;
    default: return fail(LCCRF_E_INVALID, "option %d has no process-wide default", option);
    }
}

int lccrf_trim_cache(void)
{
    std::vector<lccrf_crf *> v;
    {
        std::lock_guard<std::mutex> g(g_cache.m);
        v.swap(g_cache.parked);
    }
    for (lccrf_crf *h : v) {
        (void)hipSetDevice(h->eng.device);
        h->eng.destroy();
        delete h;
    }
    for (int d = 0; d < kMaxDevices; ++d) {               // the pose-optimisation staging areas
        PoseStage &ps = g_pose_stage[d];
        std::lock_guard<std::mutex> g(ps.m);
        if (!ps.d && !ps.h && !ps.stream) continue;
        (void)hipSetDevice(d);
        if (ps.stream) { (void)hipStreamSynchronize(ps.stream); (void)hipStreamDestroy(ps.stream); }
        if (ps.d) (void)hipFree(ps.d);
        if (ps.h) (void)hipHostFree(ps.h);
        ps.d = ps.h = nullptr; ps.stream = nullptr; ps.cap = 0;
    }
    return (int)v.size();
}

#define CHECK_H(h)                                                    \
    do {                                                              \
        if (!(h)) return fail(LCCRF_E_INVALID, "handle is NULL");     \
        HIP_TRY(hipSetDevice((h)->eng.device));                       \
        (h)->eng.idle_by_done = (h)->eng.idle_needs_done = false;     \
    } while (0)

#define CHECK_K(h, k)                                                                      \
    do {                                                                                   \
        if ((k) < 0 || (k) >= (int)(h)->eng.kernels.size())                                \
            return fail(LCCRF_E_INVALID, "kernel index %d out of range", (k));             \
    } while (0)

int lccrf_set_unary(lccrf_handle h, const float *unary)
{
    CHECK_H(h);
    if (!unary && h->N) return fail(LCCRF_E_INVALID, "unary is NULL");
    Engine &e = h->eng;
    const size_t n = (size_t)h->N * e.L;
    { int rl = e.resolve_late(); if (rl) return rl; }
    HIP_TRY(hipStreamSynchronize(e.stream));          // staging buffer may still be in flight
    if (n) {
        memcpy(h->stage_f32, unary, n * sizeof(float));
        HIP_TRY(hipMemcpyAsync(e.unary_own, h->stage_f32, n * sizeof(float), hipMemcpyHostToDevice, e.stream));
    }
    e.crf.unary = e.unary_own;
    e.unary_deferred = false;
    e.unary_is_label = false;
    e.unary_p_valid = false;
    e.unary_set = true;
    return LCCRF_OK;
}

int lccrf_set_unary_from_label(lccrf_handle h, const int16_t *label, const float *conf)
{
    CHECK_H(h);
    if ((!label && h->N) || !conf) return fail(LCCRF_E_INVALID, "label/conf is NULL");
    Engine &e = h->eng;
    { int rl = e.resolve_late(); if (rl) return rl; }
    if (e.L < 2) return fail(LCCRF_E_INVALID, "setUnaryEnergyFromLabel needs >= 2 labels");
    if (h->label_stage_busy) HIP_TRY(hipStreamSynchronize(e.stream));   // an earlier call's kernel may still read the staging buffer
    // No uploads: the 2L+1 energies travel as a kernel argument and the kernel reads the labels from
    // pinned host memory (each tiny DMA command costs ~10 us of stream time; this path is latency-bound).
    if (h->N) memcpy(h->stage_i16, label, (size_t)h->N * sizeof(int16_t));
    const int16_t *src = h->stage_i16;
    if (h->N > kObjectPinnedMaxPoints) {                  // (large frames: many kernels, thousands of wavefronts -- see lccrf_add_pairwise)
        HIP_TRY(hipMemcpyAsync(e.label_own, h->stage_i16, (size_t)h->N * sizeof(int16_t), hipMemcpyHostToDevice, e.stream));
        src = e.label_own;
    }
    e.defer_unary_from_label(src, conf);                  // launched (or folded into the frame kernel) by the first consumer
    h->label_stage_busy = true;
    return LCCRF_OK;
}

int lccrf_add_pairwise(lccrf_handle h, const float *features, int d, float w)
{
    CHECK_H(h);
    if (!features && h->N) return fail(LCCRF_E_INVALID, "features is NULL");
    Engine &e = h->eng;
    { int rl = e.resolve_late(); if (rl) return rl; }
    int rc = e.add_kernel(d, w, true, true);
    if (rc) return rc;
    const int k = (int)e.kernels.size() - 1;
    KernelState &ks = e.kernels[k];
    const size_t n = (size_t)h->N * d;
    if (n) memcpy(ks.feat_stage, features, n * sizeof(float));   // caller may free `features` right away
    // SLAM frames: the build reads the features once, straight from this pinned buffer (no upload command on a latency-bound path).
    // Frames beyond the one-launch kernel are built by a dozen kernels of thousands of wavefronts, several of which read the
    // features -- in locality mode as a gather -- and every one of which reads the point count: over PCIe that cost a 100 000-point
    // frame 0.36 ms (C5 through this API: 1.61 -> 1.25 ms host to host), so those are uploaded once.
    ks.dev.feat = ks.feat_stage;
    if (h->N > kObjectPinnedMaxPoints && n) {
        HIP_TRY(hipMemcpyAsync(ks.feat_own, ks.feat_stage, n * sizeof(float), hipMemcpyHostToDevice, e.stream));
        ks.dev.feat = ks.feat_own;
    }
    e.sync_views();
    // The lattice is built lazily, together with any other pending kernel, by the first call that
    // needs it (inference, a step, a parity probe): one launch builds all of them side by side.
    (void)k;
    e.sizes_known = false;
    return LCCRF_OK;
}

int lccrf_add_appearance_kernel(lccrf_handle h, float w, const float *vobserv, const float *verror,
                                float sd_observ, float sd_error)
{
    CHECK_H(h);
    if ((!vobserv || !verror) && h->N) return fail(LCCRF_E_INVALID, "vobserv/verror is NULL");
    std::vector<float> f((size_t)h->N * 2 + 1);
    for (int i = 0; i < h->N; ++i) {                  // pairwise3d.h:41-44
        f[2 * i + 0] = vobserv[i] / sd_observ;
        f[2 * i + 1] = verror[i] / sd_error;
    }
    return lccrf_add_pairwise(h, f.data(), 2, w);
}

int lccrf_add_smooth_kernel(lccrf_handle h, float w, const float *xy, float sd2d)
{
    CHECK_H(h);
    if (!xy && h->N) return fail(LCCRF_E_INVALID, "xy is NULL");
    std::vector<float> f((size_t)h->N * 2 + 1);
    for (int i = 0; i < h->N; ++i) {                  // pairwise3d.h:64-66
        f[2 * i + 0] = xy[2 * i + 0] / sd2d;
        f[2 * i + 1] = xy[2 * i + 1] / sd2d;
    }
    return lccrf_add_pairwise(h, f.data(), 2, w);
}

int lccrf_start_inference(lccrf_handle h)
{
    CHECK_H(h);
    return h->eng.start();
}

int lccrf_step_inference(lccrf_handle h, float relax)
{
    CHECK_H(h);
    return h->eng.step(relax);
}

int lccrf_build_map(lccrf_handle h)
{
    CHECK_H(h);
    { int rl = h->eng.resolve_late(); if (rl) return rl; }
    launch_map(h->eng.crf, h->eng.stream);
    HIP_TRY(hipGetLastError());
    return LCCRF_OK;
}

int lccrf_inference(lccrf_handle h, int n_iterations, int with_map, float relax)
{
    CHECK_H(h);
    return h->eng.inference(n_iterations, with_map, relax);
}

// ---- the reference's plug-in points on host arrays: PairwisePotential::apply and DenseCRF's protected virtuals --------
int lccrf_pairwise_apply(lccrf_handle h, int kernel, float *out_values, const float *in_values)
{
    CHECK_H(h);
    CHECK_K(h, kernel);
    if ((!out_values || !in_values) && h->N) return fail(LCCRF_E_INVALID, "out_values / in_values is NULL");
    Engine &e = h->eng;
    int rc = e.resolve_late();
    if (!rc) rc = e.ensure_plain();
    if (!rc) rc = e.learn_sizes();                        // builds the lattice if it only exists as staged features
    if (!rc) rc = e.need_io();
    if (rc) return rc;
    const size_t n = (size_t)h->N * e.L * sizeof(float);
    if (!n) return LCCRF_OK;
    HIP_TRY(hipStreamSynchronize(e.stream));
    HIP_TRY(hipMemcpy(e.io_a, in_values, n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e.io_b, out_values, n, hipMemcpyHostToDevice));
    launch_filter(e.kdevs[kernel], e.crf, e.maxV[kernel], e.io_a, e.io_b, 1, e.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e.stream));
    HIP_TRY(hipMemcpy(out_values, e.io_b, n, hipMemcpyDeviceToHost));
    return LCCRF_OK;
}

int lccrf_exp_and_normalize(lccrf_handle h, float *out, const float *in, float scale, float relax)
{
    CHECK_H(h);
    if ((!out || !in) && h->N) return fail(LCCRF_E_INVALID, "out / in is NULL");
    Engine &e = h->eng;
    int rc = e.resolve_late();
    if (!rc) rc = e.need_io();
    if (rc) return rc;
    const size_t n = (size_t)h->N * e.L * sizeof(float);
    if (!n) return LCCRF_OK;
    HIP_TRY(hipStreamSynchronize(e.stream));
    HIP_TRY(hipMemcpy(e.io_a, in, n, hipMemcpyHostToDevice));
    if (relax != 1.0f) HIP_TRY(hipMemcpy(e.io_b, out, n, hipMemcpyHostToDevice));   // the blend reads the old out (densecrf3d.h:91-94)
    launch_exp_and_normalize(e.crf, e.io_a, e.io_b, scale, relax, e.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e.stream));
    HIP_TRY(hipMemcpy(out, e.io_b, n, hipMemcpyDeviceToHost));
    return LCCRF_OK;
}

int lccrf_step_init(lccrf_handle h, float *next_out)
{
    CHECK_H(h);
    if (!next_out && h->N) return fail(LCCRF_E_INVALID, "next_out is NULL");
    Engine &e = h->eng;
    if (!e.unary_set) return fail(LCCRF_E_STATE, "unary energies not set");
    int rc = e.resolve_late();
    if (!rc) rc = e.ensure_plain();
    if (!rc) rc = e.ensure_unary();
    if (!rc) rc = e.need_io();
    if (rc) return rc;
    const size_t n = (size_t)h->N * e.L * sizeof(float);
    if (!n) return LCCRF_OK;
    launch_step_init(e.crf, e.io_a, e.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e.stream));
    HIP_TRY(hipMemcpy(next_out, e.io_a, n, hipMemcpyDeviceToHost));
    return LCCRF_OK;
}

int lccrf_map_of(lccrf_handle h, const float *prob, int16_t *map_out)
{
    CHECK_H(h);
    if ((!prob || !map_out) && h->N) return fail(LCCRF_E_INVALID, "prob / map_out is NULL");
    Engine &e = h->eng;
    int rc = e.resolve_late();
    if (!rc) rc = e.need_io();
    if (rc) return rc;
    if (!h->N) return LCCRF_OK;
    HIP_TRY(hipStreamSynchronize(e.stream));
    HIP_TRY(hipMemcpy(e.io_a, prob, (size_t)h->N * e.L * sizeof(float), hipMemcpyHostToDevice));
    launch_map_of(e.crf, e.io_a, e.io_map, e.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e.stream));
    HIP_TRY(hipMemcpy(map_out, e.io_map, (size_t)h->N * sizeof(int16_t), hipMemcpyDeviceToHost));
    return LCCRF_OK;
}

int lccrf_lattice_filter(int device_id, const float *features, int n_points, int d, const float *in, int value_size,
                         float *out, int *n_vertices)
{
    if (n_points < 0) return fail(LCCRF_E_INVALID, "n_points < 0");
    if ((!features || !in || !out) && n_points) return fail(LCCRF_E_INVALID, "features / in / out is NULL");
    lccrf_handle h = nullptr;
    int rc = lccrf_create(&h, device_id, n_points, value_size);        // value_size plays the part of the label count
    if (rc) return rc;
    rc = lccrf_add_pairwise(h, features, d, 1.0f);
    Engine &e = h->eng;
    if (!rc) rc = e.learn_sizes();
    if (!rc) rc = e.need_io();
    const size_t n = (size_t)n_points * value_size * sizeof(float);
    hipError_t er = hipSuccess;
    if (!rc && n) {
        er = hipStreamSynchronize(e.stream);
        if (er == hipSuccess) er = hipMemcpy(e.io_a, in, n, hipMemcpyHostToDevice);
        if (er == hipSuccess) {
            launch_filter(e.kdevs[0], e.crf, e.maxV[0], e.io_a, e.io_b, 0, e.stream);
            er = hipStreamSynchronize(e.stream);
        }
        if (er == hipSuccess) er = hipMemcpy(out, e.io_b, n, hipMemcpyDeviceToHost);
        if (er != hipSuccess) rc = fail(LCCRF_E_HIP, "lccrf_lattice_filter: %s", hipGetErrorString(er));
    }
    if (!rc && n_vertices) *n_vertices = e.V_host[0];
    lccrf_destroy(h);
    return rc;
}

int lccrf_get_map(lccrf_handle h, int16_t *map_out)
{
    CHECK_H(h);
    if (!map_out && h->N) return fail(LCCRF_E_INVALID, "map_out is NULL");
    Engine &e = h->eng;
    if (e.late_pending && e.labels_armed) {             // one frame in flight, its labels go straight into map_pin: take them as they land
        e.labels_armed = false;
        const volatile int16_t *m = h->map_pin;
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = true, timed_out = false;
        unsigned spins = 0;
        for (int i = 0; i < h->N && ok; ++i) {
            int16_t v;
            while ((v = m[i]) == (int16_t)-1) {           // (bounded: whatever goes wrong is left to the ordinary path below)
                if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) { ok = false; timed_out = true; break; }
                cpu_relax();
            }
            if (v < 0) ok = false;                        // -2: the frame did not fit the one-launch kernel
        }
        if (ok) {
            std::atomic_thread_fence(std::memory_order_acquire);
            memcpy(map_out, h->map_pin, (size_t)h->N * sizeof(int16_t));
            e.late_pending = false;                       // every label is there: the frame fitted and the kernel is past its last read
            e.done_armed = false;
            // ... but not formally finished: V_out, frame_status and the done word are stored behind the labels (and a helper
            // workgroup may be in its epilogue).  The handle counts as idle only once recycle() has seen the done word.
            e.idle_by_done = true;
            e.idle_needs_done = true;
            return LCCRF_OK;
        }
        if (timed_out) e.done_armed = false;              // (a delayed frame: do not spin another 2 ms on the done word, wait on the stream)
    }
    bool seen_done = false;
    { int rl = e.resolve_late(&seen_done); if (rl) return rl; }
    if (!seen_done) HIP_TRY(hipStreamSynchronize(e.stream));
    if (h->N) memcpy(map_out, h->map_pin, (size_t)h->N * sizeof(int16_t));   // written there by the kernels
    e.idle_by_done = seen_done;
    return LCCRF_OK;
}

static int copy_out_f32(lccrf_handle h, const float *dev, float *out, size_t n)
{
    Engine &e = h->eng;
    if (n) HIP_TRY(hipMemcpyAsync(h->stage_f32, dev, n * sizeof(float), hipMemcpyDeviceToHost, e.stream));
    HIP_TRY(hipStreamSynchronize(e.stream));
    if (n) memcpy(out, h->stage_f32, n * sizeof(float));
    return LCCRF_OK;
}

int lccrf_get_probability(lccrf_handle h, float *prob_out)
{
    CHECK_H(h);
    if (!prob_out && h->N) return fail(LCCRF_E_INVALID, "prob_out is NULL");
    { int rl = h->eng.resolve_late(); if (rl) return rl; }
    return copy_out_f32(h, h->eng.crf.Q, prob_out, (size_t)h->N * h->eng.L);
}

int lccrf_get_unary(lccrf_handle h, float *unary_out)
{
    CHECK_H(h);
    if (!unary_out && h->N) return fail(LCCRF_E_INVALID, "unary_out is NULL");
    { int rl = h->eng.resolve_late(); if (rl) return rl; }
    { int rp = h->eng.ensure_plain(); if (rp) return rp; }           // (energies derived from labels live in the lattices' point order)
    { int ru = h->eng.ensure_unary(); if (ru) return ru; }
    return copy_out_f32(h, h->eng.crf.unary, unary_out, (size_t)h->N * h->eng.L);
}

int lccrf_get_lattice_size(lccrf_handle h, int kernel, int *n_vertices)
{
    CHECK_H(h);
    CHECK_K(h, kernel);
    if (!n_vertices) return fail(LCCRF_E_INVALID, "n_vertices is NULL");
    { int rcf = h->eng.flush_builds(); if (rcf) return rcf; }
    HIP_TRY(hipStreamSynchronize(h->eng.stream));
    *n_vertices = h->eng.V_host[(size_t)kernel * h->eng.Fcap];
    return LCCRF_OK;
}

int lccrf_get_norm(lccrf_handle h, int kernel, float *norm_out)
{
    CHECK_H(h);
    CHECK_K(h, kernel);
    if (!norm_out && h->N) return fail(LCCRF_E_INVALID, "norm_out is NULL");
    { int rp = h->eng.ensure_plain(); if (rp) return rp; }
    { int rcf = h->eng.flush_builds(); if (rcf) return rcf; }
    HIP_TRY(hipStreamSynchronize(h->eng.stream));
    if (h->N) HIP_TRY(hipMemcpy(norm_out, h->eng.kernels[kernel].dev.norm, (size_t)h->N * sizeof(float), hipMemcpyDeviceToHost));
    return LCCRF_OK;
}

int lccrf_get_lattice(lccrf_handle h, int kernel, int32_t *offset_out, float *bary_out, int32_t *nbr_out)
{
    CHECK_H(h);
    CHECK_K(h, kernel);
    Engine &e = h->eng;
    { int rp = e.ensure_plain(); if (rp) return rp; }
    { int rcf = e.flush_builds(); if (rcf) return rcf; }
    HIP_TRY(hipStreamSynchronize(e.stream));
    const KernelDev &k = e.kernels[kernel].dev;
    const size_t ne = (size_t)h->N * k.D1;
    if (offset_out && ne) HIP_TRY(hipMemcpy(offset_out, k.offset, ne * sizeof(int), hipMemcpyDeviceToHost));
    if (bary_out && ne) HIP_TRY(hipMemcpy(bary_out, k.bary, ne * sizeof(float), hipMemcpyDeviceToHost));
    if (nbr_out) {
        const int V = e.V_host[(size_t)kernel * e.Fcap];
        for (int j = 0; j < k.D1 && V; ++j)           // device rows are strided by Epad, output by V
            HIP_TRY(hipMemcpy(nbr_out + (size_t)j * V * 2, k.nbr + (size_t)j * k.Epad * 2, (size_t)V * 2 * sizeof(int),
                              hipMemcpyDeviceToHost));
    }
    return LCCRF_OK;
}

// --------------------------------------------------------------------------------------
// batch API
// --------------------------------------------------------------------------------------
int lccrf_batch_create(lccrf_batch_handle *out, int device_id, const lccrf_batch_desc *desc)
{
    if (!out || !desc) return fail(LCCRF_E_INVALID, "out/desc is NULL");
    *out = nullptr;
    if (desc->max_frames < 1 || desc->max_points < 0) return fail(LCCRF_E_INVALID, "max_frames < 1 or max_points < 0");
    if (desc->n_labels < 1 || desc->n_labels > LCCRF_MAX_LABELS) return fail(LCCRF_E_INVALID, "n_labels out of range");
    if (desc->n_kernels < 0 || desc->n_kernels > LCCRF_MAX_KERNELS) return fail(LCCRF_E_INVALID, "n_kernels out of range");
    int rc = use_device(device_id);
    if (rc) return rc;
    lccrf_batch *b = new (std::nothrow) lccrf_batch;
    if (!b) return fail(LCCRF_E_NOMEM, "host allocation failed");
    b->desc = *desc;
    b->eng.allow_perm = true;
    b->eng.opt_single_wg = g_default_single_wg.load(std::memory_order_relaxed) != 0;
    rc = b->eng.init(device_id, desc->max_frames, desc->max_points, desc->n_labels);
    for (int k = 0; k < desc->n_kernels && !rc; ++k) rc = b->eng.add_kernel(desc->feat_dims[k], desc->weights[k], true, false);
    if (!rc && hipStreamSynchronize(b->eng.stream) != hipSuccess)      // every allocation is zeroed before the handle is handed out
        rc = fail(LCCRF_E_HIP, "hipStreamSynchronize after allocation failed");
    if (rc) {
        b->eng.destroy();
        delete b;
        return rc;
    }
    *out = b;
    return LCCRF_OK;
}

void lccrf_batch_destroy(lccrf_batch_handle b)
{
    if (!b) return;
    (void)hipSetDevice(b->eng.device);
    if (b->pipe.up) (void)hipStreamSynchronize(b->pipe.up);
    if (b->pipe.down) (void)hipStreamSynchronize(b->pipe.down);
    b->eng.destroy();
    b->pipe.destroy();
    delete b;
}

static int batch_common_inputs(lccrf_batch *b, int n_frames, const float *conf, bool have_unary, bool have_label)
{
    Engine &e = b->eng;
    if (n_frames < 1 || n_frames > e.Fcap) return fail(LCCRF_E_CAPACITY, "n_frames %d not in [1,%d]", n_frames, e.Fcap);
    if (have_unary == have_label) return fail(LCCRF_E_INVALID, "exactly one of unary / label must be given");
    if (have_label && (!conf || e.L < 2)) return fail(LCCRF_E_INVALID, "label input needs conf[n_labels] and >= 2 labels");
    { int rl = e.resolve_late(); if (rl) return rl; }
    e.F = n_frames;
    e.sync_views();
    e.unary_set = false;
    e.built = false;
    e.built_upto = 0;
    e.sizes_known = false;
    e.started = false;
    return LCCRF_OK;
}

int lccrf_batch_set_inputs_host(lccrf_batch_handle b, int n_frames, const int32_t *n_points, const float *unary,
                                const int16_t *label, const float *conf, const float *const *features)
{
    CHECK_H(b);
    if (!n_points) return fail(LCCRF_E_INVALID, "n_points is NULL");
    if (b->desc.n_kernels && !features) return fail(LCCRF_E_INVALID, "features is NULL");
    int rc = batch_common_inputs(b, n_frames, conf, unary != nullptr, label != nullptr);
    if (rc) return rc;
    Engine &e = b->eng;
    for (int f = 0; f < n_frames; ++f)
        if (n_points[f] < 0 || n_points[f] > e.maxN) return fail(LCCRF_E_CAPACITY, "n_points[%d]=%d not in [0,%d]", f, n_points[f], e.maxN);
    e.activeN = 0;
    for (int f = 0; f < n_frames; ++f) e.activeN = std::max(e.activeN, n_points[f]);
    HIP_TRY(hipStreamSynchronize(e.stream));
    HIP_TRY(hipMemcpy(e.npoints_own, n_points, sizeof(int) * n_frames, hipMemcpyHostToDevice));
    e.crf.n_points = e.npoints_own;
    e.crf.unary = e.unary_own;
    const size_t per = (size_t)e.maxN;
    if (unary) {
        HIP_TRY(hipMemcpy(e.unary_own, unary, sizeof(float) * n_frames * per * e.L, hipMemcpyHostToDevice));
        e.unary_deferred = false;
        e.unary_is_label = false;
        e.unary_p_valid = false;
    } else {
        HIP_TRY(hipMemcpy(e.label_own, label, sizeof(int16_t) * n_frames * per, hipMemcpyHostToDevice));
        e.defer_unary_from_label(e.label_own, conf);
    }
    for (int k = 0; k < b->desc.n_kernels; ++k) {
        if (!features[k]) return fail(LCCRF_E_INVALID, "features[%d] is NULL", k);
        KernelState &ks = e.kernels[k];
        HIP_TRY(hipMemcpy(ks.feat_own, features[k], sizeof(float) * n_frames * per * ks.dev.d, hipMemcpyHostToDevice));
        ks.dev.feat = ks.feat_own;
    }
    e.sync_views();
    e.unary_set = true;
    b->inputs_set = true;
    return LCCRF_OK;
}

static int pipe_refresh_after_rerun(void *ctx);

static int pipe_init(lccrf_batch *b)
{
    HostPipe &p = b->pipe;
    if (p.up) return LCCRF_OK;
    Engine &e = b->eng;
    e.after_rerun = pipe_refresh_after_rerun;
    e.after_rerun_ctx = b;
    HIP_TRY(hipStreamCreateWithFlags(&p.up, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&p.down, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_up, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_down, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_q, hipEventDisableTiming));
    return LCCRF_OK;
}

int lccrf_batch_set_inputs_host_async(lccrf_batch_handle b, int n_frames, const int32_t *n_points, const float *unary,
                                      const int16_t *label, const float *conf, const float *const *features, int flags)
{
    CHECK_H(b);
    if (!n_points) return fail(LCCRF_E_INVALID, "n_points is NULL");
    if (b->desc.n_kernels && !features) return fail(LCCRF_E_INVALID, "features is NULL");
    if (flags & ~LCCRF_HOST_PINNED) return fail(LCCRF_E_INVALID, "unknown flags 0x%x", flags);
    for (int k = 0; k < b->desc.n_kernels; ++k)
        if (!features[k]) return fail(LCCRF_E_INVALID, "features[%d] is NULL", k);
    int rc = batch_common_inputs(b, n_frames, conf, unary != nullptr, label != nullptr);   // (settles what the previous batch on this handle left pending)
    if (rc) return rc;
    Engine &e = b->eng;
    HostPipe &p = b->pipe;
    for (int f = 0; f < n_frames; ++f)
        if (n_points[f] < 0 || n_points[f] > e.maxN) return fail(LCCRF_E_CAPACITY, "n_points[%d]=%d not in [0,%d]", f, n_points[f], e.maxN);
    e.activeN = 0;
    for (int f = 0; f < n_frames; ++f) e.activeN = std::max(e.activeN, n_points[f]);
    if ((rc = pipe_init(b))) return rc;
    const bool direct = (flags & LCCRF_HOST_PINNED) != 0;
    const size_t per = (size_t)e.maxN, Fz = (size_t)n_frames, Fc = (size_t)e.Fcap;
    if (p.up_pending) {                                // the staging set is still the source of the previous upload (normally long done)
        HIP_TRY(hipEventSynchronize(p.ev_up));
        p.up_pending = false;
    }
    const int32_t *src_np = n_points;
    const float *src_un = unary;
    const int16_t *src_lb = label;
    std::vector<const float *> src_ft(features, features + b->desc.n_kernels);
    // (the point counts are always staged: a few bytes per frame, and the caller's array is usually a local one)
    if (!p.npoints && (rc = pinned_plain(e.mem, &p.npoints, Fc))) return rc;
    memcpy(p.npoints, n_points, sizeof(int) * Fz);
    src_np = p.npoints;
    if (!direct) {                                     // copy out of the caller's buffers before returning
        std::vector<CopyPool::Job> jobs;
        if (unary) {
            if (!p.unary && (rc = pinned_plain(e.mem, &p.unary, Fc * per * e.L))) return rc;
            jobs.push_back({(char *)p.unary, (const char *)unary, sizeof(float) * Fz * per * e.L});
            src_un = p.unary;
        } else {
            if (!p.label && (rc = pinned_plain(e.mem, &p.label, Fc * per))) return rc;
            jobs.push_back({(char *)p.label, (const char *)label, sizeof(int16_t) * Fz * per});
            src_lb = p.label;
        }
        p.feat.resize(b->desc.n_kernels, nullptr);
        for (int k = 0; k < b->desc.n_kernels; ++k) {
            const int d = e.kernels[k].dev.d;
            if (!p.feat[k] && (rc = pinned_plain(e.mem, &p.feat[k], Fc * per * d))) return rc;
            jobs.push_back({(char *)p.feat[k], (const char *)features[k], sizeof(float) * Fz * per * d});
            src_ft[k] = p.feat[k];
        }
        const int pool = (e.device >= 0 ? e.device : 0) % kCopyPools;
        std::lock_guard<std::mutex> g(g_copy_pool_user[pool]);
        g_copy_pool[pool].run(jobs, p.copy_threads);
    }
    // the upload waits for everything queued on the batch's stream so far (a kernel of the previous batch may still read the
    // device copies), and whatever is queued from here on waits for the upload
    HIP_TRY(hipEventRecord(p.ev_q, e.stream));
    HIP_TRY(hipStreamWaitEvent(p.up, p.ev_q, 0));
    HIP_TRY(hipMemcpyAsync(e.npoints_own, src_np, sizeof(int) * Fz, hipMemcpyHostToDevice, p.up));
    e.crf.n_points = e.npoints_own;
    e.crf.unary = e.unary_own;
    if (unary) {
        HIP_TRY(hipMemcpyAsync(e.unary_own, src_un, sizeof(float) * Fz * per * e.L, hipMemcpyHostToDevice, p.up));
        e.unary_deferred = false;
        e.unary_is_label = false;
        e.unary_p_valid = false;
    } else {
        HIP_TRY(hipMemcpyAsync(e.label_own, src_lb, sizeof(int16_t) * Fz * per, hipMemcpyHostToDevice, p.up));
        e.defer_unary_from_label(e.label_own, conf);
    }
    for (int k = 0; k < b->desc.n_kernels; ++k) {
        KernelState &ks = e.kernels[k];
        HIP_TRY(hipMemcpyAsync(ks.feat_own, src_ft[k], sizeof(float) * Fz * per * ks.dev.d, hipMemcpyHostToDevice, p.up));
        ks.dev.feat = ks.feat_own;
    }
    HIP_TRY(hipEventRecord(p.ev_up, p.up));
    HIP_TRY(hipStreamWaitEvent(e.stream, p.ev_up, 0));
    p.up_pending = true;
    e.sync_views();
    e.unary_set = true;
    b->inputs_set = true;
    return LCCRF_OK;
}

int lccrf_batch_wait_inputs(lccrf_batch_handle b)
{
    CHECK_H(b);
    if (b->pipe.up_pending) {
        HIP_TRY(hipEventSynchronize(b->pipe.ev_up));
        b->pipe.up_pending = false;
    }
    return LCCRF_OK;
}

static int pipe_queue_download(lccrf_batch *b, hipStream_t on)
{
    Engine &e = b->eng;
    HostPipe &p = b->pipe;
    const size_t Fz = (size_t)p.down_frames, per = (size_t)e.maxN;
    if (p.down_what & LCCRF_DOWNLOAD_LABEL_BITS)
        HIP_TRY(hipMemcpyAsync(p.bits, e.crf.map_bits, sizeof(uint64_t) * Fz * e.crf.bits_stride, hipMemcpyDeviceToHost, on));
    if (p.down_what & LCCRF_DOWNLOAD_MAP)
        HIP_TRY(hipMemcpyAsync(p.map, e.crf.map, sizeof(int16_t) * Fz * per, hipMemcpyDeviceToHost, on));
    if (p.down_what & LCCRF_DOWNLOAD_PROBABILITY)
        HIP_TRY(hipMemcpyAsync(p.prob, e.crf.Q, sizeof(float) * Fz * per * e.L, hipMemcpyDeviceToHost, on));
    return LCCRF_OK;
}

// Engine::after_rerun of a batch with a download pipe: a queued download copied the results BEFORE the flagged frames were re-run
// (whoever settled them -- lccrf_batch_get_fallback_frames, _synchronize, the next batch's inputs, ...): queue the copies again behind
// the re-run on the batch's stream and move the "download landed" event there, so lccrf_batch_wait_download waits for the fresh ones.
static int pipe_refresh_after_rerun(void *ctx)
{
    lccrf_batch *b = static_cast<lccrf_batch *>(ctx);
    HostPipe &p = b->pipe;
    if (!p.down_pending) return LCCRF_OK;
    HIP_TRY(hipEventSynchronize(p.ev_down));           // (the first copies still own the pinned arrays)
    int rc = pipe_queue_download(b, b->eng.stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(p.ev_down, b->eng.stream));
    return LCCRF_OK;
}

int lccrf_batch_download_async(lccrf_batch_handle b, int what)
{
    CHECK_H(b);
    Engine &e = b->eng;
    HostPipe &p = b->pipe;
    if (!what || (what & ~(LCCRF_DOWNLOAD_LABEL_BITS | LCCRF_DOWNLOAD_MAP | LCCRF_DOWNLOAD_PROBABILITY)))
        return fail(LCCRF_E_INVALID, "what = 0x%x: a combination of LCCRF_DOWNLOAD_*", what);
    if ((what & LCCRF_DOWNLOAD_LABEL_BITS) && !e.crf.map_bits) return fail(LCCRF_E_INVALID, "label bits exist for n_labels == 2 only");
    int rc = pipe_init(b);
    if (rc) return rc;
    if (p.down_pending) {                              // (the host copies of the previous download are about to be overwritten)
        HIP_TRY(hipEventSynchronize(p.ev_down));
        p.down_pending = false;
    }
    const size_t Fc = (size_t)e.Fcap, per = (size_t)e.maxN;
    if ((what & LCCRF_DOWNLOAD_LABEL_BITS) && !p.bits && (rc = pinned_plain(e.mem, &p.bits, Fc * std::max(e.crf.bits_stride, 1)))) return rc;
    if ((what & LCCRF_DOWNLOAD_MAP) && !p.map && (rc = pinned_plain(e.mem, &p.map, Fc * per))) return rc;
    if ((what & LCCRF_DOWNLOAD_PROBABILITY) && !p.prob && (rc = pinned_plain(e.mem, &p.prob, Fc * per * e.L))) return rc;
    p.down_what = what;
    p.down_frames = e.F;
    // behind everything queued on the batch's stream (calls on a caller's stream are ordered into it when they return), on the
    // download stream; the batch's stream in turn waits for the copies before anything may overwrite the results
    HIP_TRY(hipEventRecord(p.ev_q, e.stream));
    HIP_TRY(hipStreamWaitEvent(p.down, p.ev_q, 0));
    if ((rc = pipe_queue_download(b, p.down))) return rc;
    HIP_TRY(hipEventRecord(p.ev_down, p.down));
    HIP_TRY(hipStreamWaitEvent(e.stream, p.ev_down, 0));
    p.down_pending = true;
    return LCCRF_OK;
}

int lccrf_batch_wait_download(lccrf_batch_handle b, const uint64_t **label_bits, int *words_per_frame, const int16_t **map,
                              const float **probability)
{
    CHECK_H(b);
    Engine &e = b->eng;
    HostPipe &p = b->pipe;
    if (!p.down_pending) return fail(LCCRF_E_STATE, "no download queued (lccrf_batch_download_async)");
    HIP_TRY(hipEventSynchronize(p.ev_down));
    // a one-launch run may have flagged frames that did not fit: they are re-run at the first synchronisation point behind the launch
    // -- here, or in any call made since lccrf_batch_download_async -- and the re-run queues the copies again (pipe_refresh_after_rerun)
    int rc = e.resolve_late();
    if (rc) { p.down_pending = false; return rc; }
    HIP_TRY(hipEventSynchronize(p.ev_down));           // (re-recorded behind a re-run; otherwise already complete)
    p.down_pending = false;
    if (label_bits) *label_bits = (p.down_what & LCCRF_DOWNLOAD_LABEL_BITS) ? p.bits : nullptr;
    if (words_per_frame) *words_per_frame = e.crf.bits_stride;
    if (map) *map = (p.down_what & LCCRF_DOWNLOAD_MAP) ? p.map : nullptr;
    if (probability) *probability = (p.down_what & LCCRF_DOWNLOAD_PROBABILITY) ? p.prob : nullptr;
    return LCCRF_OK;
}

int lccrf_batch_bind_inputs_device(lccrf_batch_handle b, int n_frames, const int32_t *d_n_points, const float *d_unary,
                                   const int16_t *d_label, const float *conf, const float *const *d_features)
{
    CHECK_H(b);
    if (!d_n_points) return fail(LCCRF_E_INVALID, "d_n_points is NULL");
    if (b->desc.n_kernels && !d_features) return fail(LCCRF_E_INVALID, "d_features is NULL");
    int rc = batch_common_inputs(b, n_frames, conf, d_unary != nullptr, d_label != nullptr);
    if (rc) return rc;
    Engine &e = b->eng;
    e.activeN = 0;                                    // per-frame sizes live on the device: unknown here
    // the kernels index every per-frame array with n_points[f]: work on a validated copy (clamped to
    // [0, max_points]; an out-of-range entry raises LCCRF_E_CAPACITY at the next synchronisation point)
    launch_validate_npoints(d_n_points, e.npoints_own, n_frames, e.maxN, e.npoints_bad, e.stream);
    e.crf.n_points = e.npoints_own;
    if (d_unary) {
        e.crf.unary = const_cast<float *>(d_unary);   // read-only use
        e.unary_deferred = false;
        e.unary_is_label = false;
        e.unary_p_valid = false;
    } else {
        e.defer_unary_from_label(d_label, conf);
    }
    for (int k = 0; k < b->desc.n_kernels; ++k) {
        if (!d_features[k]) return fail(LCCRF_E_INVALID, "d_features[%d] is NULL", k);
        e.kernels[k].dev.feat = d_features[k];
    }
    e.sync_views();
    HIP_TRY(hipGetLastError());
    e.unary_set = true;
    b->inputs_set = true;
    return LCCRF_OK;
}

// Runs the body of a batch call on a caller-supplied stream.  The engine's own stream carries the
// zeroing of fresh allocations (Arena::alloc) and the kernels of lccrf_batch_bind_inputs_device, and
// nothing else orders a foreign stream against it: entering makes the caller's stream wait for
// everything queued on the own stream so far, leaving makes the own stream (read-backs, later
// calls) wait for the caller's.  The engine's stream is restored on every exit path.
struct StreamScope {
    Engine &e;
    hipStream_t own, use;
    StreamScope(Engine &eng, void *stream) : e(eng), own(eng.stream), use(stream ? (hipStream_t)stream : eng.stream) {}
    int enter()
    {
        if (use != own) {
            HIP_TRY(hipEventRecord(e.ev_order, own));
            HIP_TRY(hipStreamWaitEvent(use, e.ev_order, 0));
        }
        e.stream = e.mem.stream = use;                    // lazy allocations of this call are zeroed on the stream its kernels run on
        return LCCRF_OK;
    }
    ~StreamScope()
    {
        e.stream = e.mem.stream = own;
        if (use != own) {
            (void)hipEventRecord(e.ev_order, use);
            (void)hipStreamWaitEvent(own, e.ev_order, 0);
        }
    }
};

int lccrf_batch_build(lccrf_batch_handle b, void *stream)
{
    CHECK_H(b);
    if (!b->inputs_set) return fail(LCCRF_E_STATE, "inputs not set");
    Engine &e = b->eng;
    StreamScope scope(e, stream);
    int rc = scope.enter();
    if (rc) return rc;
    if (e.event_timing) HIP_TRY(hipEventRecord(e.ev[0], e.stream));
    if (!e.kernels.empty()) rc = e.build_kernels(0, (int)e.kernels.size());
    e.built_upto = (int)e.kernels.size();
    if (!rc && e.event_timing) {
        hipError_t er = hipEventRecord(e.ev[1], e.stream);
        if (er != hipSuccess) rc = fail(LCCRF_E_HIP, "hipEventRecord: %s", hipGetErrorString(er));
    }
    e.timed_build = !rc && e.event_timing;
    e.built = !rc;
    return rc;                                        // (learn_sizes() waits on the own stream, which ~StreamScope orders behind this one)
}

int lccrf_batch_inference(lccrf_batch_handle b, int n_iterations, int with_map, float relax, void *stream)
{
    CHECK_H(b);
    Engine &e = b->eng;
    if (!e.built) return fail(LCCRF_E_STATE, "lccrf_batch_build has not run for these inputs");
    int rc = e.learn_sizes();
    if (rc) return rc;
    StreamScope scope(e, stream);
    if ((rc = scope.enter())) return rc;
    if (e.event_timing) HIP_TRY(hipEventRecord(e.ev[2], e.stream));
    rc = e.inference(n_iterations, with_map, relax);
    if (!rc && e.event_timing) {
        hipError_t er = hipEventRecord(e.ev[3], e.stream);
        if (er != hipSuccess) rc = fail(LCCRF_E_HIP, "hipEventRecord: %s", hipGetErrorString(er));
    }
    e.timed_inf = !rc && e.event_timing;
    return rc;
}

int lccrf_batch_run(lccrf_batch_handle b, int n_iterations, int with_map, float relax, void *stream)
{
    CHECK_H(b);
    if (!b->inputs_set) return fail(LCCRF_E_STATE, "inputs not set");
    Engine &e = b->eng;
    StreamScope scope(e, stream);
    int rc = scope.enter();
    if (rc) return rc;
    if (e.event_timing) HIP_TRY(hipEventRecord(e.ev[2], e.stream));
    rc = e.run(n_iterations, with_map, relax);
    if (!rc && e.event_timing) {
        hipError_t er = hipEventRecord(e.ev[3], e.stream);
        if (er != hipSuccess) rc = fail(LCCRF_E_HIP, "hipEventRecord: %s", hipGetErrorString(er));
    }
    e.timed_inf = !rc && e.event_timing;
    e.built = e.built_upto == (int)e.kernels.size() && !e.kernels.empty();
    return rc;
}

int lccrf_batch_synchronize(lccrf_batch_handle b)
{
    CHECK_H(b);
    // Scoped to THIS batch: its stream (calls made on a caller's stream are ordered into it before they return) and its two copy
    // streams.  Not hipDeviceSynchronize: a host that keeps several batches in flight (tools/replay_multi.cpp, tools/host_pipeline.cpp)
    // settles one of them while the others' uploads and kernels keep running.
    HIP_TRY(hipStreamSynchronize(b->eng.stream));
    if (b->pipe.up) HIP_TRY(hipStreamSynchronize(b->pipe.up));
    if (b->pipe.down) HIP_TRY(hipStreamSynchronize(b->pipe.down));
    return b->eng.resolve_late();
}

int lccrf_batch_get_map_host(lccrf_batch_handle b, int16_t *map_out)
{
    CHECK_H(b);
    if (!map_out) return fail(LCCRF_E_INVALID, "map_out is NULL");
    Engine &e = b->eng;
    { int rl = b->eng.resolve_late(); if (rl) return rl; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(map_out, e.crf.map, sizeof(int16_t) * (size_t)e.F * e.maxN, hipMemcpyDeviceToHost));
    return LCCRF_OK;
}

int lccrf_batch_get_probability_host(lccrf_batch_handle b, float *prob_out)
{
    CHECK_H(b);
    if (!prob_out) return fail(LCCRF_E_INVALID, "prob_out is NULL");
    Engine &e = b->eng;
    { int rl = b->eng.resolve_late(); if (rl) return rl; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(prob_out, e.crf.Q, sizeof(float) * (size_t)e.F * e.maxN * e.L, hipMemcpyDeviceToHost));
    return LCCRF_OK;
}

int lccrf_batch_get_lattice_sizes_host(lccrf_batch_handle b, int kernel, int32_t *n_vertices_out)
{
    CHECK_H(b);
    CHECK_K(b, kernel);
    if (!n_vertices_out) return fail(LCCRF_E_INVALID, "n_vertices_out is NULL");
    Engine &e = b->eng;
    { int rl = b->eng.resolve_late(); if (rl) return rl; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(n_vertices_out, e.kernels[kernel].dev.V, sizeof(int) * e.F, hipMemcpyDeviceToHost));
    return LCCRF_OK;
}

int lccrf_batch_get_norm_host(lccrf_batch_handle b, int kernel, float *norm_out)
{
    CHECK_H(b);
    CHECK_K(b, kernel);
    if (!norm_out) return fail(LCCRF_E_INVALID, "norm_out is NULL");
    Engine &e = b->eng;
    { int rl = b->eng.resolve_late(); if (rl) return rl; }
    { int rf = b->eng.flush_builds(); if (rf) return rf; }   // lccrf_batch_run builds nothing in HBM
    const float *src = e.kernels[kernel].dev.norm;
    if (e.perm_on) {                                  // the lattice's point arrays are in the internal order: undo it for the caller
        int rc = e.need_io();
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(e.io_a, 0, sizeof(float) * (size_t)e.F * e.maxN, e.stream));
        launch_permute_rows(e.crf, e.io_a, src, 1, 0, e.stream);
        HIP_TRY(hipGetLastError());
        src = e.io_a;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(norm_out, src, sizeof(float) * (size_t)e.F * e.maxN, hipMemcpyDeviceToHost));
    return LCCRF_OK;
}

int lccrf_batch_device_buffers(lccrf_batch_handle b, const int16_t **d_map, const float **d_prob)
{
    CHECK_H(b);
    if (d_map) *d_map = b->eng.crf.map;
    if (d_prob) *d_prob = b->eng.crf.Q;
    return LCCRF_OK;
}

int lccrf_batch_device_label_bits(lccrf_batch_handle b, const uint64_t **d_bits, int *words_per_frame)
{
    CHECK_H(b);
    if (!d_bits || !words_per_frame) return fail(LCCRF_E_INVALID, "NULL output");
    if (!b->eng.crf.map_bits) return fail(LCCRF_E_INVALID, "packed labels exist for binary CRFs (n_labels == 2) only");
    *d_bits = reinterpret_cast<const uint64_t *>(b->eng.crf.map_bits);
    *words_per_frame = b->eng.crf.bits_stride;
    return LCCRF_OK;
}

int lccrf_batch_set_engine(lccrf_batch_handle b, int engine)
{
    CHECK_H(b);
    if (engine < 0 || engine > 2) return fail(LCCRF_E_INVALID, "engine must be 0, 1 or 2");
    b->eng.engine_pref = engine;
    b->eng.sizes_known = false;
    return LCCRF_OK;
}

int lccrf_batch_get_engine(lccrf_batch_handle b, int *engine_in_use)
{
    CHECK_H(b);
    if (!engine_in_use) return fail(LCCRF_E_INVALID, "engine_in_use is NULL");
    { int rl = b->eng.resolve_late(); if (rl) return rl; }   // a one-launch run may have fallen back
    Engine &e = b->eng;
    if (e.built && e.engine_used != 3) {                     // lattices in HBM: what an inference on them runs (or ran) on
        int rc = e.learn_sizes();
        if (rc) return rc;
        e.engine_used = e.sized_engine;
    }
    *engine_in_use = e.engine_used;
    return LCCRF_OK;
}

int lccrf_batch_get_fused_shape(lccrf_batch_handle b, int *lanes_per_frame, int *frames_per_cu)
{
    CHECK_H(b);
    if (!lanes_per_frame || !frames_per_cu) return fail(LCCRF_E_INVALID, "output pointer is NULL");
    *lanes_per_frame = b->eng.fused_shape & 0xffff;
    *frames_per_cu = b->eng.fused_shape >> 16;
    return LCCRF_OK;
}

int lccrf_batch_get_locality_mode(lccrf_batch_handle b, int *internal_point_order, int *sorted_build)
{
    CHECK_H(b);
    Engine &e = b->eng;
    if (e.built) {                                      // (a build that met an overflowing / wrapping frame is redone with the hash here)
        int rc = e.learn_sizes();
        if (rc) return rc;
    }
    if (internal_point_order) *internal_point_order = e.perm_on ? 1 : 0;
    if (sorted_build) *sorted_build = (e.perm_on && e.vorder_on && !e.vorder_broken) ? 1 : 0;
    return LCCRF_OK;
}

int lccrf_batch_get_fallback_frames(lccrf_batch_handle b, int *n_frames)
{
    CHECK_H(b);
    if (!n_frames) return fail(LCCRF_E_INVALID, "n_frames is NULL");
    { int rl = b->eng.resolve_late(); if (rl) return rl; }
    *n_frames = b->eng.fallback_frames;
    return LCCRF_OK;
}

void lccrf_default_params(lccrf_crf_params *p)       // Examples/RGB-D/TUM3.yaml:78-101
{
    if (!p) return;
    p->w1 = 10.0f; p->w2 = 30.0f;
    p->u_alpha = 1.7f; p->stdev_alpha = 0.6f;
    p->u_beta = 5.4f; p->stdev_beta = 1.5f;
    p->u_gamma = 0.3f; p->stdev_gamma = 0.2f;
    p->point3d_stdev = 0.5f; p->point2d_stdev = 18.0f;
    p->u_depth = 2.75f; p->pth = 0.8f; p->confidence = 0.7f;
}

int lccrf_unary_build(int device_id, int n_points, const float *Xw, const int32_t *obs_ptr, const int32_t *obs_kf,
                      const double *obs_kp, int n_kf, const float *kf_pose, const float *kf_intr, const float *kf_bounds,
                      const double *match_prob, const lccrf_crf_params *params, float *observs_out, float *error_out,
                      float *depth_out, int16_t *label_out)
{
    if (n_points < 0 || n_kf < 0) return fail(LCCRF_E_INVALID, "negative size");
    if (!params || !obs_ptr) return fail(LCCRF_E_INVALID, "params / obs_ptr is NULL");
    if (n_points && (!Xw || !observs_out || !error_out || !depth_out || !label_out))
        return fail(LCCRF_E_INVALID, "NULL array");
    const int n_obs = n_points ? obs_ptr[n_points] : 0;
    if (n_obs < 0 || (n_obs && (!obs_kf || !obs_kp || !kf_pose || !kf_intr || !kf_bounds)))
        return fail(LCCRF_E_INVALID, "observation arrays missing");
    for (int i = 0; i < n_points; ++i)
        if (obs_ptr[i + 1] < obs_ptr[i]) return fail(LCCRF_E_INVALID, "obs_ptr not monotone at %d", i);
    for (int o = 0; o < n_obs; ++o)
        if (obs_kf[o] < 0 || obs_kf[o] >= n_kf) return fail(LCCRF_E_INVALID, "obs_kf[%d]=%d out of range", o, obs_kf[o]);
    int rc = use_device(device_id);
    if (rc) return rc;
    hipError_t e = run_unary_build(device_id, n_points, Xw, obs_ptr, obs_kf, obs_kp, n_kf, kf_pose, kf_intr, kf_bounds,
                                   match_prob, params, observs_out, error_out, depth_out, label_out);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? LCCRF_E_NOMEM : LCCRF_E_HIP, "unary builder: %s", hipGetErrorString(e));
    return LCCRF_OK;
}

int lccrf_bf_match(int device_id, int n_query, const uint8_t *desc_query, int n_train, const uint8_t *desc_train,
                   double ratio, int32_t *train_of_query_out, int32_t *n_matches_out)
{
    if (n_query < 0 || n_train < 0) return fail(LCCRF_E_INVALID, "negative size");
    if (n_train >= (1 << 22)) return fail(LCCRF_E_CAPACITY, "at most %d train descriptors", (1 << 22) - 1);
    if ((n_query && (!desc_query || !train_of_query_out)) || (n_train && !desc_train))
        return fail(LCCRF_E_INVALID, "NULL descriptor / output array");
    if (!(ratio >= 0.0)) return fail(LCCRF_E_INVALID, "ratio must be >= 0");
    int rc = use_device(device_id);
    if (rc) return rc;
    hipError_t e = run_bf_match(device_id, n_query, desc_query, n_train, desc_train, ratio, train_of_query_out, n_matches_out);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? LCCRF_E_NOMEM : LCCRF_E_HIP, "bf_match: %s", hipGetErrorString(e));
    return LCCRF_OK;
}

int lccrf_batch_time_blur_pass(lccrf_batch_handle b, int kernel, int reps, float *ms_per_launch, int64_t *vertices_per_launch)
{
    CHECK_H(b);
    CHECK_K(b, kernel);
    if (reps < 1 || !ms_per_launch) return fail(LCCRF_E_INVALID, "reps < 1 or NULL output");
    Engine &e = b->eng;
    if (!e.built) return fail(LCCRF_E_STATE, "lccrf_batch_build has not run for these inputs");
    int rc = e.learn_sizes();
    if (rc) return rc;
    if (vertices_per_launch) {
        int64_t tot = 0;
        for (int f = 0; f < e.F; ++f) tot += e.V_host[(size_t)kernel * e.Fcap + f];
        *vertices_per_launch = tot;
    }
    hipError_t er = time_blur_pass(e.kdevs[kernel], e.F, e.maxV[kernel], e.L, reps, e.stream, ms_per_launch);
    if (er != hipSuccess) return fail(LCCRF_E_HIP, "time_blur_pass: %s", hipGetErrorString(er));
    return LCCRF_OK;
}

int lccrf_pose_optimization(int device_id, int n_points, const float *Xw, const float *kp, const float *u_right,
                            const float *inv_sigma2, const uint8_t *valid, const int16_t *label, const float *K4, float bf,
                            const float *Tcw_in, float *Tcw_out, uint8_t *outlier_out, int32_t *n_inliers_out)
{
    if (n_points < 0) return fail(LCCRF_E_INVALID, "n_points < 0");
    if (n_points > 16384) return fail(LCCRF_E_CAPACITY, "at most 16384 keypoints per frame");
    if (!K4 || !Tcw_in || !Tcw_out) return fail(LCCRF_E_INVALID, "K4 / Tcw is NULL");
    if (n_points && (!Xw || !kp || !u_right || !inv_sigma2 || !outlier_out)) return fail(LCCRF_E_INVALID, "NULL array");
    int rc = use_device(device_id);
    if (rc) return rc;
    if (device_id >= kMaxDevices) return fail(LCCRF_E_INVALID, "device_id %d beyond the staging table", device_id);
    const size_t n = (size_t)std::max(n_points, 1), n16 = (n + 15) & ~(size_t)15;
    // one staging area per call, laid out the same in pinned host memory and in device memory:
    //   [Xw | kp | ur | is2 | Tin | Tout | ints (n_points, n_inliers, n_initial, -) | outlier | label | valid]
    // one copy in (everything), one copy out ([Tout | ints | outlier]), on the context's own stream.
    const size_t off_kp = n * 12, off_ur = off_kp + n * 8, off_is2 = off_ur + n * 4, off_tin = off_is2 + n * 4, off_tout = off_tin + 64,
                 off_int = off_tout + 64, off_out = off_int + 16, off_lab = off_out + n16, off_val = off_lab + 2 * n16, total = off_val + n16;
    PoseStage &ps = g_pose_stage[device_id];
    std::lock_guard<std::mutex> guard(ps.m);
    if (ps.cap < total) {
        if (ps.d) (void)hipFree(ps.d);
        if (ps.h) (void)hipHostFree(ps.h);
        ps.d = ps.h = nullptr; ps.cap = 0;
        const size_t want = std::max<size_t>(total + total / 2, 1 << 16);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ps.d), want));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ps.h), want));
        if (!ps.stream) HIP_TRY(hipStreamCreateWithFlags(&ps.stream, hipStreamNonBlocking));
        ps.cap = want;
    }
    char *d = ps.d, *h = ps.h;
    if (n_points) {
        memcpy(h, Xw, (size_t)n_points * 12);
        memcpy(h + off_kp, kp, (size_t)n_points * 8);
        memcpy(h + off_ur, u_right, (size_t)n_points * 4);
        memcpy(h + off_is2, inv_sigma2, (size_t)n_points * 4);
        if (label) memcpy(h + off_lab, label, (size_t)n_points * 2);
        if (valid) memcpy(h + off_val, valid, (size_t)n_points);
        memcpy(h + off_out, outlier_out, (size_t)n_points);                      // entries of invalid points pass through
    }
    memcpy(h + off_tin, Tcw_in, 64);
    int *hints = reinterpret_cast<int *>(h + off_int);
    hints[0] = n_points; hints[1] = hints[2] = hints[3] = 0;
    HIP_TRY(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, ps.stream));
    int *ints = reinterpret_cast<int *>(d + off_int);
    hipError_t er = launch_pose_optimization(1, (int)n, ints, reinterpret_cast<float *>(d), reinterpret_cast<float *>(d + off_kp),
                                             reinterpret_cast<float *>(d + off_ur), reinterpret_cast<float *>(d + off_is2),
                                             valid ? reinterpret_cast<uint8_t *>(d + off_val) : nullptr,
                                             label ? reinterpret_cast<int16_t *>(d + off_lab) : nullptr, K4, bf,
                                             reinterpret_cast<float *>(d + off_tin), reinterpret_cast<float *>(d + off_tout),
                                             reinterpret_cast<uint8_t *>(d + off_out), ints + 1, ints + 2, ps.stream);
    if (er != hipSuccess) return fail(LCCRF_E_HIP, "pose optimisation: %s", hipGetErrorString(er));
    HIP_TRY(hipMemcpyAsync(h + off_tout, d + off_tout, 64 + 16 + (size_t)n_points, hipMemcpyDeviceToHost, ps.stream));
    HIP_TRY(hipStreamSynchronize(ps.stream));
    memcpy(Tcw_out, h + off_tout, 64);
    if (n_points) memcpy(outlier_out, h + off_out, (size_t)n_points);
    if (n_inliers_out) *n_inliers_out = hints[1];
    return LCCRF_OK;
}

int lccrf_batch_pose_optimization(lccrf_batch_handle b, const float *d_Xw, const float *d_kp, const float *d_u_right,
                                  const float *d_inv_sigma2, const uint8_t *d_valid, const float *K4, float bf,
                                  const float *d_Tcw_in, float *d_Tcw_out, uint8_t *d_outlier, int32_t *d_n_inliers,
                                  int32_t *d_n_initial, void *stream)
{
    CHECK_H(b);
    if (!d_Xw || !d_kp || !d_u_right || !d_inv_sigma2 || !K4 || !d_Tcw_in || !d_Tcw_out || !d_outlier || !d_n_inliers || !d_n_initial)
        return fail(LCCRF_E_INVALID, "NULL array");
    Engine &e = b->eng;
    if (!e.started) return fail(LCCRF_E_STATE, "no inference has produced labels yet");
    if (!e.last_with_map) return fail(LCCRF_E_STATE, "the last inference ran with with_map = 0: there are no labels to consume");
    if (e.maxN > 16384) return fail(LCCRF_E_CAPACITY, "at most 16384 keypoints per frame");
    // a one-launch run may still owe the labels of frames that did not fit its kernel (they are written by the
    // re-run): settle that first -- one synchronisation, and only when lccrf_batch_run is still pending
    { int rl = e.resolve_late(); if (rl) return rl; }
    StreamScope scope(e, stream);
    int rc = scope.enter();
    if (rc) return rc;
    // the labels are read where the inference kernel wrote them: no host round trip between the CRF and the pose
    hipError_t er = launch_pose_optimization(e.F, e.maxN, b->d_pose_total ? b->d_pose_total : e.crf.n_points, d_Xw, d_kp, d_u_right,
                                             d_inv_sigma2, d_valid, e.crf.map, K4, bf, d_Tcw_in, d_Tcw_out, d_outlier, d_n_inliers,
                                             d_n_initial, e.stream, b->d_pose_total ? e.crf.n_points : nullptr);
    if (er != hipSuccess) return fail(LCCRF_E_HIP, "pose optimisation: %s", hipGetErrorString(er));
    return LCCRF_OK;
}

int lccrf_batch_pose_set_crf_counts(lccrf_batch_handle b, const int32_t *d_n_total)
{
    CHECK_H(b);
    b->d_pose_total = d_n_total;
    return LCCRF_OK;
}

int lccrf_batch_last_timing(lccrf_batch_handle b, float *inference_ms, float *build_ms)
{
    CHECK_H(b);
    Engine &e = b->eng;
    { int rl = b->eng.resolve_late(); if (rl) return rl; }
    HIP_TRY(hipDeviceSynchronize());
    if (build_ms) {
        *build_ms = 0.0f;
        if (e.timed_build) HIP_TRY(hipEventElapsedTime(build_ms, e.ev[0], e.ev[1]));
    }
    if (inference_ms) {
        *inference_ms = 0.0f;
        if (e.timed_inf) HIP_TRY(hipEventElapsedTime(inference_ms, e.ev[2], e.ev[3]));
    }
    return LCCRF_OK;
}

int lccrf_batch_get_stream(lccrf_batch_handle b, void **stream)
{
    CHECK_H(b);
    if (!stream) return fail(LCCRF_E_INVALID, "stream is NULL");
    *stream = static_cast<void *>(b->eng.stream);
    return LCCRF_OK;
}

int lccrf_batch_last_prepare(lccrf_batch_handle b, float *prepare_ms, int *runs)
{
    CHECK_H(b);
    Engine &e = b->eng;
    { int rl = e.resolve_late(); if (rl) return rl; }
    if (prepare_ms) {
        *prepare_ms = 0.0f;
        if (e.lean_prep.timed) {
            HIP_TRY(hipEventSynchronize(e.lean_prep.ev1));
            HIP_TRY(hipEventElapsedTime(prepare_ms, e.lean_prep.ev0, e.lean_prep.ev1));
        }
    }
    if (runs) *runs = (int)e.lean_prep.runs;
    return LCCRF_OK;
}

}  // extern "C"
