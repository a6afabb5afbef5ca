/*
 * lccrf.h -- C-ABI of the MI355X-native dense-CRF mean-field path.
 *
 * Drop-in boundary for ONE hot path of LC-CRF-SLAM: the per-frame dense-CRF inference
 * that labels ORB keypoints static / dynamic inside Tracking::DynamicDetectionWithCRF
 * (reference src/Tracking.cc:1919-1930).  The entry points are exactly what a binding
 * of the reference's own operator interface for this path needs -- the two abstract
 * classes of Thirdparty/DenseCRF/include/densecrf_base.h:
 *
 *      class PairwisePotential   (densecrf_base.h:12-19)
 *      class DenseCRF            (densecrf_base.h:22-92)
 *
 * Plain pointers and sizes only; no C++/torch types.  Every function returns
 * LCCRF_OK (0) or a negative lccrf_status; lccrf_last_error() gives the detail for
 * the calling thread.  There is NO CPU fallback behind this ABI: without a usable
 * gfx950 device every call that needs one fails with LCCRF_E_NO_DEVICE.
 *
 * Threading: handles are thread-compatible (one handle per thread at a time), like
 * the reference's stack-local per-frame objects.
 *
 * Layouts (all little-endian, densely packed):
 *   unary / probability : float32 [N][L]   "x0l0 x0l1 .. x1l0 .." (densecrf_base.h:56)
 *   label / map         : int16   [N]      -1 = unknown (densecrf3d.h:119)
 *   features            : float32 [N][d]   already divided by the kernel's stdev
 *                                          (pairwise3d.h:41-44,64-66 do that division)
 */
#ifndef LCCRF_H
#define LCCRF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCCRF_ABI_VERSION 3   /* 3: lccrf_batch_last_prepare, lccrf_batch_get_stream, LCCRF_OPT_EVENT_TIMING, lccrf_batch_synchronize scoped to the batch's streams; 2: lccrf_batch_get_fused_shape, the asynchronous host path of the batch API, lccrf_set_option's option 3 (LCCRF_OPT_COPY_THREADS) */
#define LCCRF_MAX_KERNELS 8      /* pairwise terms per CRF                        */
#define LCCRF_MAX_DIMS    8      /* feature dimensions per kernel (reference uses <= 6) */
#define LCCRF_MAX_LABELS  64

typedef enum lccrf_status {
    LCCRF_OK            =  0,
    LCCRF_E_INVALID     = -1,    /* bad argument (NULL, negative size, d/L out of range) */
    LCCRF_E_NO_DEVICE   = -2,    /* no gfx950 device / HIP runtime unusable              */
    LCCRF_E_HIP         = -3,    /* a HIP call failed; see lccrf_last_error()            */
    LCCRF_E_NOMEM       = -4,
    LCCRF_E_STATE       = -5,    /* call order violated (e.g. inference before unary)    */
    LCCRF_E_CAPACITY    = -6     /* more kernels / frames / points than created for      */
} lccrf_status;

int          lccrf_abi_version(void);
const char  *lccrf_last_error(void);
int          lccrf_device_count(int *count);

/* ======================================================================================
 * 1. Object API -- one CRF, host buffers in / host buffers out.
 *    Mirrors DenseCRF3D<M> + PottsPotential3D<M,F> as used at src/Tracking.cc:1920-1930.
 * ==================================================================================== */
typedef struct lccrf_crf *lccrf_handle;

/* DenseCRF3D<M>::DenseCRF3D(int N)              densecrf3d.h:23-28   (M = n_labels)   */
int  lccrf_create(lccrf_handle *out, int device_id, int n_points, int n_labels);
/* ~DenseCRF3D / ~DenseCRF (owns its pairwise terms)  densecrf3d.h:30-36, base.h:41-45.
 * The reference constructs and destroys one CRF per frame (src/Tracking.cc:1920); to keep that
 * pattern cheap a destroyed handle's device memory, stream and pinned staging are parked and
 * reused by the next lccrf_create of a compatible size.  lccrf_trim_cache() frees the parked
 * handles (returns how many).                                                             */
void lccrf_destroy(lccrf_handle h);
int  lccrf_trim_cache(void);

/* Options (no reference counterpart; none can change a result).
 *   LCCRF_OPT_SINGLE_WORKGROUP   value != 0: a frame is never given two workgroups.  By default a single two-kernel frame (and a
 *       batch of up to 64) runs as TWO workgroups on two CUs, one per lattice build, handing tables over through device memory
 *       (-8 us at 2000 keypoints).  The main workgroup waits for its helper with a bounded poll: if the helper is not
 *       co-scheduled -- a GPU shared with other processes or streams whose kernels hold every CU -- the frame stalls for up to
 *       ~0.05-0.1 s (three frames of a 30 fps tracker) before it falls back to the ordinary path, with the same labels.  A tracker
 *       that shares its GPU sets this option and pays the 8 us instead.
 *   LCCRF_OPT_VERTEX_ORDER   (batches; 0 automatic = on, 1 on, 2 off; from the next lccrf_batch_build on) large frames (>= 8192 points,
 *       streaming engine) are built by SORTING the (point, corner) entries on the row-major code of their vertex in the basis of the
 *       lattice's own axes -- no hash table -- with the points processed in the same order: the vertices are numbered along those axes
 *       and a blur pass (permutohedral_cpu.h:663-679) touches half as many cache lines per gather (23 -> 19.5 us per pass over 8 frames of
 *       100 000 points, 0.61 -> 0.72 of the HBM peak).  2 selects round 3's build (hash table, vertices by first occurrence along a
 *       Z-order curve of the points).  Results never depend on how vertices are found or numbered.
 *   LCCRF_OPT_COPY_THREADS   (batches; 1 .. 64, default 8) host threads that copy the caller's arrays into the batch's pinned
 *       staging in lccrf_batch_set_inputs_host_async (one core copies ~10 GB/s, the PCIe link takes ~50).
 *   LCCRF_OPT_EVENT_TIMING   (batches; default 1) value 0: lccrf_batch_build / _inference / _run record no HIP events around their
 *       work and lccrf_batch_last_timing reads 0.  An event record is a packet of its own between two launches: ~20 us per call on a
 *       stream that is kept busy, 1-2 % of a 1.7 ms batch -- a replay loop that does not read the timings turns them off.
 * lccrf_set_option applies to one handle (set it after lccrf_create: a handle taken from the cache starts from the defaults) and is
 * per handle only for LCCRF_OPT_VERTEX_ORDER and LCCRF_OPT_COPY_THREADS; lccrf_set_default_option applies LCCRF_OPT_SINGLE_WORKGROUP
 * to every handle and batch created afterwards in this process.                                                              */
typedef enum lccrf_option {
    LCCRF_OPT_SINGLE_WORKGROUP = 1,
    LCCRF_OPT_VERTEX_ORDER     = 2,
    LCCRF_OPT_COPY_THREADS     = 3,
    LCCRF_OPT_EVENT_TIMING     = 4
} lccrf_option;
int  lccrf_set_option(lccrf_handle h, int option, int value);
int  lccrf_set_default_option(int option, int value);

/* DenseCRF::setUnaryEnergy(const float*)        densecrf3d.h:41-43                    */
int  lccrf_set_unary(lccrf_handle h, const float *unary);
/* DenseCRF::setUnaryEnergyFromLabel(const short*, float*)   densecrf3d.h:107-130;
 * conf has n_labels entries.  The scalar overload (densecrf3d.h:100-105) is the same
 * call with every entry equal.                                                         */
int  lccrf_set_unary_from_label(lccrf_handle h, const int16_t *label, const float *conf);

/* new PottsPotential3D<M,F>(features, N, w) + DenseCRF::addPairwiseEnergy(p)
 * pairwise3d.h:20-28 (lattice init + normalisation), densecrf_base.h:54.
 * The CRF owns the term.  d = F.  Kernels are applied in the order added.              */
int  lccrf_add_pairwise(lccrf_handle h, const float *features, int d, float w);

/* PottsPotential3D<M,2>::appearanceKernel(N, w, vobserv, verror, sd1, sd2)
 * pairwise3d.h:37-48 : features (vobserv/sd1, verror/sd2), then the ctor above.        */
int  lccrf_add_appearance_kernel(lccrf_handle h, float w, const float *vobserv,
                                 const float *verror, float sd_observ, float sd_error);
/* PottsPotential3D<M,2>::smoothKernel(N, w, points3d, points2d, sd3d, sd2d)
 * pairwise3d.h:51-71 : only the 2-D branch is live -> features (u/sd2d, v/sd2d).
 * xy is [N][2] (cv::Point2f layout).                                                   */
int  lccrf_add_smooth_kernel(lccrf_handle h, float w, const float *xy, float sd2d);

/* DenseCRF::startInference()                    densecrf_base.h:78-80                 */
int  lccrf_start_inference(lccrf_handle h);
/* DenseCRF::stepInference(float relax)          densecrf_base.h:82-91                 */
int  lccrf_step_inference(lccrf_handle h, float relax);
/* DenseCRF3D<M>::buildMap()                     densecrf3d.h:136-151                  */
int  lccrf_build_map(lccrf_handle h);
/* DenseCRF::inference(n_iterations, with_map, relax)   densecrf_base.h:65-73
 * Frames of >= 8192 points (far beyond a SLAM frame: BASELINE config 5) run it in the streaming engine's LOCALITY MODE -- the points
 * in an internal order, the lattice built by sorting, the first blur passes inside the splat -- when the
 * lattices are first needed by this call.  Results are the same bits in the caller's order.  The entry points that expose or
 * continue from per-point lattice state (lccrf_start_inference / lccrf_step_inference, lccrf_pairwise_apply, lccrf_step_init,
 * lccrf_get_norm / _lattice / _unary) work on the plain build: used first, the handle is built that way and stays so; used after an
 * inference() in locality mode, the lattices are re-built once (same results, one extra build).                       */
int  lccrf_inference(lccrf_handle h, int n_iterations, int with_map, float relax);

/* DenseCRF::getMap() / getProbability()         densecrf_base.h:74-75
 * (copies out; the reference returns pointers into object-owned buffers)
 * lccrf_get_map right behind lccrf_inference(h, n, 1, ...) is the tracker's sequence and the fast one: the frame kernel's
 * last stores are the labels, into pinned host memory, and the call takes them as they arrive instead of waiting for the
 * stream.  Same results as any other order of calls.                                    */
int  lccrf_get_map(lccrf_handle h, int16_t *map_out);
int  lccrf_get_probability(lccrf_handle h, float *prob_out);

/* ---- the reference's two plug-in points, on HOST arrays (device in, device out behind the call) ------------------
 * PairwisePotential::apply(out_values, in_values, tmp)        densecrf_base.h:18, pairwise3d.h:73-78
 *   out_values[i][k] += w * norm[i] * compute(in_values)[i][k] for pairwise term `kernel` of this CRF (both
 *   [N][n_labels]; the reference's `tmp` scratch lives on the device).  This is the one pure virtual of the
 *   reference's plug-in class: a caller that mixes its own PairwisePotential subclasses with ours drives the
 *   mean-field step itself (densecrf_base.h:82-91) and calls this for our terms.                               */
int  lccrf_pairwise_apply(lccrf_handle h, int kernel, float *out_values, const float *in_values);
/* DenseCRF's protected virtuals (densecrf_base.h:34-36), for exactly that kind of caller:
 *   expAndNormalize(out, in, scale, relax)    densecrf3d.h:70-98   (out is read too when relax != 1)
 *   stepInit(): next = -unary                 densecrf3d.h:154-158
 *   buildMap() on a given probability array   densecrf3d.h:136-151                                            */
int  lccrf_exp_and_normalize(lccrf_handle h, float *out, const float *in, float scale, float relax);
int  lccrf_step_init(lccrf_handle h, float *next_out);
int  lccrf_map_of(lccrf_handle h, const float *prob, int16_t *map_out);
/* PermutohedralLatticeCPU::init(features, d, N) + compute(out, in, value_size)   permutohedral_cpu.h:241-424,634-699
 * the bare lattice filter (splat, d+1 blurs, slice; no normalisation, no weight) with any value_size in
 * [1, LCCRF_MAX_LABELS]; n_vertices (may be NULL) receives the lattice size M_.                                */
int  lccrf_lattice_filter(int device_id, const float *features, int n_points, int d, const float *in,
                          int value_size, float *out, int *n_vertices);

/* Parity probes (no reference API; the reference keeps these protected):
 * lattice size M_ (permutohedral_cpu.h:398) and PottsPotential3D::norm_ (pairwise3d.h:18).
 * offset/bary are [N][d+1], nbr is [d+1][V][2]; any output pointer may be NULL.         */
int  lccrf_get_lattice_size(lccrf_handle h, int kernel, int *n_vertices);
int  lccrf_get_norm(lccrf_handle h, int kernel, float *norm_out);
int  lccrf_get_lattice(lccrf_handle h, int kernel, int32_t *offset_out, float *bary_out,
                       int32_t *nbr_out);
int  lccrf_get_unary(lccrf_handle h, float *unary_out);

/* ======================================================================================
 * 2. Batch API -- many independent frames in flight on one GPU.
 *    Every frame is one CRF of the object API; frames never interact.  Inputs may be
 *    handed over as host buffers (uploaded) or bound as DEVICE pointers (zero copy), so
 *    a caller that already holds the arrays in HBM pays no PCIe traffic.
 * ==================================================================================== */
typedef struct lccrf_batch *lccrf_batch_handle;

typedef struct lccrf_batch_desc {
    int   max_frames;                       /* frames per batch                          */
    int   max_points;                       /* per-frame stride of every array           */
    int   n_labels;
    int   n_kernels;
    int   feat_dims[LCCRF_MAX_KERNELS];
    float weights[LCCRF_MAX_KERNELS];
} lccrf_batch_desc;

int  lccrf_batch_create(lccrf_batch_handle *out, int device_id, const lccrf_batch_desc *desc);
void lccrf_batch_destroy(lccrf_batch_handle b);
/* lccrf_set_option for a batch (LCCRF_OPT_SINGLE_WORKGROUP: batches of up to 64 two-kernel frames take two workgroups per frame) */
int  lccrf_batch_set_option(lccrf_batch_handle b, int option, int value);

/* Host inputs (copied to the device).  n_points[f] <= max_points; arrays are strided by
 * max_points per frame.  Exactly one of unary / label must be non-NULL.                 */
int  lccrf_batch_set_inputs_host(lccrf_batch_handle b, int n_frames, const int32_t *n_points,
                                 const float *unary, const int16_t *label, const float *conf,
                                 const float *const *features /* [n_kernels] */);
/* Host inputs, nothing waits (round 5; the reference's per-frame cost at src/Tracking.cc:1919-1930 is host to host, so a replay
 * that wants the GPU's rate has to keep the link busy under the kernels).  Same arrays as lccrf_batch_set_inputs_host.  They are
 * copied into pinned staging owned by the batch before the call returns -- the caller's buffers may be reused at once -- and
 * uploaded on the batch's copy stream; whatever is queued on the batch afterwards (lccrf_batch_run / _build / ..., on its own or a
 * caller's stream) waits for the upload on the device, and the upload itself waits for the kernels queued before it.  With
 * LCCRF_HOST_PINNED the caller vouches that the label / unary and feature arrays are pinned (hipHostMalloc / hipHostRegister) and stay untouched until
 * lccrf_batch_wait_inputs returns (or any later result of this batch has been seen): no staging copy, the DMA reads the caller's
 * memory.  One batch is in flight per handle: to upload batch i+1 under batch i's kernels alternate between two or three handles
 * (tools/replay_multi.cpp does).                                                                       */
#define LCCRF_HOST_PINNED 1
int  lccrf_batch_set_inputs_host_async(lccrf_batch_handle b, int n_frames, const int32_t *n_points,
                                       const float *unary, const int16_t *label, const float *conf,
                                       const float *const *features /* [n_kernels] */, int flags);
int  lccrf_batch_wait_inputs(lccrf_batch_handle b);
/* Results to the host, nothing waits: queues, behind everything queued on the batch so far, copies of the chosen results into
 * pinned memory owned by the batch (label bits: 256 bytes per 2000-keypoint frame -- the layout of lccrf_batch_device_label_bits;
 * int16 labels and probabilities: the layouts of lccrf_batch_get_map_host / _get_probability_host) on the batch's download
 * stream.  lccrf_batch_wait_download waits for them -- and settles the per-frame fallback of a one-launch run first, refreshing
 * the copies if a frame had to be re-run -- and hands out the host pointers (NULL for what was not asked for), valid until the
 * next lccrf_batch_download_async on this handle.                                                                              */
#define LCCRF_DOWNLOAD_LABEL_BITS  1
#define LCCRF_DOWNLOAD_MAP         2
#define LCCRF_DOWNLOAD_PROBABILITY 4
int  lccrf_batch_download_async(lccrf_batch_handle b, int what);
int  lccrf_batch_wait_download(lccrf_batch_handle b, const uint64_t **label_bits, int *words_per_frame,
                               const int16_t **map, const float **probability);
/* Device inputs (bound, not copied; must stay valid until the batch finished).  d_n_points must be COMPLETE when
 * this is called (it is validated and clamped into a private copy on the batch's own stream right here, unordered
 * with any other stream); an entry outside [0, max_points] raises LCCRF_E_CAPACITY at the next synchronisation point. */
int  lccrf_batch_bind_inputs_device(lccrf_batch_handle b, int n_frames, const int32_t *d_n_points,
                                    const float *d_unary, const int16_t *d_label, const float *conf,
                                    const float *const *d_features /* host array of device ptrs */);

/* Per frame: every PottsPotential3D ctor (lattice + norm), pairwise3d.h:20-28.
 * Asynchronous on `stream` (a hipStream_t, or NULL for the batch's own stream).         */
int  lccrf_batch_build(lccrf_batch_handle b, void *stream);
/* Per frame: DenseCRF::inference(n_iterations, with_map, relax), densecrf_base.h:65-73. */
int  lccrf_batch_inference(lccrf_batch_handle b, int n_iterations, int with_map, float relax,
                           void *stream);
/* Per frame, in ONE kernel launch: every PottsPotential3D ctor (lattice + norm, pairwise3d.h:20-28) followed by
 * DenseCRF::inference(n_iterations, with_map, relax) (densecrf_base.h:65-73) -- the reference's whole per-frame
 * sequence src/Tracking.cc:1920-1929.  Needs inputs only (no lccrf_batch_build); the lattices never reach
 * HBM, so the parity probes (norm, lattice arrays) rebuild them on demand.  Frames the one-launch kernel
 * cannot take (n_labels != 2, a kernel with d != 2, more than two kernels, > 4096 points, or lattices too
 * large for one workgroup's LDS) run on the build + inference kernels instead: same results either way.  The
 * decision is PER FRAME: the kernel flags the frames whose lattices did not fit and, at the next synchronisation
 * point (lccrf_batch_synchronize, any getter, the next call that continues from the results), exactly those frames
 * are gathered, re-run on the two-kernel path and scattered back -- a batch with one outlier pays for one frame.
 * The device buffers of lccrf_batch_device_buffers / _label_bits are complete behind that synchronisation point.
 * Memory: the first lccrf_batch_run of a handle whose frames qualify for the two-frames-per-CU kernel (>= 256 frames, two 2-D kernels,
 * <= 2048 points) allocates that kernel's per-point record area ON THAT CALL (a synchronous hipMalloc): 96 KB per frame BOUND at that
 * moment (1.6 GB for 16384 frames, 400 MB for 4096), kept until lccrf_batch_destroy; a later run with more frames allocates a larger
 * area (the smaller one stays with the handle).  Expect the first run of a handle to be slower than the following ones.  */
int  lccrf_batch_run(lccrf_batch_handle b, int n_iterations, int with_map, float relax, void *stream);
/* How many frames of the last lccrf_batch_run had to be re-run on the two-kernel path (0 = every frame fitted). */
int  lccrf_batch_get_fallback_frames(lccrf_batch_handle b, int *n_frames);
int  lccrf_batch_synchronize(lccrf_batch_handle b);

/* Results: copy to host, or borrow the device buffers ([n_frames][max_points](xL)).     */
int  lccrf_batch_get_map_host(lccrf_batch_handle b, int16_t *map_out);
int  lccrf_batch_get_probability_host(lccrf_batch_handle b, float *prob_out);
int  lccrf_batch_get_lattice_sizes_host(lccrf_batch_handle b, int kernel, int32_t *n_vertices_out);
int  lccrf_batch_get_norm_host(lccrf_batch_handle b, int kernel, float *norm_out);
int  lccrf_batch_device_buffers(lccrf_batch_handle b, const int16_t **d_map, const float **d_prob);
/* Binary CRFs (n_labels == 2, the SLAM configuration src/Tracking.cc:1919): the MAP labels of the last
 * inference with_map as one BIT per point -- uint64 [n_frames][words_per_frame], point i = bit i%64 of
 * word i/64, bits beyond a frame's n_points are 0.  Written by the same kernel that writes the int16
 * labels; this is the payload of the multi-GPU label gather (250 bytes per
 * 2000-keypoint frame instead of 4000; RCCL has no 16-bit integer type).                          */
int  lccrf_batch_device_label_bits(lccrf_batch_handle b, const uint64_t **d_bits, int *words_per_frame);

/* Engine selection for the inference loop (all give bit-identical results):
 *   0 = automatic, 1 = streaming kernels over HBM (any size), 2 = fused one-workgroup-
 *   per-frame kernel with the lattice values in LDS (SLAM sizes only).  lccrf_batch_get_engine
 *   also reports 3 = the one-launch-per-frame kernel of lccrf_batch_run.                */
int  lccrf_batch_set_engine(lccrf_batch_handle b, int engine);
int  lccrf_batch_get_engine(lccrf_batch_handle b, int *engine_in_use);
/* Report only: the workgroup shape of the last one-workgroup-per-frame launch -- lccrf_batch_inference on the fused engine (engine 2)
 * or lccrf_batch_run's one-launch kernel (engine 3), whichever ran last -- as lanes per frame (1024 or 512) and how many frames
 * share a CU (1; 2 for the half-LDS plans: batches of at least 256 frames of up to 2048 points, csrc/fused_lean.h and
 * csrc/frame_lean.hip).  0 / 0 if no such launch happened yet.  Same results in every shape (the order of every row sum is the
 * reference's, permutohedral_cpu.h:653-661).                                                                                   */
int  lccrf_batch_get_fused_shape(lccrf_batch_handle b, int *lanes_per_frame, int *frames_per_cu);
/* Report only: how the lattices now in HBM were built (after lccrf_batch_build) -- whether the points of a frame are processed in
 * an internal order (locality mode: frames of >= 8192 points) and whether the vertices were found by the SORTED build
 * (LCCRF_OPT_VERTEX_ORDER) or, 0, by the hash table: also what an engine falls back to for good when a frame's vertex codes overflow
 * 62 bits or a feature is wide enough to wrap the reference's int16 keys.  Either pointer may be NULL.                           */
int  lccrf_batch_get_locality_mode(lccrf_batch_handle b, int *internal_point_order, int *sorted_build);

/* Measurement support for bench.py: HIP-event time of the last lccrf_batch_inference()
 * on its stream, the number of launches of the dominant kernel and their summed
 * duration as seen by events around them (0 if not instrumented).                       */
/* The batch's own stream (a hipStream_t; what `stream = NULL` means in the calls above), for a caller that orders its own work -- a
 * collective on the label bits, a copy -- behind the batch's kernels.  Several batches kept in flight overlap best on their OWN
 * streams: each was created with its handle, one after the other, and HIP spreads streams over its (four) hardware queues in that
 * order, whereas two caller-made streams may share a queue and then run their kernels one after the other (measured: bench.py with two
 * handles, 5.04e7 iterations/s on the handles' streams, 4.7-5.0e7 on two torch streams depending on what else the process created). */
int  lccrf_batch_get_stream(lccrf_batch_handle b, void **stream);
int  lccrf_batch_last_timing(lccrf_batch_handle b, float *inference_ms, float *build_ms);
/* Batches of >= 256 two-kernel frames of 513 .. 2048 points run their inference two frames per CU from PREPARED launch records: what
 * the kernel's prologue would derive from the lattices in every launch (the ranking and placement of the appearance kernel's rows,
 * every point's vertex addresses and product slots, the row and neighbour tables in their on-chip form) is written once, by the
 * FIRST lccrf_batch_inference behind a build, into one block per frame (<= 64 KB; allocated on that call, kept with the handle), and
 * every later inference on those lattices starts from it.  It is part of the lattice construction (the PottsPotential3D ctor,
 * pairwise3d.h:20-28), not of inference(): prepare_ms = HIP-event time of the last such launch (0 if none), runs = how many there
 * were on this handle.  Results are identical with and without the records.                                                         */
int  lccrf_batch_last_prepare(lccrf_batch_handle b, float *prepare_ms, int *runs);
/* Measurement support: average HIP-event duration of `reps` launches of the streaming engine's dominant kernel (one
 * Jacobi blur pass of `kernel` over every frame of the batch, permutohedral_cpu.h:663-679) and the number of lattice
 * vertices one launch processes.  Needs lccrf_batch_build; leaves the CRF state (Q, labels) untouched.            */
int  lccrf_batch_time_blur_pass(lccrf_batch_handle b, int kernel, int reps, float *ms_per_launch,
                                int64_t *vertices_per_launch);

/* ======================================================================================
 * 3. Unary builder -- the step right before the CRF (first "next" row):
 *    Tracking::ComputeMapPointErrAndObserv (src/Tracking.cc:1803-1839) for every candidate map
 *    point, then Tracking::RroughClassify (src/Tracking.cc:1961-2013).  The caller flattens the
 *    MapPoint -> observations -> KeyFrame graph of the frame into a CSR.  PARITY UNPINNED: the
 *    reference holds no test or fixture for these two functions.
 * ==================================================================================== */
typedef struct lccrf_crf_params {           /* Tracking.cc:151-171, Examples/RGB-D/TUM3.yaml:78-101 */
    float w1, w2;
    float u_alpha, stdev_alpha;             /* reprojection error mean / stdev            */
    float u_beta, stdev_beta;               /* observation count mean / stdev             */
    float u_gamma, stdev_gamma;             /* epipolar prior (read by the reference, unused here) */
    float point3d_stdev, point2d_stdev;
    float u_depth, pth, confidence;
} lccrf_crf_params;

void lccrf_default_params(lccrf_crf_params *p);       /* the TUM3.yaml / BONN.yaml values */

/* Host arrays in, host arrays out.
 *   Xw        [n_points][3]   MapPoint::GetWorldPos()
 *   obs_ptr   [n_points+1]    observations of point i are obs_*[obs_ptr[i] .. obs_ptr[i+1])
 *   obs_kf    [n_obs]         index of the observing keyframe
 *   obs_kp    [n_obs][2]      pKF->mvKeysUn[fid].pt as doubles (Point2d, Tracking.cc:1832)
 *   kf_pose   [n_kf][12]      row-major 3x4 [Rcw | tcw];  kf_intr [n_kf][4] fx fy cx cy;
 *   kf_bounds [n_kf][4]       mnMinX mnMaxX mnMinY mnMaxY
 *   match_prob[n_points]      mvFeatureMatchProb per candidate, or NULL if that map is empty
 * Outputs (each [n_points]): observation count (as float, like the reference's vector<float>),
 * mean reprojection error, mean depth, rough label (0 moving, 1 static, -1 for a point without
 * observations, which the reference drops at Tracking.cc:1858).                              */
int  lccrf_unary_build(int device_id, int n_points, const float *Xw, const int32_t *obs_ptr,
                       const int32_t *obs_kf, const double *obs_kp, int n_kf, const float *kf_pose,
                       const float *kf_intr, const float *kf_bounds, const double *match_prob,
                       const lccrf_crf_params *params, float *observs_out, float *error_out,
                       float *depth_out, int16_t *label_out);

/* ======================================================================================
 * 4. Static-feature matcher (another "next" row)
 *
 * Tracking::BfMatch (src/Tracking.cc:1747-1766): cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2)
 * of the current frame's ORB descriptors (query) against a frame 15 frames back (train), kept
 * when `match[0].distance < match[1].distance * 0.6`.  Host arrays in, host arrays out:
 *   desc_query [n_query][32], desc_train [n_train][32]   256-bit ORB descriptors (Frame::mDescriptors rows)
 *   ratio                                                0.6 in the reference
 *   train_of_query_out [n_query]                         asso[fid1] = fid2 of :1762, or -1 (no entry)
 *   n_matches_out                                        asso.size(), may be NULL
 * Ties follow OpenCV's batchDistance: the two nearest are the two smallest (distance, train
 * index) pairs.  Parity is unpinned (no OpenCV here); exact against the oracle's restatement.
 * Fewer than two train descriptors give no matches (match.size() != 2).                    */
int  lccrf_bf_match(int device_id, int n_query, const uint8_t *desc_query, int n_train,
                    const uint8_t *desc_train, double ratio, int32_t *train_of_query_out,
                    int32_t *n_matches_out);

/* ======================================================================================
 * 5. Pose optimisation -- the step right after the CRF
 *
 * Optimizer::PoseOptimization (src/Optimizer.cc:239-450, called at src/Tracking.cc:1002): motion-only bundle
 * adjustment of the frame pose on the matches the CRF left standing -- g2o Levenberg-Marquardt on one SE3 vertex,
 * pose-only reprojection edges (monocular where u_right < 0, stereo otherwise), Huber kernel, 4 rounds of 10
 * iterations from the same initial pose with chi2 re-classification (5.991 / 7.815) in between.  PARITY UNPINNED:
 * g2o needs Eigen (absent here) and the reference holds no fixture for it; csrc/pose_opt.hip states what is and is
 * not reproduced.  Double precision on the device, one workgroup per frame.
 *   Xw [n][3]         MapPoint::GetWorldPos()              kp [n][2]   mvKeysUn[i].pt
 *   u_right [n]       mvuRight[i] (< 0: monocular edge)    inv_sigma2 [n]  mvInvLevelSigma2[kpUn.octave]
 *   valid [n]         mvpMapPoints[i] != NULL, or NULL for "all"
 *   label [n]         CRF labels, or NULL: a point labelled 0 (moving) has been nulled by
 *                     Tracking::DynamicDetectionWithCRF (Tracking.cc:1945-1955) and contributes no edge
 *   K4                fx fy cx cy;  bf = mbf;  Tcw row-major 4x4 float (pFrame->mTcw in, SetPose out)
 *   outlier_out [n]   mvbOutlier (entries of points without an edge are left as they were)
 *   n_inliers_out     the return value nInitialCorrespondences - nBad (0 with < 3 correspondences: pose untouched)
 * Synchronous, host arrays in and out; thread-safe (calls on one device share a cached staging area and take turns);
 * at most 16384 keypoints (up to 4096 the frame's edges are staged in LDS).                                        */
int  lccrf_pose_optimization(int device_id, int n_points, const float *Xw, const float *kp, const float *u_right,
                             const float *inv_sigma2, const uint8_t *valid, const int16_t *label, const float *K4,
                             float bf, const float *Tcw_in, float *Tcw_out, uint8_t *outlier_out,
                             int32_t *n_inliers_out);
/* The same for every frame of a batch, on DEVICE arrays strided by the batch's max_points ([F][max_points][..],
 * Tcw [F][16], counts [F]), with the labels of the batch's last inference read where the kernel left them: the
 * labels never visit the host between the CRF and the pose.  Asynchronous on `stream` (NULL: the batch's own);
 * behind a lccrf_batch_run whose fallback is still unresolved it first settles that (one synchronisation).
 * LCCRF_E_STATE if the last inference ran with with_map = 0 (no labels).
 * CRF-ORDER CONTRACT: arrays are indexed by CRF point index i (the order in which DynamicDetectionWithCRF pushed
 * its candidates, Tracking.cc:1849-1870), label i gates edge i.  The reference's PoseOptimization instead walks all
 * pFrame->N keypoints with a non-null map point (Optimizer.cc:281-360): that includes map points the CRF skipped
 * because observs == 0 (Tracking.cc:1857-1859), which still contribute an edge there.  Append those as extra points
 * behind a frame's CRF points -- rows [n_crf, n_points) with d_valid = 1 -- and say how many CRF points each frame
 * has through lccrf_batch_pose_set_crf_counts(); points beyond that count are treated as static (label 1).          */
int  lccrf_batch_pose_optimization(lccrf_batch_handle b, const float *d_Xw, const float *d_kp, const float *d_u_right,
                                   const float *d_inv_sigma2, const uint8_t *d_valid, const float *K4, float bf,
                                   const float *d_Tcw_in, float *d_Tcw_out, uint8_t *d_outlier,
                                   int32_t *d_n_inliers, int32_t *d_n_initial, void *stream);
/* Optional: d_n_total[f] >= the CRF's n_points[f] (device array [F], bound, not copied) = CRF points + the extra
 * non-CRF edges of the contract above; NULL (default) = only the CRF's points carry edges.                          */
int  lccrf_batch_pose_set_crf_counts(lccrf_batch_handle b, const int32_t *d_n_total);

#ifdef __cplusplus
}
#endif
#endif /* LCCRF_H */
