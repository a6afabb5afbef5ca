/*
 * lccrf_record.h -- capture/replay records of the CRF call site.
 *
 * One record = everything Tracking::DynamicDetectionWithCRF hands to the dense CRF for one frame
 * (/root/reference/src/Tracking.cc:1871-1930) plus, optionally, what the reference computed from
 * it.  An instrumented reference build (on a machine with OpenCV and the TUM / Bonn sequences)
 * appends one record per frame; tools/replay.py runs the records through this library and checks
 * the labels.  INTEGRATION.md shows the dozen lines that write a record at the call site.
 *
 * File layout (little-endian, no padding beyond what is stated):
 *
 *   file    := file_header frame*
 *   frame   := frame_header
 *              float    vobservs[n]         Tracking.cc:1866   observation count per point
 *              float    verrors[n]          :1868              mean reprojection error
 *              float    vdepths[n]          :1867              mean depth
 *              float    coord2d[n][2]       :1861              undistorted keypoint (u, v)
 *              int16_t  init_label[n]       :1871              RroughClassify output (0 moving, 1 static)
 *              double   match_prob[n]       only if flags & LCCRF_REC_HAS_MATCH_PROB   (:2003, mvFeatureMatchProb[fid])
 *              int16_t  ref_label[n]        only if flags & LCCRF_REC_HAS_REF_LABEL    (:1930, crf.getMap())
 *              float    ref_prob[n][2]      only if flags & LCCRF_REC_HAS_REF_PROB     (crf.getProbability())
 *              (zero bytes up to the next multiple of 8)
 *              section*                      version 2, only if flags & LCCRF_REC_HAS_SECTIONS: frame_header.n_sections of them
 *   section := section_header payload (zero bytes up to the next multiple of 8)
 *
 * Everything is the value the reference holds at that line, bit for bit (floats are written as
 * their IEEE-754 bytes), so a replay can be compared with ref_label / ref_prob exactly.
 *
 * VERSION 2 adds optional, self-describing SECTIONS behind the CRF arrays of a frame, one per neighbouring step of the
 * tracker whose GPU twin lives in this library but whose reference code cannot be built without OpenCV / Eigen / g2o
 * (SURVEY.md section 8 rows a2/a3/f1, f3, f4).  A capture with these sections is what turns those rows from "HIP ==
 * restatement" into "HIP == reference": tools/replay.py feeds every section's inputs to the matching entry point of
 * include/lccrf.h and compares with the recorded outputs.  Sections a reader does not know are skipped by their
 * byte count.  A version-1 file is a version-2 file without sections; readers accept both.
 *
 *   LCCRF_SEC_UNARY  inputs + outputs of Tracking::ComputeMapPointErrAndObserv for every candidate map point and of
 *                    Tracking::RroughClassify (src/Tracking.cc:1803-1839, 1849-1871, 1961-2013) = lccrf_unary_build
 *   LCCRF_SEC_BFMATCH  Tracking::BfMatch (src/Tracking.cc:1747-1766) = lccrf_bf_match
 *   LCCRF_SEC_POSE   Optimizer::PoseOptimization (src/Optimizer.cc:239-450, called at src/Tracking.cc:1002)
 *                    = lccrf_pose_optimization
 */
#ifndef LCCRF_RECORD_H
#define LCCRF_RECORD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCCRF_REC_MAGIC   "LCCRFREC"     /* 8 bytes, no terminator */
#define LCCRF_REC_VERSION 2u             /* readers also accept 1 (no sections)                  */

enum {
    LCCRF_REC_HAS_MATCH_PROB = 1u << 0,
    LCCRF_REC_HAS_REF_LABEL  = 1u << 1,
    LCCRF_REC_HAS_REF_PROB   = 1u << 2,
    LCCRF_REC_HAS_SECTIONS   = 1u << 3,  /* version 2: frame_header.n_sections sections follow   */
};

/* file_header.origin: where the recorded OUTPUTS come from */
enum {
    LCCRF_REC_ORIGIN_REFERENCE = 0u,     /* an instrumented build of the reference: these pin parity               */
    LCCRF_REC_ORIGIN_SYNTHETIC = 1u,     /* this repository's own CPU restatements on synthetic inputs: format
                                            samples and plumbing tests only, they pin nothing                      */
};

typedef struct lccrf_rec_file_header {   /* 32 bytes */
    char     magic[8];                   /* LCCRF_REC_MAGIC                                    */
    uint32_t version;                    /* LCCRF_REC_VERSION                                  */
    uint32_t header_bytes;               /* sizeof(lccrf_rec_file_header): skip unknown tails  */
    uint32_t frame_header_bytes;         /* sizeof(lccrf_rec_frame_header)                     */
    uint32_t origin;                     /* LCCRF_REC_ORIGIN_* (version 1 files: 0)            */
    uint32_t reserved[2];                /* zero                                               */
} lccrf_rec_file_header;

typedef struct lccrf_rec_frame_header {  /* 80 bytes */
    uint32_t n_points;                   /* featureMapAssos.size(), Tracking.cc:1895           */
    uint32_t flags;                      /* LCCRF_REC_HAS_*                                    */
    uint32_t frame_id;                   /* mCurrentFrame.mnId, :1917                          */
    uint32_t n_iterations;               /* 5, :1929                                           */
    /* the CRF block of the settings file as Tracking holds it (Tracking.cc:151-171); same order
     * as lccrf_crf_params (include/lccrf.h) */
    float    w1, w2;
    float    u_alpha, stdev_alpha;
    float    u_beta, stdev_beta;
    float    u_gamma, stdev_gamma;
    float    point3d_stdev, point2d_stdev;
    float    u_depth, pth, confidence;
    uint32_t n_sections;                 /* version 2 with LCCRF_REC_HAS_SECTIONS, else zero   */
    float    reserved[2];                /* zero                                               */
} lccrf_rec_frame_header;

/* ---- version 2 sections ------------------------------------------------------------------------------------- */
#define LCCRF_SEC_UNARY   0x59524e55u    /* "UNRY" */
#define LCCRF_SEC_BFMATCH 0x544d4642u    /* "BFMT" */
#define LCCRF_SEC_POSE    0x45534f50u    /* "POSE" */

typedef struct lccrf_rec_section_header {   /* 16 bytes */
    uint32_t tag;                        /* LCCRF_SEC_*                                         */
    uint32_t flags;                      /* per-section, see below                              */
    uint64_t payload_bytes;              /* bytes behind this header, before the padding to 8   */
} lccrf_rec_section_header;

/* LCCRF_SEC_UNARY payload: lccrf_rec_unary_header, then
 *   float    Xw[n_cand][3]        pMP->GetWorldPos() of every keypoint i with mvpMapPoints[i] != NULL, in the order of the
 *                                 loop at Tracking.cc:1849 (BEFORE the observs == 0 filter of :1857)
 *   int32_t  fid[n_cand]          that keypoint's index i
 *   int32_t  obs_ptr[n_cand+1]    observations of candidate c are obs_*[obs_ptr[c] .. obs_ptr[c+1]), in the iteration
 *                                 order of pMP->GetObservations() (std::map<KeyFrame*, size_t>, Tracking.cc:1811-1813)
 *   int32_t  obs_kf[n_obs]        index into the keyframe arrays below
 *   double   obs_kp[n_obs][2]     pKF->mvKeysUn[it->second].pt as doubles (Point2d, :1832)
 *   float    kf_pose[n_kf][12]    row-major [Rcw | tcw] of pKF->GetPose() (:1815-1817)
 *   float    kf_intr[n_kf][4]     fx fy cx cy (:1827-1830)
 *   float    kf_bounds[n_kf][4]   mnMinX mnMaxX mnMinY mnMaxY (:1824)
 *   double   match_prob[n_cand]   only if flags & 1: mvFeatureMatchProb[fid] as read at :2003 (0 where operator[] inserted)
 *   -- outputs --
 *   float    observs[n_cand], error[n_cand], depth[n_cand]   the three by-reference results of :1856
 *   int16_t  rough_label[n_cand]  RroughClassify's label of the candidates that passed :1857, -1 for the dropped ones
 * The CRF arrays of the frame are the kept candidates in order: vobservs = observs[kept] etc.                        */
typedef struct lccrf_rec_unary_header {     /* 16 bytes */
    uint32_t n_cand, n_obs, n_kf, reserved;
} lccrf_rec_unary_header;

/* LCCRF_SEC_BFMATCH payload: lccrf_rec_bfmatch_header, then
 *   uint8_t  desc_query[n_query][32]   mCurrentFrame.mDescriptors rows (Tracking.cc:1752)
 *   uint8_t  desc_train[n_train][32]   the frame 15 back
 *   (zero bytes up to a multiple of 4)
 *   -- output --
 *   int32_t  asso[n_query]             asso[fid1] = fid2 of :1762, -1 where the map has no entry                     */
typedef struct lccrf_rec_bfmatch_header {   /* 16 bytes */
    uint32_t n_query, n_train;
    double   ratio;                      /* 0.6 at :1759                                         */
} lccrf_rec_bfmatch_header;

/* LCCRF_SEC_POSE payload: lccrf_rec_pose_header, then (n = pFrame->N keypoints, Optimizer.cc:262)
 *   float    Xw[n][3]             pMP->GetWorldPos() where valid, else zeros
 *   float    kp[n][2]             mvKeysUn[i].pt;        float u_right[n]   mvuRight[i]
 *   float    inv_sigma2[n]        mvInvLevelSigma2[kpUn.octave]
 *   uint8_t  valid[n]             mvpMapPoints[i] != NULL on entry (after the CRF's pruning, Tracking.cc:1945-1955)
 *   uint8_t  outlier[n]           mvbOutlier on RETURN (output)
 *   (zero bytes up to a multiple of 4)
 *   float    Tcw_in[16], Tcw_out[16]   pFrame->mTcw on entry / after SetPose (:443-445), row-major 4x4
 * flags & 1: int32_t crf_index[n] follows -- the CRF point index of keypoint i, or -1 (ties the section to the frame's
 * CRF arrays: label[crf_index[i]] == 0 is what nulled the map point).                                               */
typedef struct lccrf_rec_pose_header {      /* 32 bytes */
    uint32_t n_points;
    int32_t  n_inliers;                  /* the return value nInitialCorrespondences - nBad (:447) */
    float    fx, fy, cx, cy, bf;         /* pFrame->fx ... mbf                                   */
    uint32_t reserved;
} lccrf_rec_pose_header;

#ifdef __cplusplus
}
#endif
#endif
