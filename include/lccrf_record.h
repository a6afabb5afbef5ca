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
 *
 * Everything is the value the reference holds at that line, bit for bit (floats are written as
 * their IEEE-754 bytes), so a replay can be compared with ref_label / ref_prob exactly.
 */
#ifndef LCCRF_RECORD_H
#define LCCRF_RECORD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCCRF_REC_MAGIC   "LCCRFREC"     /* 8 bytes, no terminator */
#define LCCRF_REC_VERSION 1u

enum {
    LCCRF_REC_HAS_MATCH_PROB = 1u << 0,
    LCCRF_REC_HAS_REF_LABEL  = 1u << 1,
    LCCRF_REC_HAS_REF_PROB   = 1u << 2,
};

typedef struct lccrf_rec_file_header {   /* 32 bytes */
    char     magic[8];                   /* LCCRF_REC_MAGIC                                    */
    uint32_t version;                    /* LCCRF_REC_VERSION                                  */
    uint32_t header_bytes;               /* sizeof(lccrf_rec_file_header): skip unknown tails  */
    uint32_t frame_header_bytes;         /* sizeof(lccrf_rec_frame_header)                     */
    uint32_t reserved[3];                /* zero                                               */
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
    float    reserved[3];                /* zero                                               */
} lccrf_rec_frame_header;

#ifdef __cplusplus
}
#endif
#endif
