/*
 * rtdm.h -- C ABI of the MI355X stereo block-matching module (librtdm_hip.so).
 *
 * This is the drop-in boundary for the hot path of wafgo/rt-depth-map: a HIPMatcher class that
 * derives from the reference's BlockMatcher (include/stereo-matcher/stereo-matcher.h:13-19) and
 * sits next to SWMatcherKonolige / HWMatcherDisparityCoprocessor calls exactly these entry points
 * (the adapter is rt-depth-map_amd/host/bm-hip.{h,cpp}; INTEGRATION.md shows the three-line
 * change to main.cpp:134 and the Makefile.build rule).  Plain C types only: no OpenCV, no torch.
 *
 * Every function returns RTDM_OK (0) or a negative rtdm_status; nothing throws or aborts.  The
 * library has NO CPU fallback: without a usable HIP device every create call fails with
 * RTDM_ERR_NO_DEVICE.
 */
#ifndef RTDM_H_
#define RTDM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTDM_ABI_VERSION 3   /* 2: rtdm_bm_params.legacy_right_clamp, rtdm_bm_get_tuner_stats; 3: rtdm_sgm_get_pass_stats */

typedef enum rtdm_status {
    RTDM_OK = 0,
    RTDM_ERR_BAD_PARAM = -1,   /* what cv::StereoBM::compute rejects with cv::Exception */
    RTDM_ERR_BAD_SIZE = -2,    /* frame larger than the handle was created for, bad pitch ... */
    RTDM_ERR_NO_DEVICE = -3,
    RTDM_ERR_HIP = -4,         /* a HIP runtime call failed; see rtdm_last_hip_error() */
    RTDM_ERR_NOMEM = -5,
    RTDM_ERR_UNSUPPORTED = -6, /* valid for OpenCV but outside what this build implements */
    RTDM_ERR_NULL = -7
} rtdm_status;

/* Same nine knobs SWMatcherKonolige's constructor forwards to cv::StereoBM
 * (stereo-matcher/bm-sw.cpp:16-25; literals at main.cpp:134-135).  preFilterType is XSOBEL,
 * the cv::StereoBM default the reference never changes. */
typedef struct rtdm_bm_params {
    int preFilterCap;      /* 1..63 */
    int blockSize;         /* odd, 5..255, smaller than min(width,height) */
    int minDisparity;      /* >= -2047 and minDisparity + numDisparities <= 2047: the x16 output is 16 bits wide */
    int numDisparities;    /* > 0, multiple of 16 */
    int textureThreshold;  /* >= 0 */
    int uniquenessRatio;   /* >= 0 */
    int speckleWindowSize; /* <= 0 disables the speckle filter */
    int speckleRange;      /* compared unscaled against the x16 fixed-point disparities */
    int disp12MaxDiff;     /* < 0 disables the left-right check */
    /* 0 (default): the right image's window samples are clamped as in OpenCV 4.x (column <= W - numDisparities, so
     * base + d stays inside the row).  1: the rule of the 3.1-3.2 era the reference links (Makefile.include:18-23):
     * the base is clamped to W - rofs - 1 and base + d runs on into the following bytes of a step == W plane, i.e. into
     * the next row (zeros after the last row).  Only the last blockSize/2 searched columns differ, and they reach the
     * result only as voters of the left-right check (oracle/rtdm_oracle.h, hazard H1).  Still unpinned: no OpenCV here. */
    int legacy_right_clamp;
} rtdm_bm_params;

typedef struct rtdm_bm rtdm_bm;       /* one matcher = one GPU + one HIP stream + its workspace */
typedef struct rtdm_morph rtdm_morph; /* one morphological filter device */

const char* rtdm_strerror(int status);
const char* rtdm_last_hip_error(void);   /* text of the last failing HIP call on this thread */
int rtdm_abi_version(void);
int rtdm_device_count(int* count);
/* Fills the reference's literals: cap 31, block 13, minD 0, texture 10, uniqueness 10,
 * speckle 100/32, disp12MaxDiff 1 (main.cpp:134-135); numDisparities as given. */
void rtdm_bm_default_params(rtdm_bm_params* p, int numDisparities);

/* ---- BlockMatcher ------------------------------------------------------------------------
 * rtdm_bm_create   <- SWMatcherKonolige::SWMatcherKonolige (bm-sw.cpp:12-26); like the FPGA
 *                     matcher (bm-hw-ip.h:74) it is told the frame size up front and owns
 *                     device buffers for max_batch frames of max_width x max_height.
 * rtdm_bm_set_roi  <- SWMatcherKonolige::setROI1 / setROI2 (bm-sw.cpp:40-48); which = 1 | 2;
 *                     a zero-area rectangle means "whole image", as in cv::StereoBM.
 * rtdm_bm_compute  <- SWMatcherKonolige::compute (bm-sw.cpp:33-38): host 8UC1 left/right with
 *                     arbitrary row pitch (the caller passes ROI views, estimator.cpp:33,36),
 *                     host 16SC1 output, fixed point x16, invalid = (minDisparity-1)*16.
 *                     Synchronous.  Pageable planes are gathered through a page-locked staging area;
 *                     planes that are themselves page-locked (hipHostMalloc, hipHostRegister) are
 *                     read and written by DMA in place.
 */
int rtdm_bm_create(const rtdm_bm_params* params, int max_width, int max_height, int max_batch,
                   int device, rtdm_bm** out);
void rtdm_bm_destroy(rtdm_bm* bm);
int rtdm_bm_set_roi(rtdm_bm* bm, int which, int x, int y, int width, int height);
int rtdm_bm_get_params(const rtdm_bm* bm, rtdm_bm_params* out);
int rtdm_bm_compute(rtdm_bm* bm, const uint8_t* left, size_t left_pitch, const uint8_t* right,
                    size_t right_pitch, int width, int height, int16_t* disp, size_t disp_pitch);

/* Batched stream of independent pairs (BASELINE config 4).  Device-resident variant: frame i
 * of an image lives at base + i*frame_stride, rows `pitch` bytes apart; the work is enqueued on
 * `hip_stream` (a hipStream_t; NULL = the HIP null stream, which is also torch's default stream) and NOT synchronised.  n may
 * exceed max_batch; it is processed in chunks. */
int rtdm_bm_compute_device(rtdm_bm* bm, int n, const uint8_t* d_left, const uint8_t* d_right,
                           size_t pitch, size_t frame_stride, int width, int height,
                           int16_t* d_disp, size_t disp_pitch, size_t disp_frame_stride,
                           void* hip_stream);
/* Host variant: frames in host memory, processed in chunks; returns when `disp` is complete.  If left, right and disp are
 * page-locked (hipHostMalloc / hipHostRegister) the copies are DMA on streams of their own: chunk k+1 comes in and chunk
 * k-1 goes out while chunk k is computed (needs max_batch >= 2).  Pageable memory is staged by the HIP runtime, chunk by
 * chunk. */
int rtdm_bm_compute_batch(rtdm_bm* bm, int n, const uint8_t* left, const uint8_t* right,
                          size_t pitch, size_t frame_stride, int width, int height,
                          int16_t* disp, size_t disp_pitch, size_t disp_frame_stride);
int rtdm_bm_synchronize(rtdm_bm* bm);

/* Per-stage device timing with HIP events on the launching stream (bench.py's roofline leg).
 * Stages: 0 prefilter, 1 SAD search, 2 left-right check, 3 speckle filter. */
#define RTDM_STAGE_PREFILTER 0
#define RTDM_STAGE_SEARCH 1
#define RTDM_STAGE_LRCHECK 2
#define RTDM_STAGE_SPECKLE 3
#define RTDM_NUM_STAGES 4
int rtdm_bm_set_profiling(rtdm_bm* bm, int enabled);
int rtdm_bm_get_stage_time(rtdm_bm* bm, int stage, double* total_ms, long* launches, long* frames);
int rtdm_bm_reset_stage_times(rtdm_bm* bm);
/* Name of the SAD-search kernel variant the current parameters select ("generic_u16", ...). */
const char* rtdm_bm_search_variant(const rtdm_bm* bm);
/* Diagnostic counters of the strip-count tuner inside rtdm_bm_compute_device (it times a few search launches the second
 * time a batch shape is seen): shapes measured so far and the extra search launches that took.  Either pointer may be NULL. */
int rtdm_bm_get_tuner_stats(const rtdm_bm* bm, long* shapes_measured, long* timing_launches);
/* Diagnostic A/B switch, process wide: which of the hand-written search kernels may be chosen for configurations that
 * several cover.  0: k_search_fast only; 1: k_search_ring where it is instantiated; 2 / 4 / 8: as 1, with two / four / eight
 * lanes per pixel where that form of the ring kernel exists; -1 (default): the library's choice (environment RTDM_RING=0/1 and
 * RTDM_RING_LPP=2/4 override it).  Results never depend on it. */
void rtdm_debug_search_kernel(int mode);

/* ---- VideoFilterDevice (morphological open + close, 10x10 ellipse) -----------------------
 * rtdm_morph_create      <- SWMorphologicalFilter::SWMorphologicalFilter (filter/mf-sw.cpp:10-17)
 * rtdm_morph_in_buffer / _out_buffer
 *                        <- VideoFilterDevice::getVideoInBuffer / getVideoOutBuffer
 *                           (filter/filter.cpp:45-53): width*height bytes each, owned by the
 *                           device object, here page-locked host memory.
 * rtdm_morph_run         <- SWMorphologicalFilter::run (filter/mf-sw.cpp:19-28): erode, dilate,
 *                           dilate, erode with MORPH_ELLIPSE 10x10 (mf-sw.h:11-12).  Synchronous.
 */
int rtdm_morph_create(int width, int height, int max_batch, int device, rtdm_morph** out);
void rtdm_morph_destroy(rtdm_morph* mf);
uint8_t* rtdm_morph_in_buffer(rtdm_morph* mf);
uint8_t* rtdm_morph_out_buffer(rtdm_morph* mf);
int rtdm_morph_run(rtdm_morph* mf, const uint8_t* in, size_t in_pitch, uint8_t* out,
                   size_t out_pitch, int width, int height);
int rtdm_morph_run_device(rtdm_morph* mf, int n, const uint8_t* d_in, size_t in_pitch,
                          size_t in_frame_stride, uint8_t* d_out, size_t out_pitch,
                          size_t out_frame_stride, int width, int height, void* hip_stream);

/* ---- SWSemiGlobalMatcher counterpart: cv::StereoSGBM (rows S / f4, BASELINE config 5) -------------------
 * rtdm_sgm_create   <- SWSemiGlobalMatcher::SWSemiGlobalMatcher (stereo-matcher/sgbm-sw.cpp:12-25):
 *                      StereoSGBM::create(0, numDisparities, blockSize), P1 = 600, P2 = 2400 (:17-18),
 *                      then the five setters (:19-24).  preFilterCap stays 0 (=> clip at +-15), mode MODE_SGBM.
 * rtdm_sgm_compute  <- SWSemiGlobalMatcher::compute (sgbm-sw.cpp:32-37); setROI1/2 are no-ops in the
 *                      reference (sgbm-sw.h:32-33), so there is no ROI entry point.
 * The algorithm is the restatement of cv::StereoSGBM::compute in oracle/sgm_oracle.c (rules R1-R12 there: pixel cost,
 * block sum, 5 or 8 path directions, saturating sum, winner / uniqueness / sub-pixel, the always-on left-right check,
 * 3x3 median, speckle filter), integer arithmetic, bit-exact against that oracle; parity against the library itself is
 * unpinned (OpenCV is not available where this was built). */
typedef struct rtdm_sgm_params {
    int blockSize;         /* 1..255; an even size runs as the next odd one, as in the library (window = blockSize / 2 either
                            * side).  Where 93 * window^2 + P2 > 32767 (window > 17 at P2 = 2400) a block cost + P2 CAN pass
                            * 32767, where the library's 16-bit costs wrap around -- which is not reproduced: a frame in which
                            * it does is refused by the compute call (RTDM_ERR_UNSUPPORTED; it takes nearly every pixel of a
                            * window at the maximum pixel cost), and rtdm_sgm_compute_device synchronises its stream to tell */
    int minDisparity;
    int numDisparities;    /* multiple of 16, <= 256 */
    int P1, P2;            /* as the library: P1 <= 0 -> 2, P2 <= 0 -> 5, P2 >= P1 + 1 */
    int uniquenessRatio;   /* <= 100; < 0 -> 10 */
    int speckleWindowSize; /* <= 0 disables the speckle filter */
    int speckleRange;      /* multiplied by 16, as cv::StereoSGBM does */
    int disp12MaxDiff;     /* <= 0 -> 1: the library's left-right check cannot be switched off */
    int paths;             /* 5: MODE_SGBM, the mode sgbm-sw.cpp:15 creates (left, right, down, down-right, down-left);
                            * 8: MODE_HH, all eight neighbours (BASELINE config 5) */
} rtdm_sgm_params;
typedef struct rtdm_sgm rtdm_sgm;
/* blockSize as given, minD 0, P1 600, P2 2400 (sgbm-sw.cpp:17-18), uniqueness 10, speckle 100/32,
 * disp12MaxDiff 1 (the literals main.cpp:134-135 passes to the BM matcher), 8 paths. */
void rtdm_sgm_default_params(rtdm_sgm_params* p, int numDisparities, int blockSize);
int rtdm_sgm_create(const rtdm_sgm_params* params, int max_width, int max_height, int max_batch,
                    int device, rtdm_sgm** out);
void rtdm_sgm_destroy(rtdm_sgm* sg);
int rtdm_sgm_compute(rtdm_sgm* sg, const uint8_t* left, size_t left_pitch, const uint8_t* right,
                     size_t right_pitch, int width, int height, int16_t* disp, size_t disp_pitch);
int rtdm_sgm_compute_device(rtdm_sgm* sg, int n, const uint8_t* d_left, const uint8_t* d_right,
                            size_t pitch, size_t frame_stride, int width, int height,
                            int16_t* d_disp, size_t disp_pitch, size_t disp_frame_stride, void* hip_stream);
/* How the path directions of this handle's calls have run so far: *sweeps = row-synchronous passes launched (three directions
 * each: k_sgm_sweep), *gave_up = 1 once such a pass has given up waiting for a neighbouring strip (the call that finds this
 * returns RTDM_ERR_HIP once; from then on the handle runs one pass per direction).  Either pointer may be NULL. */
int rtdm_sgm_get_pass_stats(const rtdm_sgm* sg, long* sweeps, int* gave_up);

/* ---- the step after the matcher, kept on the device ------------
 * rtdm_bm_compute_depth <- estimator.cpp:56 + 75-77: bm->compute(...); left_disp /= 16.;
 *                          reprojectImageTo3D(left_disp, xyz, Q, true, CV_32F); calc_depth(...) (206-263).
 *                          The disparity map stays in HBM; only mean Z [cm = Z * unit / 10] and the pixel
 *                          count of every region come back (disp may be NULL; if given it also receives the
 *                          x16 map, like rtdm_bm_compute).  Q: 4x4 row major (stereoRectify's Q, main.cpp:92).
 *                          mask: 8UC1 host image (filter_out, estimator.cpp:45); regions: obj_boundings.
 * rtdm_depth_stats_device  the same reduction on a device-resident x16 disparity map and mask. */
typedef struct rtdm_region { int x, y, width, height; } rtdm_region;
#define RTDM_MAX_REGIONS 64
int rtdm_bm_compute_depth(rtdm_bm* bm, const uint8_t* left, size_t left_pitch, const uint8_t* right, size_t right_pitch,
                          int width, int height, const double* Q, const uint8_t* mask, size_t mask_pitch,
                          const rtdm_region* regions, int nregions, double calibration_unit,
                          double* mean_cm, int* counts, int16_t* disp, size_t disp_pitch);
int rtdm_depth_stats_device(int device, const int16_t* d_disp, size_t disp_pitch, int width, int height, const double* Q,
                            const uint8_t* d_mask, size_t mask_pitch, const rtdm_region* regions, int nregions,
                            double calibration_unit, double* mean_cm, int* counts, void* hip_stream);

/* ---- the step in front of the matcher, on the device --------------
 * rtdm_rectify_create  <- the maps the reference builds once (main.cpp:95-96, initUndistortRectifyMap(..., CV_16SC2,
 *                         map1, map2)): map1 = H x W x 2 int16 (source x, y), map2 = H x W uint16 (fy*32 + fx);
 *                         roi = the crop `roif` applied to every remapped frame (main.cpp:80-85, estimator.cpp:33,36).
 *                         Only the roi part of the maps is kept on the device.
 * rtdm_rectify_gray    <- estimator.cpp:29-36: cvtColor(img[i], gray, CV_RGB2GRAY); remap(gray, rect, map1, map2,
 *                         INTER_LINEAR); rect = rect(roif) for both cameras.  rgb: H x W x 3 bytes, first channel R;
 *                         outputs: roi_h x roi_w bytes.
 * rtdm_rectify_rgb     <- estimator.cpp:38-39: remap(img[0], img_rectified, ...)(roif); out: roi_h x roi_w x 3.
 * rtdm_bm_compute_rgb  <- estimator.cpp:29-36 + 56 in one call: the rectified gray pair never leaves HBM; the
 *                         matcher must have been created for at least roi_w x roi_h.
 * The *_device forms take n contiguous device frames (n x H x W x 3 in, n x roi_h x roi_w out). */
typedef struct rtdm_rectify rtdm_rectify;
int rtdm_rectify_create(const int16_t* map1_left, const uint16_t* map2_left, const int16_t* map1_right,
                        const uint16_t* map2_right, int width, int height, int roi_x, int roi_y, int roi_width,
                        int roi_height, int max_batch, int device, rtdm_rectify** out);
void rtdm_rectify_destroy(rtdm_rectify* rc);
int rtdm_rectify_gray(rtdm_rectify* rc, const uint8_t* rgb_left, size_t left_pitch, const uint8_t* rgb_right,
                      size_t right_pitch, uint8_t* left_rect, size_t left_rect_pitch, uint8_t* right_rect,
                      size_t right_rect_pitch);
int rtdm_rectify_rgb(rtdm_rectify* rc, int which /* 0 = left maps, 1 = right maps */, const uint8_t* rgb, size_t pitch,
                     uint8_t* out, size_t out_pitch);
int rtdm_rectify_gray_device(rtdm_rectify* rc, int n, const uint8_t* d_rgb_left, const uint8_t* d_rgb_right,
                             uint8_t* d_left_rect, uint8_t* d_right_rect, void* hip_stream);
int rtdm_bm_compute_rgb(rtdm_bm* bm, rtdm_rectify* rc, const uint8_t* rgb_left, size_t left_pitch,
                        const uint8_t* rgb_right, size_t right_pitch, int16_t* disp, size_t disp_pitch);
int rtdm_bm_compute_rgb_device(rtdm_bm* bm, rtdm_rectify* rc, int n, const uint8_t* d_rgb_left,
                               const uint8_t* d_rgb_right, int16_t* d_disp, void* hip_stream);

/* ---- the object detection that yields the matcher's ROI ---------------
 * rtdm_objects_detect <- estimator.cpp:40-53: cvtColor(RGB2BGR) + cvtColor(BGR2HSV) + inRange(low, high) -> filter_in;
 *                        morphFilter->run(filter_in, filter_out); findContours(RETR_EXTERNAL) + boundingRect per contour,
 *                        boxes below min_area dropped (fill_bounding_rects_of_contours, estimator.cpp:167-174); roi = union
 *                        of the boxes (find_relevant_matching_region, estimator.cpp:176-204).  rgb = the rectified colour
 *                        crop (height x width x 3, R first).  boxes come in the order of the reference's obj_boundings;
 *                        *nboxes = how many there are (only max_boxes are stored).  zero_border = 1 restates
 *                        OpenCV <= 3.1's findContours, which clears the outermost rows/columns first; 0 = OpenCV >= 3.2.
 * rtdm_estimate_frame <- one iteration of Estimator::run without capture, decode and drawing (estimator.cpp:29-77):
 *                        raw RGB frames in, per object the box, mean Z [cm] and pixel count out.  Everything between
 *                        stays in HBM; one small read-back (the boxes) decides the matcher's ROI1.  If no box survives
 *                        the matcher is skipped (*nboxes = 0).  At most RTDM_MAX_REGIONS objects get a depth. */
typedef struct rtdm_objects rtdm_objects;
typedef struct rtdm_hsv_range { int low[3], high[3]; } rtdm_hsv_range;   /* H, S, V inclusive; estimator.cpp:110-115: {0,150,0}..{9,255,255} */
int rtdm_objects_create(int width, int height, int device, rtdm_objects** out);
void rtdm_objects_destroy(rtdm_objects* ob);
int rtdm_objects_detect(rtdm_objects* ob, const uint8_t* rgb, size_t pitch, const rtdm_hsv_range* range, int min_area,
                        int zero_border, uint8_t* mask_out, size_t mask_pitch, rtdm_region* boxes, int max_boxes,
                        int* nboxes, rtdm_region* roi);
int rtdm_estimate_frame(rtdm_bm* bm, rtdm_rectify* rc, rtdm_objects* ob, const uint8_t* rgb_left, size_t left_pitch,
                        const uint8_t* rgb_right, size_t right_pitch, const double* Q, const rtdm_hsv_range* range,
                        int min_area, int zero_border, double calibration_unit, rtdm_region* boxes, double* mean_cm,
                        int* counts, int max_boxes, int* nboxes, int16_t* disp, size_t disp_pitch);

/* ---- synthetic rectified-pair stream (stands in for stream/ + decoder/, which are out of
 * scope): frame f of the stream uses seed + f; bit-identical to rt-depth-map_amd/synth.py. */
int rtdm_synth_pairs_device(uint64_t seed, int first_frame, int n, int width, int height,
                            int numDisparities, uint8_t* d_left, uint8_t* d_right, size_t pitch,
                            size_t frame_stride, int device, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* RTDM_H_ */
