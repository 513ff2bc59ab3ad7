/*
 * rtdm_oracle.h -- CPU oracle for the rt-depth-map block-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the HIP library (librtdm_hip.so)
 * never links, loads or calls it.
 *
 * PARITY UNPINNED.  The reference's hot path is a 48-line wrapper
 * (/root/reference/stereo-matcher/bm-sw.cpp:12-38) that forwards to OpenCV's
 * cv::StereoBM; the morphological filter (/root/reference/filter/mf-sw.cpp:19-28)
 * forwards to cv::erode / cv::dilate.  OpenCV is an un-vendored, un-pinned
 * system dependency (reference Makefile.include:18-23; era => OpenCV 3.1-3.2),
 * it is absent from the build image, and the reference holds no tests, golden
 * vectors or sample images.  This file therefore restates the *published*
 * algorithm of OpenCV calib3d (stereobm.cpp: prefilterXSobel,
 * findStereoCorrespondenceBM, getValidDisparityROI; stereosgbm.cpp:
 * validateDisparity, filterSpeckles) and imgproc (getStructuringElement,
 * erode, dilate), anchored on the reference's call sites:
 *   - parameters and call order: bm-sw.cpp:16-25,35; main.cpp:134-135
 *   - ROI forwarding:            bm-sw.cpp:40-48; estimator.cpp:54
 *   - morphology sequence:       mf-sw.cpp:22-27; mf-sw.h:11-12
 * It is cross-checked by an independent brute-force implementation
 * (tests/bruteforce.py) and by known-answer properties (tests/test_oracle_*.py).
 *
 * KNOWN VERSION HAZARDS (places where OpenCV releases differ or where this restatement had to choose; none can be
 * closed without the library, all are outside what the reference's own settings can distinguish unless noted):
 *  H1 right-image border clamp in the SAD search (bm_oracle.c, rbase[]).  The window sample column j of the right
 *     image is clamp(rofs+j, 0, W-D) + d here, as in OpenCV 4.x ("width-rofs-ndisp").  The 3.1-3.2 era the reference
 *     links clamped to W-rofs-1 and let rptr[d] run past the end of the row (into the next row / the buffer's tail).
 *     Affected: only windows that contain a sample column j > width1-1, i.e. (minD = 0) the LAST w/2 output columns
 *     [W-w/2, W) of the search.  Those columns lie outside the valid rectangle and are masked; they reach the final
 *     map only as voters of validateDisparity, whose targets are x2 in [x-D-1, x], so no pixel left of column
 *     W-w/2-D-1 can change.  tests/test_oracle_bm.py::test_right_clamp_hazard_is_confined checks both statements.
 *     orc_bm_set_legacy_right_clamp(1) selects the 3.x rule (next-row bytes as the over-read).
 *  H2 cost plane width: 3.x's scalar path stores the cost through int* into a CV_16S plane; the SIMD
 *     path and 4.x use short.  Identical whenever 2*cap*w*w < 32768 (all reference / BASELINE settings: <= 10478).
 *     The oracle keeps int32.
 *  H3 RGB2GRAY fixed point (rectify_oracle.c): 14-bit coefficients 4899/9617/1868, (.. + 8192) >> 14, as in OpenCV
 *     2.x/3.x; 4.x switched the 8-bit path to 15 bits (9798/19235/3735, >> 15).  Differences are +-1 gray level on a
 *     fraction of pixels; they would propagate into the prefilter.  The reference's era (3.1-3.2) is the 14-bit one.
 *  H4 findContours border (objects_oracle.c, zero_border): OpenCV clears the 1-pixel image border before tracing
 *     (3.x always; later releases copy with a border instead).  Affected: components touching the frame edge lose their
 *     edge pixels in the bounding box (1 px per touched side).  A switch, default = the 3.x behaviour.
 *  H5 reprojectImageTo3D (depth_oracle.c): Z = (float)(Zh / Wh) here; OpenCV multiplies by the reciprocal
 *     (Z = Zh * (1/W) in double, then float).  A last-ulp float difference per pixel is possible; the reported mean is
 *     taken over >= hundreds of pixels in double, so the effect is below the printed precision (estimator.cpp:259).
 *  H6 remap INTER_LINEAR fixed point (rectify_oracle.c): weights from the 32x32 bilinear table in 15-bit fixed point,
 *     (sum + 16384) >> 15, saturate_cast not reachable for convex weights.  OpenCV builds the table by rounding float
 *     weights to short and fixing the sum to 32768 on the largest weight; the oracle's table construction follows that
 *     rule, but the tie rule for "largest weight" is from memory.
 *  H7 StereoSGBM (sgm_oracle.c): MODE_SGBM / MODE_HH restated from memory of stereosgbm.cpp 3.x, rules R1-R12 and three
 *     knowing deviations in that file's header.  The rules most likely to differ between releases: R1's overwritten
 *     border columns, R9's always-on check and its scaled initial value, R10's median (present since 2.4).
 */
#ifndef RTDM_ORACLE_H_
#define RTDM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Parameter block; field meaning follows cv::StereoBM as set by
 * SWMatcherKonolige's constructor (bm-sw.cpp:16-25). */
typedef struct orc_bm_params {
    int preFilterCap;      /* 1..63                     (main.cpp:134 -> 31) */
    int blockSize;         /* odd, 5..255, < min(W,H)   (main.cpp:134 -> 13) */
    int minDisparity;      /*                           (main.cpp:134 -> 0)  */
    int numDisparities;    /* > 0, multiple of 16       (cmdline-parser.cpp:22 -> 192) */
    int textureThreshold;  /* >= 0                      (main.cpp:134 -> 10) */
    int uniquenessRatio;   /* >= 0                      (main.cpp:135 -> 10) */
    int speckleWindowSize; /* 0 disables                (main.cpp:135 -> 100) */
    int speckleRange;      /* unscaled, in x16 units    (main.cpp:135 -> 32) */
    int disp12MaxDiff;     /* < 0 disables              (main.cpp:135 -> 1)  */
    int roi1[4];           /* x,y,w,h; w*h == 0 means "whole image" (setROI1, bm-sw.cpp:40-43) */
    int roi2[4];           /* x,y,w,h; never set by the reference (estimator.cpp:55) */
} orc_bm_params;

enum {
    ORC_OK = 0,
    ORC_ERR_BAD_PARAM = -1,
    ORC_ERR_BAD_SIZE = -2,
    ORC_ERR_COST_OVERFLOW = -3      /* StereoSGBM: block cost + P2 > 32767 somewhere in the frame (the library's 16-bit costs would wrap) */
};

/* X-Sobel prefilter (OpenCV prefilterXSobel): clip(sobel_x, -cap, cap) + cap. */
void orc_prefilter_xsobel(const uint8_t* src, size_t sstep, int W, int H,
                          uint8_t* dst, size_t dstep, int cap);

/* Full StereoBM::compute pipeline: prefilter -> SAD search on the valid rows ->
 * left-right check -> column masking -> speckle filter.  disp is 16SC1, fixed
 * point x16, invalid = (minDisparity-1)*16.  nthreads > 1 stripes the rows. */
int orc_bm_compute(const orc_bm_params* p, const uint8_t* L, size_t lstep,
                   const uint8_t* R, size_t rstep, int W, int H,
                   int16_t* disp, size_t dstep_bytes, int nthreads);

/* Individual stages, exposed so that tests can gate them one by one. */
int orc_bm_valid_rect(const orc_bm_params* p, int W, int H, int rect[4]);
/* SAD search only, on already prefiltered images, rows [row0,row1); writes
 * disp rows [row0,row1) for every column and cost (int32, same geometry). */
void orc_bm_search(const orc_bm_params* p, const uint8_t* Lp, size_t lstep,
                   const uint8_t* Rp, size_t rstep, int W, int H,
                   int row0, int row1, int16_t* disp, size_t dstep_elems,
                   int32_t* cost, size_t cstep_elems);
/* 0 (default): right-image border clamp of OpenCV 4.x; 1: the 3.x rule that over-reads the row (hazard H1 above). */
void orc_bm_set_legacy_right_clamp(int on);
void orc_validate_disparity(int16_t* disp, size_t dstep_elems, const int32_t* cost,
                            size_t cstep_elems, int W, int rows, int minD, int numD,
                            int disp12MaxDiff);
void orc_filter_speckles(int16_t* disp, size_t dstep_elems, int W, int H, int newVal,
                         int maxSpeckleSize, int maxDiff);

/* Morphology (mf-sw.cpp:19-28).  Element = getStructuringElement(MORPH_ELLIPSE,
 * Size(kw,kh)), anchor (kw/2, kh/2), constant border that never wins. */
void orc_ellipse_element(int kw, int kh, uint8_t* elem /* kh*kw */);
void orc_erode(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int W, int H,
               int kw, int kh);
void orc_dilate(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int W, int H,
                int kw, int kh);
/* erode -> dilate -> dilate -> erode with the 10x10 ellipse (MORPH_FILTER_DX/DY). */
void orc_morph_open_close(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep,
                          int W, int H);

/* ---- cv::StereoSGBM (rows S / f4; BASELINE config 5; restated in sgm_oracle.c, rules R1-R12 there) -------------
 * Cost volumes are uint16 [H][W1][D] with W1 = the columns [minD+D, W+min(minD,0)). */
typedef struct orc_sgm_params {
    int blockSize;          /* odd; sgbm-sw.cpp passes the constructor's blockSize               */
    int minDisparity;
    int numDisparities;     /* multiple of 16                                                    */
    int P1, P2;             /* sgbm-sw.cpp:17-18 -> 600, 2400; <= 0 -> 2 / 5, P2 >= P1 + 1 (R12)   */
    int uniquenessRatio;    /* < 0 -> 10                                                           */
    int speckleWindowSize;  /* > 0 enables filterSpeckles                                          */
    int speckleRange;       /* multiplied by 16 for filterSpeckles, as cv::StereoSGBM does        */
    int disp12MaxDiff;      /* <= 0 -> 1; the left-right check cannot be switched off (R9)        */
    int paths;              /* 5 = MODE_SGBM, the mode sgbm-sw.cpp:15 gets (left, right, down, down-right, down-left);
                             * 8 (or 0) = MODE_HH, all eight neighbours (BASELINE config 5)          */
} orc_sgm_params;

void orc_sgm_pixel_cost(const uint8_t* L, size_t lstep, const uint8_t* R, size_t rstep, int W, int H,
                        int minD, int D, uint16_t* cost);
uint32_t orc_sgm_block_cost(const uint16_t* pix, int W1, int H, int D, int blockSize, uint16_t* C);   /* -> the largest block cost */
void orc_sgm_aggregate(const uint16_t* C, int W1, int H, int D, int P1, int P2, uint16_t* S);            /* 8 paths */
void orc_sgm_aggregate_paths(const uint16_t* C, int W1, int H, int D, int P1, int P2, int paths, uint16_t* S);
void orc_sgm_select(const uint16_t* S, int W, int H, int D, int minD, int uniquenessRatio, int disp12MaxDiff,
                    int16_t* disp, size_t dstep_elems);
void orc_median3x3_s16(const int16_t* src, size_t sstep_elems, int16_t* dst, size_t dstep_elems, int W, int H);
int orc_sgm_compute(const orc_sgm_params* p, const uint8_t* L, size_t lstep, const uint8_t* R, size_t rstep,
                    int W, int H, int16_t* disp, size_t dstep_bytes);

/* ---- depth statistics after the matcher (defined in depth_oracle.c) ---------
 * disp16: the matcher's x16 fixed-point output.  regions: n x (x, y, width, height) inside the image.
 * mean_cm[i] = mean Z over the valid masked pixels of region i * calibration_unit / 10 (0 if none). */
int orc_depth_stats(const int16_t* disp16, size_t dstep_elems, int W, int H, const double Q[16],
                    const uint8_t* mask, size_t mstep, const int* regions, int n,
                    double calibration_unit, double* mean_cm, int* counts);

/* ---- rectification in front of the matcher (defined in rectify_oracle.c) ----------
 * rgb: H x W x 3 bytes, first channel = R.  map1: H x W x 2 int16 (sx, sy), map2: H x W uint16 (fy*32 + fx).
 * roi: x, y, width, height of the crop taken from the remapped frame (the reference's roif, main.cpp:80-85). */
void orc_rgb2gray(const uint8_t* rgb, size_t sstep, int W, int H, uint8_t* gray, size_t dstep);
void orc_remap_bilinear(const uint8_t* src, size_t sstep, int sW, int sH, int cn, const int16_t* map1,
                        const uint16_t* map2, int dW, int dH, uint8_t* dst, size_t dstep);
int orc_rectify_gray(const uint8_t* rgb, size_t sstep, int W, int H, const int16_t* map1, const uint16_t* map2,
                     const int roi[4], uint8_t* out, size_t ostep);
int orc_rectify_rgb(const uint8_t* rgb, size_t sstep, int W, int H, const int16_t* map1, const uint16_t* map2,
                    const int roi[4], uint8_t* out, size_t ostep);
int orc_init_undistort_rectify_map(const double M[9], const double D[14], const double R[9], const double P[12],
                                   int W, int H, int16_t* map1, uint16_t* map2);

/* ---- object detection that produces the matcher's ROI (defined in objects_oracle.c) ----
 * rgb: H x W x 3, first channel R (the rectified colour crop).  lo/hi: inclusive H, S, V bounds (estimator.cpp:110-115).
 * boxes: (x, y, width, height) of the external 8-connected components with area >= min_area, in the order of the
 * reference's obj_boundings; orc_external_boxes returns how many there are (only max_boxes are stored). */
void orc_rgb2hsv(const uint8_t* rgb, size_t sstep, int W, int H, uint8_t* hsv, size_t dstep);
void orc_hsv_inrange(const uint8_t* rgb, size_t sstep, int W, int H, const int lo[3], const int hi[3], uint8_t* mask, size_t mstep);
int orc_external_boxes(const uint8_t* mask, size_t mstep, int W, int H, int zero_border, int min_area, int* boxes, int max_boxes);
void orc_union_box(const int* boxes, int n, int roi[4]);

#ifdef __cplusplus
}
#endif
#endif
