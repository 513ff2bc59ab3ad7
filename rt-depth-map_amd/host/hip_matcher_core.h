// hip_matcher_core.h -- OpenCV-free C++ host layer over the C ABI (include/rtdm.h).
//
// HIPMatcherCore has the constructor shape, method names, argument meaning and error behaviour of
// the reference's SWMatcherKonolige (/root/reference/include/stereo-matcher/bm-sw.h:28-35):
// the same twelve constructor arguments in the same order (roi1, roi2 and maxDisparity are
// accepted and ignored, as bm-sw.cpp:12-26 does), compute() returns an int that is 0 on success,
// setROI1/setROI2 forward a rectangle.  The only differences are forced by the missing OpenCV:
// images are (pointer, pitch, rows, cols) instead of cv::InputArray, rectangles are rtdm::Rect.
// bm-hip.{h,cpp} wrap this class into the real `class HIPMatcher : public BlockMatcher`.
#pragma once

#include <cstddef>
#include <cstdint>

#include "../../include/rtdm.h"

namespace rtdm {

struct Rect { int x = 0, y = 0, width = 0, height = 0; };

class HIPMatcherCore {
public:
    // Frame size is needed up front because the device buffers are allocated once, like the
    // reference's FPGA matcher (bm-hw-ip.h:74: HWMatcherDisparityCoprocessor(name, width, height)).
    HIPMatcherCore(const Rect& roi1, const Rect& roi2, int preFilterCap, int blockSize, int minDisparity,
                   int textureThreshold, int numOfDisparities, int maxDisparity, int uniquenessRatio,
                   int speckleWindowSize, int speckleRange, int disp12MaxDiff,
                   int maxWidth, int maxHeight, int maxBatch = 1, int device = 0, bool legacyRightClamp = false);
    // legacyRightClamp: rtdm_bm_params.legacy_right_clamp -- the right-border sampling rule of OpenCV 3.1-3.2, the era
    // the reference links (Makefile.include:18-23); the default is the 4.x rule.
    ~HIPMatcherCore();
    HIPMatcherCore(const HIPMatcherCore&) = delete;
    HIPMatcherCore& operator=(const HIPMatcherCore&) = delete;

    void setROI1(const Rect& roi1);
    void setROI2(const Rect& roi2);
    // 8-bit single-channel inputs with free row pitch, 16-bit signed fixed-point (x16) output.
    // Returns 0 on success, a negative rtdm_status otherwise (the reference's compute() returns
    // int and its caller ignores it, estimator.cpp:56; nothing here throws).
    int compute(const uint8_t* left, size_t leftStep, const uint8_t* right, size_t rightStep,
                int rows, int cols, int16_t* out, size_t outStep);
    // n contiguous frames (frameStride bytes apart), host memory, processed in device batches.
    int computeBatch(int n, const uint8_t* left, const uint8_t* right, size_t step, size_t frameStride,
                     int rows, int cols, int16_t* out, size_t outStep, size_t outFrameStride);

    // The caller's next three lines in one call (estimator.cpp:75-77): left_disp /= 16., reprojectImageTo3D with
    // Q (row-major 4x4), calc_depth over `regions` under `mask`; the disparity stays on the device unless `out`
    // is given.  meanCm[i] = mean Z * calibrationUnit / 10, counts[i] = pixels that entered the mean.
    int computeDepth(const uint8_t* left, size_t leftStep, const uint8_t* right, size_t rightStep, int rows, int cols,
                     const double* Q, const uint8_t* mask, size_t maskStep, const Rect* regions, int nregions,
                     double calibrationUnit, double* meanCm, int* counts, int16_t* out = nullptr, size_t outStep = 0);

    int status() const { return status_; }           // result of construction / last call
    const char* statusText() const { return rtdm_strerror(status_); }
    int filteredValue() const { return (params_.minDisparity - 1) * 16; }
    rtdm_bm* handle() { return bm_; }

private:
    rtdm_bm_params params_;
    rtdm_bm* bm_ = nullptr;
    int status_ = RTDM_OK;
};

// SWSemiGlobalMatcher counterpart (/root/reference/include/stereo-matcher/sgbm-sw.h:24-37): the same seven
// constructor arguments in the same order plus the frame size; P1 = 600 and P2 = 2400 as sgbm-sw.cpp:17-18
// hard-codes them; setROI1/2 are no-ops there too.  paths = 8 (BASELINE config 5) or 5: the directions of
// cv::StereoSGBM's default MODE_SGBM, which is what sgbm-sw.cpp:15 creates -- the adapter sgbm-hip.cpp passes 5.
class HIPSGMCore {
public:
    HIPSGMCore(int blockSize, int minDisparity, int numOfDisparities, int uniquenessRatio, int speckleWindowSize,
               int speckleRange, int disp12MaxDiff, int maxWidth, int maxHeight, int device = 0, int paths = 8);
    ~HIPSGMCore();
    HIPSGMCore(const HIPSGMCore&) = delete;
    HIPSGMCore& operator=(const HIPSGMCore&) = delete;
    int compute(const uint8_t* left, size_t leftStep, const uint8_t* right, size_t rightStep,
                int rows, int cols, int16_t* out, size_t outStep);
    int status() const { return status_; }

private:
    rtdm_sgm* sg_ = nullptr;
    int status_ = RTDM_OK;
};

// The caller's lines in front of the matcher (/root/reference/estimator.cpp:29-39): cvtColor(RGB2GRAY) + remap with
// the CV_16SC2 maps of main.cpp:95-96 + crop to roif, for both cameras, on the device.  Not behind an interface in
// the reference (OpenCV is called inline there), so using it means replacing those lines.
class HIPRectifierCore {
public:
    // map1*: rows x cols x 2 int16, map2*: rows x cols uint16 (continuous Mats: remap_left1.data ...); roif = crop
    HIPRectifierCore(const int16_t* map1Left, const uint16_t* map2Left, const int16_t* map1Right, const uint16_t* map2Right,
                     int rows, int cols, const Rect& roif, int device = 0);
    ~HIPRectifierCore();
    HIPRectifierCore(const HIPRectifierCore&) = delete;
    HIPRectifierCore& operator=(const HIPRectifierCore&) = delete;
    // rgb*: rows x cols x 3 bytes (first channel R); *Rect: roif.height x roif.width bytes
    int rectifyGray(const uint8_t* rgbLeft, size_t leftStep, const uint8_t* rgbRight, size_t rightStep,
                    uint8_t* leftRect, size_t leftRectStep, uint8_t* rightRect, size_t rightRectStep);
    // remap + crop of the colour frame (estimator.cpp:38-39); which = 0: left maps, 1: right maps
    int rectifyColour(int which, const uint8_t* rgb, size_t step, uint8_t* out, size_t outStep);
    // estimator.cpp:29-36 + 56 in one call: raw frames in, x16 disparity of the cropped pair out
    int compute(HIPMatcherCore& matcher, const uint8_t* rgbLeft, size_t leftStep, const uint8_t* rgbRight, size_t rightStep,
                int16_t* out, size_t outStep);
    int status() const { return status_; }
    rtdm_rectify* handle() { return rc_; }

private:
    rtdm_rectify* rc_ = nullptr;
    int status_ = RTDM_OK;
};

// The caller's lines between rectification and the matcher (/root/reference/estimator.cpp:40-53): HSV threshold of the
// rectified colour crop, the morphological filter, findContours(RETR_EXTERNAL) + boundingRect + the minimum-size filter
// (fill_bounding_rects_of_contours, :167-174) and the union box (find_relevant_matching_region, :176-204).
// estimateFrame() is one whole iteration of Estimator::run without capture, decode and drawing (:29-77).
class HIPObjectsCore {
public:
    HIPObjectsCore(int cols, int rows, int device = 0);       // size of the crop (roif)
    ~HIPObjectsCore();
    HIPObjectsCore(const HIPObjectsCore&) = delete;
    HIPObjectsCore& operator=(const HIPObjectsCore&) = delete;
    // thresholds as the Estimator's members iLowH ... iHighV (estimator.cpp:110-115)
    void setRange(int lowH, int lowS, int lowV, int highH, int highS, int highV);
    // rgb: rows x cols x 3 (R first); maskOut (filter_out) may be null.  Returns the number of boxes (>= 0; only
    // maxBoxes are stored) or a negative status.  zeroBorder: true = findContours of OpenCV <= 3.1.
    int detect(const uint8_t* rgb, size_t step, int minObjSize, bool zeroBorder, uint8_t* maskOut, size_t maskStep,
               Rect* boxes, int maxBoxes, Rect* matchingRoi);
    int estimateFrame(HIPMatcherCore& matcher, HIPRectifierCore& rectifier, const uint8_t* rgbLeft, size_t leftStep,
                      const uint8_t* rgbRight, size_t rightStep, const double* Q, int minObjSize, bool zeroBorder,
                      double calibrationUnit, Rect* boxes, double* meanCm, int* counts, int maxBoxes,
                      int16_t* disp = nullptr, size_t dispStep = 0);
    int status() const { return status_; }

private:
    rtdm_objects* ob_ = nullptr;
    rtdm_hsv_range range_;
    int status_ = RTDM_OK;
};

// VideoFilterDevice counterpart (/root/reference/include/filter/filter.h:13-37,
// /root/reference/filter/mf-sw.cpp:10-28): owns the frame buffers it hands out.
class HIPMorphCore {
public:
    HIPMorphCore(int w, int h, int bpp, int device = 0);
    ~HIPMorphCore();
    HIPMorphCore(const HIPMorphCore&) = delete;
    HIPMorphCore& operator=(const HIPMorphCore&) = delete;
    char* getVideoInBuffer();
    char* getVideoOutBuffer();
    int getFrameSize() const { return width_ * height_ * (bpp_ >> 3); }
    int run(const uint8_t* in, size_t inStep, uint8_t* out, size_t outStep, int rows, int cols);
    int status() const { return status_; }

private:
    rtdm_morph* mf_ = nullptr;
    int width_, height_, bpp_, status_ = RTDM_OK;
};

}  // namespace rtdm
