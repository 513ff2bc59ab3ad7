#include "hip_matcher_core.h"

#include <cstdio>

namespace rtdm {

HIPMatcherCore::HIPMatcherCore(const Rect& roi1, const Rect& roi2, int preFilterCap, int blockSize, int minDisparity,
                               int textureThreshold, int numOfDisparities, int maxDisparity, int uniquenessRatio,
                               int speckleWindowSize, int speckleRange, int disp12MaxDiff,
                               int maxWidth, int maxHeight, int maxBatch, int device, bool legacyRightClamp)
{
    (void)roi1; (void)roi2; (void)maxDisparity;   // ignored by the reference constructor as well (bm-sw.cpp:12-26)
    params_.preFilterCap = preFilterCap;
    params_.blockSize = blockSize;
    params_.minDisparity = minDisparity;
    params_.numDisparities = numOfDisparities;
    params_.textureThreshold = textureThreshold;
    params_.uniquenessRatio = uniquenessRatio;
    params_.speckleWindowSize = speckleWindowSize;
    params_.speckleRange = speckleRange;
    params_.disp12MaxDiff = disp12MaxDiff;
    params_.legacy_right_clamp = legacyRightClamp ? 1 : 0;
    status_ = rtdm_bm_create(&params_, maxWidth, maxHeight, maxBatch, device, &bm_);
    if (status_ != RTDM_OK)
        std::fprintf(stderr, "HIPMatcher: %s%s%s\n", rtdm_strerror(status_),
                     status_ == RTDM_ERR_HIP ? ": " : "", status_ == RTDM_ERR_HIP ? rtdm_last_hip_error() : "");
}

HIPMatcherCore::~HIPMatcherCore() { rtdm_bm_destroy(bm_); }

void HIPMatcherCore::setROI1(const Rect& r) { if (bm_) status_ = rtdm_bm_set_roi(bm_, 1, r.x, r.y, r.width, r.height); }
void HIPMatcherCore::setROI2(const Rect& r) { if (bm_) status_ = rtdm_bm_set_roi(bm_, 2, r.x, r.y, r.width, r.height); }

int HIPMatcherCore::compute(const uint8_t* left, size_t leftStep, const uint8_t* right, size_t rightStep,
                            int rows, int cols, int16_t* out, size_t outStep)
{
    if (!bm_) return status_ != RTDM_OK ? status_ : RTDM_ERR_NO_DEVICE;   // loud: no CPU fallback
    status_ = rtdm_bm_compute(bm_, left, leftStep, right, rightStep, cols, rows, out, outStep);
    return status_;
}

int HIPMatcherCore::computeBatch(int n, const uint8_t* left, const uint8_t* right, size_t step, size_t frameStride,
                                 int rows, int cols, int16_t* out, size_t outStep, size_t outFrameStride)
{
    if (!bm_) return status_ != RTDM_OK ? status_ : RTDM_ERR_NO_DEVICE;
    status_ = rtdm_bm_compute_batch(bm_, n, left, right, step, frameStride, cols, rows, out, outStep, outFrameStride);
    return status_;
}

int HIPMatcherCore::computeDepth(const uint8_t* left, size_t leftStep, const uint8_t* right, size_t rightStep, int rows,
                                 int cols, const double* Q, const uint8_t* mask, size_t maskStep, const Rect* regions,
                                 int nregions, double calibrationUnit, double* meanCm, int* counts, int16_t* out, size_t outStep)
{
    if (!bm_) return status_ != RTDM_OK ? status_ : RTDM_ERR_NO_DEVICE;
    static_assert(sizeof(Rect) == sizeof(rtdm_region), "Rect and rtdm_region share their layout");
    status_ = rtdm_bm_compute_depth(bm_, left, leftStep, right, rightStep, cols, rows, Q, mask, maskStep,
                                    reinterpret_cast<const rtdm_region*>(regions), nregions, calibrationUnit, meanCm, counts,
                                    out, outStep);
    return status_;
}

HIPRectifierCore::HIPRectifierCore(const int16_t* map1Left, const uint16_t* map2Left, const int16_t* map1Right,
                                   const uint16_t* map2Right, int rows, int cols, const Rect& roif, int device)
{
    status_ = rtdm_rectify_create(map1Left, map2Left, map1Right, map2Right, cols, rows, roif.x, roif.y, roif.width, roif.height,
                                  1, device, &rc_);
    if (status_ != RTDM_OK) std::fprintf(stderr, "HIPRectifier: %s\n", rtdm_strerror(status_));
}
HIPRectifierCore::~HIPRectifierCore() { rtdm_rectify_destroy(rc_); }
int HIPRectifierCore::rectifyGray(const uint8_t* rgbLeft, size_t leftStep, const uint8_t* rgbRight, size_t rightStep,
                                  uint8_t* leftRect, size_t leftRectStep, uint8_t* rightRect, size_t rightRectStep)
{
    if (!rc_) return status_ != RTDM_OK ? status_ : RTDM_ERR_NO_DEVICE;
    return status_ = rtdm_rectify_gray(rc_, rgbLeft, leftStep, rgbRight, rightStep, leftRect, leftRectStep, rightRect, rightRectStep);
}
int HIPRectifierCore::rectifyColour(int which, const uint8_t* rgb, size_t step, uint8_t* out, size_t outStep)
{
    if (!rc_) return status_ != RTDM_OK ? status_ : RTDM_ERR_NO_DEVICE;
    return status_ = rtdm_rectify_rgb(rc_, which, rgb, step, out, outStep);
}
int HIPRectifierCore::compute(HIPMatcherCore& matcher, const uint8_t* rgbLeft, size_t leftStep, const uint8_t* rgbRight,
                              size_t rightStep, int16_t* out, size_t outStep)
{
    if (!rc_ || !matcher.handle()) return status_ != RTDM_OK ? status_ : RTDM_ERR_NO_DEVICE;
    return status_ = rtdm_bm_compute_rgb(matcher.handle(), rc_, rgbLeft, leftStep, rgbRight, rightStep, out, outStep);
}

HIPObjectsCore::HIPObjectsCore(int cols, int rows, int device)
{
    setRange(0, 150, 0, 9, 255, 255);                                  // estimator.cpp:110-115
    status_ = rtdm_objects_create(cols, rows, device, &ob_);
    if (status_ != RTDM_OK) std::fprintf(stderr, "HIPObjects: %s\n", rtdm_strerror(status_));
}
HIPObjectsCore::~HIPObjectsCore() { rtdm_objects_destroy(ob_); }
void HIPObjectsCore::setRange(int lowH, int lowS, int lowV, int highH, int highS, int highV)
{
    range_.low[0] = lowH; range_.low[1] = lowS; range_.low[2] = lowV;
    range_.high[0] = highH; range_.high[1] = highS; range_.high[2] = highV;
}
int HIPObjectsCore::detect(const uint8_t* rgb, size_t step, int minObjSize, bool zeroBorder, uint8_t* maskOut, size_t maskStep,
                           Rect* boxes, int maxBoxes, Rect* matchingRoi)
{
    if (!ob_) return status_ != RTDM_OK ? status_ : RTDM_ERR_NO_DEVICE;
    static_assert(sizeof(Rect) == sizeof(rtdm_region), "Rect and rtdm_region share their layout");
    int n = 0;
    status_ = rtdm_objects_detect(ob_, rgb, step, &range_, minObjSize, zeroBorder ? 1 : 0, maskOut, maskStep,
                                  reinterpret_cast<rtdm_region*>(boxes), maxBoxes, &n, reinterpret_cast<rtdm_region*>(matchingRoi));
    return status_ == RTDM_OK ? n : status_;
}
int HIPObjectsCore::estimateFrame(HIPMatcherCore& matcher, HIPRectifierCore& rectifier, const uint8_t* rgbLeft, size_t leftStep,
                                  const uint8_t* rgbRight, size_t rightStep, const double* Q, int minObjSize, bool zeroBorder,
                                  double calibrationUnit, Rect* boxes, double* meanCm, int* counts, int maxBoxes,
                                  int16_t* disp, size_t dispStep)
{
    if (!ob_ || !matcher.handle() || !rectifier.handle()) return status_ != RTDM_OK ? status_ : RTDM_ERR_NO_DEVICE;
    int n = 0;
    status_ = rtdm_estimate_frame(matcher.handle(), rectifier.handle(), ob_, rgbLeft, leftStep, rgbRight, rightStep, Q, &range_,
                                  minObjSize, zeroBorder ? 1 : 0, calibrationUnit, reinterpret_cast<rtdm_region*>(boxes), meanCm,
                                  counts, maxBoxes, &n, disp, dispStep);
    return status_ == RTDM_OK ? n : status_;
}

HIPSGMCore::HIPSGMCore(int blockSize, int minDisparity, int numOfDisparities, int uniquenessRatio, int speckleWindowSize,
                       int speckleRange, int disp12MaxDiff, int maxWidth, int maxHeight, int device, int paths)
{
    rtdm_sgm_params p;
    rtdm_sgm_default_params(&p, numOfDisparities, blockSize);     // P1 = 8*3*5*5, P2 = 32*3*5*5 (sgbm-sw.cpp:17-18)
    p.minDisparity = minDisparity; p.uniquenessRatio = uniquenessRatio; p.speckleWindowSize = speckleWindowSize;
    p.speckleRange = speckleRange; p.disp12MaxDiff = disp12MaxDiff; p.paths = paths;
    status_ = rtdm_sgm_create(&p, maxWidth, maxHeight, 1, device, &sg_);
    if (status_ != RTDM_OK) std::fprintf(stderr, "HIPSemiGlobalMatcher: %s\n", rtdm_strerror(status_));
}
HIPSGMCore::~HIPSGMCore() { rtdm_sgm_destroy(sg_); }
int HIPSGMCore::compute(const uint8_t* left, size_t leftStep, const uint8_t* right, size_t rightStep, int rows, int cols,
                        int16_t* out, size_t outStep)
{
    if (!sg_) return status_ != RTDM_OK ? status_ : RTDM_ERR_NO_DEVICE;
    status_ = rtdm_sgm_compute(sg_, left, leftStep, right, rightStep, cols, rows, out, outStep);
    return status_;
}

HIPMorphCore::HIPMorphCore(int w, int h, int bpp, int device) : width_(w), height_(h), bpp_(bpp)
{
    status_ = (bpp == 8) ? rtdm_morph_create(w, h, 1, device, &mf_) : RTDM_ERR_UNSUPPORTED;
    if (status_ != RTDM_OK) std::fprintf(stderr, "HIPMorphologicalFilter: %s\n", rtdm_strerror(status_));
}
HIPMorphCore::~HIPMorphCore() { rtdm_morph_destroy(mf_); }
char* HIPMorphCore::getVideoInBuffer() { return (char*)rtdm_morph_in_buffer(mf_); }
char* HIPMorphCore::getVideoOutBuffer() { return (char*)rtdm_morph_out_buffer(mf_); }
int HIPMorphCore::run(const uint8_t* in, size_t inStep, uint8_t* out, size_t outStep, int rows, int cols)
{
    if (!mf_) return status_ != RTDM_OK ? status_ : RTDM_ERR_NO_DEVICE;
    status_ = rtdm_morph_run(mf_, in, inStep, out, outStep, cols, rows);
    return status_;
}

}  // namespace rtdm
