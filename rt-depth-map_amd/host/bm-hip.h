/*
 * bm-hip.h -- drop-in MI355X matcher for wafgo/rt-depth-map.
 *
 * Install as  include/stereo-matcher/bm-hip.h  next to bm-sw.h / bm-hw-ip.h of the reference tree.
 * HIPMatcher derives from the reference's BlockMatcher
 * (include/stereo-matcher/stereo-matcher.h:13-19) and has SWMatcherKonolige's constructor
 * (include/stereo-matcher/bm-sw.h:28-30) plus the frame size, so main.cpp:134-135 changes only
 * the class name and appends roif.width, roif.height; estimator.cpp is untouched.
 */
#ifndef INCLUDE_BM_BM_HIP_H_
#define INCLUDE_BM_BM_HIP_H_

#include "stereo-matcher/stereo-matcher.h"
#include "hip_matcher_core.h"

class HIPMatcher : public BlockMatcher
{
public:
	HIPMatcher(cv::Rect& roi1, cv::Rect& roi2, int preFilterCap, int blockSize, int minDisparity,
			int textureThreshold, int numOfDisparities, int maxDisparity, int uniquenessRatio, int speckleWindowSize,
			int speckleRange, int disp12MaxDiff, int width, int height, int device = 0, bool legacyRightClamp = false);
	~HIPMatcher();
	void setROI1(cv::Rect roi1);
	void setROI2(cv::Rect roi2);
	int compute(cv::InputArray left, cv::InputArray right, cv::OutputArray out);
private:
	rtdm::HIPMatcherCore* core;
};

#endif /* INCLUDE_BM_BM_HIP_H_ */
