/*
 * sgbm-hip.h -- install as include/stereo-matcher/sgbm-hip.h.  HIPSemiGlobalMatcher derives from the reference's
 * BlockMatcher with SWSemiGlobalMatcher's constructor (include/stereo-matcher/sgbm-sw.h:27-28) plus the frame size.
 * Like SWSemiGlobalMatcher it is not selected by main.cpp unless the maintainer does so.  The device module runs
 * cv::StereoSGBM's MODE_SGBM as restated in oracle/sgm_oracle.c (bit-exact against that restatement; against the
 * library itself parity is unpinned); an even blockSize runs as the next odd one (as in the library); with a window > 17 a frame whose block cost + P2 would pass 32767 (where the library's 16-bit costs wrap) makes compute return an error.
 */
#ifndef INCLUDE_BM_SGBM_HIP_H_
#define INCLUDE_BM_SGBM_HIP_H_

#include "stereo-matcher/stereo-matcher.h"
#include "hip_matcher_core.h"

class HIPSemiGlobalMatcher: public BlockMatcher
{
public:
	HIPSemiGlobalMatcher(int blockSize, int minDisparity, int numOfDisparities, int uniquenessRatio,
			int speckleWindowSize, int speckleRange, int disp12MaxDiff, int width, int height);
	~HIPSemiGlobalMatcher();
	int compute(cv::InputArray left, cv::InputArray right, cv::OutputArray out);
	void setROI1(cv::Rect roi1) {}
	void setROI2(cv::Rect roi2) {}
private:
	rtdm::HIPSGMCore* core;
};

#endif /* INCLUDE_BM_SGBM_HIP_H_ */
