/*
 * bm-hip.cpp -- install as stereo-matcher/bm-hip.cpp and add bm-hip.o to stereo-matcher/Makefile.
 * Mirrors stereo-matcher/bm-sw.cpp: the constructor forwards the nine StereoBM knobs, compute()
 * hands the Mats to the device module and returns its status (0 on success), setROI1/2 forward.
 */
#include "stereo-matcher/bm-hip.h"

static rtdm::Rect to_rect(const cv::Rect& r)
{
	rtdm::Rect o;
	o.x = r.x; o.y = r.y; o.width = r.width; o.height = r.height;
	return o;
}

HIPMatcher::HIPMatcher(cv::Rect& roi1, cv::Rect& roi2, int preFilterCap, int blockSize, int minDisparity,
		int textureThreshold, int numOfDisparities, int maxDisparity, int uniquenessRatio, int speckleWindowSize,
		int speckleRange, int disp12MaxDiff, int width, int height, int device, bool legacyRightClamp)
{
	core = new rtdm::HIPMatcherCore(to_rect(roi1), to_rect(roi2), preFilterCap, blockSize, minDisparity,
			textureThreshold, numOfDisparities, maxDisparity, uniquenessRatio, speckleWindowSize, speckleRange,
			disp12MaxDiff, width, height, 1, device, legacyRightClamp);
}

HIPMatcher::~HIPMatcher()
{
	delete core;
}

int HIPMatcher::compute(cv::InputArray left, cv::InputArray right, cv::OutputArray out)
{
	/* the caller passes ROI views (estimator.cpp:33,36): honour step, never assume continuity */
	cv::Mat l = left.getMat(), r = right.getMat();
	if (l.type() != CV_8UC1 || r.type() != CV_8UC1 || l.size() != r.size())
		return RTDM_ERR_BAD_SIZE;
	out.create(l.size(), CV_16SC1);
	cv::Mat d = out.getMat();
	return core->compute(l.data, l.step, r.data, r.step, l.rows, l.cols, (int16_t*) d.data, d.step);
}

void HIPMatcher::setROI1(cv::Rect roi1)
{
	core->setROI1(to_rect(roi1));
}

void HIPMatcher::setROI2(cv::Rect roi2)
{
	core->setROI2(to_rect(roi2));
}
