/*
 * sgbm-hip.cpp -- install as stereo-matcher/sgbm-hip.cpp and add sgbm-hip.o to stereo-matcher/Makefile.
 * Mirrors stereo-matcher/sgbm-sw.cpp: the constructor forwards the seven knobs (P1/P2 are the same literals),
 * compute() hands the Mats to the device module and returns its status.
 */
#include "stereo-matcher/sgbm-hip.h"

HIPSemiGlobalMatcher::HIPSemiGlobalMatcher(int blockSize, int minDisparity, int numOfDisparities, int uniquenessRatio,
		int speckleWindowSize, int speckleRange, int disp12MaxDiff, int width, int height)
{
	core = new rtdm::HIPSGMCore(blockSize, minDisparity, numOfDisparities, uniquenessRatio, speckleWindowSize,
			speckleRange, disp12MaxDiff, width, height, 0, 5 /* the five directions of StereoSGBM's default mode */);
}

HIPSemiGlobalMatcher::~HIPSemiGlobalMatcher()
{
	delete core;
}

int HIPSemiGlobalMatcher::compute(cv::InputArray left, cv::InputArray right, cv::OutputArray out)
{
	cv::Mat l = left.getMat(), r = right.getMat();
	if (l.type() != CV_8UC1 || r.type() != CV_8UC1 || l.size() != r.size())
		return RTDM_ERR_BAD_SIZE;
	out.create(l.size(), CV_16SC1);
	cv::Mat d = out.getMat();
	return core->compute(l.data, l.step, r.data, r.step, l.rows, l.cols, (int16_t*) d.data, d.step);
}
