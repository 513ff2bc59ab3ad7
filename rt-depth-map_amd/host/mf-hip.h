/*
 * mf-hip.h -- install as include/filter/mf-hip.h.  HIPMorphologicalFilter derives from the
 * reference's VideoFilterDevice (include/filter/filter.h:13-37) with SWMorphologicalFilter's
 * constructor (include/filter/mf-sw.h:19-24); main.cpp:133 changes only the class name.
 */
#ifndef INCLUDE_FILTER_MF_HIP_H_
#define INCLUDE_FILTER_MF_HIP_H_

#include "filter/filter.h"
#include "hip_matcher_core.h"

class HIPMorphologicalFilter: public VideoFilterDevice
{
public:
	explicit HIPMorphologicalFilter(int w, int h, int bpp);
	~HIPMorphologicalFilter();
	int run(cv::InputArray in, cv::OutputArray out);
private:
	rtdm::HIPMorphCore* core;
};

#endif /* INCLUDE_FILTER_MF_HIP_H_ */
