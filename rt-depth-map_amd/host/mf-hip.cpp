/*
 * mf-hip.cpp -- install as filter/mf-hip.cpp and add mf-hip.o to filter/Makefile.
 * Mirrors filter/mf-sw.cpp: the constructor publishes width/height/bpp and the two frame buffers
 * (page-locked here), run() performs erode, dilate, dilate, erode with the 10x10 ellipse.
 */
#include "filter/mf-hip.h"

HIPMorphologicalFilter::HIPMorphologicalFilter(int w, int h, int bpp)
{
	core = new rtdm::HIPMorphCore(w, h, bpp);
	video_in = core->getVideoInBuffer();
	video_out = core->getVideoOutBuffer();
	img_width = w;
	img_height = h;
	img_bpp = bpp;
}

HIPMorphologicalFilter::~HIPMorphologicalFilter()
{
	delete core;
}

int HIPMorphologicalFilter::run(cv::InputArray in, cv::OutputArray out)
{
	cv::Mat src = in.getMat();
	if (src.type() != CV_8UC1)
		return RTDM_ERR_BAD_SIZE;
	out.create(src.size(), CV_8UC1);
	cv::Mat dst = out.getMat();
	return core->run(src.data, src.step, dst.data, dst.step, src.rows, src.cols);
}
