// Minimal declarations of the handful of OpenCV core types that the reference's plugin headers
// (include/stereo-matcher/stereo-matcher.h, include/filter/filter.h) and this repository's
// adapters (rt-depth-map_amd/host/bm-hip.cpp, mf-hip.cpp) mention.  SYNTAX CHECK ONLY: it lets
// `g++ -fsyntax-only` verify that the adapters still match the reference's abstract interfaces
// on a machine without OpenCV.  Nothing is ever linked or run against it, and it is not used to
// build the reference or the oracle.
#pragma once
#include <cstddef>
namespace cv {
struct Rect { int x, y, width, height; };
struct Size { int width, height; bool operator!=(const Size& o) const { return width != o.width || height != o.height; } };
class Mat {
public:
    unsigned char* data; size_t step; int rows, cols;
    int type() const; Size size() const;
};
class _InputArray { public: Mat getMat() const; };
class _OutputArray : public _InputArray { public: void create(Size sz, int type) const; };
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}
#define CV_8UC1 0
#define CV_16SC1 3
