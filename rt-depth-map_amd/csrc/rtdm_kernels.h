// rtdm_kernels.h -- internal launch interface between the C-ABI layer (rtdm_api.hip) and the
// gfx950 kernels.  Everything here is device-pointer based and asynchronous on `stream`.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtdm {

// Geometry of one StereoBM search, derived once per call on the host.
struct BMGeom {
    int W, H;            // frame size
    int D, minD;         // numDisparities, minDisparity
    int w, r;            // blockSize, blockSize/2
    int cap, tex, uniq;  // preFilterCap, textureThreshold, uniquenessRatio
    int lofs, rofs, width1;
    int vx0, vx1, vy0, vy1;  // valid-disparity rectangle [vx0,vx1) x [vy0,vy1)
    int filtered;            // (minD-1)*16
    int mask_cols;           // 1: the search kernel masks columns outside [vx0,vx1) itself
    int want_cost;           // 1: a cost plane is written for the left-right check
};

struct Plane8 {  // batch of 8-bit images: frame f, row y at base + f*frame + y*pitch
    const uint8_t* base; size_t pitch, frame;
};
struct Plane8W { uint8_t* base; size_t pitch, frame; };
struct Plane16W { int16_t* base; size_t pitch_e, frame_e; };  // strides in elements

// K1: x-Sobel prefilter of n left and n right frames in one launch.
void launch_prefilter(Plane8 L, Plane8 R, Plane8W Lp, Plane8W Rp, int W, int H, int cap, int n,
                      hipStream_t stream);

// Fill a batch of disparity frames with FILTERED.
void launch_fill16(Plane16W disp, int W, int H, int n, int value, hipStream_t stream);

// K2 (generic variant): any D <= 256, any odd w, LDS column sums; writes disp (+ int32 cost).
// Returns false if the configuration does not fit (caller reports RTDM_ERR_UNSUPPORTED).
bool generic_search_supported(const BMGeom& g, bool* use16);
// [gx0, gx1) restricts the output-column range (gx1 < 0: all of [0, width1)).
void launch_search_generic(Plane8 Lp, Plane8 Rp, Plane16W disp, int32_t* cost, const BMGeom& g,
                           int n, hipStream_t stream, int gx0 = 0, int gx1 = -1);

// K2 (fast variant): packed-u8 quad-SAD kernel for the common configurations.  Works on the
// prefiltered planes and covers the output columns whose window needs no border clamping; the
// remaining border columns [lx0,lx1) and [rx0,rx1) go to the generic kernel.
bool fast_search_supported(const BMGeom& g);
void launch_search_fast(Plane8 Lp, Plane8 Rp, Plane16W disp, int32_t* cost, const BMGeom& g,
                        int n, hipStream_t stream);
void fast_border_ranges(const BMGeom& g, int* lx0, int* lx1, int* rx0, int* rx1);

// K3: row-local left-right consistency check (+ column masking to the valid rectangle).
void launch_lrcheck(Plane16W disp, const int32_t* cost, const BMGeom& g, int disp12MaxDiff, int n,
                    hipStream_t stream);

// K4: speckle filter (connected components under |a-b| <= maxDiff, size <= maxSize removed).
// label/size: n*W*H int32 each.
void launch_speckle(Plane16W disp, int32_t* label, int32_t* size, int W, int H, int n, int newVal,
                    int maxSize, int maxDiff, hipStream_t stream);

// K5: erode / dilate / dilate / erode with the 10x10 ellipse; tmp = n*W*H bytes scratch.
void launch_morph_open_close(Plane8 in, Plane8W out, uint8_t* tmp0, uint8_t* tmp1, int W, int H,
                             int n, hipStream_t stream);

// Synthetic stream generator (bit-identical to synth.py).
void launch_synth(uint64_t seed, int first_frame, int n, int W, int H, int D, Plane8W L, Plane8W R,
                  void* param_scratch, hipStream_t stream);
size_t synth_scratch_bytes(int n);

}  // namespace rtdm
