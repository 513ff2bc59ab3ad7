// rtdm_kernels.h -- internal launch interface between the C-ABI layer (rtdm_api.hip) and the
// gfx950 kernels.  Everything here is device-pointer based and asynchronous on `stream`.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

namespace rtdm {

// Process-wide state shared BETWEEN handles (two handles may be driven from two host threads): environment switches are
// read through env_int() into a function-local `static const` (C++11 initialises those exactly once, thread-safely), and
// "once per device" actions go through OncePerDevice.  Nothing else in the launch paths is static and mutable.
inline int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
// first(dev) is true for exactly one caller per device; the others may run concurrently with that caller's action, so the
// action must be idempotent (hipFuncSetAttribute with a fixed value is).  Devices >= 64: always true.
struct OncePerDevice {
    std::atomic<unsigned long long> mask{0};
    bool first()
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64) return true;
        const unsigned long long bit = 1ull << dev;
        if (mask.load(std::memory_order_acquire) & bit) return false;
        return !(mask.fetch_or(bit, std::memory_order_acq_rel) & bit);
    }
};

// Geometry of one StereoBM search, derived once per call on the host.
struct BMGeom {
    int W, H;            // frame size
    int Ws;              // row stride (elements) of the per-pixel workspace planes (cost, labels, sizes, run lists,
                         // head map): W rounded up to 8 so that their rows start 16-byte aligned
    int D, minD;         // numDisparities, minDisparity
    int w, r;            // blockSize, blockSize/2
    int cap, tex, uniq;  // preFilterCap, textureThreshold, uniquenessRatio
    int lofs, rofs, width1;
    int vx0, vx1, vy0, vy1;  // valid-disparity rectangle [vx0,vx1) x [vy0,vy1)
    int cx0, cx1;            // image columns that have to be searched: the valid columns plus, when the
                             // left-right check is on, every column that can vote for them (setROI1 skips the rest)
    int filtered;            // (minD-1)*16
    int mask_cols;           // 1: the search kernel masks columns outside [vx0,vx1) itself
    int want_cost;           // 1: a cost plane is written for the left-right check
    int cost16;              // 1: the cost plane is uint16 (all SADs < 65536), else int32
    int legacy;              // rtdm_bm_params.legacy_right_clamp: right sample base clamped to W-rofs-1, base + d wraps into the next row
};

struct Plane8 {  // batch of 8-bit images: frame f, row y at base + f*frame + y*pitch
    const uint8_t* base; size_t pitch, frame;
};
struct Plane8W { uint8_t* base; size_t pitch, frame; };
struct Plane16W { int16_t* base; size_t pitch_e, frame_e; };  // strides in elements

// The internal prefiltered planes hold clamp(sobel) + cap + PREFILTER_BIAS: every byte is >= 1.  The packed search kernels
// need that (v_mqsad_pk_u16_u8 skips ZERO reference bytes: that is how a window's unused tail bytes are masked), absolute
// differences do not see a common offset, and the texture term becomes |p - (cap + PREFILTER_BIAS)|.  (Until round 3 every
// search wave added the bias to each staged dword itself.)  cap <= 63: the bytes stay <= 127.
static constexpr int PREFILTER_BIAS = 1;

// K1: x-Sobel prefilter of n left and n right frames in one launch (writes biased values, see above).
// fill != null: launch_fill_frame's job (below) for the same n frames rides in the same launch where the strip form runs
// (a single frame is bound by the number of its launches), and is launched by itself before the other forms.
struct FillJob { Plane16W disp; int cx0, cx1, vy0, vy1, value; int32_t* rowcnt; };
void launch_prefilter(Plane8 L, Plane8 R, Plane8W Lp, Plane8W Rp, int W, int H, int cap, int n,
                      hipStream_t stream, const FillJob* fill = nullptr);

// Fill the rectangle [x0,x1) x [y0,y1) of every disparity frame with `value`.
void launch_fill16(Plane16W disp, int x0, int x1, int y0, int y1, int n, int value, hipStream_t stream);
// The complement of [cx0, cx1) x [vy0, vy1) in every frame, plus (rowcnt != null) rowcnt[n][H] = 0, in one launch.
void launch_fill_frame(Plane16W disp, int W, int H, int cx0, int cx1, int vy0, int vy1, int n, int value, int32_t* rowcnt,
                       hipStream_t stream);

// K2 (generic variant): any D <= 256, any odd w, LDS column sums; writes disp (+ int32 cost).
// Returns false if the configuration does not fit (caller reports RTDM_ERR_UNSUPPORTED).
bool generic_search_supported(const BMGeom& g, bool* use16);
// [gx0, gx1) restricts the output-column range (gx1 < 0: all of [0, width1)).
void launch_search_generic(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g,
                           int n, hipStream_t stream, int gx0 = 0, int gx1 = -1);

// K2 (fast variant): packed-u8 quad-SAD kernel for the common configurations.  Works on the
// prefiltered planes and covers the output columns whose window needs no border clamping; the
// remaining border columns [lx0,lx1) and [rx0,rx1) go to the generic kernel.
bool fast_search_supported(const BMGeom& g);
// fuse_border: the border columns are searched by extra workgroups of the same launch.
// strips_hint > 0: row strips per frame to use (a measured choice); 0: fast_strips_model(g, n).
void launch_search_fast(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g,
                        int n, hipStream_t stream, bool fuse_border, int strips_hint = 0);
int fast_strips_model(const BMGeom& g, int n);
// K2 (ring variant, k_search_ring.hip): the same columns as the fast variant, prefix sums in a register ring instead of a
// leaving-row recomputation; covers the (D, blockSize) pairs whose ring fits two waves per SIMD.  fuse_border: the border
// columns may ride in front of the tile workgroups of the same launch; returns whether they did (otherwise they go to
// launch_search_border).  ring_search_supported implies fast_search_supported's column range.
bool ring_search_supported(const BMGeom& g);
bool launch_search_ring(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g, int n, hipStream_t stream,
                        int strips_hint = 0, bool fuse_border = false);
int ring_strips_model(const BMGeom& g, int n);
void ring_set_mode(int mode);   // rtdm_debug_search_kernel
int ring_lanes_per_pixel(const BMGeom& g);   // 2 / 4: the form of k_search_ring this configuration runs (0: none)
void fast_border_ranges(const BMGeom& g, int* lx0, int* lx1, int* rx0, int* rx1);
// wave-per-column kernel for those border columns (falls back to the generic kernel if unsupported)
bool border_search_supported(const BMGeom& g);
void launch_search_border(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g, int n,
                          hipStream_t stream, int lx0, int lx1, int rx0, int rx1);
// second form (k_search_border2.hip): one wave = one border column x 64 rows, rows in the lanes, disparities unrolled; 4-5x
// fewer instructions per pixel.  Returns false (nothing launched) for what it does not cover; launch_search_border tries it first.
struct Border2Geom;
bool border2_plan(const BMGeom& g, int lx0, int lx1, int rx0, int rx1, Border2Geom* bg, int* gx, int* gy);
bool launch_search_border2(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g, int n,
                           hipStream_t stream, int lx0, int lx1, int rx0, int rx1);

// K3: row-local left-right consistency check (+ column masking to the valid rectangle).  With
// label != nullptr the speckle filter's per-row init runs on the checked row in the same pass.
// Returns 0 if the speckle head map was written per pixel; otherwise it was written as per-chunk records (compact_heads for
// launch_speckle) and the value is the number of consecutive rows, from g.vy0, whose pairs the kernel has merged itself
// (1: none; premerged_rows for launch_speckle).
int launch_lrcheck(Plane16W disp, const void* cost, const BMGeom& g, int disp12MaxDiff, int n,
                   hipStream_t stream, int32_t* label = nullptr, int32_t* size = nullptr,
                   uint32_t* runs = nullptr, int32_t* rowcnt = nullptr, int16_t* headmap = nullptr, int spkDiff = 0);

// K4: speckle filter (connected components under |a-b| <= maxDiff, size <= maxSize removed).
// label/size/runs/headmap: n*W*H elements each, rowcnt: n*H.  init_done: rows [y_lo,y_hi) were
// initialised by the caller's row kernel (rowcnt zeroed beforehand) and every other row is entirely
// `newVal`; premerged_rows > 1: that kernel also merged the row pairs inside blocks of that many rows
// (lrcheck_rows_per_block() for launch_lrcheck).
void launch_speckle(Plane16W disp, int32_t* label, int32_t* size, uint32_t* runs, int32_t* rowcnt, int16_t* headmap,
                    int W, int Ws, int H, int n, int newVal, int maxSize, int maxDiff, bool init_done, int premerged_rows,
                    int y_lo, int y_hi, hipStream_t stream, bool compact_heads = false);
// Ws = row stride of label/size/runs/headmap (>= W)
// n frames of W x H int16: src -> dst (any pitches)
void launch_copy16(Plane16W src, Plane16W dst, int W, int H, int n, hipStream_t stream);
int lrcheck_rows_per_block();

// K5: erode / dilate / dilate / erode with the 10x10 ellipse; tmp = n*W*H bytes scratch.
void launch_morph_open_close(Plane8 in, Plane8W out, uint8_t* tmp0, uint8_t* tmp1, int W, int H,
                             int n, hipStream_t stream);

// SGM-8 (BASELINE config 5).  Cost volumes live on the column domain [x0, x0+W1).
struct SGMGeom { int W, H, D, minD, x0, W1; };
struct SGMBuffers {
    uint8_t *gl, *gr;        // Birchfield-Tomasi bounds of gradient and intensity, 2 x uchar4 per pixel  [n][H][W]
    uint8_t* pix;            // pixel cost                 [n][H][W1][D]
    uint16_t *C, *S;         // block cost, aggregated     [n][H][W1][D]
    uint16_t* S2;            // the (-1, 0) direction's path costs where the two horizontal directions run side by side (or null)
    int32_t *label, *size, *rowcnt; uint32_t* runs; int16_t* headmap;   // speckle filter workspace
    int32_t* ovf;            // set to 1 by the block-cost kernel where a block cost + P2 passes 32767 (windows > 17 only)
    // row-synchronous sweep (k_sgm_sweep): edge ring between neighbouring strips, give-up flag (page-locked, host readable),
    // the handle's launch counter (tags of the ring words), how many workgroups the device holds at once per instantiation
    unsigned long long* ring; size_t ring_words;
    int32_t* abortf; uint32_t* epoch;
    int* sweep_cap;          // [36], 0 = not asked yet, < 0 = unusable
    void *ev_in, *ev_out;    // hipEvent_t: the caller's stream -> the process's sweep stream -> the caller's stream
};
size_t sgm_ring_words(int maxW, int D, int max_batch);
// cost_limit > 0: block costs above it set *b.ovf (the caller reads it back: rtdm_api.hip)
void launch_sgm(Plane8 L, Plane8 R, Plane16W disp, const SGMGeom& g, const SGMBuffers& b, int blockSize, int P1, int P2,
                int uniq, int disp12MaxDiff, int speckleWindowSize, int speckleRange, int paths, int n, hipStream_t stream,
                int cost_limit = 0);

// Depth statistics after the matcher (estimator.cpp:75-77, 206-263).  q = the 4x4 reprojection matrix Q, row major.
struct DepthQ { double q[16]; };
size_t depth_scratch_bytes(int max_regions, int maxH);
// h_regions: n x (x, y, w, h) on the host, inside the image; rows of a region are summed by one workgroup each
// (maxH >= the tallest region).  Results arrive in h_mean / h_counts after the stream is synchronised.
void launch_depth_stats(const int16_t* disp, size_t pitch_e, int W, int H, const DepthQ& q, const uint8_t* mask, size_t mpitch,
                        const int* h_regions, int n, int maxH, double unit, void* scratch, double* h_mean, int* h_counts,
                        hipStream_t stream);

// Rectification in front of the matcher (estimator.cpp:29-39).  RectifySrc: n RGB frames, pitch/frame in bytes.
struct RectifySrc { const uint8_t* base; size_t pitch, frame; };
void launch_rectify_gray(RectifySrc L, RectifySrc R, const int16_t* map1L, const uint16_t* map2L, const int16_t* map1R,
                         const uint16_t* map2R, int sW, int sH, int rw, int rh, Plane8W outL, Plane8W outR, int n,
                         hipStream_t stream);
void launch_rectify_rgb(RectifySrc S, const int16_t* map1, const uint16_t* map2, int sW, int sH, int rw, int rh, Plane8W out,
                        int n, hipStream_t stream);

// Object detection that produces the matcher's ROI (estimator.cpp:40-53).  rgb: H x W x 3, R first.
void launch_hsv_inrange(const uint8_t* rgb, size_t pitch, int W, int H, const int lo[3], const int hi[3], uint8_t* mask,
                        size_t mpitch, hipStream_t stream);
size_t cc_scratch_bytes(int W, int H, int max_records);
// records: (first pixel index, x, y, w, h, external) per 8-connected foreground component, in no particular order
void launch_cc_boxes(const uint8_t* mask, size_t mpitch, int W, int H, int zero_border, void* scratch, int max_records,
                     int** d_count, int** d_records, hipStream_t stream);

// Synthetic stream generator (bit-identical to synth.py).
void launch_synth(uint64_t seed, int first_frame, int n, int W, int H, int D, Plane8W L, Plane8W R,
                  void* param_scratch, hipStream_t stream);
size_t synth_scratch_bytes(int n);

}  // namespace rtdm
