// k_morph.hip -- K5: morphological open + close with the 10x10 MORPH_ELLIPSE element, the device
// side of SWMorphologicalFilter::run (/root/reference/filter/mf-sw.cpp:19-28; element size from
// /root/reference/include/filter/mf-sw.h:11-12).  Semantics: SURVEY.md Appendix B; oracle:
// oracle/morph_oracle.c.
//
// One workgroup produces a 64x32 output tile of ONE pass (erode or dilate); the source tile with
// its halo (5 up/left, 4 down/right) is staged in LDS, then each output is the min/max over the
// ten element rows, each row a contiguous run [j1, j2] of the staged tile.  Out-of-image samples
// are staged as the neutral value (255 for erode, 0 for dilate), which is OpenCV's constant
// border that never wins.
#include "rtdm_kernels.h"

namespace rtdm {

static constexpr int MT_W = 64, MT_H = 32;     // output tile
static constexpr int MH_L = 5, MH_R = 4;       // halo (anchor = (5,5) of a 10x10 element)
static constexpr int MS_W = MT_W + MH_L + MH_R + 3;  // 76: staged row stride (multiple of 4)
static constexpr int MS_H = MT_H + MH_L + MH_R;      // 41

// element rows of getStructuringElement(MORPH_ELLIPSE, Size(10,10)) -- SURVEY.md Appendix B
__constant__ int c_j1[10] = {5, 2, 1, 0, 0, 0, 0, 0, 1, 2};
__constant__ int c_j2[10] = {5, 8, 9, 9, 9, 9, 9, 9, 9, 8};

template <bool DILATE>
__global__ __launch_bounds__(256) void k_morph_pass(Plane8 in, Plane8W out, int W, int H)
{
    __shared__ uint8_t tile[MS_H * MS_W];
    const int f = blockIdx.z;
    const int tx0 = blockIdx.x * MT_W, ty0 = blockIdx.y * MT_H;
    const uint8_t* src = in.base + (size_t)f * in.frame;
    const uint8_t neutral = DILATE ? 0 : 255;
    for (int i = threadIdx.x; i < MS_H * MS_W; i += 256) {
        const int sy = i / MS_W, sx = i - sy * MS_W;
        const int y = ty0 + sy - MH_L, x = tx0 + sx - MH_L;
        uint8_t v = neutral;
        if (sx < MT_W + MH_L + MH_R && x >= 0 && x < W && y >= 0 && y < H) v = src[(size_t)y * in.pitch + x];
        tile[i] = v;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63;
    const int x = tx0 + lx;
    uint8_t* dst = out.base + (size_t)f * out.frame;
    for (int ly = threadIdx.x >> 6; ly < MT_H; ly += 4) {
        const int y = ty0 + ly;
        int acc = neutral;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const uint8_t* rowp = tile + (ly + i) * MS_W + lx;    // src(y + i - 5, x + j - 5)
            for (int j = c_j1[i]; j <= c_j2[i]; ++j) {
                const int v = rowp[j];
                acc = DILATE ? max(acc, v) : min(acc, v);
            }
        }
        if (x < W && y < H) dst[(size_t)y * out.pitch + x] = (uint8_t)acc;
    }
}

void launch_morph_open_close(Plane8 in, Plane8W out, uint8_t* tmp0, uint8_t* tmp1, int W, int H,
                             int n, hipStream_t stream)
{
    dim3 grid((W + MT_W - 1) / MT_W, (H + MT_H - 1) / MT_H, n), block(256);
    const size_t tp = (size_t)W, tf = (size_t)W * H;
    Plane8W t0w{tmp0, tp, tf}, t1w{tmp1, tp, tf};
    Plane8 t0r{tmp0, tp, tf}, t1r{tmp1, tp, tf};
    hipLaunchKernelGGL(k_morph_pass<false>, grid, block, 0, stream, in, t0w, W, H);    // erode
    hipLaunchKernelGGL(k_morph_pass<true>, grid, block, 0, stream, t0r, t1w, W, H);    // dilate
    hipLaunchKernelGGL(k_morph_pass<true>, grid, block, 0, stream, t1r, t0w, W, H);    // dilate
    hipLaunchKernelGGL(k_morph_pass<false>, grid, block, 0, stream, t0r, out, W, H);   // erode
}

}  // namespace rtdm
