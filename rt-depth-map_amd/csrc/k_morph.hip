// k_morph.hip -- K5: morphological open + close with the 10x10 MORPH_ELLIPSE element, the device
// side of SWMorphologicalFilter::run (/root/reference/filter/mf-sw.cpp:19-28; element size from
// /root/reference/include/filter/mf-sw.h:11-12).  Semantics: SURVEY.md Appendix B; oracle:
// oracle/morph_oracle.c.
//
// ONE launch for all four passes (erode, dilate, dilate, erode).  A workgroup owns a 64x32 output
// tile; the source tile with a (20 up/left, 16 down/right) halo is staged in LDS once and each
// pass shrinks the live region by the element's reach (5 up/left, 4 down/right), so nothing but
// the input and the final output touches HBM (algorithmic bytes: 1 read + 1 write per pixel).
// The element's ten rows are runs of length 1, 7, 9, 10 x5, 9, 7 (Appendix B), so each pass first
// builds horizontal running minima (maxima) of length 7, 9 and 10 per row, then combines ten values
// vertically: ~20 LDS accesses per pixel and pass instead of 84.
// Out-of-image samples never win: they are staged as the neutral value of the pass that reads them
// (255 before an erosion, 0 before a dilation), which is OpenCV's default constant border.
#include "rtdm_kernels.h"

namespace rtdm {

static constexpr int TW = 64, TH = 32;          // output tile
static constexpr int RL = 5, RR = 4;            // reach of one pass: left/up, right/down
static constexpr int SW = TW + 4 * (RL + RR);   // 100 staged columns
static constexpr int SH = TH + 4 * (RL + RR);   // 68 staged rows
static constexpr int SP = 104;                  // row pitch of the LDS planes

template <bool DILATE> __device__ __forceinline__ int mm(int a, int b) { return DILATE ? max(a, b) : min(a, b); }

// One pass over the live region: input plane `in` valid on columns [c0, c0+cw) x rows [r0, r0+rh)
// (tile-local coordinates), output written to `out` on the region shrunk by the reach.
template <bool DILATE>
__device__ __forceinline__ void morph_pass(const uint8_t* in, uint8_t* out, uint8_t* h7, uint8_t* h9, uint8_t* h10,
                                           int c0, int r0, int cw, int rh, int gx0, int gy0, int W, int H, int next_neutral)
{
    // horizontal running extrema; h7[x] covers [x, x+7), h9 [x, x+9), h10 [x, x+10); each is produced
    // for every x whose span stays inside the live columns [c0, c0+cw)
    const int hw = cw - 6;
    for (int i = threadIdx.x; i < rh * hw; i += 256) {
        const int y = r0 + i / hw, xr = i % hw, x = c0 + xr;
        const uint8_t* p = in + y * SP + x;
        int m = p[0];
#pragma unroll
        for (int k = 1; k < 7; ++k) m = mm<DILATE>(m, p[k]);
        h7[y * SP + x] = (uint8_t)m;
        if (xr + 9 <= cw) {
            m = mm<DILATE>(m, mm<DILATE>(p[7], p[8]));
            h9[y * SP + x] = (uint8_t)m;
            if (xr + 10 <= cw) h10[y * SP + x] = (uint8_t)mm<DILATE>(m, p[9]);
        }
    }
    __syncthreads();
    // output (y, x) for x in [c0+5, c0+cw-4), y in [r0+5, r0+rh-4):  element row i reads source row
    // y+i-5, columns x+j-5 for j in its run (Appendix B): 5 | 2..8 | 1..9 | 0..9 x5 | 1..9 | 2..8
    const int ow = cw - (RL + RR), oh = rh - (RL + RR);
    for (int i = threadIdx.x; i < ow * oh; i += 256) {
        const int y = r0 + RL + i / ow, x = c0 + RL + i % ow;
        int m = in[(y - 5) * SP + x];
        m = mm<DILATE>(m, h7[(y - 4) * SP + x - 3]);
        m = mm<DILATE>(m, h9[(y - 3) * SP + x - 4]);
#pragma unroll
        for (int k = -2; k <= 2; ++k) m = mm<DILATE>(m, h10[(y + k) * SP + x - 5]);
        m = mm<DILATE>(m, h9[(y + 3) * SP + x - 4]);
        m = mm<DILATE>(m, h7[(y + 4) * SP + x - 3]);
        const int gx = gx0 + x, gy = gy0 + y;
        if (gx < 0 || gx >= W || gy < 0 || gy >= H) m = next_neutral;   // outside the image: never wins next pass
        out[y * SP + x] = (uint8_t)m;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_morph_open_close(Plane8 in, Plane8W out, int W, int H)
{
    __shared__ uint8_t A[SH * SP], B[SH * SP], h7[SH * SP], h9[SH * SP], h10[SH * SP];
    const int f = blockIdx.z;
    const int gx0 = blockIdx.x * TW - 4 * RL, gy0 = blockIdx.y * TH - 4 * RL;   // image coords of tile-local (0,0)
    const uint8_t* src = in.base + (size_t)f * in.frame;
    for (int i = threadIdx.x; i < SH * SW; i += 256) {
        const int y = i / SW, x = i - y * SW;
        const int gx = gx0 + x, gy = gy0 + y;
        A[y * SP + x] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? src[(size_t)gy * in.pitch + gx] : (uint8_t)255;
    }
    __syncthreads();
    const int S = RL + RR;
    morph_pass<false>(A, B, h7, h9, h10, 0, 0, SW, SH, gx0, gy0, W, H, 0);                               // erode  -> dilate next
    morph_pass<true>(B, A, h7, h9, h10, RL, RL, SW - S, SH - S, gx0, gy0, W, H, 0);                      // dilate -> dilate next
    morph_pass<true>(A, B, h7, h9, h10, 2 * RL, 2 * RL, SW - 2 * S, SH - 2 * S, gx0, gy0, W, H, 255);    // dilate -> erode next
    morph_pass<false>(B, A, h7, h9, h10, 3 * RL, 3 * RL, SW - 3 * S, SH - 3 * S, gx0, gy0, W, H, 0);     // erode  -> final
    uint8_t* dst = out.base + (size_t)f * out.frame;
    for (int i = threadIdx.x; i < TW * TH; i += 256) {
        const int y = i / TW, x = i - y * TW;
        const int gx = blockIdx.x * TW + x, gy = blockIdx.y * TH + y;
        if (gx < W && gy < H) dst[(size_t)gy * out.pitch + gx] = A[(y + 4 * RL) * SP + x + 4 * RL];
    }
}

void launch_morph_open_close(Plane8 in, Plane8W out, uint8_t*, uint8_t*, int W, int H, int n, hipStream_t stream)
{
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, n);
    hipLaunchKernelGGL(k_morph_open_close, grid, dim3(256), 0, stream, in, out, W, H);
}

}  // namespace rtdm
