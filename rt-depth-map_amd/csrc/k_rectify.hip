// k_rectify.hip -- the step right in front of the matcher, on the device:
//     cvtColor(img, gray, CV_RGB2GRAY); remap(gray, rect, map1, map2, INTER_LINEAR); rect = rect(roif);
// (/root/reference/estimator.cpp:29-36) for both cameras in one launch, and the colour remap of the left frame
// (estimator.cpp:38-39).  Semantics: oracle/rectify_oracle.c (8-bit fixed point, bit-exact).
// gray() is a per-pixel function, so gray-then-remap equals remap of the four gray-converted samples: the kernel
// reads the RGB frame once, never materialises the gray image, and only produces the cropped region.
// HBM-bound gather: per output pixel 6 map bytes + 12 source bytes (neighbouring pixels share them through L2) + 1.
#include "rtdm_kernels.h"

namespace rtdm {

__device__ __forceinline__ int gray_of(int c0, int c1, int c2) { return (c0 * 4899 + c1 * 9617 + c2 * 1868 + (1 << 13)) >> 14; }

// 6 bytes (two RGB pixels) starting at byte offset a of frame `src`; bytes past `limit` are never touched.
__device__ __forceinline__ void load6(const uint8_t* src, size_t a, size_t limit, int* p)
{
    const size_t a0 = a & ~(size_t)3;
    if (a0 + 12 <= limit && ((size_t)src & 3) == 0) {
        const uint32_t* q = (const uint32_t*)(src + a0);
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
        const int sh = (int)(a - a0);
        const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh), hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
        p[0] = lo & 0xff; p[1] = (lo >> 8) & 0xff; p[2] = (lo >> 16) & 0xff; p[3] = lo >> 24; p[4] = hi & 0xff; p[5] = (hi >> 8) & 0xff;
    } else {
        for (int k = 0; k < 6; ++k) p[k] = (a + k < limit) ? src[a + k] : 0;
    }
}

// grid: (ceil(rw*rh / 256), n frames, 2 cameras).  maps are stored for the roi only (rh x rw).
__global__ __launch_bounds__(256) void k_rectify_gray(RectifySrc L, RectifySrc R, const int16_t* map1L, const uint16_t* map2L,
                                                      const int16_t* map1R, const uint16_t* map2R, int sW, int sH,
                                                      int rw, int rh, Plane8W outL, Plane8W outR)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rw * rh) return;
    const int y = idx / rw, x = idx - y * rw;
    const int f = blockIdx.y;
    const bool right = blockIdx.z != 0;
    const RectifySrc S = right ? R : L;
    const int16_t* m1 = right ? map1R : map1L;
    const uint16_t* m2 = right ? map2R : map2L;
    const Plane8W O = right ? outR : outL;
    const uint8_t* src = S.base + (size_t)f * S.frame;
    const int sx = m1[2 * idx], sy = m1[2 * idx + 1];
    const int fr = m2[idx] & 1023, fx = fr & 31, fy = fr >> 5;
    int g[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int yy = sy + r;
        if (yy < 0 || yy >= sH || sx + 1 < 0 || sx >= sW) continue;
        if (sx >= 0 && sx + 1 < sW) {
            int p[6];
            load6(src, (size_t)yy * S.pitch + (size_t)sx * 3, S.frame, p);
            g[2 * r] = gray_of(p[0], p[1], p[2]);
            g[2 * r + 1] = gray_of(p[3], p[4], p[5]);
        } else {
            const int xx = sx >= 0 ? sx : sx + 1;                       // the one sample that is inside
            const uint8_t* q = src + (size_t)yy * S.pitch + (size_t)xx * 3;
            g[2 * r + (sx >= 0 ? 0 : 1)] = gray_of(q[0], q[1], q[2]);
        }
    }
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    O.base[(size_t)f * O.frame + (size_t)y * O.pitch + x] =
        (uint8_t)((w00 * g[0] + w01 * g[1] + w10 * g[2] + w11 * g[3] + (1 << 14)) >> 15);
}

// colour remap + crop of one camera: out is rh x rw x 3 (pitch in bytes).
__global__ __launch_bounds__(256) void k_rectify_rgb(RectifySrc S, const int16_t* m1, const uint16_t* m2, int sW, int sH,
                                                     int rw, int rh, Plane8W O)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rw * rh) return;
    const int y = idx / rw, x = idx - y * rw;
    const int f = blockIdx.y;
    const uint8_t* src = S.base + (size_t)f * S.frame;
    const int sx = m1[2 * idx], sy = m1[2 * idx + 1];
    const int fr = m2[idx] & 1023, fx = fr & 31, fy = fr >> 5;
    int s[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int yy = sy + r;
        if (yy < 0 || yy >= sH || sx + 1 < 0 || sx >= sW) continue;
        if (sx >= 0 && sx + 1 < sW) {
            int p[6];
            load6(src, (size_t)yy * S.pitch + (size_t)sx * 3, S.frame, p);
            for (int c = 0; c < 3; ++c) { s[2 * r][c] = p[c]; s[2 * r + 1][c] = p[3 + c]; }
        } else {
            const int xx = sx >= 0 ? sx : sx + 1;
            const uint8_t* q = src + (size_t)yy * S.pitch + (size_t)xx * 3;
            for (int c = 0; c < 3; ++c) s[2 * r + (sx >= 0 ? 0 : 1)][c] = q[c];
        }
    }
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    uint8_t* o = O.base + (size_t)f * O.frame + (size_t)y * O.pitch + (size_t)x * 3;
    for (int c = 0; c < 3; ++c)
        o[c] = (uint8_t)((w00 * s[0][c] + w01 * s[1][c] + w10 * s[2][c] + w11 * s[3][c] + (1 << 14)) >> 15);
}

void launch_rectify_gray(RectifySrc L, RectifySrc R, const int16_t* map1L, const uint16_t* map2L, const int16_t* map1R,
                         const uint16_t* map2R, int sW, int sH, int rw, int rh, Plane8W outL, Plane8W outR, int n,
                         hipStream_t stream)
{
    hipLaunchKernelGGL(k_rectify_gray, dim3((rw * rh + 255) / 256, n, 2), dim3(256), 0, stream, L, R, map1L, map2L, map1R, map2R,
                       sW, sH, rw, rh, outL, outR);
}

void launch_rectify_rgb(RectifySrc S, const int16_t* map1, const uint16_t* map2, int sW, int sH, int rw, int rh, Plane8W out,
                        int n, hipStream_t stream)
{
    hipLaunchKernelGGL(k_rectify_rgb, dim3((rw * rh + 255) / 256, n), dim3(256), 0, stream, S, map1, map2, sW, sH, rw, rh, out);
}

}  // namespace rtdm
