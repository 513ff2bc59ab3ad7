/*
 * rectify_oracle.c -- CPU oracle for the step right in front of the matcher:
 *     cvtColor(img[i], gray, CV_RGB2GRAY);                                   (estimator.cpp:29-30)
 *     remap(gray, rect, map1, map2, INTER_LINEAR);  rect = rect(roif);       (estimator.cpp:32-36)
 *     remap(img[0], img_rectified, map1, map2, INTER_LINEAR); (roif)         (estimator.cpp:38-39)
 * and for the one-off map construction at start-up:
 *     initUndistortRectifyMap(M, D, R, P, size, CV_16SC2, map1, map2);       (main.cpp:95-96)
 *
 * TEST INFRASTRUCTURE, PARITY UNPINNED -- see rtdm_oracle.h.  The reference only calls OpenCV here; what is
 * restated is the published behaviour of OpenCV 3.x imgproc:
 *   RGB2GRAY, 8 bit   : (c0*4899 + c1*9617 + c2*1868 + (1<<13)) >> 14          (c0 is the first channel in memory:
 *                       CV_RGB2GRAY gives it the R weight)
 *   remap, CV_16SC2 + CV_16UC1 maps, INTER_LINEAR, BORDER_CONSTANT(0), 8 bit:
 *                       map1 = integer source coordinates (sx, sy), map2 = fy*32 + fx with fx, fy in 1/32 pixel;
 *                       weights = the 2x2 bilinear table in 15-bit fixed point, which for 5-bit fractions is exactly
 *                       (32-fx)(32-fy)*32, fx(32-fy)*32, (32-fx)fy*32, fx*fy*32 (sum 32768, no fix-up needed);
 *                       dst = (sum w*s + (1<<14)) >> 15; samples outside the source count as 0.
 *   initUndistortRectifyMap: per pixel the ray iR*[j i 1]^T accumulated along the row (x += ir[0] ...), the
 *                       14-coefficient distortion model (k1 k2 p1 p2 k3 k4 k5 k6 s1..s4; tilt taken as identity),
 *                       u = fx*xd + cx; fixed point: iu = round_half_even(u*32), map1 = iu >> 5, map2 = (iv&31)*32 + (iu&31).
 */
#include "rtdm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int gray_of(const uint8_t* p) { return (p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + (1 << 13)) >> 14; }

void orc_rgb2gray(const uint8_t* rgb, size_t sstep, int W, int H, uint8_t* gray, size_t dstep)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) gray[(size_t)y * dstep + x] = (uint8_t)gray_of(rgb + (size_t)y * sstep + 3 * (size_t)x);
}

/* src: sH x sW x cn (cn = 1 or 3); maps and dst: dH x dW. */
void orc_remap_bilinear(const uint8_t* src, size_t sstep, int sW, int sH, int cn, const int16_t* map1 /* dH x dW x 2 */,
                        const uint16_t* map2 /* dH x dW */, int dW, int dH, uint8_t* dst, size_t dstep)
{
    for (int y = 0; y < dH; ++y)
        for (int x = 0; x < dW; ++x) {
            const int sx = map1[((size_t)y * dW + x) * 2], sy = map1[((size_t)y * dW + x) * 2 + 1];
            const int f = map2[(size_t)y * dW + x] & 1023, fx = f & 31, fy = f >> 5;
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            for (int c = 0; c < cn; ++c) {
                int s[4];
                for (int k = 0; k < 4; ++k) {
                    const int xx = sx + (k & 1), yy = sy + (k >> 1);
                    s[k] = (xx >= 0 && xx < sW && yy >= 0 && yy < sH) ? src[(size_t)yy * sstep + (size_t)xx * cn + c] : 0;
                }
                dst[(size_t)y * dstep + (size_t)x * cn + c] =
                    (uint8_t)((w00 * s[0] + w01 * s[1] + w10 * s[2] + w11 * s[3] + (1 << 14)) >> 15);
            }
        }
}

/* The two calls the reference makes per camera, then the crop: gray(rgb) -> remap -> roi.  out: roi_h x roi_w. */
int orc_rectify_gray(const uint8_t* rgb, size_t sstep, int W, int H, const int16_t* map1, const uint16_t* map2,
                     const int roi[4], uint8_t* out, size_t ostep)
{
    if (!rgb || !map1 || !map2 || !roi || !out || W <= 0 || H <= 0) return ORC_ERR_BAD_SIZE;
    if (roi[0] < 0 || roi[1] < 0 || roi[2] <= 0 || roi[3] <= 0 || roi[0] + roi[2] > W || roi[1] + roi[3] > H) return ORC_ERR_BAD_SIZE;
    uint8_t* gray = (uint8_t*)malloc((size_t)W * H);
    uint8_t* rect = (uint8_t*)malloc((size_t)W * H);
    if (!gray || !rect) { free(gray); free(rect); return ORC_ERR_BAD_SIZE; }
    orc_rgb2gray(rgb, sstep, W, H, gray, (size_t)W);
    orc_remap_bilinear(gray, (size_t)W, W, H, 1, map1, map2, W, H, rect, (size_t)W);
    for (int y = 0; y < roi[3]; ++y) memcpy(out + (size_t)y * ostep, rect + (size_t)(roi[1] + y) * W + roi[0], (size_t)roi[2]);
    free(gray); free(rect);
    return ORC_OK;
}

/* remap of the colour frame itself + crop (estimator.cpp:38-39).  out: roi_h x roi_w x 3. */
int orc_rectify_rgb(const uint8_t* rgb, size_t sstep, int W, int H, const int16_t* map1, const uint16_t* map2,
                    const int roi[4], uint8_t* out, size_t ostep)
{
    if (!rgb || !map1 || !map2 || !roi || !out || W <= 0 || H <= 0) return ORC_ERR_BAD_SIZE;
    if (roi[0] < 0 || roi[1] < 0 || roi[2] <= 0 || roi[3] <= 0 || roi[0] + roi[2] > W || roi[1] + roi[3] > H) return ORC_ERR_BAD_SIZE;
    uint8_t* rect = (uint8_t*)malloc((size_t)W * H * 3);
    if (!rect) return ORC_ERR_BAD_SIZE;
    orc_remap_bilinear(rgb, sstep, W, H, 3, map1, map2, W, H, rect, (size_t)W * 3);
    for (int y = 0; y < roi[3]; ++y)
        memcpy(out + (size_t)y * ostep, rect + ((size_t)(roi[1] + y) * W + roi[0]) * 3, (size_t)roi[2] * 3);
    free(rect);
    return ORC_OK;
}

static int inv3x3(const double a[9], double o[9])
{
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
    if (det == 0.0) return -1;
    const double id = 1.0 / det;
    o[0] = c0 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c1 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c2 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return 0;
}

/* M: 3x3 camera matrix, D: 14 distortion coefficients (unused tail = 0), R: 3x3 rectification, P: 3x4 (or the 3x3
 * new camera matrix padded) -- its left 3x3 block is used.  map1: H x W x 2 int16, map2: H x W uint16. */
int orc_init_undistort_rectify_map(const double M[9], const double D[14], const double R[9], const double P[12],
                                   int W, int H, int16_t* map1, uint16_t* map2)
{
    if (!M || !D || !R || !P || !map1 || !map2 || W <= 0 || H <= 0) return ORC_ERR_BAD_SIZE;
    double Ar[9], ArR[9], ir[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ar[3 * r + c] = P[4 * r + c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) ArR[3 * r + c] = Ar[3 * r] * R[c] + Ar[3 * r + 1] * R[3 + c] + Ar[3 * r + 2] * R[6 + c];
    if (inv3x3(ArR, ir)) return ORC_ERR_BAD_PARAM;
    const double fx = M[0], fy = M[4], u0 = M[2], v0 = M[5];
    const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4], k4 = D[5], k5 = D[6], k6 = D[7];
    const double s1 = D[8], s2 = D[9], s3 = D[10], s4 = D[11];
    for (int i = 0; i < H; ++i) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < W; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2;
            const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2;
            const double u = fx * xd + u0, v = fy * yd + v0;
            double su = nearbyint(u * 32.0), sv = nearbyint(v * 32.0);      /* round half to even */
            su = fmin(fmax(su, -2147483648.0), 2147483647.0);
            sv = fmin(fmax(sv, -2147483648.0), 2147483647.0);
            const int iu = (int)su, iv = (int)sv;
            map1[((size_t)i * W + j) * 2] = (int16_t)(iu >> 5);
            map1[((size_t)i * W + j) * 2 + 1] = (int16_t)(iv >> 5);
            map2[(size_t)i * W + j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
    return ORC_OK;
}
