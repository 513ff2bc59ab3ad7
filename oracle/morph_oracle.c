/*
 * morph_oracle.c -- CPU restatement of SWMorphologicalFilter::run
 * (/root/reference/filter/mf-sw.cpp:19-28): erode, dilate, dilate, erode, each with
 * getStructuringElement(MORPH_ELLIPSE, Size(MORPH_FILTER_DX, MORPH_FILTER_DY)) = 10x10
 * (/root/reference/include/filter/mf-sw.h:11-12).
 *
 * TEST INFRASTRUCTURE, PARITY UNPINNED -- see rtdm_oracle.h.  Semantics restated from the
 * published OpenCV imgproc behaviour: the element row i spans columns
 * [max(c-dx,0), min(c+dx+1,kw)) with dx = round(c*sqrt((r*r-dy*dy)/(r*r))), r = kh/2,
 * c = kw/2, dy = i-r; anchor = (kw/2, kh/2); the element is NOT reflected for dilation;
 * out-of-image samples never win (constant border of +inf for erode, -inf for dilate).
 */
#include "rtdm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_ellipse_element(int kw, int kh, uint8_t* elem)
{
    int r = kh / 2, c = kw / 2;
    double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < kh; ++i) {
        int j1 = 0, j2 = 0;
        int dy = i - r;
        if (abs(dy) <= r) {
            int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
            j1 = c - dx > 0 ? c - dx : 0;
            j2 = c + dx + 1 < kw ? c + dx + 1 : kw;
        }
        for (int j = 0; j < kw; ++j) elem[i * kw + j] = (uint8_t)(j >= j1 && j < j2);
    }
}

static void morph(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int W, int H,
                  int kw, int kh, int is_dilate)
{
    uint8_t* elem = (uint8_t*)malloc((size_t)kw * kh);
    uint8_t* tmp = (uint8_t*)malloc((size_t)W * H); /* allows src == dst */
    orc_ellipse_element(kw, kh, elem);
    const int ax = kw / 2, ay = kh / 2;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        int acc = is_dilate ? 0 : 255;
        for (int i = 0; i < kh; ++i) {
            int yy = y + i - ay;
            if (yy < 0 || yy >= H) continue;
            for (int j = 0; j < kw; ++j) {
                if (!elem[i * kw + j]) continue;
                int xx = x + j - ax;
                if (xx < 0 || xx >= W) continue;
                int v = src[(size_t)yy * sstep + xx];
                if (is_dilate ? v > acc : v < acc) acc = v;
            }
        }
        tmp[(size_t)y * W + x] = (uint8_t)acc;
    }
    for (int y = 0; y < H; ++y) memcpy(dst + (size_t)y * dstep, tmp + (size_t)y * W, (size_t)W);
    free(tmp); free(elem);
}

void orc_erode(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int W, int H, int kw, int kh)
{ morph(src, sstep, dst, dstep, W, H, kw, kh, 0); }

void orc_dilate(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int W, int H, int kw, int kh)
{ morph(src, sstep, dst, dstep, W, H, kw, kh, 1); }

void orc_morph_open_close(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int W, int H)
{
    orc_erode(src, sstep, dst, dstep, W, H, 10, 10);
    orc_dilate(dst, dstep, dst, dstep, W, H, 10, 10);
    orc_dilate(dst, dstep, dst, dstep, W, H, 10, 10);
    orc_erode(dst, dstep, dst, dstep, W, H, 10, 10);
}
