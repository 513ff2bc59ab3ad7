/*
 * objects_oracle.c -- CPU oracle for the object detection that produces the matcher's ROI:
 *     cvtColor(img_rectified, img_rectified, COLOR_RGB2BGR); cvtColor(img_rectified, imgHSV, COLOR_BGR2HSV);   (estimator.cpp:40,42)
 *     inRange(imgHSV, Scalar(lowH, lowS, lowV), Scalar(highH, highS, highV), filter_in);                        (estimator.cpp:43)
 *     morphFilter->run(filter_in, filter_out);                                    (estimator.cpp:45; oracle: morph_oracle.c)
 *     findContours(contInput, contours, hierarchy, CV_RETR_EXTERNAL, CV_CHAIN_APPROX_SIMPLE);                   (estimator.cpp:47)
 *     fill_bounding_rects_of_contours(...); find_relevant_matching_region(...); bm->setROI1(matching_roi);     (estimator.cpp:51-54)
 *
 * TEST INFRASTRUCTURE, PARITY UNPINNED -- see rtdm_oracle.h.  fill_bounding_rects_of_contours and
 * find_relevant_matching_region are the reference's own code (estimator.cpp:167-204) and are restated from it; the
 * OpenCV calls are restated from their published behaviour:
 *   BGR2HSV, 8 bit : v = max, diff = v - min; s = (diff * sdiv[v] + 2048) >> 12 with sdiv[v] = round(255*4096 / v);
 *                    h = (r == v ? g - b : g == v ? b - r + 2 diff : r - g + 4 diff); h = (h * hdiv[diff] + 2048) >> 12 with
 *                    hdiv[d] = round(180*4096 / (6 d)); h += 180 if negative.  (The RGB2BGR swap in front of it only
 *                    reorders the bytes; this file takes the frame with R first, as the decoder delivers it.)
 *   inRange        : 255 where low <= value <= high in all three channels, else 0.
 *   findContours(RETR_EXTERNAL) + boundingRect: one box per 8-connected foreground component that is not enclosed by
 *                    another component (i.e. whose surrounding 4-connected background reaches the frame); box =
 *                    (min x, min y, max x - min x + 1, max y - min y + 1); contours come out in REVERSE order of
 *                    discovery (discovery = raster order of each component's first pixel).  zero_border = 1 restates
 *                    OpenCV <= 3.1, where the function first clears the image's outermost rows and columns (the
 *                    reference copies filter_out before the call for that reason, estimator.cpp:46); 0 restates >= 3.2.
 */
#include "rtdm_oracle.h"

#include <stdlib.h>
#include <string.h>

static int sdiv_tab[256], hdiv_tab[256], tabs_ready;

static void init_tabs(void)
{
    if (tabs_ready) return;
    sdiv_tab[0] = hdiv_tab[0] = 0;
    for (int i = 1; i < 256; ++i) {
        /* saturate_cast<int>(double) rounds half to even; none of these quotients is a tie */
        sdiv_tab[i] = (int)((255 << 12) / (1. * i) + 0.5);
        hdiv_tab[i] = (int)((180 << 12) / (6. * i) + 0.5);
    }
    tabs_ready = 1;
}

void orc_rgb2hsv(const uint8_t* rgb, size_t sstep, int W, int H, uint8_t* hsv, size_t dstep)
{
    init_tabs();
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const uint8_t* p = rgb + (size_t)y * sstep + 3 * (size_t)x;
            const int r = p[0], g = p[1], b = p[2];
            int v = b, vmin = b;
            if (g > v) v = g;
            if (r > v) v = r;
            if (g < vmin) vmin = g;
            if (r < vmin) vmin = r;
            const int diff = v - vmin;
            const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
            const int s = (diff * sdiv_tab[v] + (1 << 11)) >> 12;
            int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
            h = (h * hdiv_tab[diff] + (1 << 11)) >> 12;
            h += h < 0 ? 180 : 0;
            uint8_t* o = hsv + (size_t)y * dstep + 3 * (size_t)x;
            o[0] = (uint8_t)(h < 0 ? 0 : h > 255 ? 255 : h); o[1] = (uint8_t)s; o[2] = (uint8_t)v;
        }
}

void orc_hsv_inrange(const uint8_t* rgb, size_t sstep, int W, int H, const int lo[3], const int hi[3], uint8_t* mask, size_t mstep)
{
    uint8_t* hsv = (uint8_t*)malloc((size_t)W * 3);
    for (int y = 0; y < H; ++y) {
        orc_rgb2hsv(rgb + (size_t)y * sstep, 0, W, 1, hsv, 0);
        for (int x = 0; x < W; ++x) {
            int ok = 1;
            for (int c = 0; c < 3; ++c) ok &= hsv[3 * x + c] >= lo[c] && hsv[3 * x + c] <= hi[c];
            mask[(size_t)y * mstep + x] = ok ? 255 : 0;
        }
    }
    free(hsv);
}

/* boxes: up to max_boxes x (x, y, w, h), in the order the reference's obj_boundings would have; returns the number of
 * boxes found (which may exceed max_boxes; only the first max_boxes are stored), or a negative error. */
int orc_external_boxes(const uint8_t* mask, size_t mstep, int W, int H, int zero_border, int min_area, int* boxes, int max_boxes)
{
    if (!mask || W <= 0 || H <= 0 || (!boxes && max_boxes > 0)) return ORC_ERR_BAD_SIZE;
    const size_t N = (size_t)W * H;
    uint8_t* fg = (uint8_t*)malloc(N);
    int* lab = (int*)malloc(N * sizeof(int));        /* component id per pixel (fg and bg ids share one counter) */
    int* stack = (int*)malloc(N * sizeof(int));
    if (!fg || !lab || !stack) { free(fg); free(lab); free(stack); return ORC_ERR_BAD_SIZE; }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int v = mask[(size_t)y * mstep + x] != 0;
            if (zero_border && (x == 0 || y == 0 || x == W - 1 || y == H - 1)) v = 0;
            fg[(size_t)y * W + x] = (uint8_t)v;
            lab[(size_t)y * W + x] = -1;
        }
    /* background components, 4-connected; outer[id] = touches the frame */
    int ncomp = 0;
    uint8_t* outer = (uint8_t*)calloc(N + 1, 1);
    int* first = (int*)malloc((N + 1) * sizeof(int));
    int* bb = (int*)malloc((N + 1) * 4 * sizeof(int));
    if (!outer || !first || !bb) { free(fg); free(lab); free(stack); free(outer); free(first); free(bb); return ORC_ERR_BAD_SIZE; }
    for (size_t p0 = 0; p0 < N; ++p0) {
        if (lab[p0] >= 0) continue;
        const int id = ncomp++, isfg = fg[p0];
        int sp = 0, x0 = W, y0 = H, x1 = -1, y1 = -1, touches = 0;
        stack[sp++] = (int)p0; lab[p0] = id; first[id] = (int)p0;
        while (sp) {
            const int p = stack[--sp], y = p / W, x = p - y * W;
            if (x < x0) x0 = x;
            if (x > x1) x1 = x;
            if (y < y0) y0 = y;
            if (y > y1) y1 = y;
            if (x == 0 || y == 0 || x == W - 1 || y == H - 1) touches = 1;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!dx && !dy) continue;
                    if (!isfg && dx && dy) continue;                   /* background: 4-connected */
                    const int xx = x + dx, yy = y + dy;
                    if (xx < 0 || yy < 0 || xx >= W || yy >= H) continue;
                    const int q = yy * W + xx;
                    if (lab[q] >= 0 || fg[q] != isfg) continue;
                    lab[q] = id; stack[sp++] = q;
                }
        }
        outer[id] = (uint8_t)(!isfg && touches);
        bb[4 * id] = x0; bb[4 * id + 1] = y0; bb[4 * id + 2] = x1 - x0 + 1; bb[4 * id + 3] = y1 - y0 + 1;
    }
    /* external foreground components, reverse discovery order (ids are handed out in raster order of the first pixel) */
    int n = 0;
    for (int id = ncomp - 1; id >= 0; --id) {
        const int p = first[id];
        if (!fg[p]) continue;
        const int y = p / W, x = p - y * W;
        const int external = x == 0 || outer[lab[p - 1]];          /* the pixel left of the first pixel is background */
        if (!external) continue;
        if (bb[4 * id + 2] * bb[4 * id + 3] < min_area) continue;     /* region.area() < minSize, estimator.cpp:173 */
        if (n < max_boxes) memcpy(boxes + 4 * n, bb + 4 * id, 4 * sizeof(int));
        ++n;
    }
    free(fg); free(lab); free(stack); free(outer); free(first); free(bb);
    return n;
}

/* find_relevant_matching_region (estimator.cpp:176-204): the union of the boxes. */
void orc_union_box(const int* boxes, int n, int roi[4])
{
    int max_x = -1000000, max_y = -1000000, min_x = 1000000, min_y = 1000000;
    for (int i = 0; i < n; ++i) {
        const int x = boxes[4 * i], y = boxes[4 * i + 1], x2 = x + boxes[4 * i + 2], y2 = y + boxes[4 * i + 3];
        if (x < min_x) min_x = x;
        if (y < min_y) min_y = y;
        if (x2 > max_x) max_x = x2;
        if (y2 > max_y) max_y = y2;
    }
    roi[0] = min_x; roi[1] = min_y; roi[2] = max_x - min_x; roi[3] = max_y - min_y;
}
