/*
 * sgm_oracle.c -- CPU oracle for BASELINE config 5: "SWSemiGlobalMatcher-equivalent (SGM 8-path cost
 * aggregation), d=128".
 *
 * TEST INFRASTRUCTURE, PARITY UNPINNED -- see rtdm_oracle.h.  The reference's SWSemiGlobalMatcher
 * (/root/reference/stereo-matcher/sgbm-sw.cpp:12-37) is a wrapper over cv::StereoSGBM::create(0, nd,
 * blockSize) with P1 = 8*3*5*5 = 600 (:17), P2 = 32*3*5*5 = 2400 (:18), default mode (OpenCV MODE_SGBM,
 * 5 directions) and is never instantiated by main.cpp.  BASELINE config 5 asks for the 8-path variant,
 * so this file DEFINES the algorithm the HIP kernels are checked against ("SGM-8"), built from the
 * published pieces of cv::StereoSGBM, all in integer arithmetic:
 *
 *   pixel cost   Birchfield-Tomasi on the x-Sobel image clipped to +-15 (preFilterCap 0 -> ftzero 15)
 *                plus Birchfield-Tomasi on the raw intensities >> 2          (calcPixelCostBT)
 *   block cost   C(p,d) = sum of the pixel cost over blockSize x blockSize, coordinates clamped to
 *                the image (edge replication)
 *   path cost    L_r(p,d) = C(p,d) + min(L_r(q,d), L_r(q,d-1)+P1, L_r(q,d+1)+P1, min_k L_r(q,k)+P2)
 *                           - min_k L_r(q,k),  q = p - r;  L_r = C where q is outside the image;
 *                8 directions r; S = sum_r L_r
 *   selection    d* = first minimum of S; uniqueness: any |d-d*| > 1 with S[d]*(100-u) < S[d*]*100
 *                rejects; quadratic sub-pixel d*16 + ((S[d-1]-S[d+1])*16 + den)/(2*den),
 *                den = max(S[d-1]+S[d+1]-2S[d],1); left-right check on the integer winners
 *                (disp12MaxDiff); speckle filter with 16*speckleRange.
 * Domain: only columns x in [minD + D, W + min(minD,0)) can see every disparity, so -- like cv::StereoSGBM --
 * the cost volume, the block sums' edge replication and the paths live on that column range only
 * (W1 = its width); every other column is INVALID = (minD-1)*16.
 * Tolerance against this oracle: 0 (integer algorithm).  Against a real cv::StereoSGBM the result is
 * expected to differ (different direction set and border handling) -- that comparison is unpinned.
 */
#include "rtdm_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int iabs(int v) { return v < 0 ? -v : v; }

#define SGM_FTZERO 15

/* gradient image: clip(sobel_x with vertical edge replication, +-15) + 15; borders = 15 */
static void sgm_gradient(const uint8_t* img, size_t step, int W, int H, uint8_t* g)
{
    for (int y = 0; y < H; ++y) {
        const uint8_t* r1 = img + (size_t)y * step;
        const uint8_t* r0 = img + (size_t)(y > 0 ? y - 1 : y) * step;
        const uint8_t* r2 = img + (size_t)(y < H - 1 ? y + 1 : y) * step;
        uint8_t* o = g + (size_t)y * W;
        o[0] = SGM_FTZERO;
        if (W > 1) o[W - 1] = SGM_FTZERO;
        for (int x = 1; x < W - 1; ++x) {
            int v = (r1[x + 1] - r1[x - 1]) * 2 + (r0[x + 1] - r0[x - 1]) + (r2[x + 1] - r2[x - 1]);
            o[x] = (uint8_t)(iclamp(v, -SGM_FTZERO, SGM_FTZERO) + SGM_FTZERO);
        }
    }
}

/* Birchfield-Tomasi dissimilarity between a[xa] and b[xb] on rows of width W (integer halves) */
static inline int bt(const uint8_t* a, int xa, const uint8_t* b, int xb, int W)
{
    const int u = a[xa];
    const int ul = xa > 0 ? (u + a[xa - 1]) / 2 : u, ur = xa < W - 1 ? (u + a[xa + 1]) / 2 : u;
    const int u0 = imin(imin(ul, ur), u), u1 = imax(imax(ul, ur), u);
    const int v = b[xb];
    const int vl = xb > 0 ? (v + b[xb - 1]) / 2 : v, vr = xb < W - 1 ? (v + b[xb + 1]) / 2 : v;
    const int v0 = imin(imin(vl, vr), v), v1 = imax(imax(vl, vr), v);
    const int c0 = imax(0, imax(u - v1, v0 - u));
    const int c1 = imax(0, imax(v - u1, u0 - v));
    return imin(c0, c1);
}

void orc_sgm_pixel_cost(const uint8_t* L, size_t lstep, const uint8_t* R, size_t rstep, int W, int H,
                        int minD, int D, uint16_t* cost /* H * W1 * D, W1 = columns [minD+D, W+min(minD,0)) */)
{
    const int x0 = imax(minD + D, 0), x1 = W + imin(minD, 0), W1 = x1 - x0;
    uint8_t* gl = (uint8_t*)malloc((size_t)W * H);
    uint8_t* gr = (uint8_t*)malloc((size_t)W * H);
    sgm_gradient(L, lstep, W, H, gl);
    sgm_gradient(R, rstep, W, H, gr);
    for (int y = 0; y < H; ++y)
        for (int x = x0; x < x1; ++x)
            for (int d = 0; d < D; ++d) {
                const int xr = x - (d + minD);      /* always inside the image on this domain */
                const int c = bt(gl + (size_t)y * W, x, gr + (size_t)y * W, xr, W) +
                              (bt(L + (size_t)y * lstep, x, R + (size_t)y * rstep, xr, W) >> 2);
                cost[((size_t)y * W1 + (x - x0)) * D + d] = (uint16_t)c;
            }
    free(gr); free(gl);
}

void orc_sgm_block_cost(const uint16_t* pix, int W, int H, int D, int blockSize, uint16_t* C)
{
    const int r = blockSize / 2;
    /* separable box sum with clamped coordinates */
    uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)W * H * D);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int d = 0; d < D; ++d) {
                uint32_t s = 0;
                for (int k = -r; k <= r; ++k) s += pix[((size_t)y * W + iclamp(x + k, 0, W - 1)) * D + d];
                tmp[((size_t)y * W + x) * D + d] = s;
            }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int d = 0; d < D; ++d) {
                uint32_t s = 0;
                for (int k = -r; k <= r; ++k) s += tmp[((size_t)iclamp(y + k, 0, H - 1) * W + x) * D + d];
                C[((size_t)y * W + x) * D + d] = (uint16_t)s;
            }
    free(tmp);
}

/* S += L_r for one direction (dx, dy) */
static void sgm_path(const uint16_t* C, int W, int H, int D, int dx, int dy, int P1, int P2, uint16_t* S)
{
    int* prev = (int*)malloc(sizeof(int) * (size_t)(D + 2));
    int* cur = (int*)malloc(sizeof(int) * (size_t)(D + 2));
    for (int sy = 0; sy < H; ++sy)
        for (int sx = 0; sx < W; ++sx) {
            const int px = sx - dx, py = sy - dy;
            if (px >= 0 && px < W && py >= 0 && py < H) continue;   /* not the start of a path */
            int x = sx, y = sy, first = 1, minprev = 0;
            while (x >= 0 && x < W && y >= 0 && y < H) {
                const uint16_t* c = C + ((size_t)y * W + x) * D;
                uint16_t* s = S + ((size_t)y * W + x) * D;
                int mincur = INT_MAX;
                for (int d = 0; d < D; ++d) {
                    int l;
                    if (first) l = c[d];
                    else {
                        int best = prev[d + 1];
                        if (d > 0) best = imin(best, prev[d] + P1);
                        if (d < D - 1) best = imin(best, prev[d + 2] + P1);
                        best = imin(best, minprev + P2);
                        l = c[d] + best - minprev;
                    }
                    cur[d + 1] = l;
                    mincur = imin(mincur, l);
                    s[d] = (uint16_t)(s[d] + l);
                }
                int* t = prev; prev = cur; cur = t;
                minprev = mincur; first = 0;
                x += dx; y += dy;
            }
        }
    free(cur); free(prev);
}

void orc_sgm_aggregate_paths(const uint16_t* C, int W, int H, int D, int P1, int P2, int paths, uint16_t* S)
{
    static const int dirs[8][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {-1, 1}, {1, -1}, {-1, -1}};
    memset(S, 0, sizeof(uint16_t) * (size_t)W * H * D);
    for (int k = 0; k < 8; ++k) {
        if (paths == 5 && dirs[k][1] < 0) continue;        /* MODE_SGBM: no path that runs upwards */
        sgm_path(C, W, H, D, dirs[k][0], dirs[k][1], P1, P2, S);
    }
}

void orc_sgm_aggregate(const uint16_t* C, int W, int H, int D, int P1, int P2, uint16_t* S)
{ orc_sgm_aggregate_paths(C, W, H, D, P1, P2, 8, S); }

void orc_sgm_select(const uint16_t* S, int W, int H, int D, int minD, int uniquenessRatio, int disp12MaxDiff,
                    int16_t* disp, size_t dstep)
{
    const int INVALID = (minD - 1) * 16;
    const int minX1 = imax(minD + D, 0), maxX1 = W + imin(minD, 0), W1 = maxX1 - minX1;
    int* d2 = (int*)malloc(sizeof(int) * 2 * (size_t)W);
    int* c2 = d2 + W;
    for (int y = 0; y < H; ++y) {
        int16_t* out = disp + (size_t)y * dstep;
        for (int x = 0; x < W; ++x) { out[x] = (int16_t)INVALID; d2[x] = minD - 1; c2[x] = INT_MAX; }
        for (int x = minX1; x < maxX1; ++x) {
            const uint16_t* s = S + ((size_t)y * W1 + (x - minX1)) * D;
            int mins = INT_MAX, bd = -1;
            for (int d = 0; d < D; ++d) if (s[d] < mins) { mins = s[d]; bd = d; }
            int d;
            for (d = 0; d < D; ++d)
                if (iabs(d - bd) > 1 && (int)s[d] * (100 - uniquenessRatio) < mins * 100) break;
            if (d < D) continue;
            {   /* vote for the matching right-image column with the integer winner */
                const int x2 = x - (bd + minD);
                if (x2 >= 0 && x2 < W && c2[x2] > mins) { c2[x2] = mins; d2[x2] = bd + minD; }
            }
            int d16;
            if (bd > 0 && bd < D - 1) {
                const int den = imax((int)s[bd - 1] + s[bd + 1] - 2 * s[bd], 1);
                d16 = bd * 16 + (((int)s[bd - 1] - s[bd + 1]) * 16 + den) / (den * 2);
            } else d16 = bd * 16;
            out[x] = (int16_t)(d16 + minD * 16);
        }
        if (disp12MaxDiff >= 0)
            for (int x = minX1; x < maxX1; ++x) {
                const int d1 = out[x];
                if (d1 == INVALID) continue;
                const int da = d1 >> 4, db = (d1 + 15) >> 4;
                const int xa = x - da, xb = x - db;
                if (0 <= xa && xa < W && d2[xa] >= minD && iabs(d2[xa] - da) > disp12MaxDiff &&
                    0 <= xb && xb < W && d2[xb] >= minD && iabs(d2[xb] - db) > disp12MaxDiff)
                    out[x] = (int16_t)INVALID;
            }
    }
    free(d2);
}

int orc_sgm_compute(const orc_sgm_params* p, const uint8_t* L, size_t lstep, const uint8_t* R, size_t rstep,
                    int W, int H, int16_t* disp, size_t dstep_bytes)
{
    if (!p || !L || !R || !disp || W <= 0 || H <= 0) return ORC_ERR_BAD_SIZE;
    const int D = p->numDisparities, minD = p->minDisparity;
    if (D <= 0 || D % 16 != 0 || p->blockSize < 1 || (p->blockSize & 1) == 0) return ORC_ERR_BAD_PARAM;
    if (p->P1 <= 0 || p->P2 <= p->P1 || p->uniquenessRatio < 0 || p->uniquenessRatio > 100) return ORC_ERR_BAD_PARAM;
    if (p->paths != 0 && p->paths != 5 && p->paths != 8) return ORC_ERR_BAD_PARAM;
    const int W1 = (W + imin(minD, 0)) - imax(minD + D, 0);
    if (W1 <= 0) {
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) disp[(size_t)y * (dstep_bytes / 2) + x] = (int16_t)((minD - 1) * 16);
        return ORC_OK;
    }
    const size_t vol = (size_t)W1 * H * D;
    uint16_t* pix = (uint16_t*)malloc(vol * 2);
    uint16_t* C = (uint16_t*)malloc(vol * 2);
    uint16_t* S = (uint16_t*)malloc(vol * 2);
    orc_sgm_pixel_cost(L, lstep, R, rstep, W, H, minD, D, pix);
    orc_sgm_block_cost(pix, W1, H, D, p->blockSize, C);
    orc_sgm_aggregate_paths(C, W1, H, D, p->P1, p->P2, p->paths == 5 ? 5 : 8, S);
    orc_sgm_select(S, W, H, D, minD, p->uniquenessRatio, p->disp12MaxDiff, disp, dstep_bytes / 2);
    if (p->speckleWindowSize > 0 && p->speckleRange >= 0)
        orc_filter_speckles(disp, dstep_bytes / 2, W, H, (minD - 1) * 16, p->speckleWindowSize, 16 * p->speckleRange);
    free(S); free(C); free(pix);
    return ORC_OK;
}
