/*
 * sgm_oracle.c -- CPU restatement of what SWSemiGlobalMatcher::compute delegates to
 * (/root/reference/stereo-matcher/sgbm-sw.cpp:32-37 -> cv::StereoSGBM::compute), rows S / f4 of SURVEY.md section 8.
 *
 * TEST INFRASTRUCTURE, PARITY UNPINNED -- see rtdm_oracle.h (hazard H7).  OpenCV is absent from the image and the
 * reference holds no fixtures; what follows restates the PUBLISHED algorithm of OpenCV 3.x calib3d/stereosgbm.cpp
 * (calcPixelCostBT, computeDisparitySGBM, StereoSGBMImpl::compute) from memory, in this project's own formulation
 * (whole cost volumes, one pass per direction), anchored on the reference's call site:
 *   sgbm-sw.cpp:15      StereoSGBM::create(0, nd, blockSize): preFilterCap 0, mode MODE_SGBM
 *   sgbm-sw.cpp:16-24   P1 = 8*3*5*5 = 600, P2 = 32*3*5*5 = 2400, minDisparity, numDisparities, uniquenessRatio,
 *                       speckleWindowSize, speckleRange, disp12MaxDiff from the constructor
 *
 * paths = 5 is MODE_SGBM (what the reference creates): the directions left->right, right->left and the three that
 * come down from the row above.  paths = 8 (BASELINE config 5, "8-path") is MODE_HH: all eight neighbours.  Rules
 * restated, each with the place it has in the library:
 *   R1 calcPixelCostBT   x-Sobel with the rows above / below REPLICATED at the frame edge, clipped to +-ftzero and
 *                        offset by ftzero, ftzero = max(preFilterCap, 15) | 1 = 15; columns 0 and W-1 of BOTH the
 *                        gradient row and the raw-intensity row are overwritten with ftzero before anything reads them;
 *                        Birchfield-Tomasi dissimilarity with integer half-way points ((a + b) / 2), no half-way point
 *                        beyond columns 0 / W-1; cost = BT(gradient) + (BT(raw) >> 2).
 *   R2 domain            only columns x in [minX1, maxX1) = [max(minD+D,0), W+min(minD,0)) carry costs (width1 wide);
 *                        everything else ends up INVALID = (minD-1)*16.
 *   R3 block cost        sum of the pixel cost over SADWindowSize^2 with the first / last column of the DOMAIN and the
 *                        first / last row of the frame replicated (the hsum/pixAdd/pixSub clamps); SADWindowSize =
 *                        blockSize (the constructor argument).  The library adds P2 to every C and subtracts
 *                        min_k + P2: the same numbers.
 *   R4 path cost         L_r(p,d) = C(p,d) + min(L_r(q,d), L_r(q,d-1)+P1, L_r(q,d+1)+P1, min_k L_r(q,k)+P2) - min_k L_r(q,k),
 *                        q = p - r; d-1 / d+1 outside [0,D) never win (MAX_COST sentinels); where q lies outside the
 *                        domain the library's zeroed border buffers make L_r(p,d) = C(p,d).
 *   R5 sum               S accumulates with saturate_cast<short>: since every L_r >= 0, S = min(sum_r L_r, 32767) however
 *                        the additions are grouped.
 *   R6 winner            scanning d upwards with a strict "<": the FIRST minimum; uniqueness: some d with |d - best| > 1
 *                        and S[d]*(100-u) < minS*100 rejects; u = uniquenessRatio >= 0 ? it : 10.
 *   R7 right view        columns are visited from RIGHT to LEFT; a winner votes for x2 = x - (best + minD) and replaces
 *                        the vote there only if its minS is strictly smaller (ties: the larger x stays).
 *   R8 sub-pixel         0 < d < D-1: d*16 + ((S[d-1]-S[d+1])*16 + den2) / (den2*2), den2 = max(S[d-1]+S[d+1]-2S[d], 1),
 *                        C division; else d*16; + minD*16.
 *   R9 left-right check  ALWAYS on, with disp12MaxDiff > 0 ? it : 1: a pixel is invalidated iff both x - (d>>4) and
 *                        x - ((d+15)>>4) lie in the row, hold a vote ">= minD" and differ from it by more than the limit.
 *                        The vote array is initialised with the SCALED invalid value (minD-1)*16, which is ">= minD"
 *                        for minD >= 2 (a quirk that is restated, not repaired).
 *   R10 median           medianBlur(disp, disp, 3): 3x3 median of the int16 map, coordinates clamped (BORDER_REPLICATE).
 *   R11 speckle          if speckleWindowSize > 0: filterSpeckles(disp, (minD-1)*16, window, 16 * speckleRange).
 *   R12 parameters       P1 = P1 > 0 ? P1 : 2, P2 = max(P2 > 0 ? P2 : 5, P1+1).
 * Knowing deviations: (a) the library's CostType is short and wraps above 32767; a frame whose largest block cost + P2
 * exceeds that is refused (ORC_ERR_COST_OVERFLOW) instead of restating the wrap-around -- impossible for windows <= 17
 * at P2 = 2400 (93 * 17^2 + 2400 < 32767), data dependent above; (b) [round 3: closed] even block sizes run as the library runs them, with the odd window blockSize/2*2+1;
 * (c) for minD != 0 the library precomputes the right image's Birchfield-Tomasi bounds over [minX2, maxX2) only,
 * which does not cover every column the cost loop reads -- the bounds are computed for every column here.
 * Tolerance of the HIP kernels against this file: 0 (integer algorithm).
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
    uint8_t* il = (uint8_t*)malloc((size_t)W * H);      /* R1: raw rows with columns 0 and W-1 overwritten by ftzero */
    uint8_t* ir = (uint8_t*)malloc((size_t)W * H);
    sgm_gradient(L, lstep, W, H, gl);
    sgm_gradient(R, rstep, W, H, gr);
    for (int y = 0; y < H; ++y) {
        memcpy(il + (size_t)y * W, L + (size_t)y * lstep, (size_t)W);
        memcpy(ir + (size_t)y * W, R + (size_t)y * rstep, (size_t)W);
        il[(size_t)y * W] = il[(size_t)y * W + W - 1] = SGM_FTZERO;
        ir[(size_t)y * W] = ir[(size_t)y * W + W - 1] = SGM_FTZERO;
    }
    for (int y = 0; y < H; ++y)
        for (int x = x0; x < x1; ++x)
            for (int d = 0; d < D; ++d) {
                const int xr = x - (d + minD);      /* always inside the image on this domain */
                const int c = bt(gl + (size_t)y * W, x, gr + (size_t)y * W, xr, W) +
                              (bt(il + (size_t)y * W, x, ir + (size_t)y * W, xr, W) >> 2);
                cost[((size_t)y * W1 + (x - x0)) * D + d] = (uint16_t)c;
            }
    free(ir); free(il); free(gr); free(gl);
}

/* Returns the largest block cost (a caller that needs 16-bit path costs checks it against 32767 - P2: deviation (a)). */
uint32_t orc_sgm_block_cost(const uint16_t* pix, int W, int H, int D, int blockSize, uint16_t* C)
{
    uint32_t cmax = 0;
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
                if (s > cmax) cmax = s;
            }
    free(tmp);
    return cmax;
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
                    s[d] = (uint16_t)imin((int)s[d] + l, 32767);      /* R5: saturate_cast<short>, all terms >= 0 */
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
        if (paths == 5 && dirs[k][1] < 0) continue;        /* MODE_SGBM: no path that runs upwards (R4) */
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
    const int uniq = uniquenessRatio >= 0 ? uniquenessRatio : 10;          /* R6 */
    const int maxDiff = disp12MaxDiff > 0 ? disp12MaxDiff : 1;            /* R9: never off */
    int* d2 = (int*)malloc(sizeof(int) * 2 * (size_t)W);
    int* c2 = d2 + W;
    for (int y = 0; y < H; ++y) {
        int16_t* out = disp + (size_t)y * dstep;
        for (int x = 0; x < W; ++x) { out[x] = (int16_t)INVALID; d2[x] = INVALID; c2[x] = SHRT_MAX; }   /* R9: the scaled value */
        for (int x = maxX1 - 1; x >= minX1; --x) {                                                      /* R7: right to left */
            const uint16_t* s = S + ((size_t)y * W1 + (x - minX1)) * D;
            int mins = SHRT_MAX, bd = -1;
            for (int d = 0; d < D; ++d) if (s[d] < mins) { mins = s[d]; bd = d; }
            if (bd < 0) continue;            /* every S saturated at 32767 (a large P2, 8 paths): the library's bestDisp stays -1, it
                                              * casts no vote and what it writes, (-1 + minD) * 16, IS the invalid value */
            int d;
            for (d = 0; d < D; ++d)
                if ((int)s[d] * (100 - uniq) < mins * 100 && iabs(bd - d) > 1) break;
            if (d < D) continue;
            {
                const int x2 = x - bd - minD;                     /* inside the row on this domain */
                if (x2 >= 0 && x2 < W && c2[x2] > mins) { c2[x2] = mins; d2[x2] = bd + minD; }
            }
            int d16;
            if (bd > 0 && bd < D - 1) {
                const int den = imax((int)s[bd - 1] + s[bd + 1] - 2 * s[bd], 1);
                d16 = bd * 16 + (((int)s[bd - 1] - s[bd + 1]) * 16 + den) / (den * 2);
            } else d16 = bd * 16;
            out[x] = (int16_t)(d16 + minD * 16);
        }
        for (int x = minX1; x < maxX1; ++x) {
            const int d1 = out[x];
            if (d1 == INVALID) continue;
            const int da = d1 >> 4, db = (d1 + 15) >> 4;
            const int xa = x - da, xb = x - db;
            if (0 <= xa && xa < W && d2[xa] >= minD && iabs(d2[xa] - da) > maxDiff &&
                0 <= xb && xb < W && d2[xb] >= minD && iabs(d2[xb] - db) > maxDiff)
                out[x] = (int16_t)INVALID;
        }
    }
    free(d2);
}

/* R10: 3x3 median of an int16 image, coordinates clamped to the frame (what medianBlur's BORDER_REPLICATE does). */
void orc_median3x3_s16(const int16_t* src, size_t sstep, int16_t* dst, size_t dstep, int W, int H)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int v[9], n = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    v[n++] = src[(size_t)iclamp(y + dy, 0, H - 1) * sstep + iclamp(x + dx, 0, W - 1)];
            for (int i = 1; i < 9; ++i) {                    /* insertion sort: the definition, not a network */
                const int t = v[i];
                int j = i - 1;
                while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; --j; }
                v[j + 1] = t;
            }
            dst[(size_t)y * dstep + x] = (int16_t)v[4];
        }
}

int orc_sgm_compute(const orc_sgm_params* p, const uint8_t* L, size_t lstep, const uint8_t* R, size_t rstep,
                    int W, int H, int16_t* disp, size_t dstep_bytes)
{
    if (!p || !L || !R || !disp || W <= 0 || H <= 0) return ORC_ERR_BAD_SIZE;
    const int D = p->numDisparities, minD = p->minDisparity;
    if (D <= 0 || D % 16 != 0 || p->blockSize < 1) return ORC_ERR_BAD_PARAM;
    /* the library never checks the parity: SW2 = SH2 = SADWindowSize / 2, an even size is the next odd one */
    if (p->blockSize / 2 * 2 + 1 > 255) return ORC_ERR_BAD_PARAM;
    if (p->uniquenessRatio > 100) return ORC_ERR_BAD_PARAM;
    if (p->paths != 0 && p->paths != 5 && p->paths != 8) return ORC_ERR_BAD_PARAM;
    const int P1 = p->P1 > 0 ? p->P1 : 2;                                 /* R12 */
    const int P2 = imax(p->P2 > 0 ? p->P2 : 5, P1 + 1);
    const size_t dstep = dstep_bytes / 2;
    const int INVALID = (minD - 1) * 16;
    const int W1 = (W + imin(minD, 0)) - imax(minD + D, 0);
    if (W1 <= 0) {      /* the library returns before the median and the speckle filter; they would change nothing */
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) disp[(size_t)y * dstep + x] = (int16_t)INVALID;
        return ORC_OK;
    }
    const size_t vol = (size_t)W1 * H * D;
    uint16_t* pix = (uint16_t*)malloc(vol * 2);
    uint16_t* C = (uint16_t*)malloc(vol * 2);
    uint16_t* S = (uint16_t*)malloc(vol * 2);
    int16_t* raw = (int16_t*)malloc(sizeof(int16_t) * (size_t)W * H);
    orc_sgm_pixel_cost(L, lstep, R, rstep, W, H, minD, D, pix);
    /* deviation (a): a path cost is at most block cost + P2; where that passes 32767 the library's short arithmetic wraps,
     * which is not restated -- such a FRAME is refused (round 3; until then every window > 17 was refused at P2 = 2400,
     * although a block cost near its bound 93 * window^2 needs every pixel of the window at the maximum pixel cost) */
    if ((long)orc_sgm_block_cost(pix, W1, H, D, p->blockSize, C) + P2 > 32767) {
        free(raw); free(S); free(C); free(pix);
        return ORC_ERR_COST_OVERFLOW;
    }
    orc_sgm_aggregate_paths(C, W1, H, D, P1, P2, p->paths == 5 ? 5 : 8, S);
    orc_sgm_select(S, W, H, D, minD, p->uniquenessRatio, p->disp12MaxDiff, raw, (size_t)W);
    orc_median3x3_s16(raw, (size_t)W, disp, dstep, W, H);                 /* R10 */
    if (p->speckleWindowSize > 0)                                         /* R11 */
        orc_filter_speckles(disp, dstep, W, H, INVALID, p->speckleWindowSize, 16 * p->speckleRange);
    free(raw); free(S); free(C); free(pix);
    return ORC_OK;
}
