/*
 * bm_oracle.c -- CPU restatement of what SWMatcherKonolige::compute delegates to
 * (/root/reference/stereo-matcher/bm-sw.cpp:33-38 -> cv::StereoBM::compute).
 *
 * TEST INFRASTRUCTURE, PARITY UNPINNED -- see rtdm_oracle.h for the full note.
 *
 * The formulation is this project's own (column sums over the window rows first,
 * then a horizontal sliding sum), chosen because it is easy to check against the
 * mathematical definition.  The semantics restated are those published for
 * OpenCV calib3d (3.x/4.x):
 *
 *   lofs = max(D-1+minD,0), rofs = -min(D-1+minD,0), width1 = W-rofs-D+1
 *   sample column j (j = x+dx, x = output column index, left column = lofs+x):
 *       left column  = clamp(lofs+j, 0, W-1)
 *       right column = clamp(rofs+j, 0, W-D) + d          (d = reversed index)
 *   sad[d]   = sum over the w x w window of |Lp - Rp|
 *   mind     = first d with the smallest sad (=> largest disparity on ties)
 *   texture  = sum over the window of |Lp - cap| < textureThreshold -> FILTERED
 *   unique   = any d outside [mind-1,mind+1] with sad[d] <= minsad+minsad*ratio/100 -> FILTERED
 *   subpixel = ((D-mind-1+minD)*256 + (den ? (p-n)*256/den : 0) + 15) >> 4
 *
 * Documented deviations from OpenCV (both are cases the reference never relies on):
 *   - rows outside the valid-disparity rectangle are always written FILTERED
 *     (OpenCV leaves a row stripe that misses the rectangle unwritten);
 *   - the cost plane is int32 (OpenCV 3.x's scalar path writes int through a
 *     16-bit plane; its SIMD path and 4.x use 16-bit, which is identical whenever
 *     2*cap*w*w < 32768, i.e. for every configuration the reference or BASELINE use).
 */
#include "rtdm_oracle.h"

#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* ---- A.3a prefilter (OpenCV prefilterXSobel) ------------------------------------------ */
void orc_prefilter_xsobel(const uint8_t* src, size_t sstep, int W, int H,
                          uint8_t* dst, size_t dstep, int cap)
{
    /* Rows are produced in pairs; a trailing odd row is all `cap`.  Row -1 mirrors to
     * row 1 and row H mirrors to row H-2 (reflection without repeating the edge).      */
    int npair_rows = (H >= 2) ? (H & ~1) : 0;
    for (int y = 0; y < H; ++y) {
        uint8_t* d = dst + (size_t)y * dstep;
        if (y >= npair_rows) {
            for (int x = 0; x < W; ++x) d[x] = (uint8_t)cap;
            continue;
        }
        int ya = (y > 0) ? y - 1 : 1;
        int yb = (y < H - 1) ? y + 1 : H - 2;
        const uint8_t* ra = src + (size_t)ya * sstep;
        const uint8_t* rc = src + (size_t)y * sstep;
        const uint8_t* rb = src + (size_t)yb * sstep;
        d[0] = (uint8_t)cap;
        if (W > 1) d[W - 1] = (uint8_t)cap;
        for (int x = 1; x < W - 1; ++x) {
            int g = (ra[x + 1] - ra[x - 1]) + 2 * (rc[x + 1] - rc[x - 1]) + (rb[x + 1] - rb[x - 1]);
            d[x] = (uint8_t)(iclamp(g, -cap, cap) + cap);
        }
    }
}

/* ---- A.2 valid rectangle (OpenCV getValidDisparityROI, clipped to the image) ---------- */
int orc_bm_valid_rect(const orc_bm_params* p, int W, int H, int rect[4])
{
    int r1[4] = {0, 0, W, H}, r2[4] = {0, 0, W, H};
    if (p->roi1[2] > 0 && p->roi1[3] > 0) memcpy(r1, p->roi1, sizeof r1);
    if (p->roi2[2] > 0 && p->roi2[3] > 0) memcpy(r2, p->roi2, sizeof r2);
    int sw2 = p->blockSize / 2;
    int minD = p->minDisparity, maxD = minD + p->numDisparities - 1;
    int xmin = imax(r1[0], r2[0] + maxD) + sw2;
    int xmax = imin(r1[0] + r1[2], r2[0] + r2[2] - minD) - sw2;
    int ymin = imax(r1[1], r2[1]) + sw2;
    int ymax = imin(r1[1] + r1[3], r2[1] + r2[3]) - sw2;
    /* the stripe invoker intersects with Rect(0,row0,cols,rows); ROIs that overhang the
     * image are additionally held to rows with a full halo ([sw2, H-sw2)).               */
    xmin = imax(xmin, 0); xmax = imin(xmax, W);
    ymin = imax(ymin, sw2); ymax = imin(ymax, H - sw2);
    if (xmax - xmin <= 0 || ymax - ymin <= 0) {
        rect[0] = rect[1] = rect[2] = rect[3] = 0;
        return 0;
    }
    rect[0] = xmin; rect[1] = ymin; rect[2] = xmax - xmin; rect[3] = ymax - ymin;
    return 1;
}

/* ---- A.3b SAD search ------------------------------------------------------------------ */
static int g_legacy_right_clamp = 0;   /* rtdm_oracle.h, hazard H1 */
void orc_bm_set_legacy_right_clamp(int on) { g_legacy_right_clamp = on != 0; }

typedef struct {
    int W, H, D, minD, w, r, cap, tex, uniq, lofs, rofs, width1, nj;
    const uint8_t *Lp, *Rp;
    size_t lstep, rstep;
    int *lcol, *rbase;
} search_ctx;

static void add_row(const search_ctx* c, int row, int sign, int32_t* V, int32_t* T)
{
    const uint8_t* lrow = c->Lp + (size_t)row * c->lstep;
    const uint8_t* rrow = c->Rp + (size_t)row * c->rstep;
    const int D = c->D;
    for (int jj = 0; jj < c->nj; ++jj) {
        int lv = lrow[c->lcol[jj]];
        const uint8_t* rp = rrow + c->rbase[jj];
        int32_t* v = V + (size_t)jj * D;
        if (sign > 0) for (int d = 0; d < D; ++d) v[d] += iabs(lv - rp[d]);
        else          for (int d = 0; d < D; ++d) v[d] -= iabs(lv - rp[d]);
        T[jj] += sign * iabs(lv - c->cap);
    }
}

void orc_bm_search(const orc_bm_params* p, const uint8_t* Lp, size_t lstep,
                   const uint8_t* Rp, size_t rstep, int W, int H,
                   int row0, int row1, int16_t* disp, size_t dstep,
                   int32_t* cost, size_t cstep)
{
    search_ctx c;
    c.W = W; c.H = H; c.D = p->numDisparities; c.minD = p->minDisparity;
    c.w = p->blockSize; c.r = c.w / 2; c.cap = p->preFilterCap;
    c.tex = p->textureThreshold; c.uniq = p->uniquenessRatio;
    c.lofs = imax(c.D - 1 + c.minD, 0);
    c.rofs = -imin(c.D - 1 + c.minD, 0);
    c.width1 = W - c.rofs - c.D + 1;
    c.nj = c.width1 + 2 * c.r;
    c.Lp = Lp; c.Rp = Rp; c.lstep = lstep; c.rstep = rstep;
    const int D = c.D, r = c.r;
    const int16_t FILTERED = (int16_t)((c.minD - 1) * 16);
    if (row1 <= row0) return;

    c.lcol = (int*)malloc(sizeof(int) * (size_t)c.nj);
    c.rbase = (int*)malloc(sizeof(int) * (size_t)c.nj);
    for (int jj = 0; jj < c.nj; ++jj) {
        int j = jj - r;
        c.lcol[jj] = iclamp(c.lofs + j, 0, W - 1);
        /* 4.x: the last sample base keeps base + (D-1) inside the row.  3.x (legacy switch): base up to W-rofs-1, so
         * rp[d] runs on into the following bytes (the next row when the step equals W).                              */
        c.rbase[jj] = iclamp(c.rofs + j, 0, g_legacy_right_clamp ? W - c.rofs - 1 : W - D);
    }
    int32_t* V = (int32_t*)calloc((size_t)c.nj * D, sizeof(int32_t));
    int32_t* T = (int32_t*)calloc((size_t)c.nj, sizeof(int32_t));
    int32_t* S = (int32_t*)malloc(sizeof(int32_t) * (size_t)D);

    for (int row = row0 - r; row <= row0 + r; ++row) add_row(&c, row, +1, V, T);

    for (int y = row0; y < row1; ++y) {
        if (y > row0) {
            add_row(&c, y + r, +1, V, T);
            add_row(&c, y - r - 1, -1, V, T);
        }
        int16_t* drow = disp + (size_t)y * dstep;
        int32_t* crow = cost ? cost + (size_t)y * cstep : NULL;
        for (int x = 0; x < c.lofs; ++x) drow[x] = FILTERED;
        for (int x = c.lofs + c.width1; x < W; ++x) drow[x] = FILTERED;

        int tsum = 0;
        memset(S, 0, sizeof(int32_t) * (size_t)D);
        for (int jj = 0; jj < 2 * r + 1; ++jj) {
            const int32_t* v = V + (size_t)jj * D;
            for (int d = 0; d < D; ++d) S[d] += v[d];
            tsum += T[jj];
        }
        for (int x = 0; x < c.width1; ++x) {
            if (x > 0) {
                const int32_t* va = V + (size_t)(x + 2 * r) * D;
                const int32_t* vs = V + (size_t)(x - 1) * D;
                for (int d = 0; d < D; ++d) S[d] += va[d] - vs[d];
                tsum += T[x + 2 * r] - T[x - 1];
            }
            /* minD > 0 makes lofs + width1 exceed W by minD columns; OpenCV's column-major
             * loop lets those writes spill into the next row's first (later masked) columns.
             * The observable result is that they are dropped.                               */
            if (c.lofs + x >= W) break;
            int16_t* out = drow + c.lofs + x;
            int minsad = INT_MAX, mind = -1;
            for (int d = 0; d < D; ++d)
                if (S[d] < minsad) { minsad = S[d]; mind = d; }
            if (tsum < c.tex) { *out = FILTERED; continue; }
            if (c.uniq > 0) {
                int thresh = minsad + (minsad * c.uniq / 100);
                int d;
                for (d = 0; d < D; ++d)
                    if ((d < mind - 1 || d > mind + 1) && S[d] <= thresh) break;
                if (d < D) { *out = FILTERED; continue; }
            }
            int pp = (mind + 1 < D) ? S[mind + 1] : S[D - 2];
            int nn = (mind > 0) ? S[mind - 1] : S[1];
            int den = pp + nn - 2 * S[mind] + iabs(pp - nn);
            int v = (D - mind - 1 + c.minD) * 256 + (den != 0 ? (pp - nn) * 256 / den : 0) + 15;
            *out = (int16_t)(v >> 4);
            if (crow) crow[c.lofs + x] = S[mind];
        }
    }
    free(S); free(T); free(V); free(c.rbase); free(c.lcol);
}

/* ---- A.4 left-right check (OpenCV validateDisparity) ----------------------------------- */
void orc_validate_disparity(int16_t* disp, size_t dstep, const int32_t* cost, size_t cstep,
                            int W, int rows, int minD, int numD, int disp12MaxDiff)
{
    const int maxD = minD + numD;
    const int minX1 = imax(maxD, 0), maxX1 = W + imin(minD, 0);
    const int INVALID = (minD - 1) * 16;
    const int maxDiff16 = disp12MaxDiff * 16;
    int* disp2 = (int*)malloc(sizeof(int) * 2 * (size_t)W);
    int* cost2 = disp2 + W;
    for (int y = 0; y < rows; ++y) {
        int16_t* dp = disp + (size_t)y * dstep;
        const int32_t* cp = cost + (size_t)y * cstep;
        for (int x = 0; x < W; ++x) { disp2[x] = INVALID; cost2[x] = INT_MAX; }
        for (int x = minX1; x < maxX1; ++x) {
            int d = dp[x];
            if (d == INVALID) continue;
            int c = cp[x];
            int x2 = x - ((d + 8) >> 4);
            if (x2 < 0 || x2 >= W) continue; /* unreachable for in-range disparities */
            if (cost2[x2] > c) { cost2[x2] = c; disp2[x2] = d; }
        }
        for (int x = minX1; x < maxX1; ++x) {
            int d = dp[x];
            if (d == INVALID) continue;
            int x0 = x - (d >> 4), x1 = x - ((d + 15) >> 4);
            if ((0 <= x0 && x0 < W && disp2[x0] > INVALID && iabs(disp2[x0] - d) > maxDiff16) &&
                (0 <= x1 && x1 < W && disp2[x1] > INVALID && iabs(disp2[x1] - d) > maxDiff16))
                dp[x] = (int16_t)INVALID;
        }
    }
    free(disp2);
}

/* ---- A.5 speckle filter (OpenCV filterSpeckles) ----------------------------------------- */
void orc_filter_speckles(int16_t* disp, size_t dstep, int W, int H, int newVal,
                         int maxSpeckleSize, int maxDiff)
{
    /* 4-connected components of pixels != newVal under |a-b| <= maxDiff; every component of
     * size <= maxSpeckleSize becomes newVal.  Breadth-first fill on a snapshot of the input,
     * so the result cannot depend on traversal order.                                        */
    size_t n = (size_t)W * H;
    int32_t* label = (int32_t*)calloc(n, sizeof(int32_t));
    int32_t* queue = (int32_t*)malloc(n * sizeof(int32_t));
    int16_t* snap = (int16_t*)malloc(n * sizeof(int16_t));
    for (int y = 0; y < H; ++y) memcpy(snap + (size_t)y * W, disp + (size_t)y * dstep, sizeof(int16_t) * (size_t)W);
    int32_t cur = 0;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        size_t s = (size_t)y * W + x;
        if (snap[s] == newVal || label[s]) continue;
        ++cur;
        size_t head = 0, tail = 0;
        queue[tail++] = (int32_t)s; label[s] = cur;
        while (head < tail) {
            int32_t q = queue[head++];
            int qy = q / W, qx = q % W, v = snap[q];
            const int nx[4] = {qx + 1, qx - 1, qx, qx};
            const int ny[4] = {qy, qy, qy + 1, qy - 1};
            for (int k = 0; k < 4; ++k) {
                if (nx[k] < 0 || nx[k] >= W || ny[k] < 0 || ny[k] >= H) continue;
                size_t t = (size_t)ny[k] * W + nx[k];
                if (label[t] || snap[t] == newVal || iabs(v - snap[t]) > maxDiff) continue;
                label[t] = cur; queue[tail++] = (int32_t)t;
            }
        }
        if ((int)tail <= maxSpeckleSize)
            for (size_t k = 0; k < tail; ++k) {
                int32_t q = queue[k];
                disp[(size_t)(q / W) * dstep + (q % W)] = (int16_t)newVal;
            }
    }
    free(snap); free(queue); free(label);
}

/* ---- A.1/A.2 whole pipeline ------------------------------------------------------------- */
typedef struct {
    const orc_bm_params* p; const uint8_t *Lp, *Rp; int W, H, row0, row1;
    int16_t* disp; size_t dstep; int32_t* cost;
} stripe_job;

static void* stripe_main(void* arg)
{
    stripe_job* j = (stripe_job*)arg;
    orc_bm_search(j->p, j->Lp, (size_t)j->W, j->Rp, (size_t)j->W, j->W, j->H, j->row0, j->row1,
                  j->disp, j->dstep, j->cost, (size_t)j->W);
    return NULL;
}

int orc_bm_compute(const orc_bm_params* p, const uint8_t* L, size_t lstep,
                   const uint8_t* R, size_t rstep, int W, int H,
                   int16_t* disp, size_t dstep_bytes, int nthreads)
{
    if (!p || !L || !R || !disp || W <= 0 || H <= 0) return ORC_ERR_BAD_SIZE;
    if (dstep_bytes % sizeof(int16_t)) return ORC_ERR_BAD_SIZE;
    const int D = p->numDisparities, minD = p->minDisparity, w = p->blockSize;
    if (p->preFilterCap < 1 || p->preFilterCap > 63) return ORC_ERR_BAD_PARAM;
    if (w < 5 || w > 255 || (w & 1) == 0 || w >= imin(W, H)) return ORC_ERR_BAD_PARAM;
    if (D <= 0 || D % 16 != 0) return ORC_ERR_BAD_PARAM;
    if (p->textureThreshold < 0 || p->uniquenessRatio < 0) return ORC_ERR_BAD_PARAM;

    const size_t dstep = dstep_bytes / sizeof(int16_t);
    const int16_t FILTERED = (int16_t)((minD - 1) * 16);
    const int lofs = imax(D - 1 + minD, 0), rofs = -imin(D - 1 + minD, 0);
    const int width1 = W - rofs - D + 1;
    int rect[4];
    int have = orc_bm_valid_rect(p, W, H, rect);
    if (lofs >= W || rofs >= W || width1 < 1 || !have) {
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) disp[(size_t)y * dstep + x] = FILTERED;
        return ORC_OK;
    }
    uint8_t* Lp = (uint8_t*)malloc((size_t)W * H);
    uint8_t* Rp = (uint8_t*)calloc((size_t)W * H + (size_t)D, 1);   /* + D: the legacy clamp (H1) over-reads the last row */
    int32_t* cost = (int32_t*)malloc(sizeof(int32_t) * (size_t)W * H);
    orc_prefilter_xsobel(L, lstep, W, H, Lp, (size_t)W, p->preFilterCap);
    orc_prefilter_xsobel(R, rstep, W, H, Rp, (size_t)W, p->preFilterCap);

    const int vy0 = rect[1], vy1 = rect[1] + rect[3];
    for (int y = 0; y < H; ++y) {
        if (y >= vy0 && y < vy1) continue;
        for (int x = 0; x < W; ++x) disp[(size_t)y * dstep + x] = FILTERED;
    }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > vy1 - vy0) nthreads = vy1 - vy0;
    if (nthreads > 64) nthreads = 64;
    stripe_job jobs[64]; pthread_t tids[64];
    for (int t = 0; t < nthreads; ++t) {
        stripe_job* j = &jobs[t];
        j->p = p; j->Lp = Lp; j->Rp = Rp; j->W = W; j->H = H;
        j->row0 = vy0 + (int)((long)(vy1 - vy0) * t / nthreads);
        j->row1 = vy0 + (int)((long)(vy1 - vy0) * (t + 1) / nthreads);
        j->disp = disp; j->dstep = dstep; j->cost = cost;
        if (nthreads == 1) stripe_main(j);
        else pthread_create(&tids[t], NULL, stripe_main, j);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; ++t) pthread_join(tids[t], NULL);

    if (p->disp12MaxDiff >= 0)
        orc_validate_disparity(disp + (size_t)vy0 * dstep, dstep, cost + (size_t)vy0 * W, (size_t)W,
                               W, vy1 - vy0, minD, D, p->disp12MaxDiff);
    for (int y = vy0; y < vy1; ++y) {
        int16_t* dp = disp + (size_t)y * dstep;
        for (int x = 0; x < rect[0]; ++x) dp[x] = FILTERED;
        for (int x = rect[0] + rect[2]; x < W; ++x) dp[x] = FILTERED;
    }
    if (p->speckleRange >= 0 && p->speckleWindowSize > 0)
        orc_filter_speckles(disp, dstep, W, H, FILTERED, p->speckleWindowSize, p->speckleRange);
    free(cost); free(Rp); free(Lp);
    return ORC_OK;
}
