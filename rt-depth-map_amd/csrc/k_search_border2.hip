// k_search_border2.hip -- K2, border columns, second form: ROWS in the lanes.
//
// The border columns (the first and last blockSize/2 searched columns, whose windows clamp at the image edge; they lie
// outside the valid rectangle but vote in validateDisparity, so they must be exact) are 0.7 % of a 720p frame's pixels --
// and cost the search stage 7 % of its time at 720p, 13 % at 640x480 and 29 % at 320x240 in the form of k_search_border
// (one WAVE per column, lanes = disparities, one pixel per walk step: 84 wave-instructions per pixel against 4.5 in
// k_search_ring; measured with the border launch removed, profiles/r03_border_cost.txt).  Here a wave takes one border
// column and 64 consecutive image ROWS, one row per lane, and walks the DISPARITIES (fully unrolled, compile-time D):
//   * a lane keeps its row's right-image span (D + 4 NSP bytes) in registers; the window of reversed disparity e is the
//     static byte span [e, e + 4 NSP) of it (v_alignbyte with immediate shifts);
//   * the clamp: the sample base is clamp(rofs + j, 0, W - D), so the window is a CONTIGUOUS run plus one byte replicated
//     (at the left edge the first k samples all read R[0 + e], at the right edge the last ones all read R[W - D + e]).  The
//     left-image bytes are split accordingly once per wave into Lc (aligned with the span, zero elsewhere) and Lrep (the
//     replicated samples): v_msad_u8 skips zero reference bytes (the planes are biased by +1, rtdm_kernels.h), so
//     SAD_e = msad(span_e, Lc) + msad(R_e x 4, Lrep) with no per-disparity selects;
//   * the vertical window sum is a prefix sum over the LANES (DPP row shifts / broadcasts) of two disparities packed as
//     u16 pairs, S(top row c) = P(c + w - 1) - P(c) + h(c): one ds_bpermute per pair; the packed pairs are exactly the
//     layout select_disparity (rtdm_select.h) wants, which then runs once per lane = per output pixel.
// About 19 wave-instructions per pixel at D = 64, w = 9.  Outputs: 64 - (w - 1) rows per wave.
// Not covered (k_search_border keeps them): the 3.x right-border rule (legacy_right_clamp), windows that clamp at both
// edges, w > 16, prefix sums that could leave 16 bits (64 w 2 cap > 65535), D outside the table below.
#include "rtdm_border2.h"

namespace rtdm {

template <int D, int NSP>
__global__ __launch_bounds__(256) void k_search_border2(Plane8 Lp, Plane8 Rp, Plane16W disp, uint16_t* cost, BMGeom g, Border2Geom bg)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    border2_body<D, NSP>(Lp, Rp, disp, cost, g, bg, (int)blockIdx.x * 4 + wv, (int)blockIdx.y, (int)blockIdx.z);
}

// ---- host side --------------------------------------------------------------------------------
#define RTDM_BORDER2_TABLE(X) X(16) X(32) X(48) X(64) X(96) X(128) X(192) X(256)

static int border2_nsp(const BMGeom& g) { return (g.w + 3) / 4; }

// every column of [lx0, lx1) u [rx0, rx1) must clamp at exactly one edge, in the way the kernel assumes
static bool border2_columns_ok(const BMGeom& g, int lx0, int lx1, int rx0, int rx1)
{
    const int nsp = border2_nsp(g);
    for (int pass = 0; pass < 2; ++pass) {
        for (int x = pass ? rx0 : lx0; x < (pass ? rx1 : lx1); ++x) {
            const int j0 = g.rofs + x - g.r, j1 = j0 + g.w - 1;
            const bool lo = j0 < 0, hi = j1 > g.W - g.D;
            if (lo == hi) return false;                              // clamps at both edges or at none
            if (hi && (g.W - g.D) - (4 * nsp - 1) < 0) return false;  // the anchored span would start left of the row
            if (hi && j0 > g.W - g.D) return false;                   // (cannot happen for searched columns: sample 0 is never clamped high)
            if (lo && -j0 > g.r + 1) return false;
        }
    }
    return true;
}

// Does this form cover the configuration?  If so: its geometry and its grid (gx workgroups of four columns x gy row blocks
// per frame; gx = 0: no border columns at all).  Also used by k_search_ring to put these workgroups in front of its own.
bool border2_plan(const BMGeom& g, int lx0, int lx1, int rx0, int rx1, Border2Geom* bg, int* gx, int* gy)
{
    static const int enabled = env_int("RTDM_BORDER2", 1);            // A/B switch: 0 = k_search_border for everything
    if (!enabled || g.legacy || g.w > 16 || g.w < 5) return false;
    if (64L * g.w * 2 * g.cap > 65535) return false;                  // packed u16 prefix sums over 64 rows
    if (2L * g.cap * g.w * g.w > 32766) return false;                 // select_disparity's T + 1 <= 32767
    if (!g.cost16 && g.want_cost) return false;
    bool have = false;
#define X(DD) have |= g.D == DD;
    RTDM_BORDER2_TABLE(X)
#undef X
    if (!have) return false;
    lx1 = max(lx1, lx0); rx1 = max(rx1, rx0);
    const int ncols = (lx1 - lx0) + (rx1 - rx0);
    *bg = Border2Geom{lx0, lx1, rx0, rx1, 64 - (g.w - 1)};
    *gx = *gy = 0;
    if (ncols <= 0) return true;
    if (!border2_columns_ok(g, lx0, lx1, rx0, rx1)) return false;
    *gx = (ncols + 3) / 4;
    *gy = ((g.vy1 - g.vy0) + bg->ro - 1) / bg->ro;
    return true;
}

bool launch_search_border2(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g, int n, hipStream_t stream,
                           int lx0, int lx1, int rx0, int rx1)
{
    Border2Geom bg;
    int gx = 0, gy = 0;
    if (!border2_plan(g, lx0, lx1, rx0, rx1, &bg, &gx, &gy)) return false;
    if (gx == 0) return true;
    const dim3 grid(gx, gy, n), block(256);
    const int nsp = border2_nsp(g);
#define X(DD)                                                                                                                   \
    if (g.D == DD) {                                                                                                            \
        if (nsp == 2) hipLaunchKernelGGL((k_search_border2<DD, 2>), grid, block, 0, stream, Lp, Rp, disp, (uint16_t*)cost, g, bg);       \
        else if (nsp == 3) hipLaunchKernelGGL((k_search_border2<DD, 3>), grid, block, 0, stream, Lp, Rp, disp, (uint16_t*)cost, g, bg);  \
        else hipLaunchKernelGGL((k_search_border2<DD, 4>), grid, block, 0, stream, Lp, Rp, disp, (uint16_t*)cost, g, bg);                \
        return true;                                                                                                            \
    }
    RTDM_BORDER2_TABLE(X)
#undef X
    return false;
}

}  // namespace rtdm
