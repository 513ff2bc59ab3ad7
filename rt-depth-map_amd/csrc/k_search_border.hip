// k_search_border.hip -- K2, border columns of the fast path.
//
// k_search_fast covers the output columns whose 9x9 (w x w) window needs no border clamping.  The
// first and last w/2 computed columns do clamp (left column = clamp(lofs+j,0,W-1), right column =
// clamp(rofs+j,0,W-D)+e); they lie outside the valid rectangle, so they
// are masked in the end, but validateDisparity's first pass lets them vote, so they must be exact.
// They are only 2*(w/2) columns, which suits the transposed mapping: one WAVE per border column,
// LANES = reversed disparity index e (e = lane + 64*c), walking down a strip of rows.
//   * per row the wave stages the (w-1+D)-byte right span (whole dwords) and the w left samples in LDS (8 rows
//     per batch so that global latency is paid once per batch); lane e's w window bytes R[rb(dx)+e] are gathered
//     from 8-byte spans by v_alignbyte + v_perm with wave-uniform selectors (they only depend on where the clamp
//     bites), so SAD(e) is ceil(w/4) four-byte v_sad_u8;
//   * vertical sliding window through an LDS ring of the last w row-SADs per e;
//   * selection with wave-level operations: min-reduce of the key sad<<8|e (first minimum),
//     __any() for the uniqueness test, readlane for sad[mind +- 1].
#include "rtdm_border.h"

#include <cstdlib>
#include <type_traits>

namespace rtdm {

bool border_search_supported(const BMGeom& g)
{
    return g.w <= 21 && g.D <= 256 && 2L * g.cap * g.w * g.w <= 65535;
}

// Geometry + LDS bytes of the border work; returns the grid (x = column groups, y = strips), 0 columns -> false.
// Rows per border workgroup: 128 for batches (the w-1 halo rows are then 6 % of the visits), shorter when the whole launch
// would otherwise consist of a handful of long, latency-bound walks (single frames: the reference's real-time case).
static int border_rows(int nrows, int ncols, int n)
{
    static const int rs_env = env_int("RTDM_BORDER_RS", 0);
    if (rs_env >= 8) return rs_env;
    const long want = (long)nrows * ((ncols + 3) / 4) * n / 256;     // rows per workgroup that still leave >= 256 workgroups
    return (int)std::min(128L, std::max(16L, want));
}

bool border_geometry(const BMGeom& g, int lx0, int lx1, int rx0, int rx1, int n, BorderGeom* out, int* gx, int* gy, size_t* lds_bytes)
{
    const int ncols = max(0, lx1 - lx0) + max(0, rx1 - rx0);
    if (ncols <= 0) return false;
    BorderGeom bg;
    bg.lx0 = lx0; bg.lx1 = max(lx1, lx0); bg.rx0 = rx0; bg.rx1 = max(rx1, rx0);
    const int nrows = g.vy1 - g.vy0;
    bg.rs = border_rows(nrows, ncols, n);
    bg.rsp = (g.D + g.w + 3 + 63 + 64) & ~3;       // + one chunk of slack for lanes with e >= D
    const int nch = (g.D + 63) / 64;
    const size_t per_wave = ((size_t)RB * bg.rsp + (size_t)RB * 32 + (size_t)g.w * (nch * 64) * 2 + (size_t)g.w * 4 + 15) & ~(size_t)15;
    *out = bg; *gx = (ncols + 3) / 4; *gy = (nrows + bg.rs - 1) / bg.rs; *lds_bytes = per_wave * 4;
    return true;
}

void launch_search_border(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g, int n,
                          hipStream_t stream, int lx0, int lx1, int rx0, int rx1)
{
    const int ncols = max(0, lx1 - lx0) + max(0, rx1 - rx0);
    if (ncols <= 0) return;
    if (launch_search_border2(Lp, Rp, disp, cost, g, n, stream, lx0, lx1, rx0, rx1)) return;   // rows in the lanes: far fewer instructions
    BorderGeom bg;
    bg.lx0 = lx0; bg.lx1 = max(lx1, lx0); bg.rx0 = rx0; bg.rx1 = max(rx1, rx0);
    const int nrows = g.vy1 - g.vy0;
    bg.rs = border_rows(nrows, ncols, n);
    bg.rsp = (g.D + g.w + 3 + 63 + 64) & ~3;       // + one chunk of slack for lanes with e >= D
    const int nch = (g.D + 63) / 64;
    const size_t per_wave = ((size_t)RB * bg.rsp + (size_t)RB * 32 + (size_t)g.w * (nch * 64) * 2 + (size_t)g.w * 4 + 15) & ~(size_t)15;
    const size_t lds = per_wave * 4;
    dim3 grid((ncols + 3) / 4, (nrows + bg.rs - 1) / bg.rs, n), block(256);
    const auto go = [&](auto nc, auto lg) {
        hipLaunchKernelGGL((k_search_border<decltype(nc)::value, decltype(lg)::value>), grid, block, lds, stream, Lp, Rp, disp, (uint16_t*)cost, g, bg);
    };
    const auto pick = [&](auto lg) {
        switch (nch) {
            case 1: go(std::integral_constant<int, 1>{}, lg); break;
            case 2: go(std::integral_constant<int, 2>{}, lg); break;
            case 3: go(std::integral_constant<int, 3>{}, lg); break;
            default: go(std::integral_constant<int, 4>{}, lg); break;
        }
    };
    if (g.legacy) pick(std::true_type{});
    else pick(std::false_type{});
}

}  // namespace rtdm
