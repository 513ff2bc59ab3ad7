// k_search_fast.hip -- K2, fast variant: the SAD window search as packed-u8 quad-SAD on CDNA4.
//
// Mapping (one workgroup = 256 output columns x `rs` output rows of one frame, 4 waves):
//   * wave = phase phi in 0..3, lane l handles output column x = x_tile + phi + 4*l.  All lanes of
//     a wave therefore share one byte alignment, so each wave reads its own byte-shifted copy of
//     the staged rows with plain aligned dword LDS reads (consecutive lanes -> consecutive banks).
//   * per lane, ALL D reversed-disparity SADs live in registers as packed u16 (D/2 VGPRs).
//   * horizontal: one v_qsad_pk_u16_u8 gives 4 consecutive disparities x 4 window bytes; the
//     window row is ceil(w/4) pieces, the last one masked (v_mqsad_pk_u16_u8 skips zero reference
//     bytes, so staged bytes carry a +1 bias and masked bytes are 0).
//   * vertical: sliding window down the strip.  The entering row accumulates straight into the
//     running sums (the quad-SAD's accumulator operand); the leaving row is recomputed from the
//     LDS ring and subtracted with v_pk_sub_u16.
//   * selection per output pixel, all in registers: rtdm_select.h (shared with k_search_ring).
//   * launch: 1-D grid, XCD-aware (see FastGeom); row strips per frame from a cost model, measured once per
//     batch shape by the caller (rtdm_api.hip, tune_strips).
//   Only columns whose whole window is free of border clamping are handled here; the 2*(w/2)
//   border columns are searched by extra workgroups of the same grid (rtdm_border.h; they are outside the
//   valid rectangle but feed the left-right check).  Semantics: SURVEY.md Appendix A.3b; oracle: oracle/bm_oracle.c.
#include "rtdm_border.h"
#include "rtdm_select.h"

#include <cmath>
#include <cstdlib>

namespace rtdm {

struct FastGeom {
    int x0, nx;          // output-column range [x0, x0+nx) handled by this kernel
    int rs;              // output rows per workgroup
    uint32_t lastmask;   // byte mask of the last (partial) window piece
    int tiles, strips;   // column tiles x row strips per frame
    int bgx, bgy;        // grid of the border-column work per frame (rtdm_border.h; 0 x 0: none)
    // 1-D launch grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs, so work is dealt in GROUPS of 8
    // consecutive ids that are all tiles or all border work; the two kinds of groups are interleaved evenly
    // (Bresenham).  Every XCD then gets the same share of both kinds whatever tiles / strips / batch are -- with a 3-D
    // grid the share depended on gridDim.x mod 8 and the search ran up to 1.5x slower for unlucky shapes.
    unsigned nfast, nborder;   // workgroups of each kind over the whole batch
    unsigned gfast, gborder;   // groups of 8
    int xcd_local;             // tiles only: contiguous runs of items per XCD (RTDM_FAST_XCD=0: plain order)
};

template <int D, int NP>
struct FastCfg {
    static constexpr int NG = D / 4;            // quad-SAD groups (4 disparities each)
    static constexpr int NW = NG + NP - 1;      // distinct 8-byte right windows per row visit
    static constexpr int NR = D / 2;            // packed u16x2 registers holding sad[0..D)
    static constexpr int LWD = 64 + NP;         // dwords per byte-shifted copy of the left row
    static constexpr int RWD = 64 + NG + NP;    // dwords per byte-shifted copy of the right row
    static constexpr int SLOT = 4 * (LWD + RWD);
    static constexpr int ITEMS = (SLOT + 255) / 256;
};

// Horizontal SADs of one staged row for this lane's column, disparity groups [G0, G0+GN):
// acc[g - G0] += the 4 packed SADs of group g.  l[] = the NP window pieces of the left row.
template <int D, int NP, int G0, int GN>
__device__ __forceinline__ void row_sads(const uint32_t (&l)[NP], const uint32_t* __restrict__ rp, uint64_t (&acc)[GN])
{
    // gfx950 wants 64-bit operands in even-aligned VGPR pairs.  Windows at odd dword offsets are
    // loaded through a second pointer whose index is laundered, so that the compiler issues its own
    // ds_read2_b32 for them instead of rebuilding the pair from already-loaded dwords with v_mov
    // (the laundered value is the INDEX, so the pointer keeps its LDS address space).
    int one = 1;
    asm volatile("" : "+v"(one));
    const uint32_t* rpo = rp + one;
#pragma unroll
    for (int j = G0; j < G0 + GN + NP - 1; ++j) {
        const uint32_t* wp = (j & 1) ? rpo + (j - 1) : rp + j;
        const uint64_t win = (uint64_t)wp[0] | ((uint64_t)wp[1] << 32);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int gi = j - k - G0;
            if (gi >= 0 && gi < GN) {
                if (k == NP - 1) acc[gi] = __builtin_amdgcn_mqsad_pk_u16_u8(win, l[k], acc[gi]);
                else             acc[gi] = __builtin_amdgcn_qsad_pk_u16_u8(win, l[k], acc[gi]);
            }
        }
    }
}

// The NP window pieces of the left row (last one masked) and their texture sum |L - cap|.
template <int NP>
__device__ __forceinline__ void left_pieces(const uint32_t* __restrict__ lp, uint32_t lastmask, uint32_t capb,
                                            uint32_t (&l)[NP], uint32_t& tacc)
{
#pragma unroll
    for (int k = 0; k < NP; ++k) l[k] = lp[k];
    l[NP - 1] &= lastmask;
#pragma unroll
    for (int k = 0; k < NP - 1; ++k) tacc = __builtin_amdgcn_sad_u8(l[k], capb, tacc);
    tacc = __builtin_amdgcn_msad_u8(capb, l[NP - 1], tacc);   // zero bytes of the reference are skipped
}

#ifndef FAST_TIGHT_BOUNDS     // 1: register budgets one occupancy step tighter for D >= 128 (A/B: tools/fast_ab.sh)
#define FAST_TIGHT_BOUNDS 0
#endif

template <int D, int NP>
__global__ __launch_bounds__(256, (FAST_TIGHT_BOUNDS ? (D >= 192 ? 3 : D >= 128 ? 4 : 1) : 1)) void k_search_fast(Plane8 Lp, Plane8 Rp, Plane16W disp, uint16_t* cost,
                                                     BMGeom g, FastGeom fg, BorderGeom bg)
{
    using C = FastCfg<D, NP>;
    constexpr int NG = C::NG, NR = C::NR, LWD = C::LWD, RWD = C::RWD, SLOT = C::SLOT, ITEMS = C::ITEMS;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

    const unsigned grp = blockIdx.x >> 3, slot = blockIdx.x & 7, ngroups = fg.gfast + fg.gborder;
    const unsigned nb = (unsigned)(((unsigned long long)grp * fg.gborder) / ngroups);            // border groups before this one
    if ((unsigned)(((unsigned long long)(grp + 1) * fg.gborder) / ngroups) > nb) {
        // border columns: latency-bound, so they ride in the same grid and overlap the VALU-bound tiles
        const unsigned bi = nb * 8 + slot;
        if (bi >= fg.nborder) return;
        const unsigned per = (unsigned)(fg.bgx * fg.bgy), fr = bi / per, id = bi - fr * per;
        border_body<(D + 63) / 64>((unsigned char*)lds, Lp, Rp, disp, cost, g, bg, (int)(id % fg.bgx), (int)(id / fg.bgx), (int)fr);
        return;
    }
    unsigned fi = (grp - nb) * 8 + slot;
    // no border workgroups in the grid (batches: they run on a side stream): XCD `slot` takes the contiguous run of items
    // [slot gfast, (slot + 1) gfast), so that the tiles of a strip and the strips of a frame share one L2 (k_search_ring.hip)
    if (fg.gborder == 0 && fg.xcd_local) fi = slot * fg.gfast + grp;
    if (fi >= fg.nfast) return;
    const int b_tile = (int)(fi % fg.tiles), b_strip = (int)((fi / fg.tiles) % fg.strips), b_frame = (int)(fi / (fg.tiles * fg.strips));

    const int tid = threadIdx.x, lane = tid & 63;
    const int phi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x_tile = fg.x0 + b_tile * 256;
    const int x = x_tile + phi + 4 * lane;
    const bool active = x < fg.x0 + fg.nx;
    const int ys0 = g.vy0 + b_strip * fg.rs;
    const int ys1 = min(ys0 + fg.rs, g.vy1);
    const int f = b_frame;
    const int w = g.w, r = g.r, RING = g.w + 2;
    const uint8_t* Lb = Lp.base + (size_t)f * Lp.frame;
    const uint8_t* Rb = Rp.base + (size_t)f * Rp.frame;
    const int Lbase = g.lofs + x_tile - r;   // image column of tile-row byte 0 (left)
    const int Rbase = g.rofs + x_tile - r;   //                                  (right)
    const uint32_t capb = (uint32_t)(g.cap + PREFILTER_BIAS) * 0x01010101u;

    // --- staging: every item = one dword of one byte-shifted copy ------------------------------
    // item idx in [0, 4*LWD): left copy c = idx / LWD, dword m = idx % LWD; then the right copies.
    // copy c, dword m holds tile-row bytes [c + 4m, c + 4m + 4) (the planes are biased by +1: rtdm_kernels.h).
    int it_q[ITEMS], it_sh[ITEMS];
    bool it_r[ITEMS], it_ok[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int idx = tid + it * 256;
        const bool isr = idx >= 4 * LWD;
        const int ii = isr ? idx - 4 * LWD : idx;
        const int c = isr ? ii / RWD : ii / LWD;
        const int m = isr ? ii - c * RWD : ii - c * LWD;
        const int col = (isr ? Rbase : Lbase) + c + 4 * m;
        it_r[it] = isr; it_q[it] = col & ~3; it_sh[it] = col & 3; it_ok[it] = idx < SLOT;
    }
    // The loads are unconditional: bytes past a row's end only ever reach lanes that are not `active` (their windows lie
    // outside the searched columns), and the prefiltered planes are allocated with 1 KB of slack behind the last row.
    // (Per-lane load predicates would sit in SGPR pairs across the whole row loop; with this kernel's scalar pressure
    // they were spilled to VGPR lanes and cost a v_readlane pair per load and row.)
    uint32_t pre[ITEMS][2];
    auto issue = [&](int row) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint8_t* rowp = (it_r[it] ? Rb : Lb) + (size_t)row * Lp.pitch;
            const int q = it_q[it];
            pre[it][0] = *(const uint32_t*)(rowp + q);
            pre[it][1] = *(const uint32_t*)(rowp + q + 4);
        }
    };
    auto commit = [&](int slot) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t v = __builtin_amdgcn_alignbyte(pre[it][1], pre[it][0], (uint32_t)it_sh[it]);
            if (it_ok[it]) lds[slot * SLOT + tid + it * 256] = v;           // (the planes carry the +1 bias themselves)
        }
    };

    uint64_t S[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) S[i] = 0;
    uint32_t tsum = 0;

    issue(ys0 - r);
    commit(0);
    __syncthreads();

    const int nsteps = (ys1 - ys0) + w - 1;
    int slot_in = 0, slot_out = 0, slot_next = 1;   // slot_out trails slot_in by w steps
    int16_t* db = disp.base + (size_t)f * disp.frame_e;
    const int col = g.lofs + x;
    const bool masked_col = g.mask_cols && (col < g.vx0 || col >= g.vx1);

    for (int s = 0; s < nsteps; ++s) {
        const int row_in = ys0 - r + s;
        const bool more = s + 1 < nsteps;
        if (more) issue(row_in + 1);

        {   // entering row: accumulate straight into the running sums
            // (the index is laundered so that the compiler keeps ONE base register per copy and
            //  folds the window offsets into the ds_read immediates instead of one v_add per read)
            int li = slot_in * SLOT + phi * LWD + lane, ri = slot_in * SLOT + 4 * LWD + phi * RWD + lane;
            asm volatile("" : "+v"(li), "+v"(ri));
            uint32_t l[NP];
            left_pieces<NP>(lds + li, fg.lastmask, capb, l, tsum);
            row_sads<D, NP, 0, NG>(l, lds + ri, S);
        }
        if (s >= w) {   // leaving row: recompute and subtract
            int li = slot_out * SLOT + phi * LWD + lane, ri = slot_out * SLOT + 4 * LWD + phi * RWD + lane;
            asm volatile("" : "+v"(li), "+v"(ri));
            uint32_t l[NP];
            uint32_t told = 0;
            left_pieces<NP>(lds + li, fg.lastmask, capb, l, told);
            // in chunks of 16 disparity groups, so that the temporaries stay at 32 VGPRs for any D
            constexpr int CG = NG < 16 ? NG : 16;
#pragma unroll
            for (int c0 = 0; c0 < NG; c0 += CG) {
                uint64_t T[CG];
#pragma unroll
                for (int i = 0; i < CG; ++i) T[i] = 0;
                if (c0 == 0)       row_sads<D, NP, 0, CG>(l, lds + ri, T);
                else if (c0 == 16) row_sads<D, NP, (NG > 16 ? 16 : 0), CG>(l, lds + ri, T);
                else if (c0 == 32) row_sads<D, NP, (NG > 32 ? 32 : 0), CG>(l, lds + ri, T);
                else               row_sads<D, NP, (NG > 48 ? 48 : 0), CG>(l, lds + ri, T);
#pragma unroll
                for (int i = 0; i < CG; ++i) {
                    // plain 32-bit subtractions of the packed pairs (v_sub_u32 issues 1.6x faster than v_pk_sub_u16): each
                    // half of the window sum is >= that half of the leaving row's sum, so no borrow crosses the halves
                    const uint32_t lo = (uint32_t)S[c0 + i] - (uint32_t)T[i];
                    const uint32_t hi = (uint32_t)(S[c0 + i] >> 32) - (uint32_t)(T[i] >> 32);
                    S[c0 + i] = (uint64_t)lo | ((uint64_t)hi << 32);
                }
            }
            tsum -= told;
            slot_out = (slot_out + 1 == RING) ? 0 : slot_out + 1;
        }

        // Selection is skipped for a wave none of whose 64 columns can produce a disparity here: untextured (flat walls,
        // sky: the texture test fails before anything else is looked at), outside the tile, or masked.  Exact: such a
        // pixel is FILTERED and writes no cost whatever its SADs are.
        const bool dead = !active || masked_col || (int)tsum < g.tex;
        if (s >= w - 1 && __builtin_amdgcn_ballot_w64(!dead) == 0) {
            if (active) db[(size_t)(row_in - r) * disp.pitch_e + col] = (int16_t)g.filtered;
        } else if (s >= w - 1) {
            const int y = row_in - r;
            uint32_t rr[NR];
#pragma unroll
            for (int i = 0; i < NG; ++i) { rr[2 * i] = (uint32_t)S[i]; rr[2 * i + 1] = (uint32_t)(S[i] >> 32); }
            // (minsad, FIRST argmin), uniqueness, sub-pixel: rtdm_select.h (shared with k_search_ring)
            int m1; bool fail;
            const int out = select_disparity<D>(rr, (int)tsum, g, &m1, &fail);
            if (active) {
                if (!fail && g.want_cost) cost[((size_t)f * g.H + y) * g.Ws + col] = (uint16_t)m1;
                db[(size_t)y * disp.pitch_e + col] = (int16_t)(masked_col ? g.filtered : out);
            }
        }

        if (more) commit(slot_next);
        slot_in = slot_next;
        slot_next = (slot_next + 1 == RING) ? 0 : slot_next + 1;
        __syncthreads();
    }
}

// ---- host side ----------------------------------------------------------------------------
static bool fast_range(const BMGeom& g, int* x0, int* nx)
{
    const int r = g.r;
    int xl = 0, xh = g.width1 - 1;
    xl = max(xl, r - g.lofs); xl = max(xl, r - g.rofs);
    xh = min(xh, g.W - 1 - g.lofs - r); xh = min(xh, g.W - g.D - g.rofs - r);
    xl = max(xl, g.cx0 - g.lofs); xh = min(xh, g.cx1 - g.lofs - 1);      // setROI1: only the needed columns
    if (x0) *x0 = xl;
    if (nx) *nx = xh - xl + 1;
    return xh >= xl;
}

void fast_border_ranges(const BMGeom& g, int* lx0, int* lx1, int* rx0, int* rx1);
int fast_strips_model(const BMGeom& g, int n);

static int fast_np(const BMGeom& g) { return (g.w + 3) / 4; }

template <int D, int NP>
static void launch_one(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g, int n, hipStream_t stream, bool fuse_border,
                       int strips_hint)
{
    using C = FastCfg<D, NP>;
    FastGeom fg;
    fast_range(g, &fg.x0, &fg.nx);
    const int rem = g.w - 4 * (NP - 1);
    fg.lastmask = rem >= 4 ? 0xffffffffu : ((1u << (8 * rem)) - 1u);
    const int nrows = g.vy1 - g.vy0;
    const int tiles = (fg.nx + 255) / 256;
    // Row strips per frame: a strip pays w-1 extra (entering-only, ~half price) row visits, a long workgroup leaves a long
    // tail in the last scheduling round (4 workgroups/CU resident => 1024 slots).  With K = tiles*n/1024 rounds per strip
    // the time is ~ (nrows/strips + h)(K*strips + 1/2), h = (w-1)/2, minimal at strips = sqrt(nrows / ((w-1) K)).
    // RTDM_FAST_WGS=<total workgroups> overrides (sweeps: tools/sweep_strips.sh, tools/sweep_wgs_small.py).
    static const int target_wgs = env_int("RTDM_FAST_WGS", 0);
    int strips;
    if (strips_hint > 0) strips = strips_hint;                      // measured choice (rtdm_api.hip, tune_strips)
    else if (target_wgs > 0) strips = (target_wgs + tiles * n - 1) / (tiles * n);
    else strips = fast_strips_model(g, n);
    strips = max(1, min(strips, (nrows + 15) / 16));
    fg.rs = (nrows + strips - 1) / strips;
    strips = (nrows + fg.rs - 1) / fg.rs;
    size_t ldsb = (size_t)(g.w + 2) * C::SLOT * 4;
    BorderGeom bg = {};
    fg.tiles = tiles; fg.strips = strips; fg.bgx = fg.bgy = 0;
    if (fuse_border) {
        int lx0, lx1, rx0, rx1; size_t blds = 0;
        fast_border_ranges(g, &lx0, &lx1, &rx0, &rx1);
        if (border_geometry(g, lx0, lx1, rx0, rx1, n, &bg, &fg.bgx, &fg.bgy, &blds)) ldsb = max(ldsb, blds);
        else fg.bgx = fg.bgy = 0;
    }
    fg.nfast = (unsigned)tiles * strips * n; fg.nborder = (unsigned)(fg.bgx * fg.bgy) * n;
    fg.gfast = (fg.nfast + 7) / 8; fg.gborder = (fg.nborder + 7) / 8;
    static const int xcd_local = [] { const char* e = getenv("RTDM_FAST_XCD"); return e ? atoi(e) : 1; }();
    fg.xcd_local = xcd_local;
    if (ldsb > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_search_fast<D, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipLaunchKernelGGL((k_search_fast<D, NP>), dim3((fg.gfast + fg.gborder) * 8), dim3(256), ldsb, stream, Lp, Rp, disp, (uint16_t*)cost, g, fg, bg);
}

// Instantiations: every (D, pieces) with D in {16,32,48,64,96,128,192,256} and 2..4 pieces (w = 5..15), plus
// 5 and 6 pieces (w = 17..21) for D in {32,64,128}.  Anything else runs the generic kernel.
#define RTDM_FAST_TABLE(X)                                                                               \
    X(16, 2) X(16, 3) X(16, 4) X(32, 2) X(32, 3) X(32, 4) X(32, 5) X(32, 6) X(48, 2) X(48, 3) X(48, 4)    \
    X(64, 2) X(64, 3) X(64, 4) X(64, 5) X(64, 6) X(96, 2) X(96, 3) X(96, 4)                               \
    X(128, 2) X(128, 3) X(128, 4) X(128, 5) X(128, 6) X(192, 2) X(192, 3) X(192, 4) X(256, 2) X(256, 3) X(256, 4)    \
    X(80, 2) X(80, 3) X(80, 4) X(112, 2) X(112, 3) X(112, 4) X(144, 2) X(144, 3) X(144, 4) X(160, 2) X(160, 3) X(160, 4)  \
    X(176, 2) X(176, 3) X(176, 4) X(208, 2) X(208, 3) X(208, 4) X(224, 2) X(224, 3) X(224, 4) X(240, 2) X(240, 3) X(240, 4)  \
    X(16, 5) X(16, 6) X(48, 5) X(48, 6) X(80, 5) X(80, 6) X(96, 5) X(96, 6) X(112, 5) X(112, 6) X(144, 5) X(144, 6)          \
    X(160, 5) X(160, 6) X(176, 5) X(176, 6) X(192, 5) X(192, 6)

bool fast_search_supported(const BMGeom& g)
{
    if (2L * g.cap * g.w * g.w > 32766) return false;       // packed u16 sums + the T+1 <= 32767 argument
    if (!fast_range(g, nullptr, nullptr)) return false;
    const int np = fast_np(g);
#define X(DD, PP) if (g.D == DD && np == PP) return true;
    RTDM_FAST_TABLE(X)
#undef X
    return false;
}

int fast_strips_model(const BMGeom& g, int n)
{
    int x0 = 0, nx = 0;
    if (!fast_range(g, &x0, &nx)) return 1;
    const int tiles = (nx + 255) / 256, nrows = g.vy1 - g.vy0;
    const int s = (int)(sqrtf((float)nrows * 1024.0f / ((float)(g.w - 1) * (float)tiles * (float)n)) + 0.5f);
    return max(1, min(s, (nrows + 15) / 16));
}

void launch_search_fast(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g, int n, hipStream_t stream, bool fuse_border,
                        int strips_hint)
{
    const int np = fast_np(g);
#define X(DD, PP) if (g.D == DD && np == PP) { launch_one<DD, PP>(Lp, Rp, disp, cost, g, n, stream, fuse_border, strips_hint); return; }
    RTDM_FAST_TABLE(X)
#undef X
}

void fast_border_ranges(const BMGeom& g, int* lx0, int* lx1, int* rx0, int* rx1)
{
    int x0 = 0, nx = 0;
    fast_range(g, &x0, &nx);
    *lx0 = max(0, g.cx0 - g.lofs); *lx1 = x0; *rx0 = x0 + nx; *rx1 = min(g.width1, g.cx1 - g.lofs);
}

}  // namespace rtdm
