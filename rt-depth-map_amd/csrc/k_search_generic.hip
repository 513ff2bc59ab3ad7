// k_search_generic.hip -- K2, generic variant: SAD window search for ANY valid StereoBM
// configuration with numDisparities <= 256 (any odd blockSize, any minDisparity, any cap).
// It is the fallback behind k_search_fast.hip and a second, structurally different device
// implementation that the parity suite runs against the same oracle.
//
// Work decomposition (one workgroup = 64 output columns x RS output rows of one frame):
//   LDS  V[e][j]   column sums over the w window rows, for every reversed disparity index e and
//                  every sample column j of the tile (64 + w - 1 of them)
//        Ssc[e][c] the w-wide horizontal sums = sad[e] of output column c (kept so that the
//                  uniqueness and sub-pixel steps can re-read them)
//   per output row: slide V down by one row (add row y+r, subtract row y-r-1), then every thread
//   sums w entries of V, 4 waves split the disparity range, partial (min, argmin) pairs are merged
//   in index order so that the FIRST minimum wins (= largest disparity).
// Column clamping at the image border follows the reference formulation exactly:
//   left column = clamp(lofs + j, 0, W-1), right column = clamp(rofs + j, 0, W-D) + e.
#include "rtdm_kernels.h"

namespace rtdm {

static constexpr int RS = 24;   // output rows per workgroup

// TC = output columns per workgroup: 64 for whole frames, 8 for the narrow border strips that
// the fast kernel leaves over (work per workgroup scales with TC + w - 1 sample columns).
template <typename T, int TC>
__global__ __launch_bounds__(256) void k_search_generic(Plane8 Lp, Plane8 Rp, Plane16W disp, T* cost, BMGeom g,
                                                        int gx0, int gx1)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int D = g.D, w = g.w, r = g.r;
    const int TCH = TC + w - 1;               // sample columns per tile
    const int RW = (TCH + D + 3) & ~3;        // staged right-row bytes
    const int LW = (TCH + 3) & ~3;
    // carve-up (all offsets multiples of 4)
    T* V = (T*)smem;                                     // D * TCH
    T* Ssc = V + (size_t)D * TCH + ((D * TCH) & 1);      // D * TC
    int* Tcol = (int*)(Ssc + (size_t)D * TC);            // TCH
    constexpr int NQ = 256 / TC;                         // slices of the disparity range
    int* pmin = Tcol + TCH;                              // NQ * TC = 256
    int* pidx = pmin + 256;                              // 256
    int* uflag = pidx + 256;                             // TC
    int* m1s = uflag + TC;                               // TC
    int* mis = m1s + TC;                                 // TC
    short* lidx = (short*)(mis + TC);                    // TCH (left column of sample j)
    short* ridx = lidx + TCH + (TCH & 1);                // TCH (right base of sample j, tile-relative)
    uint8_t* Ln = (uint8_t*)(ridx + TCH + (TCH & 1));    // LW
    uint8_t* Lo = Ln + LW;
    uint8_t* Rn = Lo + LW;                               // RW
    uint8_t* Ro = Rn + RW;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = tid % TC, qs = tid / TC;               // output column within the tile, disparity slice
    const int x_tile = gx0 + blockIdx.x * TC;            // first output column index of the tile
    const int ys0 = g.vy0 + blockIdx.y * RS;
    const int ys1 = min(ys0 + RS, g.vy1);
    const int f = blockIdx.z;
    const uint8_t* Lb = Lp.base + (size_t)f * Lp.frame;
    const uint8_t* Rb = Rp.base + (size_t)f * Rp.frame;
    int16_t* db = disp.base + (size_t)f * disp.frame_e;

    // sample-column tables; rb0 = right base of the first sample column (clamps are monotone)
    const int Wc = g.legacy ? g.W - g.rofs - 1 : g.W - D;      // largest right sample base (rtdm_bm_params.legacy_right_clamp)
    const int rb0 = min(max(g.rofs + x_tile - r, 0), Wc);
    for (int jj = tid; jj < TCH; jj += 256) {
        const int j = x_tile + jj - r;
        lidx[jj] = (short)min(max(g.lofs + j, 0), g.W - 1);
        ridx[jj] = (short)(min(max(g.rofs + j, 0), Wc) - rb0);
        Tcol[jj] = 0;
    }
    for (int i = tid; i < D * TCH; i += 256) V[i] = 0;
    __syncthreads();

    const int nsteps = (ys1 - ys0) + w - 1;
    for (int s = 0; s < nsteps; ++s) {
        const int row_in = ys0 - r + s;
        const bool sub = s >= w;
        const int row_out = row_in - w;
        // stage the entering (and leaving) prefiltered rows
        {
            const uint8_t* lrow = Lb + (size_t)row_in * Lp.pitch;
            const uint8_t* rrow = Rb + (size_t)row_in * Rp.pitch;
            const uint8_t* lrow_o = Lb + (size_t)(sub ? row_out : row_in) * Lp.pitch;
            const uint8_t* rrow_o = Rb + (size_t)(sub ? row_out : row_in) * Rp.pitch;
            for (int jj = tid; jj < TCH; jj += 256) { Ln[jj] = lrow[lidx[jj]]; Lo[jj] = lrow_o[lidx[jj]]; }
            // legacy clamp: base + e addresses a plane of step W -- past the row's end come the next row's first bytes
            const int rin1 = row_in + 1, rout1 = (sub ? row_out : row_in) + 1;
            const uint8_t* rnext = Rb + (size_t)min(rin1, g.H - 1) * Rp.pitch;
            const uint8_t* rnext_o = Rb + (size_t)min(rout1, g.H - 1) * Rp.pitch;
            for (int k = tid; k < TCH + D; k += 256) {
                const int x = rb0 + k;
                const bool wrap = g.legacy && x >= g.W && x - g.W < g.W;
                // (past the last row the 3.x code reads what follows the plane: zeros in the oracle = the bias here)
                Rn[k] = x < g.W ? rrow[x] : (wrap && rin1 < g.H) ? rnext[x - g.W] : PREFILTER_BIAS;
                Ro[k] = x < g.W ? rrow_o[x] : (wrap && rout1 < g.H) ? rnext_o[x - g.W] : PREFILTER_BIAS;
            }
        }
        __syncthreads();
        // slide the column sums
        for (int e = wv; e < D; e += 4) {
            T* v = V + (size_t)e * TCH;
            for (int jj = lane; jj < TCH; jj += 64) {
                const int ri = ridx[jj] + e;
                int a = abs((int)Ln[jj] - (int)Rn[ri]);
                if (sub) a -= abs((int)Lo[jj] - (int)Ro[ri]);
                v[jj] = (T)(v[jj] + a);
            }
        }
        for (int jj = tid; jj < TCH; jj += 256) {
            int a = abs((int)Ln[jj] - (g.cap + PREFILTER_BIAS));          // (the planes are biased: rtdm_kernels.h)
            if (sub) a -= abs((int)Lo[jj] - (g.cap + PREFILTER_BIAS));
            Tcol[jj] += a;
        }
        __syncthreads();
        if (s < w - 1) continue;

        const int y = row_in - r;
        const int x = x_tile + c;                 // output column index
        const int col = g.lofs + x;               // image column
        const bool active = (x < gx1) && (col < g.W);
        // horizontal sums for this wave's slice of the disparity range
        const int e0 = (D * qs) / NQ, e1 = (D * (qs + 1)) / NQ;
        int best = 0x7fffffff, besti = -1;
        for (int e = e0; e < e1; ++e) {
            const T* v = V + (size_t)e * TCH + c;
            int sum = 0;
            for (int k = 0; k < w; ++k) sum += (int)v[k];
            Ssc[(size_t)e * TC + c] = (T)sum;
            if (sum < best) { best = sum; besti = e; }
        }
        pmin[qs * TC + c] = best; pidx[qs * TC + c] = besti;
        if (qs == 0) uflag[c] = 0;
        __syncthreads();
        int m1 = 0x7fffffff, mi = -1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int b = pmin[q * TC + c];
            if (b < m1) { m1 = b; mi = pidx[q * TC + c]; }
        }
        if (g.uniq > 0) {
            const int thresh = m1 + (m1 * g.uniq / 100);
            int hit = 0;
            for (int e = e0; e < e1; ++e)
                hit |= ((e < mi - 1 || e > mi + 1) && (int)Ssc[(size_t)e * TC + c] <= thresh);
            if (hit) atomicOr(&uflag[c], 1);
        }
        __syncthreads();
        if (qs == 0 && active) {
            int tsum = 0;
            for (int k = 0; k < w; ++k) tsum += Tcol[c + k];
            int out = g.filtered;
            if (tsum >= g.tex && !uflag[c]) {
                const int pp = (int)Ssc[(size_t)((mi + 1 < D) ? mi + 1 : D - 2) * TC + c];
                const int nn = (int)Ssc[(size_t)((mi > 0) ? mi - 1 : 1) * TC + c];
                const int den = pp + nn - 2 * m1 + abs(pp - nn);
                const int v = (D - mi - 1 + g.minD) * 256 + (den != 0 ? (pp - nn) * 256 / den : 0) + 15;
                out = v >> 4;
                if (g.want_cost) cost[((size_t)f * g.H + y) * g.Ws + col] = (T)m1;
            }
            if (g.mask_cols && (col < g.vx0 || col >= g.vx1)) out = g.filtered;
            db[(size_t)y * disp.pitch_e + col] = (int16_t)out;
        }
        // Ssc / pmin / uflag are rewritten only after the next step's two barriers
    }
}

static size_t generic_lds_bytes(const BMGeom& g, bool use16, int TC = 64)
{
    const size_t ts = use16 ? 2 : 4;
    const int TCH = TC + g.w - 1;
    const int RW = (TCH + g.D + 3) & ~3, LW = (TCH + 3) & ~3;
    size_t b = ((size_t)g.D * TCH + ((g.D * TCH) & 1)) * ts + (size_t)g.D * TC * ts;
    b += (size_t)(TCH + 2 * 256 + 3 * TC) * 4;
    b += (size_t)2 * (TCH + (TCH & 1)) * 2;
    b += (size_t)2 * LW + 2 * RW;
    return (b + 15) & ~(size_t)15;
}

bool generic_search_supported(const BMGeom& g, bool* use16)
{
    const long maxsum = 2L * g.cap * g.w * g.w;
    const bool u16 = maxsum < 65536;
    if (use16) *use16 = u16;
    if (g.D > 256 || g.W > 32767) return false;
    return generic_lds_bytes(g, u16) <= 160 * 1024;
}

template <typename T, int TC>
static void launch_generic_t(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g, int n,
                             hipStream_t stream, int gx0, int gx1, size_t lds)
{
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_search_generic<T, TC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((gx1 - gx0 + TC - 1) / TC, (g.vy1 - g.vy0 + RS - 1) / RS, n);
    hipLaunchKernelGGL((k_search_generic<T, TC>), grid, dim3(256), lds, stream, Lp, Rp, disp, (T*)cost, g, gx0, gx1);
}

void launch_search_generic(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g,
                           int n, hipStream_t stream, int gx0, int gx1)
{
    if (gx1 < 0) gx1 = g.width1;
    if (gx1 <= gx0) return;
    bool u16 = false;
    generic_search_supported(g, &u16);
    const bool narrow = (gx1 - gx0) <= 16;
    const size_t lds = generic_lds_bytes(g, u16, narrow ? 8 : 64);
    if (u16) {
        if (narrow) launch_generic_t<uint16_t, 8>(Lp, Rp, disp, cost, g, n, stream, gx0, gx1, lds);
        else        launch_generic_t<uint16_t, 64>(Lp, Rp, disp, cost, g, n, stream, gx0, gx1, lds);
    } else {
        if (narrow) launch_generic_t<uint32_t, 8>(Lp, Rp, disp, cost, g, n, stream, gx0, gx1, lds);
        else        launch_generic_t<uint32_t, 64>(Lp, Rp, disp, cost, g, n, stream, gx0, gx1, lds);
    }
}

}  // namespace rtdm
