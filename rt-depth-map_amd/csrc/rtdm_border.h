// rtdm_border.h -- device body of the border-column search (see k_search_border.hip for the design);
// shared by the stand-alone kernel and by k_search_fast, which runs it in extra workgroups of its own grid
// so that this latency-bound work overlaps the VALU-bound main search.
#pragma once

#include "rtdm_kernels.h"

namespace rtdm {

static constexpr int RB = 8;   // rows staged per batch

struct BorderGeom { int lx0, lx1, rx0, rx1, rs, rsp; };

// host side (k_search_border.hip): geometry, grid and LDS bytes of the border work; false if there is none
bool border_geometry(const BMGeom& g, int lx0, int lx1, int rx0, int rx1, int n, BorderGeom* out, int* gx, int* gy, size_t* lds_bytes);

// Wave-wide unsigned minimum with DPP row operations (no LDS round trips); result is uniform.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#define RTDM_DPP_MIN(ctrl, rmask)                                                                      \
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, ctrl, rmask, 0xf, false))
    RTDM_DPP_MIN(0xB1, 0xf);    // quad_perm [1,0,3,2]
    RTDM_DPP_MIN(0x4E, 0xf);    // quad_perm [2,3,0,1]
    RTDM_DPP_MIN(0x141, 0xf);   // row_half_mirror
    RTDM_DPP_MIN(0x140, 0xf);   // row_mirror          -> every lane holds its 16-lane row minimum
    RTDM_DPP_MIN(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    RTDM_DPP_MIN(0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave minimum
#undef RTDM_DPP_MIN
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int border_div_trunc(int num, int den)   // den > 0, |num| < 2^24: truncating quotient without the
{                                                                    // 30-instruction integer division sequence
    const unsigned an = (unsigned)(num < 0 ? -num : num);
    unsigned q = (unsigned)((float)an * __builtin_amdgcn_rcpf((float)den));
    const int rem = (int)an - (int)(q * (unsigned)den);          // |quotient| <= 128 here: the estimate is within one
    if (rem < 0) --q;
    else if (rem >= den) ++q;
    return num < 0 ? -(int)q : (int)q;
}

// Body of the border search for one workgroup (4 waves = 4 border columns x bg.rs rows).  bx/by = the
// workgroup's column-group / strip index, f = frame.  No workgroup barrier inside.
//
// These workgroups share the SIMDs with the VALU-bound tile workgroups, so what they cost is their instruction count.
// Per row visit and lane (= disparity e) the window's right-image bytes are R[rb(dx) + e] with rb(dx) = the clamped,
// monotone sample base: the w bytes come out of 8-byte spans of the staged row by v_perm with WAVE-UNIFORM selectors
// (which only depend on where the clamp bites), so a row visit is ceil(w/4) x (LDS reads, 2 v_alignbyte, v_perm,
// v_sad_u8 on 4 bytes) instead of w x (byte read, index clamp, single-byte SAD).
// LEGACY (stand-alone kernel only): rtdm_bm_params.legacy_right_clamp -- the sample base is clamped to W - rofs - 1 and
// base + e addresses a plane of step W, so bytes past the row's end are the next row's first bytes (zeros after the last).
template <int NCH, bool LEGACY = false>
__device__ __forceinline__ void border_body(unsigned char* smem, Plane8 Lp, Plane8 Rp, Plane16W disp, uint16_t* cost,
                                            const BMGeom& g, const BorderGeom& bg, int bx, int by, int f)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = g.D, w = g.w, r = g.r, W = g.W;
    const int nl = bg.lx1 - bg.lx0, ncols = nl + (bg.rx1 - bg.rx0);
    const int wcol = bx * 4 + wv;
    if (wcol >= ncols) return;                       // no workgroup barrier anywhere below
    const int x = wcol < nl ? bg.lx0 + wcol : bg.rx0 + (wcol - nl);
    const int col = g.lofs + x;
    if (col >= W) return;
    const int ys0 = g.vy0 + by * bg.rs;
    const int ys1 = min(ys0 + bg.rs, g.vy1);
    // per-wave LDS carve-up
    const int rsp = bg.rsp;                                            // bytes per staged right row
    const size_t per_wave = (size_t)RB * rsp + (size_t)RB * 32 + (size_t)w * (NCH * 64) * 2 + (size_t)w * 4;
    unsigned char* base = smem + (size_t)wv * ((per_wave + 15) & ~(size_t)15);
    uint8_t* Rbuf = base;                                              // RB x rsp
    uint8_t* Lbuf = Rbuf + (size_t)RB * rsp;                           // RB x 32   (w <= 21 on this path)
    uint16_t* Hring = (uint16_t*)(Lbuf + RB * 32);                     // w x (NCH*64)
    int* Tring = (int*)(Hring + (size_t)w * (NCH * 64));               // w

    const uint8_t* Lb = Lp.base + (size_t)f * Lp.frame;
    const uint8_t* Rb = Rp.base + (size_t)f * Rp.frame;
    const int j0 = x - r;                                              // first sample index
    const int Wc = LEGACY ? W - g.rofs - 1 : W - D;                    // largest right sample base
    const int rbmin = min(max(g.rofs + j0, 0), Wc);
    const int rbmax = min(max(g.rofs + j0 + w - 1, 0), Wc);
    const int ra = rbmin & ~3, rsh = rbmin & 3;                        // rows are staged from the aligned byte ra on
    const int ndw = (w + 3) >> 2;                                      // window dwords (<= 6)
    const int stage_dw = (rsh + NCH * 64 + (rbmax - rbmin) + 12 + 3) >> 2;   // dwords staged per right row (<= rsp / 4)
    // wave-uniform gather plan: window dword q = bytes idx(4q) .. idx(4q+3) of the lane's span, idx(dx) = rb(dx) - rbmin
    constexpr int MAXQ = 6;
    int aq[MAXQ];
    unsigned sel[MAXQ], capm[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int p0 = min(max(g.rofs + j0 + 4 * q, 0), Wc) - rbmin;
        aq[q] = p0 >> 2;
        unsigned sv = 0, cm = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dx = 4 * q + k;
            const int pk = min(max(g.rofs + j0 + dx, 0), Wc) - rbmin - 4 * aq[q];         // 0..7
            sv |= (dx < w ? (unsigned)pk : 0x0cu) << (8 * k);                              // 0x0c selects the constant 0
            cm |= (dx < w ? (unsigned)(g.cap + PREFILTER_BIAS) : 0u) << (8 * k);            // (biased planes: rtdm_kernels.h)
        }
        sel[q] = sv; capm[q] = cm;
    }
    // the lane's span starts at byte rsh + e of the staged row
    int dwi[NCH], bsh[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { const int ba = rsh + lane + 64 * c; dwi[c] = ba >> 2; bsh[c] = ba & 3; }

    for (int i = lane; i < w * NCH * 64; i += 64) Hring[i] = 0;
    if (lane < w) Tring[lane] = 0;
    int S[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) S[c] = 0;
    int tsum = 0;
    __builtin_amdgcn_wave_barrier();

    const int nsteps = (ys1 - ys0) + w - 1;
    int slot = 0;
    int16_t* db = disp.base + (size_t)f * disp.frame_e;
    const bool masked_col = g.mask_cols && (col < g.vx0 || col >= g.vx1);
    // What follows the wave-wide reductions of a row -- tests on three numbers, the sub-pixel division, the stores -- is
    // scalar work: done per row it would occupy the VALU with one live lane.  The rows of a batch leave their numbers in
    // lane k of five registers and the lanes finish the batch's rows side by side.
    int rec_m1 = 0, rec_a = 0, rec_pp = 0, rec_nn = 0, rec_yf = 0;
    for (int s0 = 0; s0 < nsteps; s0 += RB) {
        const int nb = min(RB, nsteps - s0);
        int nrec = 0;
        // the batch's rows are all requested before any of them is stored (a load, a wait and a store per row made the walk
        // eight dependent memory round trips per batch: the kernel's whole latency); stage_dw <= 128: two dwords per lane and row
        uint32_t rv[RB][2];
        uint8_t lv[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int row = ys0 - r + s0 + min(b, nb - 1);
            const uint32_t* rrow = (const uint32_t*)(Rb + (size_t)row * Rp.pitch + ra);    // plane pitches are multiples of 64
            if constexpr (LEGACY) {
                // staged dword i = bytes ra + 4i .. ra + 4i + 3 of the row CONTINUED by the next row (a plane of step W)
                const uint8_t* r0 = Rb + (size_t)row * Rp.pitch;
                const uint8_t* r1 = Rb + (size_t)min(row + 1, g.H - 1) * Rp.pitch;
                const bool more = row + 1 < g.H;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = lane + 64 * h, c0 = ra + 4 * i;
                    uint32_t v = 0;
                    if (i < stage_dw) {
                        if (c0 + 3 < W) v = rrow[i];
                        else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int c = c0 + k;
                                const uint32_t by = c < W ? r0[c] : (more && c - W < W) ? r1[c - W] : (uint32_t)PREFILTER_BIAS;   // (a zero of the oracle's plane)
                                v |= by << (8 * k);
                            }
                        }
                    }
                    rv[b][h] = v;
                }
            } else {
                rv[b][0] = lane < stage_dw ? rrow[lane] : 0u;
                rv[b][1] = lane + 64 < stage_dw ? rrow[lane + 64] : 0u;
            }
            lv[b] = lane < w ? Lb[(size_t)row * Lp.pitch + min(max(g.lofs + j0 + lane, 0), W - 1)] : (uint8_t)0;
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            uint32_t* rdst = (uint32_t*)(Rbuf + b * rsp);
            if (lane < stage_dw) rdst[lane] = rv[b][0];
            if (lane + 64 < stage_dw) rdst[lane + 64] = rv[b][1];
            if (lane < 4 * ndw) Lbuf[b * 32 + lane] = lv[b];
        }
        __builtin_amdgcn_wave_barrier();
        for (int b = 0; b < nb; ++b) {
            const int s = s0 + b;
            unsigned t = 0;
            unsigned h[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) h[c] = 0;
            const uint32_t* lrow = (const uint32_t*)(Lbuf + b * 32);
            const uint32_t* rrow = (const uint32_t*)(Rbuf + b * rsp);
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                if (q < ndw) {
                    const unsigned lv = lrow[q];
                    t = __builtin_amdgcn_sad_u8(lv, capm[q], t);
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const uint32_t* sp = rrow + dwi[c] + aq[q];
                        const unsigned s0w = sp[0], s1w = sp[1], s2w = sp[2];
                        const unsigned lo = __builtin_amdgcn_alignbyte(s1w, s0w, bsh[c]);
                        const unsigned hi = __builtin_amdgcn_alignbyte(s2w, s1w, bsh[c]);
                        h[c] = __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(hi, lo, sel[q]), lv, h[c]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                uint16_t* hp = &Hring[slot * (NCH * 64) + e];
                const int old = *hp;
                *hp = (uint16_t)h[c];
                S[c] += (int)h[c] - old;
            }
            const int told = Tring[slot];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) Tring[slot] = (int)t;
            tsum += (int)t - told;
            slot = (slot + 1 == w) ? 0 : slot + 1;
            if (s < w - 1) continue;

            const int y = ys0 - r + s - r;
            unsigned k = 0xffffffffu;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                if (e < D) k = min(k, ((unsigned)S[c] << 8) | (unsigned)e);
            }
            k = wave_min_u32(k);
            const int m1 = (int)(k >> 8), a = (int)(k & 0xffu);
            bool fail = tsum < g.tex;
            if (g.uniq > 0) {
                const int thresh = m1 + (int)((unsigned)(m1 * g.uniq) / 100u);     // wave-uniform: scalar ALU
                bool hit = false;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int e = lane + 64 * c;
                    hit |= (e < D) && (e < a - 1 || e > a + 1) && S[c] <= thresh;
                }
                fail |= __any(hit) != 0;
            }
            const int ip = (a + 1 < D) ? a + 1 : D - 2;
            const int in = (a > 0) ? a - 1 : 1;
            int pp = 0, nn = 0;                                        // ip, in are wave-uniform
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if ((ip >> 6) == c) pp = __builtin_amdgcn_readlane(S[c], ip & 63);
                if ((in >> 6) == c) nn = __builtin_amdgcn_readlane(S[c], in & 63);
            }
            const bool me = lane == nrec;                               // (one compare, five selects with scalar sources)
            rec_m1 = me ? m1 : rec_m1; rec_a = me ? a : rec_a; rec_pp = me ? pp : rec_pp; rec_nn = me ? nn : rec_nn;
            rec_yf = me ? y * 2 + (fail ? 1 : 0) : rec_yf;
            ++nrec;
        }
        if (lane < nrec) {
            const int y = rec_yf >> 1;
            int out = g.filtered;
            if (!(rec_yf & 1)) {
                const int den = rec_pp + rec_nn - 2 * rec_m1 + abs(rec_pp - rec_nn);
                const int v = (D - rec_a - 1 + g.minD) * 256 + (den != 0 ? border_div_trunc((rec_pp - rec_nn) * 256, den) : 0) + 15;
                out = v >> 4;
                if (g.want_cost) cost[((size_t)f * g.H + y) * g.Ws + col] = (uint16_t)rec_m1;
            }
            if (masked_col) out = g.filtered;
            db[(size_t)y * disp.pitch_e + col] = (int16_t)out;
        }
        __builtin_amdgcn_wave_barrier();
    }
}


template <int NCH, bool LEGACY>
__global__ __launch_bounds__(256) void k_search_border(Plane8 Lp, Plane8 Rp, Plane16W disp, uint16_t* cost,
                                                       BMGeom g, BorderGeom bg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    border_body<NCH, LEGACY>(smem, Lp, Rp, disp, cost, g, bg, blockIdx.x, blockIdx.y, blockIdx.z);
}

}  // namespace rtdm
