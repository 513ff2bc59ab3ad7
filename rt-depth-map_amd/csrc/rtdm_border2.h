// rtdm_border2.h -- the device side of k_search_border2.hip (border columns with the ROWS in the lanes; see that file's
// header for the method), as a per-wave body: k_search_border2 runs it as a kernel of its own (batches: on a side stream),
// k_search_ring<.., FUSE> as the first workgroups of its own grid (single frames and small batches: a second launch costs a
// single 720p frame 8 us of its 84).
#pragma once
#include "rtdm_select.h"

#include <type_traits>
#include <utility>

namespace rtdm {

struct Border2Geom { int lx0, lx1, rx0, rx1, ro; };   // output-column ranges (as BorderGeom), output rows per wave

__device__ __forceinline__ uint32_t b2_scan_add(uint32_t t)   // inclusive prefix sum over the 64 lanes
{
#define RTDM_B2_SCAN(ctrl, rmask) t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, ctrl, rmask, 0xf, false)
    RTDM_B2_SCAN(0x111, 0xf); RTDM_B2_SCAN(0x112, 0xf); RTDM_B2_SCAN(0x114, 0xf); RTDM_B2_SCAN(0x118, 0xf);   // row_shr:1,2,4,8
    RTDM_B2_SCAN(0x142, 0xa);                                                                                  // row_bcast:15
    RTDM_B2_SCAN(0x143, 0xc);                                                                                  // row_bcast:31
#undef RTDM_B2_SCAN
    return t;
}

// one byte of a register array at a compile-time byte offset, replicated into all four bytes
template <int OFS, int N>
__device__ __forceinline__ uint32_t b2_rep_byte(const uint32_t (&rr)[N])
{
    constexpr uint32_t b = OFS & 3, sel = b * 0x01010101u;
    return __builtin_amdgcn_perm(0u, rr[OFS >> 2], sel);
}
// four bytes of a register array starting at a compile-time byte offset
template <int OFS, int N>
__device__ __forceinline__ uint32_t b2_span(const uint32_t (&rr)[N])
{
    if constexpr ((OFS & 3) == 0) return rr[OFS >> 2];
    else return __builtin_amdgcn_alignbyte(rr[(OFS >> 2) + 1], rr[OFS >> 2], (uint32_t)(OFS & 3));
}

template <int E, int NSP, int NDW, int... Q>
__device__ __forceinline__ uint32_t b2_span_sad(std::integer_sequence<int, Q...>, const uint32_t (&rr)[NDW], const uint32_t (&Lc)[NSP])
{
    uint32_t h = 0;
    ((h = __builtin_amdgcn_msad_u8(b2_span<E + 4 * Q>(rr), Lc[Q], h)), ...);
    return h;
}

template <int D, int NSP, bool RIGHT, int E, int NDW>
__device__ __forceinline__ uint32_t b2_row_sad(const uint32_t (&rr)[NDW], const uint32_t (&Lc)[NSP], const uint32_t (&Lrep)[2], int nrep)
{
    uint32_t h = b2_span_sad<E>(std::make_integer_sequence<int, NSP>{}, rr, Lc);
    const uint32_t rep = b2_rep_byte<E + (RIGHT ? 4 * NSP - 1 : 0)>(rr);
    h = __builtin_amdgcn_msad_u8(rep, Lrep[0], h);
    if (nrep > 4) h = __builtin_amdgcn_msad_u8(rep, Lrep[1], h);          // (wave-uniform; blockSize >= 11 only)
    return h;
}

template <int D, int NSP, bool RIGHT, int... EP>
__device__ __forceinline__ void b2_all_pairs(std::integer_sequence<int, EP...>, const uint32_t (&rr)[D / 4 + NSP + 1], const uint32_t (&Lc)[NSP],
                                             const uint32_t (&Lrep)[2], int nrep, int up_addr, uint32_t (&S)[D / 2])
{
    // pair EP: reversed disparities 2 EP (low half) and 2 EP + 1 (high half)
    ((void)([&] {
        const uint32_t h0 = b2_row_sad<D, NSP, RIGHT, 2 * EP>(rr, Lc, Lrep, nrep);
        const uint32_t h1 = b2_row_sad<D, NSP, RIGHT, 2 * EP + 1>(rr, Lc, Lrep, nrep);
        const uint32_t hp = h0 | (h1 << 16);
        const uint32_t P = b2_scan_add(hp);                                // (no carry between the halves: 64 w 2 cap <= 65535)
        const uint32_t Pu = (uint32_t)__builtin_amdgcn_ds_bpermute(up_addr, (int)P);   // P of the window's last row
        S[EP] = Pu - P + hp;                                               // per half: P(c + w - 1) - P(c - 1) >= 0, no borrow
    }()), ...);
}

// One WAVE: border column number wcol (left columns first), row block yblk (bg.ro output rows), frame f.  No LDS, no
// workgroup barrier: callable from k_search_border2 and from the first workgroups of k_search_ring's grid alike.
template <int D, int NSP>
__device__ __forceinline__ void border2_body(const Plane8& Lp, const Plane8& Rp, const Plane16W& disp, uint16_t* cost, const BMGeom& g, const Border2Geom& bg,
                                             int wcol, int yblk, int f)
{
    constexpr int NDW = D / 4 + NSP + 1;                 // staged right-row dwords per lane (+1: the span of e = D - 1 reads one dword further)
    const int lane = threadIdx.x & 63;
    const int w = g.w, r = g.r, W = g.W;
    const int nl = bg.lx1 - bg.lx0, ncols = nl + (bg.rx1 - bg.rx0);
    if (wcol >= ncols) return;
    const int x = wcol < nl ? bg.lx0 + wcol : bg.rx0 + (wcol - nl);
    const int col = g.lofs + x;
    if (col >= W) return;
    const int ys0 = g.vy0 + yblk * bg.ro;                // first output row of this wave; lane c = window rows ys0 - r + c ...
    const int nout = min(bg.ro, g.vy1 - ys0);
    const uint8_t* Lb = Lp.base + (size_t)f * Lp.frame;
    const uint8_t* Rb = Rp.base + (size_t)f * Rp.frame;
    const int yr = min(max(ys0 - r + lane, 0), g.H - 1);   // this lane's image row (rows past the frame: lanes that produce nothing)
    const uint8_t* lrow = Lb + (size_t)yr * Lp.pitch;
    const uint8_t* rrow = Rb + (size_t)yr * Rp.pitch;

    // ---- the column's clamp pattern (wave-uniform) ----------------------------------------------
    const int j0 = g.rofs + x - r;                       // unclamped base of sample 0
    const bool right = j0 >= 0;                          // (the host sends only columns that clamp at exactly one edge)
    // LEFT:  samples dx < k read R[0 + e] (replicated), sample dx >= k reads R[(dx - k) + e]: span byte dx - k.
    // RIGHT: samples dx <= m read R[j0 + dx + e]: span byte (4 NSP - 1 - m) + dx with the span anchored so that R[W - D + e]
    //        is its byte 4 NSP - 1 + e; samples dx > m read that byte (replicated).
    const int k = right ? 0 : -j0;
    const int m = right ? (W - D) - j0 : 0;
    const int anchor = right ? (W - D) - (4 * NSP - 1) : 0;           // image column of staged byte 0
    const int shift = right ? (4 * NSP - 1 - m) : -k;                 // span byte of sample dx = dx + shift
    const int rep_lo = right ? m + 1 : 0, rep_hi = right ? w : k;     // replicated samples [rep_lo, rep_hi)
    const int nrep = rep_hi - rep_lo;
    const auto lbyte = [&](int dx) -> uint32_t { return lrow[min(max(g.lofs + x - r + dx, 0), W - 1)]; };
    uint32_t Lc[NSP], Lrep[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < NSP; ++q) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int dx = 4 * q + b - shift;                         // the sample that sits at span byte 4 q + b
            const bool in = dx >= 0 && dx < w && !(dx >= rep_lo && dx < rep_hi);
            v |= (in ? lbyte(dx) : 0u) << (8 * b);
        }
        Lc[q] = v;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int dx = rep_lo + i;
        Lrep[i >> 2] |= (dx < rep_hi ? lbyte(dx) : 0u) << (8 * (i & 3));
    }
    // texture: every window sample once (Lc and Lrep partition them)
    const uint32_t capb = (uint32_t)(g.cap + PREFILTER_BIAS) * 0x01010101u;
    uint32_t trow = __builtin_amdgcn_msad_u8(capb, Lrep[0], 0u);
    trow = __builtin_amdgcn_msad_u8(capb, Lrep[1], trow);
#pragma unroll
    for (int q = 0; q < NSP; ++q) trow = __builtin_amdgcn_msad_u8(capb, Lc[q], trow);

    // ---- this lane's right-row span in registers (unaligned dword loads: the memory pipeline takes them) -------------
    uint32_t rr[NDW];
#pragma unroll
    for (int i = 0; i < NDW; ++i) rr[i] = *(const uint32_t*)(rrow + anchor + 4 * i);

    const int up_addr = min(lane + w - 1, 63) * 4;        // the lane that holds the window's last row
    uint32_t S[D / 2];
    if (right) b2_all_pairs<D, NSP, true>(std::make_integer_sequence<int, D / 2>{}, rr, Lc, Lrep, nrep, up_addr, S);
    else       b2_all_pairs<D, NSP, false>(std::make_integer_sequence<int, D / 2>{}, rr, Lc, Lrep, nrep, up_addr, S);
    const uint32_t Pt = b2_scan_add(trow);
    const int tsum = (int)((uint32_t)__builtin_amdgcn_ds_bpermute(up_addr, (int)Pt) - Pt + trow);

    // ---- selection: lane c < nout owns output row ys0 + c (all lanes run it: it votes wave-wide) ------------------------
    int m1; bool fail;
    const int out = select_disparity<D>(S, tsum, g, &m1, &fail);
    if (lane < nout) {
        const int y = ys0 + lane;
        const bool masked_col = g.mask_cols && (col < g.vx0 || col >= g.vx1);
        if (!fail && g.want_cost) cost[((size_t)f * g.H + y) * g.Ws + col] = (uint16_t)m1;
        disp.base[(size_t)f * disp.frame_e + (size_t)y * disp.pitch_e + col] = (int16_t)(masked_col ? g.filtered : out);
    }
}


}  // namespace rtdm
