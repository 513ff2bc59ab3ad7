// rtdm_select.h -- per-pixel selection of the SAD search, all in registers (oracle:
// oracle/bm_oracle.c:175-191): given the D window SADs of one pixel as packed u16 pairs -- rr[i] holds sad[2i] (low
// half) and sad[2i+1] (high half), index = REVERSED disparity -- produce
//   (minsad, FIRST argmin)            ties => the smallest index = the largest disparity
//   texture / uniqueness rejection    any index outside [a-1, a+1] with sad <= minsad + minsad*ratio/100 => FILTERED
//   sub-pixel disparity x16           ((D-a-1+minD)*256 + (p-n)*256/den + 15) >> 4, truncating division
// Lane = pixel; the 64 lanes of the wave must all call it (it uses wave-wide votes to skip work).
//
// Two levels: packed minima of groups of four registers (eight disparities; v_pk_min_u16 serves two values per
// instruction and needs no key), 32-bit keys (min << 8 | group) built by v_perm and reduced with v_min3_u32 give
// (minsad, first group), the six registers around that group come out of a v_cndmask tree and the eight in-group keys
// (sad << 8 | e) give the FIRST argmin; uniqueness is the identity
//     sum_e max(T+1 - sad[e], 0)  ==  the same sum over {a-1, a, a+1}
// evaluated with saturating packed u16 ops (groups above the threshold in the whole wave are skipped);
// sad[a +- 1] are among the six fetched registers.  Needs every sad <= 32766.
#pragma once

#include "rtdm_kernels.h"

namespace rtdm {

typedef unsigned short sel_us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t sel_pk_min(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(sel_us2, a), __builtin_bit_cast(sel_us2, b))); }
__device__ __forceinline__ uint32_t sel_pk_sub_sat(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(sel_us2, a), __builtin_bit_cast(sel_us2, b))); }
__device__ __forceinline__ uint32_t sel_pk_add_sat(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(sel_us2, a), __builtin_bit_cast(sel_us2, b))); }

// The sub-pixel term of A.3b, trunc((p - n) * 256 / den) with den = p + n - 2 m + |p - n| (0 if den == 0), for m <= min(p, n):
// den = 2 (max(p, n) - m) and |p - n| <= max(p, n) - m, so the magnitude is floor(|p - n| * 128 / u) <= 128 with
// u = max(p, n) - m -- one float estimate (both operands < 2^22: exact) and one correction each way, no branches.
__device__ __forceinline__ int sel_subpixel(int p, int n, int m)
{
    const int hi = max(p, n), lo = min(p, n);
    const unsigned an = (unsigned)(hi - lo) << 7, u = (unsigned)max(hi - m, 1);      // (u = 0 only with |p - n| = 0: quotient 0 either way)
    unsigned q = (unsigned)((float)an * __builtin_amdgcn_rcpf((float)u));
    const int rem = (int)an - (int)(q * u);
    q += (unsigned)(rem >= (int)u);
    q += (unsigned)(rem >> 31);                                                        // -1 if the estimate was one too large
    return p < n ? -(int)q : (int)q;
}

// Returns the x16 disparity (g.filtered if rejected); *minsad = the winning SAD, *rejected = the pixel failed a test.
// SEQ_TREES: build the six fetch trees one after the other (fewer live registers, less instruction-level parallelism).
template <int D, bool SEQ_TREES = (D >= 96)>
__device__ __forceinline__ int select_disparity(const uint32_t (&rr)[D / 2], int tsum, const BMGeom& g, int* minsad, bool* rejected)
{
    constexpr int NR = D / 2, NGp = NR / 4;
    uint32_t kacc[2] = {0xffffffffu, 0xffffffffu};   // two chains: no back-to-back dependency
    uint32_t gmin[NGp];                               // kept: the uniqueness test skips groups above its threshold
#pragma unroll
    for (int gq = 0; gq < NGp; ++gq) {
        const uint32_t gm = sel_pk_min(sel_pk_min(rr[4 * gq], rr[4 * gq + 1]), sel_pk_min(rr[4 * gq + 2], rr[4 * gq + 3]));
        gmin[gq] = gm;
        const uint32_t gc = (uint32_t)gq | ((uint32_t)gq << 8);
        const uint32_t klo = __builtin_amdgcn_perm(gm, gc, 0x0C050400u);
        const uint32_t khi = __builtin_amdgcn_perm(gm, gc, 0x0C070601u);
        kacc[gq & 1] = min(min(kacc[gq & 1], klo), khi);
    }
    const uint32_t kmin = min(kacc[0], kacc[1]);
    const int m1 = (int)(kmin >> 8);
    const int gs = (int)(kmin & 0xffu);
    // six[k] = rr[4 gs - 1 + k], k = 0..5 (0 outside the array): binary select on the bits of gs.  One output at a time
    // (the empty asm keeps the trees from being interleaved): interleaved, the six trees keep 6 * NGp/2 temporaries alive
    // next to the D/2 SAD registers, which is what pushed D = 128 over 128 VGPRs.
    uint32_t six[6];
    {
        constexpr int HBG = (NGp - 1) >= 16 ? 16 : (NGp - 1) >= 8 ? 8 : (NGp - 1) >= 4 ? 4 : (NGp - 1) >= 2 ? 2 : 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            uint32_t cand[NGp];
#pragma unroll
            for (int gq = 0; gq < NGp; ++gq) {
                const int idx = 4 * gq - 1 + k;
                cand[gq] = (idx >= 0 && idx < NR) ? rr[idx] : 0u;
            }
            int len = NGp;
#pragma unroll
            for (int bit = HBG; bit >= 1; bit >>= 1) {
                const bool up = (gs & bit) != 0;
#pragma unroll
                for (int i = 0; i < bit; ++i)
                    if (i < len) cand[i] = up ? ((i + bit < len) ? cand[i + bit] : 0u) : cand[i];
                len = bit < len ? bit : len;
            }
            six[k] = cand[0];
            if (SEQ_TREES) asm volatile("" : "+v"(six[k]));
        }
    }
    uint32_t k3[2] = {0xffffffffu, 0xffffffffu};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t ec = (uint32_t)(2 * q) | ((uint32_t)(2 * q + 1) << 8);
        const uint32_t klo = __builtin_amdgcn_perm(six[1 + q], ec, 0x0C050400u);
        const uint32_t khi = __builtin_amdgcn_perm(six[1 + q], ec, 0x0C070601u);
        k3[q & 1] = min(min(k3[q & 1], klo), khi);
    }
    const int a = 8 * gs + (int)(min(k3[0], k3[1]) & 0xffu);
    // the two packed registers that hold sad[a-1 .. a+1] are among the six
    const int am1 = a > 0 ? a - 1 : 0;
    const int jl = (am1 >> 1) - (4 * gs - 1);                 // 0..4
    const bool j1 = (jl & 1) != 0, j2 = (jl & 2) != 0, j4 = (jl & 4) != 0;
    const uint32_t e0 = j4 ? six[4] : (j2 ? (j1 ? six[3] : six[2]) : (j1 ? six[1] : six[0]));
    const uint32_t e1 = j4 ? six[5] : (j2 ? (j1 ? six[4] : six[3]) : (j1 ? six[2] : six[1]));
    const int posc = a > 0 ? (am1 & 1) + 1 : 0;         // position of sad[a] among the 4 fetched
    const auto elem = [&](int pos) -> int {
        return (int)__builtin_amdgcn_perm(e1, e0, 0x0C0C0100u + 0x0202u * (uint32_t)pos);
    };
    const bool has_n = a > 0, has_p = a + 1 < D;
    const int n_real = elem(has_n ? posc - 1 : 0);
    const int p_real = elem(has_p ? posc + 1 : 0);
    bool fail = tsum < g.tex;
    if (g.uniq > 0) {
        uint32_t T = (uint32_t)m1 + ((uint32_t)m1 * (uint32_t)g.uniq) / 100u;
        T = min(T, 32766u);
        const uint32_t T1 = T + 1u, T1pk = T1 * 0x00010001u;
        // saturating sums of non-negative terms are order independent: four chains
        // (the empty asm pins "four subtractions, then four additions": back-to-back dependent
        //  packed ops cost a wait state each on gfx950)
        uint32_t zz[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NR; i += 4) {
            // a group none of whose eight values reaches the threshold in any lane adds nothing (exact)
            if (__builtin_amdgcn_ballot_w64(sel_pk_sub_sat(T1pk, gmin[i >> 2]) != 0u) == 0) continue;
            uint32_t t0 = sel_pk_sub_sat(T1pk, rr[i]), t1 = sel_pk_sub_sat(T1pk, rr[i + 1]);
            uint32_t t2 = sel_pk_sub_sat(T1pk, rr[i + 2]), t3 = sel_pk_sub_sat(T1pk, rr[i + 3]);
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
            zz[0] = sel_pk_add_sat(zz[0], t0); zz[1] = sel_pk_add_sat(zz[1], t1);
            zz[2] = sel_pk_add_sat(zz[2], t2); zz[3] = sel_pk_add_sat(zz[3], t3);
        }
        const uint32_t z = sel_pk_add_sat(sel_pk_add_sat(zz[0], zz[1]), sel_pk_add_sat(zz[2], zz[3]));
        const auto term = [&](int v) -> uint32_t { return T1 > (uint32_t)v ? T1 - (uint32_t)v : 0u; };
        const uint32_t wsame = term(m1);
        const uint32_t wother = (has_n ? term(n_real) : 0u) + (has_p ? term(p_real) : 0u);
        const uint32_t zlo = z & 0xffffu, zhi = z >> 16;
        const bool even = (a & 1) == 0;
        fail |= (even ? zlo : zhi) != wsame;
        fail |= (even ? zhi : zlo) != wother;
    }
    const int pp = has_p ? p_real : n_real;
    const int nn = has_n ? n_real : p_real;
    const int out = ((D - a - 1 + g.minD) * 256 + sel_subpixel(pp, nn, m1) + 15) >> 4;
    *minsad = m1;
    *rejected = fail;
    return fail ? g.filtered : out;
}

// The same selection with the data-dependent register fetches done through LDS: the D values are written once to the
// lane's scratch record (conflict-free 16-byte stores at a 144-byte lane stride), the winning group comes back with ONE
// 16-byte read and sad[a-1], sad[a+1] with two 2-byte reads -- in place of the 42 + 14 v_cndmask of the register tree --
// and both round trips are in flight while the uniqueness sum (which needs only minsad) is computed.
// scr: this LANE's record, SelRecord<D>::DWORDS dwords apart from its neighbours', 16-byte aligned; private to the wave.
// D/2 SAD dwords + D/16 dwords of packed group minima (GroupSelectRec), rounded up to an ODD number of 16-byte quads so that
// eight consecutive lanes' 16-byte stores start in eight different bank quads: 12, 20, 28, 36 dwords for D = 16 .. 64 (the
// minima fill what used to be padding), 60 / 76 / 108 / 148 for D = 96 / 128 / 192 / 256
// (MINIMA = false, GroupSelect at D = 96 / 128: D/2 + 4 = 52 / 68 dwords)
// B64: the record is written and read in 8-byte pieces instead (GroupSelectRec): the stride only has to be an odd number of
// 8-byte pairs -- 34 dwords at D = 64, which is what lets four workgroups of k_search_ring<64, 9, 4> share a CU's 160 KB.
template <int D, bool MINIMA = true, bool B64 = false> struct SelRecord {
    static constexpr int QUADS = MINIMA ? (D / 2 + D / 16 + 3) / 4 : D / 8 + 1;
    static constexpr int PAIRS = (D / 2 + D / 16 + 1) / 2;
    static constexpr int DWORDS = B64 ? 2 * (PAIRS | 1) : 4 * (QUADS | 1);
};
template <int D>
__device__ __forceinline__ int select_disparity_lds(const uint32_t (&rr)[D / 2], int tsum, const BMGeom& g, uint32_t* scr,
                                                    int* minsad, bool* rejected)
{
    constexpr int NR = D / 2, NGp = NR / 4;
    static_assert(D == 16 || D == 32 || D == 48 || D == 64, "record stride checked for these sizes only");
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < NR; i += 4) *(u4*)(scr + i) = u4{rr[i], rr[i + 1], rr[i + 2], rr[i + 3]};
    uint32_t kacc[2] = {0xffffffffu, 0xffffffffu};
    uint32_t gmin[NGp];
#pragma unroll
    for (int gq = 0; gq < NGp; ++gq) {
        const uint32_t gm = sel_pk_min(sel_pk_min(rr[4 * gq], rr[4 * gq + 1]), sel_pk_min(rr[4 * gq + 2], rr[4 * gq + 3]));
        gmin[gq] = gm;
        const uint32_t gc = (uint32_t)gq | ((uint32_t)gq << 8);
        const uint32_t klo = __builtin_amdgcn_perm(gm, gc, 0x0C050400u);
        const uint32_t khi = __builtin_amdgcn_perm(gm, gc, 0x0C070601u);
        kacc[gq & 1] = min(min(kacc[gq & 1], klo), khi);
    }
    const uint32_t kmin = min(kacc[0], kacc[1]);
    const int m1 = (int)(kmin >> 8);
    const int gs = (int)(kmin & 0xffu);
    const u4 grp = *(const u4*)(scr + 4 * gs);                  // the four registers of the winning group
    // uniqueness: sum_e max(T+1 - sad[e], 0) over ALL e, compared below with the same sum over {a-1, a, a+1}
    uint32_t z = 0;
    uint32_t T1 = 0;
    if (g.uniq > 0) {
        uint32_t T = (uint32_t)m1 + ((uint32_t)m1 * (uint32_t)g.uniq) / 100u;
        T = min(T, 32766u);
        T1 = T + 1u;
        const uint32_t T1pk = T1 * 0x00010001u;
        uint32_t zz[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NR; i += 4) {
            // a group none of whose eight values reaches the threshold in any lane adds nothing (exact)
            if (__builtin_amdgcn_ballot_w64(sel_pk_sub_sat(T1pk, gmin[i >> 2]) != 0u) == 0) continue;
            uint32_t t0 = sel_pk_sub_sat(T1pk, rr[i]), t1 = sel_pk_sub_sat(T1pk, rr[i + 1]);
            uint32_t t2 = sel_pk_sub_sat(T1pk, rr[i + 2]), t3 = sel_pk_sub_sat(T1pk, rr[i + 3]);
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
            zz[0] = sel_pk_add_sat(zz[0], t0); zz[1] = sel_pk_add_sat(zz[1], t1);
            zz[2] = sel_pk_add_sat(zz[2], t2); zz[3] = sel_pk_add_sat(zz[3], t3);
        }
        const uint32_t zp = sel_pk_add_sat(sel_pk_add_sat(zz[0], zz[1]), sel_pk_add_sat(zz[2], zz[3]));
        // both halves saturate at 65535, above anything the three-term sum of a parity can reach: the total stays >= the
        // expected total, with equality only if no half saturated and nothing outside {a-1, a, a+1} contributed
        z = (zp & 0xffffu) + (zp >> 16);
    }
    uint32_t k3[2] = {0xffffffffu, 0xffffffffu};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t ec = (uint32_t)(2 * q) | ((uint32_t)(2 * q + 1) << 8);
        const uint32_t klo = __builtin_amdgcn_perm(grp[q], ec, 0x0C050400u);
        const uint32_t khi = __builtin_amdgcn_perm(grp[q], ec, 0x0C070601u);
        k3[q & 1] = min(min(k3[q & 1], klo), khi);
    }
    const int a = 8 * gs + (int)(min(k3[0], k3[1]) & 0xffu);
    const bool has_n = a > 0, has_p = a + 1 < D;
    const unsigned short* sv = (const unsigned short*)scr;
    const int n_real = sv[has_n ? a - 1 : a];
    const int p_real = sv[has_p ? a + 1 : a];
    bool fail = tsum < g.tex;
    if (g.uniq > 0) {
        const auto term = [&](int v) -> uint32_t { return T1 > (uint32_t)v ? T1 - (uint32_t)v : 0u; };
        const uint32_t want = term(m1) + (has_n ? term(n_real) : 0u) + (has_p ? term(p_real) : 0u);
        fail |= z != want;
    }
    const int pp = has_p ? p_real : n_real;
    const int nn = has_n ? n_real : p_real;
    const int out = ((D - a - 1 + g.minD) * 256 + sel_subpixel(pp, nn, m1) + 15) >> 4;
    *minsad = m1;
    *rejected = fail;
    return fail ? g.filtered : out;
}

// ---- selection for pixels whose D values are spread over LPP lanes (k_search_ring), without transposing them --------
// The LPP lanes {p + h * 64/LPP, h = 0..LPP-1} of a wave each hold one SLICE of the D values -- indices [h D/LPP, (h+1) D/LPP)
// -- of LPP pixel rows: S[r] belongs to the row owned by the lane with h = r.  Each lane writes its slices into the owners'
// LDS records, runs level one (group minima and keys) and the uniqueness sums on what it holds -- in total the work of one
// whole row -- and only three values per row cross between the lanes (v_permlane32_swap / v_permlane16_swap): the partial
// key minima (reduced towards the owner), the owners' thresholds T+1 (broadcast back) and the partial sums (reduced).
// Everything after that is the owner's: winning group back from its record, argmin inside it, sad[a-1], sad[a+1], the
// tests, the sub-pixel step.  Results are those of select_disparity_lds on the gathered values.  (A first version kept the
// slices in registers and evaluated the sum identity of select_disparity_lds on them: 13 % more instructions and 10 more
// VGPRs at D = 64, four lanes, 0.968 vs 1.018 ms per 64 pairs.)

// xr_reduce: v[r] = this lane's partial value for row r; returns op over the pixel's LPP lanes of v[h] -- the full value of
// the row this lane owns.  xr_bcast: t = the owner's value; out[r] = the value of the owner of row r, in every lane.
template <int LPP, class Op>
__device__ __forceinline__ uint32_t xr_reduce(const uint32_t (&v)[LPP], Op op)
{
    static_assert(LPP == 2 || LPP == 4 || LPP == 8, "lane bits 5, 4 and 3");
    if constexpr (LPP == 2) {
        // lower lane: {own v0, partner's v0}; upper lane: {partner's v1, own v1}
        const auto s = __builtin_amdgcn_permlane32_swap(v[0], v[1], false, false);
        return op(s[0], s[1]);
    } else if constexpr (LPP == 4) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);   // h < 2: row 0 over {h, h+2}; h >= 2: row 2
        const auto s1 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);   //        row 1                       row 3
        const uint32_t a = op(s0[0], s0[1]), b = op(s1[0], s1[1]);
        const auto s2 = __builtin_amdgcn_permlane16_swap(a, b, false, false);         // even h: {own a, partner's a}; odd h: {partner's b, own b}
        return op(s2[0], s2[1]);
    } else {
        // h = lane >> 3: bit 2 = lane bit 5, bit 1 = lane bit 4, bit 0 = lane bit 3
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                                 // keeps rows (h & 4) + k, over {h, h ^ 4}
            const auto sk = __builtin_amdgcn_permlane32_swap(v[k], v[k + 4], false, false);
            w[k] = op(sk[0], sk[1]);
        }
        const auto t0 = __builtin_amdgcn_permlane16_swap(w[0], w[2], false, false);   // keeps rows (h & 6) + 0 and + 1, over four lanes
        const auto t1 = __builtin_amdgcn_permlane16_swap(w[1], w[3], false, false);
        const uint32_t x0 = op(t0[0], t0[1]), x1 = op(t1[0], t1[1]);
        const bool odd = (__lane_id() & 8) != 0;
        const uint32_t send = odd ? x0 : x1, own = odd ? x1 : x0;                     // the partner (lane ^ 8) owns the other row
        const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0x128, 0xf, 0xf, false);   // row_ror:8
        return op(own, recv);
    }
}
template <int LPP>
__device__ __forceinline__ void xr_bcast(uint32_t t, uint32_t (&out)[LPP])
{
    if constexpr (LPP == 2) {
        const auto s = __builtin_amdgcn_permlane32_swap(t, t, false, false);
        out[0] = s[0]; out[1] = s[1];
    } else if constexpr (LPP == 4) {
        const auto s = __builtin_amdgcn_permlane16_swap(t, t, false, false);          // {value of the even h of the pair, of the odd h}
        const auto se = __builtin_amdgcn_permlane32_swap(s[0], s[0], false, false);   // {h = 0, h = 2}
        const auto so = __builtin_amdgcn_permlane32_swap(s[1], s[1], false, false);   // {h = 1, h = 3}
        out[0] = se[0]; out[2] = se[1]; out[1] = so[0]; out[3] = so[1];
    } else {
        const bool odd = (__lane_id() & 8) != 0;
        const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x128, 0xf, 0xf, false);     // row_ror:8: lane ^ 8
        const uint32_t e = odd ? other : t, o = odd ? t : other;                      // rows (h & 6) + 0 and + 1
        const auto se = __builtin_amdgcn_permlane16_swap(e, e, false, false);         // rows (h & 4) + {0, 2}
        const auto so = __builtin_amdgcn_permlane16_swap(o, o, false, false);         //               {1, 3}
        const uint32_t q[4] = {se[0], so[0], se[1], so[1]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const auto sk = __builtin_amdgcn_permlane32_swap(q[k], q[k], false, false);
            out[k] = sk[0]; out[k + 4] = sk[1];
        }
    }
}

// GroupSelect is fed row by row (k_search_ring: the SADs of a row are dead as soon as its step is over, so a group of LPP
// rows holds LPP * NGH + LPP registers of state instead of LPP * D / (2 LPP)), with the uniqueness test split in two so that
// it needs no SAD value a second time:
//   (A) no group of eight other than the winner's and the one its neighbour a-1 / a+1 may lie in has a minimum <= T:
//       every lane counts the groups of its slices whose minimum is <= T, the counts are reduced to the owner, and the owner
//       expects 1 (+ 1 if the neighbour group's minimum is <= T);
//   (B) inside those one or two groups -- read back from the owner's record -- the identity
//       sum_e max(T+1 - sad[e], 0) == the same sum over {a-1, a, a+1}.
// Exact: {a-1, a, a+1} lies inside the two groups, so an index outside it with sad <= T is either in another group (A) or
// one of the up to sixteen values of (B).
template <int D, int LPP>
struct GroupSelect {
    static constexpr int NRL = D / (2 * LPP), NGH = NRL / 4;
    static_assert(D == 16 || D == 32 || D == 48 || D == 64 || D == 96 || D == 128, "record stride checked for these sizes only");
    static_assert(NRL % 4 == 0, "a lane's slice must be whole groups of eight disparities");
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    uint32_t mm[LPP][NGH];      // minimum of each group of this lane's slice, per row
    uint32_t kpart[LPP];        // min over the slice of (minimum << 8 | group), per row

    // row R of the group: sv = this lane's slice of its window sums; wr = where the slice goes in the record of the row's
    // owner; hofs = h * NGH, the number of the slice's first group
    template <int R>
    __device__ __forceinline__ void row(const uint32_t (&sv)[NRL], uint32_t* wr, uint32_t hofs)
    {
#pragma unroll
        for (int i = 0; i < NRL; i += 4) *(u4*)(wr + i) = u4{sv[i], sv[i + 1], sv[i + 2], sv[i + 3]};
        uint32_t k = 0xffffffffu;
#pragma unroll
        for (int gq = 0; gq < NGH; ++gq) {
            const uint32_t m = sel_pk_min(sel_pk_min(sv[4 * gq], sv[4 * gq + 1]), sel_pk_min(sv[4 * gq + 2], sv[4 * gq + 3]));
            const uint32_t m1 = min(m & 0xffffu, m >> 16);
            mm[R][gq] = m1;
            k = min(k, (m1 << 8) | (uint32_t)gq);
        }
        kpart[R] = k + hofs;
    }

    // after the LPP rows: the result for the row this lane owns (rec_own = its record)
    __device__ __forceinline__ int finish(int tsum, const BMGeom& g, const uint32_t* rec_own, int* minsad, bool* rejected)
    {
        const uint32_t kmin = xr_reduce<LPP>(kpart, [](uint32_t a, uint32_t b) { return min(a, b); });
        const int m1 = (int)(kmin >> 8);
        const int gs = (int)(kmin & 0xffu);
        const u4 grp = *(const u4*)(rec_own + 4 * gs);               // the four registers of the winning group
        uint32_t T1 = 0, cnt = 0;
        if (g.uniq > 0) {
            uint32_t T = (uint32_t)m1 + ((uint32_t)m1 * (uint32_t)g.uniq) / 100u;
            T = min(T, 32766u);
            T1 = T + 1u;
            uint32_t t1r[LPP], cpart[LPP];
            xr_bcast<LPP>(T1, t1r);
#pragma unroll
            for (int r = 0; r < LPP; ++r) {
                uint32_t c = 0;
#pragma unroll
                for (int gq = 0; gq < NGH; ++gq) c += (uint32_t)(mm[r][gq] < t1r[r]);
                cpart[r] = c;
            }
            cnt = xr_reduce<LPP>(cpart, [](uint32_t a, uint32_t b) { return a + b; });
        }
        uint32_t k3[2] = {0xffffffffu, 0xffffffffu};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t ec = (uint32_t)(2 * q) | ((uint32_t)(2 * q + 1) << 8);
            const uint32_t klo = __builtin_amdgcn_perm(grp[q], ec, 0x0C050400u);
            const uint32_t khi = __builtin_amdgcn_perm(grp[q], ec, 0x0C070601u);
            k3[q & 1] = min(min(k3[q & 1], klo), khi);
        }
        const int e = (int)(min(k3[0], k3[1]) & 0xffu);
        const int a = 8 * gs + e;
        const bool has_n = a > 0, has_p = a + 1 < D;
        const unsigned short* sv = (const unsigned short*)rec_own;
        const int n_real = sv[has_n ? a - 1 : a];
        const int p_real = sv[has_p ? a + 1 : a];
        bool fail = tsum < g.tex;
        if (g.uniq > 0) {
            // the group a-1 or a+1 falls into, if that is not the winner's
            const int nbq = (e == 0 && has_n) ? gs - 1 : (e == 7 && has_p) ? gs + 1 : gs;
            const bool has_nb = nbq != gs;
            const u4 nbg = *(const u4*)(rec_own + 4 * nbq);
            const uint32_t nbm = sel_pk_min(sel_pk_min(nbg[0], nbg[1]), sel_pk_min(nbg[2], nbg[3]));
            const uint32_t nbmin = min(nbm & 0xffffu, nbm >> 16);
            const uint32_t expect = 1u + (uint32_t)(has_nb && nbmin < T1);
            const uint32_t T1pk = T1 * 0x00010001u, nbmask = has_nb ? 0xffffffffu : 0u;
            uint32_t zz[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                zz[q] = sel_pk_add_sat(sel_pk_sub_sat(T1pk, grp[q]), sel_pk_sub_sat(T1pk, nbg[q]) & nbmask);
            const uint32_t zp = sel_pk_add_sat(sel_pk_add_sat(zz[0], zz[1]), sel_pk_add_sat(zz[2], zz[3]));
            // a 16-bit half saturates at 65535, more than the (at most two) terms of {a-1, a, a+1} it can hold add up to:
            // the total stays >= the expected total, with equality only if nothing saturated and nothing else contributed
            const uint32_t z = (zp & 0xffffu) + (zp >> 16);
            const auto term = [&](int v) -> uint32_t { return T1 > (uint32_t)v ? T1 - (uint32_t)v : 0u; };
            const uint32_t want = term(m1) + (has_n ? term(n_real) : 0u) + (has_p ? term(p_real) : 0u);
            fail |= (z != want) | (cnt != expect);
        }
        const int pp = has_p ? p_real : n_real;
        const int nn = has_n ? n_real : p_real;
        const int out = ((D - a - 1 + g.minD) * 256 + sel_subpixel(pp, nn, m1) + 15) >> 4;
        *minsad = m1;
        *rejected = fail;
        return fail ? g.filtered : out;
    }
};

// GroupSelectRec: GroupSelect with the group minima kept in the OWNER's record as well (D <= 64: the D/8 packed u16 minima
// are D/16 <= 4 dwords and live in the four padding dwords behind the D/2 SAD dwords).  A lane's part of a row shrinks to
// the packed minima of its groups (3 v_pk_min + one SDWA minimum per group, written with ds_write_b16) and NOTHING crosses
// between the lanes any more: the owner reads the minima of all D/8 groups back with one 16-byte read and builds the keys
// (v_perm, v_min3: 12 instructions at D = 64 against 16 per-row key instructions + an 8-instruction lane reduction), and
// evaluates test (A) itself as a second sum identity over those minima (8 packed instructions against a 9-instruction
// broadcast of T+1, 16 compares in the other lanes and a 10-instruction reduction).  Per 64 pixel-rows at D = 64, four lanes per pixel: about 45 VALU instructions
// and 12 registers of per-group state (mm, kpart) less.  Same results: the tests (A) and (B) of GroupSelect, same values.
template <int D, int LPP, bool B64 = false>
struct GroupSelectRec {
    static constexpr int NRL = D / (2 * LPP), NGH = NRL / 4, NG = D / 8, NGD = NG / 2, NGQ = (NGD + 3) / 4;
    static_assert(D % 16 == 0 && D / 2 + (B64 ? 2 * ((NGD + 1) / 2) : 4 * NGQ) <= SelRecord<D, true, B64>::DWORDS, "the minima live behind the SADs of the record");
    static_assert(NRL % 4 == 0, "a lane's slice must be whole groups of eight disparities");
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    // four dwords of a record at a 16-byte (8-byte if B64) aligned offset
    static __device__ __forceinline__ u4 ld4(const uint32_t* q)
    {
        if constexpr (B64) { const u2 a = *(const u2*)q, b = *(const u2*)(q + 2); return u4{a[0], a[1], b[0], b[1]}; }
        else return *(const u4*)q;
    }
    static __device__ __forceinline__ void st4(uint32_t* q, u4 v)
    {
        if constexpr (B64) { *(u2*)q = u2{v[0], v[1]}; *(u2*)(q + 2) = u2{v[2], v[3]}; }
        else *(u4*)q = v;
    }

    // once per kernel: unused minima slots of this lane's record never win and never count
    static __device__ __forceinline__ void init(uint32_t* rec_own)
    {
        if constexpr (B64) {
#pragma unroll
            for (int i = 0; i < (NGD + 1) / 2; ++i) *(u2*)(rec_own + D / 2 + 2 * i) = u2{0xffffffffu, 0xffffffffu};
        } else {
#pragma unroll
            for (int i = 0; i < NGQ; ++i) *(u4*)(rec_own + D / 2 + 4 * i) = u4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        }
    }

    // row R of the group: sv = this lane's slice of its window sums; wr = where the slice goes in the record of the row's
    // owner; wm = where the minima of the slice's groups go there (u16 each)
    template <int R>
    __device__ __forceinline__ void row(const uint32_t (&sv)[NRL], uint32_t* wr, unsigned short* wm)
    {
#pragma unroll
        for (int i = 0; i < NRL; i += 4) st4(wr + i, u4{sv[i], sv[i + 1], sv[i + 2], sv[i + 3]});
#pragma unroll
        for (int gq = 0; gq < NGH; ++gq) {
            const uint32_t m = sel_pk_min(sel_pk_min(sv[4 * gq], sv[4 * gq + 1]), sel_pk_min(sv[4 * gq + 2], sv[4 * gq + 3]));
            wm[gq] = (unsigned short)min(m & 0xffffu, m >> 16);
        }
    }

    // after the LPP rows: the result for the row this lane owns (rec_own = its record)
    __device__ __forceinline__ int finish(int tsum, const BMGeom& g, const uint32_t* rec_own, int* minsad, bool* rejected)
    {
        uint32_t mn[4 * NGQ];                                        // minima of the groups 2q (low half) and 2q + 1 (high half)
#pragma unroll
        for (int i = 0; i < NGQ; ++i) {
            // (B64 with an odd number of minima pairs: the last two dwords lie past the record's end -- the neighbour's
            //  first SADs or, behind the last record, nothing that is read: they are never used, NGD counts the real ones)
            u4 v;
            if constexpr (B64 && (NGD % 4 == 1 || NGD % 4 == 2)) {
                if (i == NGQ - 1) { const u2 a = *(const u2*)(rec_own + D / 2 + 4 * i); v = u4{a[0], a[1], 0xffffffffu, 0xffffffffu}; }
                else v = ld4(rec_own + D / 2 + 4 * i);
            } else v = ld4(rec_own + D / 2 + 4 * i);
            mn[4 * i] = v[0]; mn[4 * i + 1] = v[1]; mn[4 * i + 2] = v[2]; mn[4 * i + 3] = v[3];
        }
        uint32_t kk[2] = {0xffffffffu, 0xffffffffu};
#pragma unroll
        for (int q = 0; q < NGD; ++q) {
            const uint32_t gc = (uint32_t)(2 * q) | ((uint32_t)(2 * q + 1) << 8);
            const uint32_t klo = __builtin_amdgcn_perm(mn[q], gc, 0x0C050400u);
            const uint32_t khi = __builtin_amdgcn_perm(mn[q], gc, 0x0C070601u);
            kk[q & 1] = min(min(kk[q & 1], klo), khi);
        }
        const uint32_t kmin = min(kk[0], kk[1]);
        const int m1 = (int)(kmin >> 8);
        const int gs = (int)(kmin & 0xffu);
        const u4 grp = ld4(rec_own + 4 * gs);                        // the four registers of the winning group
        // (A) as a sum identity too: sum over ALL groups of max(T+1 - group minimum, 0) must equal what the winner's group and
        // the neighbour's group contribute -- any other group with a minimum <= T adds a positive term.  Groups 2q / 2q+1
        // sit in the low / high halves, so the (adjacent) winner and neighbour groups fall into different halves: a half that
        // saturates (65535) holds more than its one expected term (<= 32767) -- the total stays above the expected total.
        uint32_t T1 = 0, zg = 0;
        if (g.uniq > 0) {
            uint32_t T = (uint32_t)m1 + ((uint32_t)m1 * (uint32_t)g.uniq) / 100u;
            T = min(T, 32766u);
            T1 = T + 1u;
            const uint32_t T1pk = T1 * 0x00010001u;
            uint32_t z2[2] = {0, 0};
#pragma unroll
            for (int q = 0; q < NGD; ++q) z2[q & 1] = sel_pk_add_sat(z2[q & 1], sel_pk_sub_sat(T1pk, mn[q]));   // unused halves (0xffff) add 0
            const uint32_t zq = NGD > 1 ? sel_pk_add_sat(z2[0], z2[1]) : z2[0];
            zg = (zq & 0xffffu) + (zq >> 16);
        }
        uint32_t k3[2] = {0xffffffffu, 0xffffffffu};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t ec = (uint32_t)(2 * q) | ((uint32_t)(2 * q + 1) << 8);
            const uint32_t klo = __builtin_amdgcn_perm(grp[q], ec, 0x0C050400u);
            const uint32_t khi = __builtin_amdgcn_perm(grp[q], ec, 0x0C070601u);
            k3[q & 1] = min(min(k3[q & 1], klo), khi);
        }
        const int e = (int)(min(k3[0], k3[1]) & 0xffu);
        const int a = 8 * gs + e;
        const bool has_n = a > 0, has_p = a + 1 < D;
        const unsigned short* sv = (const unsigned short*)rec_own;
        const int n_real = sv[max(a - 1, 0)];                        // (sad[a] itself where the neighbour does not exist)
        const int p_real = sv[min(a + 1, D - 1)];
        bool fail = tsum < g.tex;
        if (g.uniq > 0) {
            // the group a-1 or a+1 falls into, if that is not the winner's
            const int nbq = (e == 0 && has_n) ? gs - 1 : (e == 7 && has_p) ? gs + 1 : gs;
            const bool has_nb = nbq != gs;
            const u4 nbg = ld4(rec_own + 4 * nbq);
            const uint32_t nbmin = sv[D + nbq];                      // its minimum, from the minima behind the SADs
            const uint32_t T1pk = T1 * 0x00010001u, nbmask = has_nb ? 0xffffffffu : 0u;
            uint32_t zz[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                zz[q] = sel_pk_add_sat(sel_pk_sub_sat(T1pk, grp[q]), sel_pk_sub_sat(T1pk, nbg[q]) & nbmask);
            const uint32_t zp = sel_pk_add_sat(sel_pk_add_sat(zz[0], zz[1]), sel_pk_add_sat(zz[2], zz[3]));
            // a 16-bit half saturates at 65535, more than the (at most two) terms of {a-1, a, a+1} it can hold add up to:
            // the total stays >= the expected total, with equality only if nothing saturated and nothing else contributed
            const uint32_t z = (zp & 0xffffu) + (zp >> 16);
            const auto term = [&](int v) -> uint32_t { return T1 > (uint32_t)v ? T1 - (uint32_t)v : 0u; };
            const uint32_t tm1 = term(m1);
            const uint32_t want = tm1 + (has_n ? term(n_real) : 0u) + (has_p ? term(p_real) : 0u);     // (B): inside the one or two groups
            const uint32_t wantg = tm1 + (has_nb ? term((int)nbmin) : 0u);                             // (A): over the group minima
            fail |= (z + zg) != (want + wantg);                      // z >= want and zg >= wantg: equal sums <=> both equal
        }
        // sad[-1] := sad[1], sad[D] := sad[D-2] (A.3b): the missing neighbour takes the other one's value
        const int pp = has_p ? p_real : n_real;
        const int nn = has_n ? n_real : p_real;
        const int out = ((D - a - 1 + g.minD) * 256 + sel_subpixel(pp, nn, m1) + 15) >> 4;
        *minsad = m1;
        *rejected = fail;
        return fail ? g.filtered : out;
    }
};

}  // namespace rtdm
