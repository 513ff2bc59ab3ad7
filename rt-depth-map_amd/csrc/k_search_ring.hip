// k_search_ring.hip -- K2, ring variant: the SAD window search with NO leaving-row recomputation.
//
// k_search_fast keeps the w-row window sums of a pixel in registers and pays for every row twice: once when it enters
// the window and once, w rows later, when it is recomputed from an LDS ring and subtracted.  Here a pixel's registers
// hold the running PREFIX sums over the rows of its strip instead,
//       P(t)[d] = sum_{rows 0..t} H(row)[d]            H = the horizontal w-byte SAD of one row
//       S(t)[d] = P(t)[d] - P(t-w)[d]                   the w x w window sum
// in a ring of w+1 register slots: v_qsad_pk_u16_u8 takes P(t-1) as its accumulator operand and writes P(t) into the
// slot of the dead P(t-w-1), so a row costs its quad-SADs once plus one subtraction per two disparities.  Ring slots
// are compile-time registers: one trip of the row loop is a whole number of rounds of the ring, unrolled, every row step a
// copy with its own register operands (a switch over the slot inside a rolled loop made the register allocator copy and spill
// the ring at every merge).
//
// A pixel's ring is (w+1) * D/2 registers (320 at D = 64, w = 9), so the D disparities of a pixel are split over LPP lanes
// of a wave -- p + h * 64/LPP, h = 0..LPP-1, each with a slice of D/LPP disparities -- and rows are processed in GROUPS of LPP:
// the lane with h = k owns row k of the group and produces its output pixel.
//   LPP = 2 (D <= 48; D = 64 as an A/B form): ring 160 registers at D = 64, w = 9 => two waves per SIMD
//   LPP = 4 (D = 64, 96): ring 80 registers at (64, 9), 148 VGPRs => three waves (four for w <= 7; two for D = 96)
//   LPP = 8 (D = 128): ring 96 registers at w = 11; two waves (the selection records, 17 KB per wave, allow no more)
// Selection wants all D values of a pixel in one place.  Two forms (rtdm_select.h):
//   * transposing (LPP = 2, D <= 32): one v_permlane32_swap per register hands the lower lane both halves of row t and the
//     upper lane both halves of row t+1; each lane runs the single-lane selection select_disparity_lds;
//   * GroupSelect (everything else): nothing is transposed.  Each lane writes its slice of a row into the owner's LDS
//     record and folds it into per-group minima as soon as the row's step is over (the SADs are dead after that), and per
//     row only three words cross between the lanes: partial key minima, thresholds, partial counts.
//
// Mapping: workgroup = 4 * 64/LPP output columns x `rs` rows of one frame; wave = byte phase phi, lane (p, h) = column
// x_tile + phi + 4 p.  The four waves never synchronise: each stages ITS byte-shifted copy of the entering rows (64 or 128
// dwords, padded) into its own LDS slice, two rows ahead of their use, and the LDS queue of a wave is in order.  The
// texture sum is a prefix ring too (one dword per pixel and slot, in LDS).  Only columns whose window needs no border
// clamping are handled here; the border columns go to k_search_border.
// Semantics: SURVEY.md Appendix A.3b; oracle: oracle/bm_oracle.c.
//
// Measured instruction costs on gfx950 that shaped this (tools/ubench_valu.hip, SIMD cycles per wave-instruction):
// v_qsad/v_mqsad_pk_u16_u8 16.4; v_add/sub/and/xor/mov 2.6; everything else used here (packed u16 ops, v_perm,
// v_cndmask, v_min3, v_cmp, shifts, v_sad_u8) 4.2-4.7; v_permlane32_swap 8.1.
#include "rtdm_select.h"
#include "rtdm_border.h"
#include "rtdm_border2.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

#ifndef RING_ABL          // timing-only ablations (tools/ring_ablate.sh): 1 no selection, 2 no stores, 3 no global loads,
#define RING_ABL 0        // 4 no quad-SADs, 5 no swaps, 6 no swaps + no selection.  Outputs are wrong for n != 0.
#endif
#ifndef RING_LDS_SELECT   // 1: select_disparity_lds (fetches through LDS), 0: select_disparity (v_cndmask tree)
#define RING_LDS_SELECT 1
#endif
#ifndef RING_MINREC       // 0: GroupSelect (lane reductions) also where GroupSelectRec (minima in the owner's record) applies: A/B
#define RING_MINREC 1
#endif
#ifndef RING_W3_LIMIT      // two-lane forms: most ring registers a three-wave form may hold.  (32, 13) sits AT 112 and spills ten dwords
#define RING_W3_LIMIT 112  // at 168 VGPRs -- and is still 9 % faster than as a two-wave form without (111): profiles/r03_ring_spills.txt
#endif
#ifndef RING_STATIC_SLOTS // 1: four staging slots where a trip of the row loop is a multiple of four rows: the slot of every unrolled row step
#define RING_STATIC_SLOTS 1 // is a compile-time constant and all LDS addresses are base + immediate (3-4 VALU per row less)
#endif
#ifndef RING_DIRECT_LOADS // 1: a staged dword is ONE (unaligned) global load at uniform row base + lane offset, stored as it is; 0: two
#define RING_DIRECT_LOADS 1 // aligned dwords per lane and row, shifted into place by v_alignbyte
#endif
#ifndef RING_B64REC       // 1: D = 64 writes / reads its selection records in 8-byte pieces (34-dword stride: four workgroups per CU at (64, 9, 4))
#define RING_B64REC 1
#endif
#ifndef RING_MINREC96     // 1: GroupSelectRec at D = 96 too (records of 240 instead of 208 bytes)
#define RING_MINREC96 1
#endif
#ifndef RING_SPLIT_SELECT // 1: GroupSelect also for the two-lane configurations that default to the transposing selection
#define RING_SPLIT_SELECT 0 // (D = 16, D = 32 except w = 9: measured 0-8 % slower there, profiles/r02_ring_split_select_ab.txt;
#endif                      //  tools/ring_split_ab.sh)

#ifdef RING_STAMPS        // diagnostic build only (tools/ring_stamps.sh): where a row pair spends its cycles
__device__ unsigned long long ring_stamps[8];
#define RING_STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                           stamp_sum[i] += t_ - stamp_last; stamp_last = t_; } while (0)
extern "C" void rtdm_debug_ring_stamps(unsigned long long* out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ring_stamps), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(ring_stamps), z, sizeof z); }
}
#else
#define RING_STAMP(i) do { } while (0)
#endif

namespace rtdm {

struct RingGeom {
    int x0, nx;          // output-column range [x0, x0+nx) handled by this kernel
    int rs;              // output rows per workgroup
    uint32_t lastmask;   // byte mask of the last (partial) window piece
    int tiles, strips;   // column tiles x row strips per frame
    unsigned nitems;     // workgroups over the whole batch
    unsigned chunk;      // ceil(nitems / 8): consecutive items one XCD works through (0: no remapping)
    uint32_t rdelta;     // the right prefiltered plane lies this many bytes behind the left one (direct loads)
    int rebase;          // > 0: every `rebase` trips of the row loop the ring is rebased (P -= the oldest live prefix sum), so a
                         // strip may be as long as the frame; 0: strips are short enough for 16-bit prefix sums (ring_rows_cap)
    unsigned nborder;    // border-column workgroups in FRONT of the tile workgroups (single frames and small batches, where a
    int bgx, bgy;        // second launch or a side stream costs more than it hides); 0: none
    int b2;              // 1: they run border2_body (rtdm_border2.h: rows in the lanes), 0: border_body (rtdm_border.h)
};

// LPP = lanes per pixel: the D disparities of a pixel are split over LPP lanes of a wave (p + h * 64/LPP, h = 0..LPP-1), and
// rows are processed in groups of LPP, one owner lane per row for the selection.
// RPG = rows per group: the lanes with h < RPG own the group's rows; RPG < LPP (GroupSelectRec only: its lanes exchange
// nothing) halves / quarters the number of selection records a wave needs -- what lets D = 192 and D = 256 (records of 432 /
// 592 bytes) keep two workgroups per CU -- at the price of a selection that runs with RPG / LPP of its lanes.
#ifndef RING_RPG96        // A/B (tools/ring_dev.py with `make variant`): rows per group at D = 96, 128, 256
#define RING_RPG96 4
#endif
#ifndef RING_RPG128       // (measured: four rows per group + GroupSelectRec 6-8 % faster than eight + GroupSelect; D = 96 with two
#define RING_RPG128 4     //  rows and D = 256 with eight are slower than four: profiles/r03_ring_rows_per_group_ab.txt)
#endif
#ifndef RING_RPG256
#define RING_RPG256 4
#endif
__host__ __device__ constexpr int ring_rpg(int D, int LPP)
{ return D == 256 ? RING_RPG256 : D >= 192 ? 4 : D == 128 ? RING_RPG128 : D == 96 ? RING_RPG96 : LPP; }

template <int D, int WS, int LPP = 2>
struct RingCfg {
    static constexpr int RPG = ring_rpg(D, LPP);
    // selection without transposing the lanes' slices (GroupSelect / GroupSelectRec): always for four and more lanes per pixel;
    // with two lanes where it measured faster than transposing (tools/ring_split_ab.sh, profiles/r02_ring_split_select_ab.txt)
    // (round 3 re-measured with GroupSelectRec: D = 32, w = 5 joins (+10 %); the other D = 16 / 32 forms stay within +-1-5 %
    //  of transposing, profiles/r03_ring_split_select_ab.txt)
    static constexpr bool SPLIT = LPP != 2 || D >= 48 || (D == 32 && (WS == 9 || WS == 5)) || (RING_SPLIT_SELECT && RING_LDS_SELECT);
    // the group minima live in the owner's record too and nothing crosses between the lanes (GroupSelectRec).  Not at D = 128
    // with eight rows per group: 64 records of 304 bytes would leave one workgroup per CU.
    static constexpr bool MINREC = SPLIT && RING_MINREC && !(D == 128 && RPG == 8) && !(D == 96 && !RING_MINREC96);
    static_assert(RPG == LPP || MINREC, "GroupSelect's lane reductions need one owner per lane of a pixel");
    // D = 64: records in 8-byte pieces at a 34-dword stride (rtdm_select.h)
    static constexpr bool B64 = MINREC && D == 64 && RING_B64REC;
    using Rec = SelRecord<D, MINREC || !SPLIT, B64>;
    static constexpr int NP = (WS + 3) / 4;        // 4-byte pieces of a window row
    static constexpr int W1 = WS + 1;              // ring slots
    static constexpr int PPW = 64 / LPP;           // pixels (columns) per wave
    static constexpr int DL = D / LPP;             // disparities per lane
    static constexpr int NGL = DL / 4;             // quad-SAD groups per lane
    static constexpr int NRL = DL / 2;             // packed u16x2 registers per lane and slot
    static constexpr int NW = NGL + NP - 1;        // distinct 8-byte right windows per lane and row
    static constexpr int LWD = PPW + NP;           // dwords of the wave's copy of the left row
    static constexpr int RWD = PPW + D / 4 + NP;   //                                 right row
    static constexpr int ITEMS = (LWD + RWD + 63) / 64;
    static constexpr int SLOT = ITEMS * 64;        // padded: every lane stores every item, no exec masking
    // rows per trip of the unrolled row loop: whole rounds of the ring AND whole groups
    static constexpr int TRIP = (W1 % RPG == 0) ? W1 : (2 * W1 % RPG == 0) ? 2 * W1 : 4 * W1;
    static_assert(W1 % 2 == 0 && TRIP % RPG == 0 && TRIP % W1 == 0, "block sizes are odd");
    // staged rows in flight per wave: three (row t is read while t+1 waits and t+2 arrives), or four where that makes the
    // slot of every unrolled step a constant (one item per row only: ds_read2's 8-bit dword offsets must reach the slots)
    static constexpr bool STATIC_SLOTS = RING_STATIC_SLOTS && ITEMS == 1 && TRIP % 4 == 0;
    static constexpr int NSLOT = STATIC_SLOTS ? 4 : 3;
    // dwords per wave: staged rows + the texture prefix ring [W1][PPW] + the window texture sums of a group's rows [RPG][PPW] (u16)
    static constexpr int STG = (NSLOT * SLOT + (W1 + RPG) * PPW / 2 + 3) & ~3;
    static constexpr int WAVE_LDS = STG + PPW * RPG * Rec::DWORDS;   // + the selection's records, one per owner lane (rtdm_select.h)
    // waves per SIMD the register budget is set for: the ring takes W1 * NRL registers, the rest of the kernel about 50
    static constexpr int RING_REGS = W1 * NRL;
    // (tighter bounds spill; eight lanes per pixel = D = 128: the selection records, 17 KB per wave, allow two workgroups per CU)
    static constexpr int WAVES = LPP >= 8 ? 2 : LPP == 4 ? (RING_REGS <= (MINREC ? 80 : 64) ? 4 : (RING_REGS <= 96 && NRL <= 8) ? 3 : 2) : (RING_REGS <= 72 && NRL <= 8) ? 4 : RING_REGS <= RING_W3_LIMIT ? 3 : 2;
    static constexpr int TILE = 4 * PPW;           // four byte phases
    // border-column workgroups may ride in this kernel's grid (small launches): border2_body in every form (it needs no LDS
    // and 52-120 VGPRs for D <= 64); border_body (configurations border2 does not cover) not in the four-wave forms, whose
    // 128 registers its code would overflow
    static constexpr bool FUSE_BORDER = WAVES < 4;
    static constexpr bool FUSE_BORDER2 = D <= 192;  // (border2_body<256, 4> needs 261 VGPRs: it stays a launch of its own)
    // LDS read addresses of a row: three registers that advance (3 VALU per row) or recomputed from the slot index (6 VALU,
    // no registers held) -- the latter for the two-lane configurations that sit at their three-wave register limit
    static constexpr bool ROW_PTRS = !(LPP == 2 && RING_REGS > 88 && WAVES == 3);
};

// The form with the border-column workgroups in its grid (BORDER: single frames, batches < 16 -- a handful of scheduling rounds,
// where the occupancy bound buys nothing) is compiled for one wave per SIMD less: border2_body on top of a form that sits at
// its register limit spilled two dwords ((64, 9, 4) at 128 VGPRs, (192, 15, 8) at 256).
template <int D, int WS, int LPP, bool BORDER>
#ifndef RING_BORDER_RELAX   // A/B: 0 = the same bound for both forms (round 3 until then)
#define RING_BORDER_RELAX 1
#endif
struct RingBounds { static constexpr int WAVES = RING_BORDER_RELAX && BORDER && RingCfg<D, WS, LPP>::WAVES > 1 ? RingCfg<D, WS, LPP>::WAVES - 1 : RingCfg<D, WS, LPP>::WAVES; };

template <int D, int WS, int LPP = 2>
struct RingState { uint64_t P[RingCfg<D, WS, LPP>::W1][RingCfg<D, WS, LPP>::NGL]; };   // only ever indexed with constants: registers

// The registers a lane needs of one staged row: its NW right windows (8 bytes each) and its NP left pieces.
template <int D, int WS, int LPP = 2>
struct RowRegs { uint64_t win[RingCfg<D, WS, LPP>::NW]; uint32_t l[RingCfg<D, WS, LPP>::NP]; };

typedef __attribute__((address_space(3))) const uint32_t* lds_cptr;    // an LDS address held as what it is: 32 bits
__device__ __forceinline__ uint32_t lds_addr(const uint32_t* q) { return (uint32_t)(uintptr_t)(lds_cptr)q; }
__device__ __forceinline__ lds_cptr lds_at(uint32_t a) { return (lds_cptr)(uintptr_t)a; }

// lp / rp: this lane's left pieces / right windows of the staged row in LDS; rpo = rp + 1, as a pointer of its own.
template <int D, int WS, int LPP, typename PTR>
__device__ __forceinline__ void ring_load_row(PTR lp, PTR rp, PTR rpo, RowRegs<D, WS, LPP>& rw)
{
    using C = RingCfg<D, WS, LPP>;
    // 64-bit operands want even-aligned VGPR pairs: windows at odd dword offsets are loaded through a second pointer
    // (laundered by the caller), so that they get their own ds_read2_b32 instead of v_mov rebuilds.
    // Issue order = order of first use (the LDS returns data in order): the left pieces, then the windows by index.
#pragma unroll
    for (int k = 0; k < C::NP; ++k) rw.l[k] = lp[k];
#pragma unroll
    for (int j = 0; j < C::NW; ++j) {
        const PTR wp = (j & 1) ? rpo + (j - 1) : rp + j;
        rw.win[j] = (uint64_t)wp[0] | ((uint64_t)wp[1] << 32);
    }
}

// One row step on ring slot K: P[K] = P[K-1] + H(row), S = P[K] - P[K+1] (the slot that holds P(t-w));
// tnew += the row's texture term sum |L - cap|.
template <int D, int WS, int LPP, int K, int NRLc>
__device__ __forceinline__ void ring_step(RingState<D, WS, LPP>& st, const RowRegs<D, WS, LPP>& rw, uint32_t lastmask, uint32_t capb,
                                          uint32_t& tnew, uint32_t (&S)[NRLc])
{
    using C = RingCfg<D, WS, LPP>;
    auto& P = st.P;
    constexpr int NP = C::NP, NGL = C::NGL, W1 = C::W1;
    constexpr int KP = (K + W1 - 1) % W1, KO = (K + 1) % W1;
    uint32_t l[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) l[k] = rw.l[k];
    l[NP - 1] &= lastmask;
#pragma unroll
    for (int k = 0; k < NP - 1; ++k) tnew = __builtin_amdgcn_sad_u8(l[k], capb, tnew);
    tnew = __builtin_amdgcn_msad_u8(capb, l[NP - 1], tnew);   // zero bytes of the reference are skipped
#pragma unroll
    for (int j = 0; j < C::NW; ++j) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int gi = j - k;
            if (gi >= 0 && gi < NGL) {
                const uint64_t acc = k == 0 ? P[KP][gi] : P[K][gi];
#if RING_ABL == 4
                P[K][gi] = acc + (rw.win[j] ^ l[k]);
#else
                if (k == NP - 1) P[K][gi] = __builtin_amdgcn_mqsad_pk_u16_u8(rw.win[j], l[k], acc);
                else             P[K][gi] = __builtin_amdgcn_qsad_pk_u16_u8(rw.win[j], l[k], acc);
#endif
            }
        }
    }
#pragma unroll
    for (int gi = 0; gi < NGL; ++gi) {
        // plain 32-bit subtractions of the packed pairs (v_sub_u32 issues at 1.6x the rate of v_pk_sub_u16): a strip is
        // short enough that no prefix sum leaves 16 bits (ring_rows_cap), so each half of P(t) is >= that half of P(t-w)
        // and no borrow crosses from the low half into the high one
        S[2 * gi] = (uint32_t)P[K][gi] - (uint32_t)P[KO][gi];
        S[2 * gi + 1] = (uint32_t)(P[K][gi] >> 32) - (uint32_t)(P[KO][gi] >> 32);
    }
}

// Orders everything that produces v[] before whatever follows (no instruction is emitted).
template <int N>
__device__ __forceinline__ void pin(uint32_t (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// compile-time loops: over the row groups of one trip of the row loop (f returns false to stop) and over a group's rows
template <typename F, int... U>
__device__ __forceinline__ void ring_for_groups(std::integer_sequence<int, U...>, F&& f)
{ (void)(f(std::integral_constant<int, U>{}) && ...); }
template <typename F, int... R>
__device__ __forceinline__ void ring_for_rows(std::integer_sequence<int, R...>, F&& f)
{ (f(std::integral_constant<int, R>{}), ...); }

// FUSE: the grid starts with rg.nborder border-column workgroups (an instantiation of its own: with the border body inside,
// the tile loop of the batch form came out 1.6 % slower)
template <int D, int WS, int LPP, bool FUSE>
__global__ __launch_bounds__(256, (RingBounds<D, WS, LPP, FUSE>::WAVES)) void k_search_ring(Plane8 Lp, Plane8 Rp, Plane16W disp, uint16_t* cost, BMGeom g, RingGeom rg,
                                                                                   BorderGeom bg, Border2Geom b2g)
{
    using C = RingCfg<D, WS, LPP>;
    constexpr int NGL = C::NGL, NRL = C::NRL, W1 = C::W1, LWD = C::LWD, SLOT = C::SLOT, ITEMS = C::ITEMS, PPW = C::PPW, RPG = C::RPG;
    constexpr bool SPLIT = C::SPLIT, MINREC = C::MINREC;
    constexpr int RECD = C::Rec::DWORDS;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

    // Workgroup ids go round-robin over the 8 XCDs, each with its own L2.  Item = (frame, strip, tile), tile fastest:
    // XCD k takes the items [k chunk, (k+1) chunk) in order, so the tiles of a strip -- which share their D + w column halo
    // -- and the strips of a frame -- which share w-1 rows -- meet in ONE L2 instead of being fetched into eight.
    unsigned fi = blockIdx.x;
    if constexpr (FUSE) {
        if (fi < rg.nborder) {
            // border columns: a latency-bound walk, dispatched first so that it runs under the tiles
            const unsigned per = (unsigned)(rg.bgx * rg.bgy), fr = fi / per, id = fi - fr * per;
            if (C::FUSE_BORDER2 && rg.b2) {
                const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
                border2_body<D, C::NP>(Lp, Rp, disp, cost, g, b2g, (int)(id % rg.bgx) * 4 + wave, (int)(id / rg.bgx), (int)fr);
            } else if constexpr (C::FUSE_BORDER) {
                border_body<(D + 63) / 64>((unsigned char*)lds, Lp, Rp, disp, cost, g, bg, (int)(id % rg.bgx), (int)(id / rg.bgx), (int)fr);
            }
            return;
        }
        fi -= rg.nborder;
    }
    if (rg.chunk) fi = (fi & 7u) * rg.chunk + (fi >> 3);
    if (fi >= rg.nitems) return;
    const int b_tile = (int)(fi % rg.tiles), b_strip = (int)((fi / rg.tiles) % rg.strips), f = (int)(fi / (rg.tiles * rg.strips));

    const int tid = threadIdx.x, lane = tid & 63;
    const int phi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & (PPW - 1), h = lane / PPW;
    const int x_tile = rg.x0 + b_tile * C::TILE;
    const int x = x_tile + phi + 4 * p;
    const bool active = x < rg.x0 + rg.nx;
    const int ys0 = g.vy0 + b_strip * rg.rs;
    const int ys1 = min(ys0 + rg.rs, g.vy1);
    if (ys0 >= ys1) return;
    const int r = g.r;
    const uint8_t* Lb = Lp.base + (size_t)f * Lp.frame;
#if !RING_DIRECT_LOADS
    const uint8_t* Rb = Rp.base + (size_t)f * Rp.frame;
#endif
    const int Lbase = g.lofs + x_tile - r + phi;   // image column of byte 0 of this wave's copy (left)
    const int Rbase = g.rofs + x_tile - r + phi;   //                                            (right)
    const uint32_t capb = (uint32_t)(g.cap + PREFILTER_BIAS) * 0x01010101u;

    uint32_t* stg = lds + phi * C::WAVE_LDS;       // this wave's slice: nothing below is shared between waves
    // texture prefix ring [W1][PPW], 16 bits each (the lanes of a pixel write the same value; window sums are < 2^16 and
    // differences of prefix sums are taken modulo 2^16, so the prefix sums themselves may wrap)
    unsigned short* ptr = (unsigned short*)(stg + C::NSLOT * SLOT);
    // the window texture sum of row R of a group goes to ptw[R][p] (every lane of the pixel writes the same 16 bits) and the
    // row's owner reads its own at the end of the group: no per-row select in registers
    unsigned short* ptw = ptr + W1 * PPW;
    const bool owner = RPG == LPP || h < RPG;       // this lane owns row h of every group
    const unsigned short* ptw_own = ptw + (owner ? h : 0) * PPW + p;
    uint32_t* scr = stg + C::STG + (owner ? lane : 0) * RECD;        // this lane's selection record (lanes that own nothing: never used)
    uint32_t* scr_w = stg + C::STG + p * RECD + h * NRL;             // SPLIT: where this lane's slice of the group's first row goes
    using GSel = std::conditional_t<MINREC, GroupSelectRec<D, LPP, C::B64>, GroupSelect<D, LPP>>;
    unsigned short* scr_m = (unsigned short*)(stg + C::STG + p * RECD + D / 2) + h * (NRL / 4);   // ... and the minima of its groups
    if constexpr (MINREC) GSel::init(scr);

    // --- staging: item idx = one dword of the wave's copy; dword m holds copy bytes [4m, 4m+4) (biased planes) ---------
    // Unconditional loads: bytes past a row's end only ever reach lanes that are not `active`, and the prefiltered planes
    // are allocated with 1 KB of slack behind the last row (rtdm_api.hip).  Two rows of loads are in flight (pre[0/1]).
    const int row0 = ys0 - r;
    const int Hm1 = g.H - 1;
#if RING_DIRECT_LOADS
    // One load per staged dword: address = the left plane's row (wave-uniform: lives in SGPRs and advances on the scalar
    // unit) + a 32-bit lane offset -- the right plane lies rg.rdelta bytes behind the left one (one allocation, rtdm_api.hip).
    // The byte phase makes the address unaligned; the memory pipeline takes unaligned dwords, and what arrives is what gets
    // staged: no second dword, no v_alignbyte, no 64-bit pointer arithmetic per lane.
    const uint8_t* rowp = Lb + (size_t)row0 * Lp.pitch;
    uint32_t loff[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int idx = lane + it * 64;
        const bool isr = idx >= LWD;
        const int m = isr ? idx - LWD : idx;
        loff[it] = (uint32_t)((isr ? Rbase : Lbase) + 4 * m) + (isr ? rg.rdelta : 0u);
    }
    int next_row = row0;
    uint32_t pre[2][ITEMS];
    auto issue = [&](auto Sc) {                     // loads of the next row into register set Sc; rows past H-1 repeat H-1
        constexpr int SET = decltype(Sc)::value;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
#if RING_ABL == 3
            pre[SET][it] = (uint32_t)next_row * 0x01020304u + lane;
#else
            // (the empty asm keeps the zero-extension of the offset next to the load: hoisted out of the loop as a 64-bit
            //  value it hides the "SGPR base + 32-bit VGPR offset" addressing mode and costs a 64-bit VALU add per load)
            asm volatile("" : "+v"(loff[it]));
            pre[SET][it] = *(const uint32_t*)(rowp + loff[it]);
#endif
        }
        rowp += next_row < Hm1 ? Lp.pitch : 0;
        ++next_row;
    };
    auto commit = [&](auto Sc, auto slot) {        // slot: an int, or an integral_constant where the slots are static
        constexpr int SET = decltype(Sc)::value;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) stg[slot * SLOT + lane + it * 64] = pre[SET][it];   // (the planes carry the +1 bias themselves)
        __builtin_amdgcn_wave_barrier();           // the wave's later reads stay behind these writes (LDS is in order per wave)
    };
#else
    const uint8_t* src[ITEMS];                      // where the next row to be issued starts, per item
    int it_sh[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int idx = lane + it * 64;
        const bool isr = idx >= LWD;
        const int m = isr ? idx - LWD : idx;
        const int col = (isr ? Rbase : Lbase) + 4 * m;
        src[it] = (isr ? Rb : Lb) + (size_t)row0 * Lp.pitch + (col & ~3);
        it_sh[it] = col & 3;
    }
    int next_row = row0;
    uint32_t pre[2][ITEMS][2];
    auto issue = [&](auto Sc) {                     // loads of the next row into register set Sc; rows past H-1 repeat H-1
        constexpr int SET = decltype(Sc)::value;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
#if RING_ABL == 3
            pre[SET][it][0] = (uint32_t)next_row * 0x01020304u + lane; pre[SET][it][1] = (uint32_t)next_row * 0x04030201u ^ lane;
#else
            pre[SET][it][0] = *(const uint32_t*)(src[it]);
            pre[SET][it][1] = *(const uint32_t*)(src[it] + 4);
#endif
        }
        const size_t adv = next_row < Hm1 ? Lp.pitch : 0;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) src[it] += adv;
        ++next_row;
    };
    auto commit = [&](auto Sc, auto slot) {        // slot: an int, or an integral_constant where the slots are static
        constexpr int SET = decltype(Sc)::value;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t v = __builtin_amdgcn_alignbyte(pre[SET][it][1], pre[SET][it][0], (uint32_t)it_sh[it]);
            stg[slot * SLOT + lane + it * 64] = v;                          // (the planes carry the +1 bias themselves)
        }
        __builtin_amdgcn_wave_barrier();           // the wave's later reads stay behind these writes (LDS is in order per wave)
    };
#endif
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;

    RingState<D, WS, LPP> st;
#pragma unroll
    for (int k = 0; k < W1; ++k)
#pragma unroll
        for (int i = 0; i < NGL; ++i) st.P[k][i] = 0;
#pragma unroll
    for (int k = 0; k < W1; ++k) ptr[k * PPW + p] = 0;
    uint32_t pt = 0;                                // texture prefix sum of the rows so far

    const int nsteps = (ys1 - ys0) + WS - 1;
    const int nstepsg = (nsteps + RPG - 1) / RPG * RPG;   // rows go in groups of RPG; padded last rows are computed and dropped
    issue(Set0{}); issue(Set1{});
    commit(Set0{}, 0); commit(Set1{}, 1);           // rows 0 and 1 -> slots 0 and 1
    issue(Set0{});                                  // row 2: committed at the end of step 0

    // Output addresses: a wave-uniform frame base plus a 32-bit byte offset per lane that advances LPP rows per group
    // (the host checks that a frame's planes stay below 4 GB) -- no 64-bit multiply per store.
    char* const db = (char*)(disp.base + (size_t)f * disp.frame_e);
    char* const cb = (char*)(cost + (size_t)f * g.H * g.Ws);
    const int col = g.lofs + x;
    constexpr int TF = (WS - 1) / RPG * RPG;        // first group with an output row; its row h is strip row TF + h - (WS - 1)
    uint32_t dofs = (uint32_t)(((long long)(ys0 + TF + h - (WS - 1)) * (long long)disp.pitch_e + col) * 2);   // (mod 2^32; a row
    uint32_t cofs = (uint32_t)(((long long)(ys0 + TF + h - (WS - 1)) * (long long)g.Ws + col) * 2);           //  above the strip is never stored)
    const uint32_t dstep = (uint32_t)(disp.pitch_e * 2 * RPG), cstep = (uint32_t)(g.Ws * 2 * RPG);
    const bool masked_col = g.mask_cols && (col < g.vx0 || col >= g.vx1);

    uint32_t S[RPG][NRL];
#ifdef RING_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory");
#endif
    // Row t is read from LDS slot t mod 3, where it was put two steps earlier.  Step t issues the global loads of row
    // t+3 at its start and, at its end, writes row t+2 (issued one step earlier) to LDS -- BEHIND the step's quad-SADs
    // (the empty asm on S pins that order: left alone the compiler sinks the SADs below the commit and every step then
    // waits for its loads).  Rows past the strip's last are loaded and staged too (clamped to the frame) and never read:
    // no branch in the step.
    int sl_cur = 0;                                 // LDS slot of row t
    // this lane's read addresses in the slot of row t: three registers that move on by one slot per row (one v_add each
    // with a wave-uniform step; laundered so that every window offset folds into a ds_read immediate of ITS base)
    // (32-bit LDS addresses, not C++ pointers: a generic pointer that went through an asm loses its address space and
    //  turns every read into a flat load -- measured 2x slower)
    // Static slots (four of them, trip a multiple of four rows): the slot is a constant of the unrolled step and the three
    // registers never move -- every read is base + immediate.
    uint32_t la_cur = lds_addr(stg + p), ra_cur = lds_addr(stg + LWD + p + h * NGL), ro_cur = ra_cur + 4;
    if constexpr (C::STATIC_SLOTS) asm volatile("" : "+v"(la_cur), "+v"(ra_cur), "+v"(ro_cur));
    const auto lds_row = [&](auto SLc, RowRegs<D, WS, LPP>& rw) {
        if constexpr (C::STATIC_SLOTS) {
            constexpr uint32_t off = (uint32_t)(decltype(SLc)::value * SLOT * 4);
            ring_load_row<D, WS, LPP>(lds_at(la_cur + off), lds_at(ra_cur + off), lds_at(ro_cur + off), rw);
        } else if constexpr (C::ROW_PTRS) {
            asm volatile("" : "+v"(la_cur), "+v"(ra_cur), "+v"(ro_cur));
            ring_load_row<D, WS, LPP>(lds_at(la_cur), lds_at(ra_cur), lds_at(ro_cur), rw);
            const uint32_t adv = (uint32_t)((sl_cur == 2 ? -2 * SLOT : SLOT) * 4);   // (before the step moves sl_cur on)
            la_cur += adv; ra_cur += adv; ro_cur += adv;
        } else {
            int li = sl_cur * SLOT + p, ri = sl_cur * SLOT + LWD + p + h * NGL, one = 1;
            asm volatile("" : "+v"(li), "+v"(ri), "+v"(one));
            ring_load_row<D, WS, LPP, const uint32_t*>(stg + li, stg + ri, stg + ri + one, rw);
        }
    };
    auto step = [&](auto Kc, auto SLc, auto Rc, const RowRegs<D, WS, LPP>& rw, uint32_t (&Sr)[NRL]) {
        constexpr int K = decltype(Kc)::value;            // ring slot of row t: a compile-time register set
        using SetIn = std::integral_constant<int, (K + 1) & 1>;    // row t+3 goes where row t+1 was (W1 is even: K and t have the same parity)
        using SetOut = std::integral_constant<int, K & 1>;         // row t+2
        issue(SetIn{});
        uint32_t tnew = 0;
        ring_step<D, WS, LPP, K>(st, rw, rg.lastmask, capb, tnew, Sr);
        constexpr int KO = (K + 1) % W1;
        pt += tnew;
        const uint32_t told = ptr[KO * PPW + p];                  // prefix sum w rows back (0 while the window fills)
        ptr[K * PPW + p] = (unsigned short)pt;
        ptw[decltype(Rc)::value * PPW + p] = (unsigned short)(pt - told);   // (mod 2^16: the window sum itself is < 2^16)
        pin(Sr);
        if constexpr (C::STATIC_SLOTS) {
            commit(SetOut{}, std::integral_constant<int, (decltype(SLc)::value + 2) % 4>{});   // row t+2
        } else {
            const int sl_new = sl_cur == 0 ? 2 : sl_cur - 1;        // slot of row t+2 = (t + 2) mod 3
            sl_cur = sl_cur == 2 ? 0 : sl_cur + 1;
            commit(SetOut{}, sl_new);
        }
    };

    // One trip of the outer loop = whole rounds of the ring AND whole row groups (TRIP rows), unrolled: every row step has
    // its ring slot -- its registers -- fixed at compile time.
    int trips_since_rebase = 0;
    for (int t0 = 0; t0 < nstepsg; t0 += C::TRIP) {
        // Rebase (round 3): a trip starts on ring slot 0, whose content P(t-w-1) is dead; slot 1 holds the oldest prefix sum
        // still needed, P(t-w).  Subtracting it from every live slot (per 16-bit half, no borrow: prefix sums only grow)
        // leaves all window sums unchanged and keeps the halves below 2^16 for another rg.rebase trips -- strips no longer have
        // to be short, and a long strip pays the w-1 window-filling rows once instead of once per ~100 rows.
        if (rg.rebase && ++trips_since_rebase > rg.rebase) {
            trips_since_rebase = 1;
#pragma unroll
            for (int i = 0; i < NGL; ++i) {
                const uint32_t blo = (uint32_t)st.P[1][i], bhi = (uint32_t)(st.P[1][i] >> 32);
#pragma unroll
                for (int k = 1; k < W1; ++k) {
                    const uint32_t lo = (uint32_t)st.P[k][i] - blo, hi = (uint32_t)(st.P[k][i] >> 32) - bhi;
                    st.P[k][i] = (uint64_t)lo | ((uint64_t)hi << 32);
                }
            }
        }
        ring_for_groups(std::make_integer_sequence<int, C::TRIP / RPG>{}, [&](auto Uc) -> bool {
            constexpr int U = RPG * decltype(Uc)::value;
            const int t = t0 + U;
            if (t >= nstepsg) return false;
            RING_STAMP(0);                                          // (loop overhead + whatever precedes the group)
            GSel gsel;
            ring_for_rows(std::make_integer_sequence<int, RPG>{}, [&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                RowRegs<D, WS, LPP> rw;
                using SL = std::integral_constant<int, (U + R) % 4>;    // (static slots: TRIP % 4 == 0, so t % 4 == (U + R) % 4)
                lds_row(SL{}, rw);
                RING_STAMP(1);                                      // LDS reads of the row (the stamp waits for them)
                step(std::integral_constant<int, (U + R) % W1>{}, SL{}, Rc, rw, S[R]);
                RING_STAMP(2);
                if constexpr (SPLIT && RING_ABL == 0) {
                    // the row's slice goes to its owner's record and into the group minima at once: S[R] is dead after this
                    // (also while the window fills: a branch around it turns into selects on all of gsel's state)
                    if constexpr (MINREC) gsel.template row<R>(S[R], scr_w + R * (PPW * RECD), scr_m + R * (PPW * RECD * 2));
                    else gsel.template row<R>(S[R], scr_w + R * (PPW * RECD), (uint32_t)(h * (NRL / 4)));
                }
            });
            if (t + RPG - 1 < WS - 1) return true;                  // the window is still filling
            // the lanes p + h PPW hold the LPP slices of a pixel for the rows t .. t+RPG-1; the lane with h = k < RPG owns row t+k
            const int y = ys0 + (t - (WS - 1)) + h;
            const bool row_ok = owner && y >= ys0 && y < ys1;
            const uint32_t dof = dofs, cof = cofs;
            dofs += dstep; cofs += cstep;
            // Selection is skipped for a wave none of whose pixels can produce a disparity here (untextured, outside the
            // tile, masked): exact, such a pixel is FILTERED and writes no cost whatever its SADs are.
            const int tsum = *ptw_own;                               // the texture sum of the row this lane owns
            const bool dead = !active || !row_ok || masked_col || tsum < g.tex;
            if (__builtin_amdgcn_ballot_w64(!dead) == 0) {
                if (active && row_ok) *(int16_t*)(db + dof) = (int16_t)g.filtered;
            } else {
                int m1; bool fail;
                int out;
                if constexpr (SPLIT && RING_ABL == 0) {
                    out = gsel.finish(tsum, g, scr, &m1, &fail);
                    RING_STAMP(5);
                } else {
                    // two lanes per pixel: after the swap the lower lane has both halves of row t and the upper lane both
                    // halves of row t+1
                    uint32_t rr[D / 2];
#pragma unroll
                    for (int i = 0; i < NRL; ++i) {
#if RING_ABL == 5 || RING_ABL == 6
                        rr[i] = S[0][i]; rr[NRL + i] = S[RPG - 1][i];
#else
                        const auto sw = __builtin_amdgcn_permlane32_swap(S[0][i], S[RPG - 1][i], false, false);
                        rr[i] = sw[0]; rr[NRL + i] = sw[1];
#endif
                    }
                    RING_STAMP(5);                                  // swaps
#if RING_ABL == 1 || RING_ABL == 6
                    uint32_t xo = 0;
#pragma unroll
                    for (int i = 0; i < D / 2; ++i) xo ^= rr[i];
                    out = (int)xo; m1 = (int)(xo >> 3); fail = (xo & 1) != 0;
#elif RING_LDS_SELECT
                    out = select_disparity_lds<D>(rr, tsum, g, scr, &m1, &fail);
#else
                    out = select_disparity<D>(rr, tsum, g, &m1, &fail);
#endif
                }
#if RING_ABL == 2
                if (active && row_ok && out == 0x12345678) {
#else
                if (active && row_ok) {
#endif
                    if (!fail && g.want_cost) *(uint16_t*)(cb + cof) = (uint16_t)m1;
                    *(int16_t*)(db + dof) = (int16_t)(masked_col ? g.filtered : out);
                }
            }
            RING_STAMP(6);                                          // selection + stores
            return true;
        });
    }
#ifdef RING_STAMPS
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&ring_stamps[i], stamp_sum[i]);
#endif
}

// ---- host side ----------------------------------------------------------------------------
static bool ring_range(const BMGeom& g, int* x0, int* nx)
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

// rows a strip may have so that no prefix sum leaves 16 bits: a row adds at most w * 2 cap per disparity, and a strip of
// rs output rows walks rs + w - 1 rows plus up to seven padded rows (groups of eight)
static int ring_rows_cap(const BMGeom& g) { return 65535 / (g.w * 2 * g.cap) - g.w - 6; }
static int ring_lpp(const BMGeom& g);
// trips of the row loop between two rebases of the ring (0: one trip is longer than the cap -- no rebasing, short strips)
static int ring_rebase_trips(const BMGeom& g)
{
    static const int enabled = env_int("RTDM_RING_REBASE", 1);        // A/B switch: 0 = short strips (round 2)
    const int lpp = ring_lpp(g);
    if (!enabled || !lpp) return 0;
    const int W1 = g.w + 1, rpg = ring_rpg(g.D, lpp);
    const int trip = (W1 % rpg == 0) ? W1 : (2 * W1 % rpg == 0) ? 2 * W1 : 4 * W1;   // RingCfg::TRIP
    return ring_rows_cap(g) / trip;
}
// strips a frame must at least be cut into
static int ring_min_strips(const BMGeom& g, int nrows)
{ return ring_rebase_trips(g) > 0 ? 1 : max(1, (nrows + ring_rows_cap(g) - 1) / ring_rows_cap(g)); }

// Instantiations: (D, blockSize, lanes per pixel).  Two lanes per pixel: every (D, blockSize) whose ring (blockSize+1) * D/4
// registers per lane leaves room for two waves per SIMD.  Four lanes per pixel: the D = 64 ones, whose two-lane ring holds
// them at two waves, and D = 96 (D = 32 with four lanes measured 0-8 % slower than with two: not instantiated).  Eight
// lanes: D = 128.
#ifdef RTDM_RING_DEV      // development builds (tools/ring_isa.sh): -DRTDM_RING_DEV="X(64, 9, 4)" = these instantiations only
#define RTDM_RING_TABLE(X) RTDM_RING_DEV
#else
#define RTDM_RING_TABLE(X) X(64, 9, 2) X(64, 7, 2) X(64, 5, 2) X(32, 7, 2) X(32, 9, 2) X(32, 11, 2) X(32, 13, 2) X(48, 7, 2) X(48, 9, 2) \
                           X(16, 5, 2) X(16, 7, 2) X(16, 9, 2) X(64, 9, 4) X(64, 7, 4) X(64, 5, 4) X(64, 11, 4) X(64, 13, 4) \
                           X(128, 7, 8) X(128, 9, 8) X(128, 11, 8) X(128, 13, 8) X(96, 7, 4) X(96, 9, 4) X(96, 11, 4) X(96, 13, 4) X(48, 11, 2) X(48, 13, 2) \
                           X(16, 11, 2) X(16, 13, 2) X(32, 5, 2) X(32, 15, 2) X(48, 5, 2) X(64, 15, 4) X(128, 15, 8) \
                           X(192, 9, 8) X(192, 11, 8) X(192, 13, 8) X(192, 15, 8) X(256, 9, 16) X(256, 11, 16) X(256, 13, 16) X(256, 15, 16)
#endif

// rtdm_debug_search_kernel (a process-wide A/B switch; one atomic word so that a launch on another thread sees a consistent
// pair): low byte = mode + 1 (0 never, 1 wherever instantiated, -1 default), next byte = lanes per pixel to force (0: none)
static std::atomic<int> g_ring_switch{0};
void ring_set_mode(int mode)
{
    const int m = mode < 0 ? -1 : mode == 0 ? 0 : 1, lpp = (mode == 2 || mode == 4 || mode == 8) ? mode : 0;
    g_ring_switch.store((m + 1) | (lpp << 8), std::memory_order_relaxed);
}
static int ring_mode() { return (g_ring_switch.load(std::memory_order_relaxed) & 0xff) - 1; }
static int ring_forced_lpp() { return g_ring_switch.load(std::memory_order_relaxed) >> 8; }

// lanes per pixel for a configuration (0: not instantiated).  RTDM_RING_LPP = 2 / 4 forces one form where it exists (A/B).
static int ring_lpp(const BMGeom& g)
{
    static const int env = [] { const char* e = getenv("RTDM_RING_LPP"); return e ? atoi(e) : 0; }();
    int have = 0;                          // bit mask of the forms instantiated for (D, w)
#define X(DD, WW, LL) if (g.D == DD && g.w == WW) have |= LL;
    RTDM_RING_TABLE(X)
#undef X
    const int forced = ring_forced_lpp(), want = forced ? forced : env;
    if (want && (have & want) == want) return want;
    // measured (tools/ab_ring.py): more lanes per pixel win where the ring holds the two-lane form at two waves per SIMD
    return (have & 16) ? 16 : (have & 8) ? 8 : (have & 4) ? 4 : (have & 2) ? 2 : 0;
}

int ring_lanes_per_pixel(const BMGeom& g) { return ring_lpp(g); }

static int ring_tile(const BMGeom& g) { return 256 / ring_lpp(g); }

// workgroups resident per CU (= waves per SIMD: a workgroup is four waves, one per SIMD) of the form (D, w) runs
static int ring_wgs_per_cu(const BMGeom& g)
{
    const int lpp = ring_lpp(g);
#define X(DD, WW, LL) if (g.D == DD && g.w == WW && lpp == LL) return RingCfg<DD, WW, LL>::WAVES;
    RTDM_RING_TABLE(X)
#undef X
    return 2;
}

int ring_strips_model(const BMGeom& g, int n)
{
    int x0 = 0, nx = 0;
    if (!ring_range(g, &x0, &nx) || !ring_lpp(g)) return 1;
    const int tile = ring_tile(g), tiles = (nx + tile - 1) / tile, nrows = g.vy1 - g.vy0;
    const int slots = 256 * ring_wgs_per_cu(g);             // workgroups in flight on the chip
    const int smin = ring_min_strips(g, nrows), smax = max(smin, (nrows + 15) / 16);
    // a strip pays w-1 filling rows at about half the price of an output row
    const float fill = 0.5f * (float)(g.w - 1);
    int s = (int)(sqrtf((float)nrows * (float)slots / (fill * (float)tiles * (float)n)) + 0.5f);
    s = max(smin, min(s, smax));
    if ((long)tiles * s * n < 6L * slots) {
        // Small jobs (single frames, small batches -- nothing measures their strip count): the grid is a handful of
        // scheduling rounds, and a round that is started is paid in full.  Time ~ rounds x (rows per workgroup + fill + a
        // fixed two rows' worth of prologue); take the strip count that minimises it.
        float best = 1e30f;
        for (int c = smin; c <= smax; ++c) {
            const int rs = (nrows + c - 1) / c, ce = (nrows + rs - 1) / rs;
            const long rounds = ((long)tiles * ce * n + slots - 1) / slots;
            const float t = (float)rounds * ((float)rs + fill + 2.0f);
            if (t < best) { best = t; s = ce; }
        }
    }
    return s;
}

template <int D, int WS, int LPP>
static bool ring_launch_one(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g, int n, hipStream_t stream, int strips_hint,
                            bool fuse_border)
{
    using C = RingCfg<D, WS, LPP>;
    RingGeom rg;
    rg.rdelta = (uint32_t)(Rp.base - Lp.base);       // (launch_search_ring has checked that it fits)
    ring_range(g, &rg.x0, &rg.nx);
    const int rem = g.w - 4 * (C::NP - 1);
    rg.lastmask = rem >= 4 ? 0xffffffffu : ((1u << (8 * rem)) - 1u);
    const int nrows = g.vy1 - g.vy0;
    const int tiles = (rg.nx + C::TILE - 1) / C::TILE;
    int strips = strips_hint > 0 ? strips_hint : ring_strips_model(g, n);
    strips = max(strips, ring_min_strips(g, nrows));
    rg.rebase = ring_rebase_trips(g);                 // (= ring_rows_cap / C::TRIP)
    strips = max(1, min(strips, nrows));
    rg.rs = (nrows + strips - 1) / strips;
    strips = (nrows + rg.rs - 1) / rg.rs;
    rg.tiles = tiles; rg.strips = strips;
    rg.nitems = (unsigned)tiles * strips * n;
    static const bool xcd_local = [] { const char* e = getenv("RTDM_RING_XCD"); return !e || atoi(e) != 0; }();   // A/B switch
    rg.chunk = xcd_local ? (rg.nitems + 7) / 8 : 0;
    const unsigned grid = rg.chunk ? rg.chunk * 8 : rg.nitems;
    BorderGeom bg{};
    Border2Geom b2g{};
    size_t blds = 0;
    rg.nborder = 0; rg.bgx = rg.bgy = 0; rg.b2 = 0;
    bool fused = false;                               // the border columns are dealt with by this launch
    if (fuse_border) {
        int lx0, lx1, rx0, rx1;
        fast_border_ranges(g, &lx0, &lx1, &rx0, &rx1);
        static const int fuse2 = env_int("RTDM_RING_FUSE_BORDER2", 1);   // A/B: 0 = border2 never rides in this grid (round 3's first form)
        if (fuse2 && C::FUSE_BORDER2 && border2_plan(g, lx0, lx1, rx0, rx1, &b2g, &rg.bgx, &rg.bgy)) {
            rg.b2 = 1; fused = true;
            rg.nborder = (unsigned)(rg.bgx * rg.bgy) * (unsigned)n;
        } else if constexpr (C::FUSE_BORDER) {
            if (border_geometry(g, lx0, lx1, rx0, rx1, n, &bg, &rg.bgx, &rg.bgy, &blds)) { rg.nborder = (unsigned)(rg.bgx * rg.bgy) * (unsigned)n; fused = true; }
        }
    }
    static const size_t ldspad = [] { const char* e = getenv("RTDM_RING_LDSPAD"); return e ? (size_t)atol(e) : (size_t)0; }();
    const size_t ldsb = max((size_t)4 * C::WAVE_LDS * sizeof(uint32_t), blds) + ldspad;   // (padding: occupancy experiments)
    const auto launch = [&](auto Fc) {
        constexpr bool F = decltype(Fc)::value;
        if (ldsb > 48 * 1024) {                     // once per device of this process (a handle lives on one device)
            // (one flag per instantiation; a second thread that races the first caller on the same device sets the same value)
            static OncePerDevice once;
            static std::atomic<size_t> granted{0};
            const size_t want = max(ldsb, (size_t)64 * 1024);
            if (once.first() || granted.load(std::memory_order_acquire) < want) {
                (void)hipFuncSetAttribute((const void*)k_search_ring<D, WS, LPP, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
                size_t g0 = granted.load(std::memory_order_relaxed);
                while (g0 < want && !granted.compare_exchange_weak(g0, want, std::memory_order_release)) {}
            }
        }
        hipLaunchKernelGGL((k_search_ring<D, WS, LPP, F>), dim3(grid + rg.nborder), dim3(256), ldsb, stream, Lp, Rp, disp, (uint16_t*)cost, g, rg, bg, b2g);
    };
    if (rg.nborder) { launch(std::true_type{}); return true; }
    launch(std::false_type{});
    return fused;                                     // (true without border workgroups: there are no border columns)
}

bool ring_search_supported(const BMGeom& g)
{
    static const int env = [] { const char* e = getenv("RTDM_RING"); return e ? atoi(e) : -1; }();   // A/B switch
    const int sw = ring_mode(), mode = sw >= 0 ? sw : env;
    if (mode == 0) return false;
    if (2L * g.cap * g.w * g.w > 32766) return false;       // packed u16 sums + the T+1 <= 32767 argument of the selection
    if (ring_rows_cap(g) < 2) return false;
    if (!ring_range(g, nullptr, nullptr)) return false;
    if ((size_t)g.H * (size_t)g.Ws * 2 >= ((size_t)1 << 32)) return false;   // 32-bit byte offsets inside a frame (pitch <= Ws)
    return ring_lpp(g) != 0;
}

bool launch_search_ring(Plane8 Lp, Plane8 Rp, Plane16W disp, void* cost, const BMGeom& g, int n, hipStream_t stream, int strips_hint,
                        bool fuse_border)
{
    const int lpp = ring_lpp(g);
    // the kernel addresses both planes from the left plane's rows with 32-bit lane offsets: they must be one allocation, the
    // right plane behind the left one, with the same strides (rtdm_api.hip allocates them that way)
    if (RING_DIRECT_LOADS && !(Rp.base > Lp.base && (size_t)(Rp.base - Lp.base) < ((size_t)1 << 32) - ((size_t)1 << 20) && Rp.pitch == Lp.pitch && Rp.frame == Lp.frame)) {
        launch_search_fast(Lp, Rp, disp, cost, g, n, stream, fuse_border, 0);
        return fuse_border;
    }
#define X(DD, WW, LL) if (g.D == DD && g.w == WW && lpp == LL) return ring_launch_one<DD, WW, LL>(Lp, Rp, disp, cost, g, n, stream, strips_hint, fuse_border);
    RTDM_RING_TABLE(X)
#undef X
    return false;
}

}  // namespace rtdm
