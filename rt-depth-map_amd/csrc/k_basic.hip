// k_basic.hip -- the HBM-bound per-pixel / per-row stages of the pipeline for gfx950:
//   K1 x-Sobel prefilter, FILTERED fill, K3 left-right check, K4 speckle filter.
// Semantics: SURVEY.md Appendix A.3a / A.4 / A.5 (what cv::StereoBM does behind
// /root/reference/stereo-matcher/bm-sw.cpp:35); oracle: oracle/bm_oracle.c.
#include "rtdm_kernels.h"
#include "rtdm_device.h"

#include <cstdlib>

namespace rtdm {

// ---------------------------------------------------------------------------------------------
// K1 prefilter: x-Sobel, clip to +-cap, + cap.  Rows come in pairs; a trailing odd row is all
// `cap`; row -1 mirrors to 1, row H to H-2; columns 0 and W-1 are `cap`.
// Fast variant: one thread = 8 consecutive output bytes from three 8-byte loads per source row
// (needs 8-byte aligned base/pitch/frame); byte variant for arbitrary caller pitches.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int colsum3(unsigned long long a, unsigned long long c, unsigned long long b, int k)
{ return (int)((a >> (8 * k)) & 0xff) + 2 * (int)((c >> (8 * k)) & 0xff) + (int)((b >> (8 * k)) & 0xff); }

__global__ __launch_bounds__(256) void k_prefilter8(Plane8 L, Plane8 R, Plane8W Lp, Plane8W Rp,
                                                    int W, int H, int cap, int n, int nxb)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;        // over (row, 8-byte block)
    if (idx >= nxb * H) return;
    const int y = idx / nxb, xb = idx - y * nxb, x0 = xb * 8;
    int f = blockIdx.y;
    const bool right = f >= n;
    if (right) f -= n;
    const Plane8 S = right ? R : L;
    const Plane8W O = right ? Rp : Lp;
    const uint8_t* src = S.base + (size_t)f * S.frame;
    uint8_t* dst = O.base + (size_t)f * O.frame + (size_t)y * O.pitch + x0;
    const int npair = (H >= 2) ? (H & ~1) : 0;
    const int off = cap + PREFILTER_BIAS;
    unsigned long long out = (unsigned long long)off * 0x0101010101010101ull;
    if (y < npair) {
        const int ya = (y > 0) ? y - 1 : 1;
        const int yb = (y < H - 1) ? y + 1 : H - 2;
        const uint8_t* ra = src + (size_t)ya * S.pitch + x0;
        const uint8_t* rc = src + (size_t)y * S.pitch + x0;
        const uint8_t* rb = src + (size_t)yb * S.pitch + x0;
        const bool has_prev = x0 > 0, has_next = x0 + 16 <= (int)S.pitch;
        const unsigned long long a1 = *(const unsigned long long*)ra, c1 = *(const unsigned long long*)rc,
                                 b1 = *(const unsigned long long*)rb;
        const unsigned long long a0 = has_prev ? *(const unsigned long long*)(ra - 8) : 0ull,
                                 c0 = has_prev ? *(const unsigned long long*)(rc - 8) : 0ull,
                                 b0 = has_prev ? *(const unsigned long long*)(rb - 8) : 0ull;
        const unsigned long long a2 = has_next ? *(const unsigned long long*)(ra + 8) : 0ull,
                                 c2 = has_next ? *(const unsigned long long*)(rc + 8) : 0ull,
                                 b2 = has_next ? *(const unsigned long long*)(rb + 8) : 0ull;
        int s[10];                                          // column sums for x0-1 .. x0+8
        s[0] = colsum3(a0, c0, b0, 7);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k + 1] = colsum3(a1, c1, b1, k);
        s[9] = colsum3(a2, c2, b2, 0);
        out = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int x = x0 + k;
            int g = s[k + 2] - s[k];
            g = g < -cap ? -cap : (g > cap ? cap : g);
            const int v = (x == 0 || x >= W - 1) ? off : g + off;
            out |= (unsigned long long)v << (8 * k);
        }
    }
    *(unsigned long long*)dst = out;                        // plane pitch is a multiple of 64: in bounds
}

// The unit of work of k_fill_frame (16 pixels of a row outside what the search writes, or one row's run count), also run by the
// extra workgroups of k_prefilter16<.., true>: a single frame is bound by its launches, not by its kernels.
struct FillArgs { Plane16W d; int cx0, cx1, vy0, vy1, value; int32_t* rowcnt; int first_block; };
__device__ __forceinline__ void fill_frame_units(Plane16W d, int W, int H, int cx0, int cx1, int vy0, int vy1, int value, int32_t* rowcnt, int idx, int frame)
{
    const int nw = (W + 15) / 16, nl = (cx0 + 15) / 16, nr = (W - cx1 + 15) / 16, nv = vy1 - vy0;
    const int u0 = nw * vy0, u1 = u0 + nw * (H - vy1), u2 = u1 + nl * nv, u3 = u2 + nr * nv;
    if (idx >= u3) {
        idx -= u3;
        if (rowcnt && idx < H) rowcnt[frame * H + idx] = 0;
        return;
    }
    int y, xs, xe;
    if (idx < u0)      { y = idx / nw; xs = (idx - y * nw) * 16; xe = W; }
    else if (idx < u1) { idx -= u0; y = idx / nw; xs = (idx - y * nw) * 16; xe = W; y += vy1; }
    else if (idx < u2) { idx -= u1; y = idx / nl; xs = (idx - y * nl) * 16; xe = cx0; y += vy0; }
    else               { idx -= u2; y = idx / nr; xs = cx1 + (idx - y * nr) * 16; xe = W; y += vy0; }
    int16_t* p = d.base + (size_t)frame * d.frame_e + (size_t)y * d.pitch_e;
    for (int x = xs; x < min(xs + 16, xe); ++x) p[x] = (int16_t)value;
}

// Strip variant for 16-byte aligned sources: one thread = 16 columns x RY rows.  Rows stream through registers (each
// source row is loaded once per strip as ONE 128-bit load; k_prefilter8 issues nine 64-bit loads per 8 output
// bytes and is bound by the load-issue rate), the bytes left and right of the 16 come from the neighbouring lanes.
template <int RY, bool FILL>
__global__ __launch_bounds__(256) void k_prefilter16(Plane8 L, Plane8 R, Plane8W Lp, Plane8W Rp,
                                                     int W, int H, int cap, int n, int nxb, FillArgs fa)
{
    if constexpr (FILL) {
        if ((int)blockIdx.x >= fa.first_block) {            // the extra workgroups: k_fill_frame's work for frame blockIdx.y
            if ((int)blockIdx.y < n)
                fill_frame_units(fa.d, W, H, fa.cx0, fa.cx1, fa.vy0, fa.vy1, fa.value, fa.rowcnt, ((int)blockIdx.x - fa.first_block) * 256 + threadIdx.x, blockIdx.y);
            return;
        }
    }
    const int nstrip = (H + RY - 1) / RY;
    const int idx = blockIdx.x * 256 + threadIdx.x;        // over (strip, 16-byte block)
    const bool inb = idx < nxb * nstrip;
    const int cidx = inb ? idx : 0;
    const int strip = cidx / nxb, x0 = (cidx - strip * nxb) * 16, ys = strip * RY;
    const int lane = threadIdx.x & 63;
    int f = blockIdx.y;
    const bool right = f >= n;
    if (right) f -= n;
    const Plane8 S = right ? R : L;
    const Plane8W O = right ? Rp : Lp;
    const uint8_t* src = S.base + (size_t)f * S.frame + x0;
    uint8_t* dst = O.base + (size_t)f * O.frame + (size_t)ys * O.pitch + x0;
    const int npair = (H >= 2) ? (H & ~1) : 0;
    const bool has_prev = x0 > 0, has_next = x0 + 16 < W;
    const uint32_t capb = (uint32_t)(cap + PREFILTER_BIAS) * 0x01010101u;   // what edge columns and an odd last row hold
    const uint4 capv = make_uint4(capb, capb, capb, capb);
    // Packed 16-bit arithmetic, two columns per instruction (the scalar form ran 27 VALU instructions per pixel and was
    // VALU bound at 87 % busy -- not HBM bound, as a prefilter should be): the 18 bytes b[0..17] = left neighbour, the 16
    // of this thread, right neighbour become nine pairs P[i] = (b[2i], b[2i+1]) by v_perm; hd pair i = P[i+1] - P[i] =
    // (b[2i+2] - b[2i], b[2i+3] - b[2i+1]), the x-differences of columns 2i and 2i+1.
    typedef short s2 __attribute__((ext_vector_type(2)));
    const auto pk = [](uint32_t v) { return __builtin_bit_cast(s2, v); };
    const auto un = [](s2 v) { return __builtin_bit_cast(uint32_t, v); };
    const s2 capp = pk((uint32_t)cap * 0x00010001u), ncapp = pk((uint32_t)(-cap & 0xffff) * 0x00010001u);
    const s2 offp = pk((uint32_t)(cap + PREFILTER_BIAS) * 0x00010001u);
    // bytes of the output that are the frame's first / last column (or padding): they hold `cap`
    uint32_t em[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int x = x0 + 4 * q + k; m |= (x == 0 || x >= W - 1) ? (0xffu << (8 * k)) : 0u; }
        em[q] = m;
    }
    // all RY + 2 source rows are requested before the first one is used: a wave that waits for each row before asking for
    // the next keeps 1 KB in flight, and the kernel then runs at what eight such waves per SIMD can pull (5 TB/s)
    uint4 qs[RY + 2];
    int le[RY + 2], re[RY + 2];                             // bytes across the wave's edges (lanes 0 and 63 only)
#pragma unroll
    for (int j = 0; j < RY + 2; ++j) {
        int yy = ys + j - 1;                                // source row of this step (mirrored at the frame edge)
        yy = yy < 0 ? 1 : (yy > H - 1 ? H - 2 : yy);
        if (H < 2) yy = 0;
        const uint8_t* rp = src + (size_t)yy * S.pitch;
        qs[j] = make_uint4(0, 0, 0, 0); le[j] = 0; re[j] = 0;
        if (inb) qs[j] = *(const uint4*)rp;
        if (inb && lane == 0 && has_prev) le[j] = rp[-1];
        if (inb && lane == 63 && has_next) re[j] = rp[16];
    }
    s2 hd[3][8];                                            // x-differences of the last three rows
#pragma unroll
    for (int j = 0; j < RY + 2; ++j) {
        const uint4 q = qs[j];
        int lb = __shfl_up((int)(q.w >> 24), 1), rb = __shfl_down((int)(q.x & 0xff), 1);
        if (lane == 0) lb = le[j];
        if (lane == 63) rb = re[j];
        uint32_t P[9];
        P[0] = __builtin_amdgcn_perm(q.x, (uint32_t)lb, 0x0c040c00u);
        P[1] = __builtin_amdgcn_perm(0u, q.x, 0x0c020c01u);
        P[2] = __builtin_amdgcn_perm(q.y, q.x, 0x0c040c03u);
        P[3] = __builtin_amdgcn_perm(0u, q.y, 0x0c020c01u);
        P[4] = __builtin_amdgcn_perm(q.z, q.y, 0x0c040c03u);
        P[5] = __builtin_amdgcn_perm(0u, q.z, 0x0c020c01u);
        P[6] = __builtin_amdgcn_perm(q.w, q.z, 0x0c040c03u);
        P[7] = __builtin_amdgcn_perm(0u, q.w, 0x0c020c01u);
        P[8] = __builtin_amdgcn_perm((uint32_t)rb, q.w, 0x0c040c03u);
#pragma unroll
        for (int i = 0; i < 8; ++i) hd[j % 3][i] = pk(P[i + 1]) - pk(P[i]);
        if (j >= 2) {
            const int y = ys + j - 2;                       // output row: rows (j-2, j-1, j) are (above, centre, below)
            if (inb && y < H) {
                uint4 o = capv;
                if (y < npair) {
                    uint32_t v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const s2 c = hd[(j - 1) % 3][i];
                        s2 g = hd[(j - 2) % 3][i] + hd[j % 3][i] + c + c;          // |g| <= 1020
                        g = __builtin_elementwise_min(__builtin_elementwise_max(g, ncapp), capp) + offp;
                        v[i] = un(g);
                    }
                    uint32_t w[4];
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const uint32_t by = __builtin_amdgcn_perm(v[2 * qd + 1], v[2 * qd], 0x06040200u);   // four low bytes of two pairs
                        w[qd] = (by & ~em[qd]) | (capb & em[qd]);
                    }
                    o = make_uint4(w[0], w[1], w[2], w[3]);
                }
                *(uint4*)(dst + (size_t)(j - 2) * O.pitch) = o;   // plane pitch is a multiple of 64: in bounds
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_prefilter1(Plane8 L, Plane8 R, Plane8W Lp, Plane8W Rp,
                                                    int W, int H, int cap, int n)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    int f = blockIdx.z;
    if (x >= W) return;
    const bool right = f >= n;
    if (right) f -= n;
    const Plane8 S = right ? R : L;
    const Plane8W O = right ? Rp : Lp;
    const uint8_t* src = S.base + (size_t)f * S.frame;
    const int npair = (H >= 2) ? (H & ~1) : 0;
    int v = cap + PREFILTER_BIAS;
    if (y < npair && x > 0 && x < W - 1) {
        const int ya = (y > 0) ? y - 1 : 1;
        const int yb = (y < H - 1) ? y + 1 : H - 2;
        const uint8_t* ra = src + (size_t)ya * S.pitch;
        const uint8_t* rc = src + (size_t)y * S.pitch;
        const uint8_t* rb = src + (size_t)yb * S.pitch;
        int g = ((int)ra[x + 1] - (int)ra[x - 1]) + 2 * ((int)rc[x + 1] - (int)rc[x - 1]) + ((int)rb[x + 1] - (int)rb[x - 1]);
        g = g < -cap ? -cap : (g > cap ? cap : g);
        v = g + cap + PREFILTER_BIAS;
    }
    O.base[(size_t)f * O.frame + (size_t)y * O.pitch + x] = (uint8_t)v;
}

// fill != null: the frame fill (launch_fill_frame's arguments) rides in the prefilter's grid where the strip kernel runs, and is a
// launch of its own before the other forms
void launch_prefilter(Plane8 L, Plane8 R, Plane8W Lp, Plane8W Rp, int W, int H, int cap, int n,
                      hipStream_t stream, const FillJob* fill)
{
    const auto al16 = [](const Plane8& p) { return (((size_t)p.base | p.pitch | p.frame) & 15) == 0; };
    const size_t w16 = (size_t)((W + 15) & ~15);
    static const int strip16 = env_int("RTDM_PREFILTER_STRIP", 1);
    if (strip16 && al16(L) && al16(R) && L.pitch >= w16 && R.pitch >= w16 && ((size_t)Lp.base & 15) == 0 && ((size_t)Rp.base & 15) == 0) {
        constexpr int RY = 8;
        const int nxb = (W + 15) / 16;
        const int pblocks = (nxb * ((H + RY - 1) / RY) + 255) / 256;
        FillArgs fa{};
        static const int fuse_fill = env_int("RTDM_FILL_IN_PREFILTER", 1);      // A/B: 0 = k_fill_frame as a launch of its own
        if (fill && fuse_fill) {
            const int nw = (W + 15) / 16, nl = (fill->cx0 + 15) / 16, nr = (W - fill->cx1 + 15) / 16, nv = fill->vy1 - fill->vy0;
            const int units = nw * fill->vy0 + nw * (H - fill->vy1) + (nl + nr) * nv + (fill->rowcnt ? H : 0);
            if (units > 0) {
                fa = FillArgs{fill->disp, fill->cx0, fill->cx1, fill->vy0, fill->vy1, fill->value, fill->rowcnt, pblocks};
                hipLaunchKernelGGL((k_prefilter16<RY, true>), dim3(pblocks + (units + 255) / 256, 2 * n), dim3(256), 0, stream, L, R, Lp, Rp, W, H, cap, n, nxb, fa);
                return;
            }
        } else if (fill) launch_fill_frame(fill->disp, W, H, fill->cx0, fill->cx1, fill->vy0, fill->vy1, n, fill->value, fill->rowcnt, stream);
        hipLaunchKernelGGL((k_prefilter16<RY, false>), dim3(pblocks, 2 * n), dim3(256), 0, stream, L, R, Lp, Rp, W, H, cap, n, nxb, fa);
        return;
    }
    if (fill) launch_fill_frame(fill->disp, W, H, fill->cx0, fill->cx1, fill->vy0, fill->vy1, n, fill->value, fill->rowcnt, stream);
    const auto al8 = [](const Plane8& p) { return (((size_t)p.base | p.pitch | p.frame) & 7) == 0; };
    if (al8(L) && al8(R) && L.pitch >= (size_t)((W + 7) & ~7) && R.pitch >= (size_t)((W + 7) & ~7)) {
        const int nxb = (W + 7) / 8;
        hipLaunchKernelGGL(k_prefilter8, dim3((nxb * H + 255) / 256, 2 * n), dim3(256), 0, stream, L, R, Lp, Rp, W, H, cap, n, nxb);
    } else {
        hipLaunchKernelGGL(k_prefilter1, dim3((W + 255) / 256, H, 2 * n), dim3(256), 0, stream, L, R, Lp, Rp, W, H, cap, n);
    }
}

// ---------------------------------------------------------------------------------------------
// FILTERED fill of a rectangle [x0,x1) x [y0,y1) of every frame (16 pixels per thread).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill16(Plane16W d, int x0, int x1, int y0, int y1, int value)
{
    const int nxb = (x1 - x0 + 15) / 16;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nxb * (y1 - y0)) return;
    const int y = y0 + idx / nxb, xs = x0 + (idx % nxb) * 16;
    int16_t* p = d.base + (size_t)blockIdx.y * d.frame_e + (size_t)y * d.pitch_e;
    for (int x = xs; x < min(xs + 16, x1); ++x) p[x] = (int16_t)value;
}

// Everything of a frame the search does not write, in ONE launch (four launches + a memset cost a single frame 17 us): the
// rows above / below [vy0, vy1), the columns left / right of [cx0, cx1) inside them, and (rowcnt != null) the per-row run
// counts of the speckle filter, which the left-right check only writes for the valid rows.
__global__ __launch_bounds__(256) void k_fill_frame(Plane16W d, int W, int H, int cx0, int cx1, int vy0, int vy1, int value, int32_t* rowcnt)
{
    fill_frame_units(d, W, H, cx0, cx1, vy0, vy1, value, rowcnt, blockIdx.x * 256 + threadIdx.x, blockIdx.y);
}

void launch_fill_frame(Plane16W disp, int W, int H, int cx0, int cx1, int vy0, int vy1, int n, int value, int32_t* rowcnt, hipStream_t stream)
{
    const int nw = (W + 15) / 16, nl = (cx0 + 15) / 16, nr = (W - cx1 + 15) / 16, nv = vy1 - vy0;
    const int units = nw * vy0 + nw * (H - vy1) + (nl + nr) * nv + (rowcnt ? H : 0);
    if (units <= 0) return;
    hipLaunchKernelGGL(k_fill_frame, dim3((units + 255) / 256, n), dim3(256), 0, stream, disp, W, H, cx0, cx1, vy0, vy1, value, rowcnt);
}

__global__ __launch_bounds__(256) void k_copy16(Plane16W src, Plane16W dst, int W, int H)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W * H) return;
    const int y = idx / W, x = idx - y * W;
    dst.base[(size_t)blockIdx.y * dst.frame_e + (size_t)y * dst.pitch_e + x] = src.base[(size_t)blockIdx.y * src.frame_e + (size_t)y * src.pitch_e + x];
}

void launch_copy16(Plane16W src, Plane16W dst, int W, int H, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(k_copy16, dim3((W * H + 255) / 256, n), dim3(256), 0, stream, src, dst, W, H);
}

void launch_fill16(Plane16W disp, int x0, int x1, int y0, int y1, int n, int value, hipStream_t stream)
{
    if (x1 <= x0 || y1 <= y0) return;
    const int nxb = (x1 - x0 + 15) / 16;
    hipLaunchKernelGGL(k_fill16, dim3((nxb * (y1 - y0) + 255) / 256, n), dim3(256), 0, stream, disp, x0, x1, y0, y1, value);
}

// ---------------------------------------------------------------------------------------------
// K3 left-right check (cv::validateDisparity): one workgroup per (valid
// row, frame).  LDS holds a snapshot of the row and one 64-bit key per column: (cost << 32 | x);
// ds_min_u64 reproduces pass 1 ("strictly smaller cost wins, first x wins ties").  Pass 2 reads the
// snapshot, so the in-place update cannot race.  Columns outside the valid rectangle are masked in
// the same pass.  With SPK the final row is handed straight to the speckle filter's init step.
// ---------------------------------------------------------------------------------------------
// KT = key type: 32-bit keys (cost << 16 | x) when the cost plane is 16-bit, else 64-bit (cost << 32 | x).
// R = rows per workgroup.  With the speckle filter on (SPK) the R checked rows stay in LDS together with their
// head maps, so the R-1 row pairs inside the block are merged right here (one union per vertical contact
// segment) and only every R-th pair is left to k_spk_merge.
static int lr_rows()   // rows per workgroup when the speckle init is fused (RTDM_LR_ROWS = 1 | 2 | 4)
{
    static const int r = [] { const int v = env_int("RTDM_LR_ROWS", 1); return (v == 1 || v == 2 || v == 4) ? v : 1; }();
    return r;
}

template <bool SPK, typename CT, typename KT, int R>
__global__ __launch_bounds__(256) void k_lrcheck(Plane16W disp, const CT* cost, BMGeom g, int maxDiff16,
                                                 int32_t* label, int32_t* size, uint32_t* runs, int32_t* rowcnt,
                                                 int16_t* headmap, int spkDiff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KSH = sizeof(KT) * 4;                               // bit position of the cost inside a key
    constexpr KT NONE = (KT)~(KT)0, XMASK = ((KT)1 << KSH) - 1;
    constexpr int KB = sizeof(KT) > 4 ? 8 : 4;
    const int W = g.W, INV = g.filtered;
    int* sc = (int*)smem;                                             // W ints: keys first, then scan scratch
    KT* key = (KT*)smem;                                              // (64-bit keys need 2W ints)
    int16_t* snap = (int16_t*)(smem + (size_t)W * KB);                // W
    int16_t* finb = snap + W;                                         // R x W final rows
    int16_t* hmb = finb + (size_t)R * W;                              // R x W head maps (SPK only)
    __shared__ int wsum[4];
    const int nt = blockDim.x;
    const int f = blockIdx.z;
    const int yb = g.vy0 + blockIdx.y * R;                            // first row of this block
    const int nr = min(R, g.vy1 - yb);
    const int minX1 = max(g.minD + g.D, 0), maxX1 = W + min(g.minD, 0);
    for (int r = 0; r < nr; ++r) {
        const int y = yb + r;
        int16_t* row = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
        const CT* crow = cost + ((size_t)f * g.H + y) * g.Ws;
        int16_t* fin = finb + (size_t)r * W;
        for (int x = threadIdx.x; x < W; x += nt) { key[x] = NONE; snap[x] = row[x]; }
        __syncthreads();
        for (int x = minX1 + threadIdx.x; x < maxX1; x += nt) {
            const int d = snap[x];
            if (d == INV) continue;
            const int x2 = x - ((d + 8) >> 4);
            if (x2 < 0 || x2 >= W) continue;
            atomicMin(&key[x2], ((KT)(unsigned)crow[x] << KSH) | (KT)(unsigned)x);
        }
        __syncthreads();
        for (int x = threadIdx.x; x < W; x += nt) {
            int d = snap[x];
            bool kill = (x < g.vx0 || x >= g.vx1);
            if (!kill && d != INV && x >= minX1 && x < maxX1) {
                const int x0 = x - (d >> 4), x1 = x - ((d + 15) >> 4);
                bool bad0 = false, bad1 = false;
                if (x0 >= 0 && x0 < W && key[x0] != NONE) bad0 = abs((int)snap[(unsigned)(key[x0] & XMASK)] - d) > maxDiff16;
                if (x1 >= 0 && x1 < W && key[x1] != NONE) bad1 = abs((int)snap[(unsigned)(key[x1] & XMASK)] - d) > maxDiff16;
                kill = bad0 && bad1;
            }
            if (kill && d != INV) { row[x] = (int16_t)INV; d = INV; }
            if (SPK) fin[x] = (int16_t)d;
        }
        __syncthreads();
        if (SPK) {
            const int base = (f * g.H + y) * g.Ws;        // the keys are no longer needed: scan scratch
            // only the block's first and last row meet rows of other blocks: they alone need the global map
            spk_row_init(fin, sc, wsum, W, base, label, size, runs, rowcnt + (f * g.H + y), headmap, INV, spkDiff,
                         hmb + (size_t)r * W, r == 0 || r == R - 1 || r == nr - 1);
            __syncthreads();
        }
    }
    if (SPK) {
        // labels written above are plain stores (write-through) followed by barriers; uf_find reads with
        // agent-scope loads, so they are visible here
        for (int r = 0; r + 1 < nr; ++r) {
            const int16_t *d0 = finb + (size_t)r * W, *d1 = d0 + W, *h0 = hmb + (size_t)r * W, *h1 = h0 + W;
            const int base0 = (f * g.H + yb + r) * g.Ws, base1 = base0 + g.Ws;
            for (int x = threadIdx.x; x < W; x += nt) {
                if (!conn(d0[x], d1[x], INV, spkDiff)) continue;
                const bool dup = x > 0 && conn(d0[x - 1], d1[x - 1], INV, spkDiff) && h0[x - 1] == h0[x] && h1[x - 1] == h1[x];
                if (!dup) uf_union(label, base0 + h0[x], base1 + h1[x]);
            }
        }
    }
}

struct alignas(16) Short8 { int16_t v[8]; };

// A contact between two runs of which at least one is longer than maxSize needs no union: only "component size <= maxSize"
// is ever asked, the long run's component is large whatever else it touches, and so is the other run's -- which is MARKED
// instead (its size becomes maxSize + 1: no find, no hook; k_spk_count adds a non-root's size to its root, so the mark
// reaches the root of whatever small runs are united with it, and a later contact of a marked run marks its neighbour in
// turn).  Exact: marks only ever appear in components that contain a long run, and every short run of such a component is
// reached from the long run through contacts that either united the two trees or marked the far end.  size[] holds the run
// lengths (and marks) until k_spk_count runs.  Returns true if the contact is dealt with.
__device__ __forceinline__ bool spk_large_contact(int32_t* size, int a, int b, int maxSize)
{
    const bool la = ld_relaxed(&size[a]) > maxSize, lb = ld_relaxed(&size[b]) > maxSize;
    if (la != lb) atomicMax(&size[la ? b : a], maxSize + 1);
    return la || lb;
}

// Vector form of k_lrcheck<SPK, uint16_t, uint32_t, 1> for 16-byte-aligned rows with W % 8 == 0: one thread owns
// 8 consecutive columns, so the row, its costs, the write-back and the head map each move as ONE 128-bit access
// per thread (the scalar kernel spends its time issuing 2-byte accesses), and the run scan works on one
// aggregate per thread instead of one LDS element per column.  Same results as the scalar kernel.
// TWO (round 3): two rows per workgroup, one per HALF-wave -- lanes 0..31 of every wave take 32 chunks of row y, lanes 32..63
// the same chunks of row y + 1.  A 1280-wide row is 160 chunks = five half-waves: five full waves per row pair instead of
// three waves of which one is half empty per row (17 % of the lanes idle in a VALU-saturated kernel), and half the
// barriers per row.  Scans stop at the half-wave boundary (no row_bcast:31 step), everything else is per thread.
// NIT > 1 (SPK and TWO only): the workgroup walks NIT row pairs, 2 NIT consecutive rows, and finds the speckle filter's
// vertical contacts on the way -- while a row and the row above it are both in LDS -- for every pair of rows inside the block:
// the chunk's head record gets 8 more bits, `cand`: bit k = "the pixel at column x0 + k touches the pixel above it, and that
// contact is not the continuation of the contact to its left" (one union per contact segment).  k_spk_merge_rec then reads
// the 4-byte records only (0.5 bytes per pixel; k_spk_merge_strip read the disparity plane again and was HBM bound at
// 0.75 ms per 1024 720p pairs) and k_spk_merge_strip<1> is left with the one pair in 2 NIT that crosses two blocks.
// A thread compares ITS row (registers) with the row above: the other half-wave's current row (upper half) or the upper
// half's row of the previous trip (lower half) -- one 16-byte LDS read; the partner lane's run starts come by v_permlane32_swap.
template <bool SPK, bool TWO, int NIT>
__global__ __launch_bounds__(512) void k_lrcheck_vec(Plane16W disp, const uint16_t* cost, BMGeom g, int maxDiff16,
                                                     int32_t* label, int32_t* size, uint32_t* runs, int32_t* rowcnt,
                                                     int16_t* headmap, int spkDiff)
{
    static_assert(NIT == 1 || (SPK && TWO), "merging walks row pairs");
    constexpr bool MERGE = NIT > 1;
    constexpr int NB = MERGE ? 2 : 1;                                 // buffers of the final rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = g.W, INV = g.filtered;
    const int Wp = (W + 7) & ~7;                                      // LDS rows hold whole 8-column chunks
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = TWO ? lane >> 5 : 0;                             // which of the workgroup's rows this lane works on
    const int hl = TWO ? lane & 31 : lane;                            // lane inside the (half-)wave
    const int chunk = TWO ? wv * 32 + hl : tid;
    // per row: Wp keys: cost << 16 | (d + 0x8000); key[W] stays "none", key[W+1] takes the votes nobody uses; then the row
    // after the check (SPK; NB of them)
    const size_t per_half = (size_t)Wp * 4 + 16 + (size_t)NB * Wp * 2;
    uint32_t* key = (uint32_t*)(smem + (size_t)half * per_half);
    int16_t* fin0 = (int16_t*)(key + Wp + 4);                         // NB x Wp
    // the other half's rows: the row above an upper-half row is the lower half's current row, the row above a lower-half
    // row is the upper half's row of the previous trip
    const int16_t* ofin0 = (const int16_t*)((uint32_t*)(smem + (size_t)(half ^ 1) * per_half) + Wp + 4);
    __shared__ int wsum[2][8];
    const int x0 = chunk * 8;
    const int f = blockIdx.z;
    const int minX1 = max(g.minD + g.D, 0), maxX1 = W + min(g.minD, 0);
    // per-thread column masks (bit k = column x0 + k): inside the image / allowed to vote / inside the valid rectangle
    const auto span = [&](int lo, int hi) -> unsigned {
        const int a = min(max(lo - x0, 0), 8), b = min(max(hi - x0, 0), 8);
        return b > a ? ((1u << b) - 1u) & ~((1u << a) - 1u) : 0u;
    };
    const unsigned inimg = span(0, W), votem = span(minX1, maxX1), keepm = span(g.vx0, g.vx1);
    if (chunk == 0) { key[Wp] = ~0u; key[Wp + 1] = ~0u; }            // (W == Wp: the two extra slots lie behind the chunks)
    uint32_t prev_mine = 0;                                           // MERGE: run starts | disparity mask << 8 of the previous trip's row
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
        const int ypair = TWO ? 2 * ((int)blockIdx.y * NIT + it) : (int)blockIdx.y;   // first row of this trip, from vy0
        if (MERGE && g.vy0 + ypair >= g.vy1) break;                   // (uniform) the frame's rows end inside this block
        const int yu = g.vy0 + ypair + half;
        const bool active = x0 < W && yu < g.vy1;                     // (an odd row count leaves the last trip's second half idle)
        const int y = min(yu, g.vy1 - 1);
        const int cur = MERGE ? it & 1 : 0;
        int16_t* fin = fin0 + (size_t)cur * Wp;
        int16_t* row = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
        const uint16_t* crow = cost + ((size_t)f * g.H + y) * g.Ws;
        Short8 d8, c8;
        unsigned im = 0;                                              // bit k: d8.v[k] is a disparity (not INV)
        if (active) {
            d8 = *(const Short8*)(row + x0);
            c8 = *(const Short8*)(crow + x0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (!((inimg >> k) & 1)) d8.v[k] = (int16_t)INV;      // ragged last chunk: padding columns do not exist
                im |= (unsigned)(d8.v[k] != INV) << k;
            }
            const uint4 none = make_uint4(~0u, ~0u, ~0u, ~0u);
            ((uint4*)(key + x0))[0] = none; ((uint4*)(key + x0))[1] = none;
        }
        __syncthreads();
        // Votes and look-ups are straight-line code for all eight columns (per-column branches cost more in exec-mask
        // bookkeeping than the work they skip).  The key of a vote carries the voter's DISPARITY, not its column: among the
        // voters of one right column a smaller x means a smaller disparity (x - x2 is its rounded integer part), so the
        // minimum still prefers the lower cost and then the first voter, and a look-up has the winner's disparity without a
        // second read.  A column that may not vote, or whose target lies outside the row, votes into key[W+1]; a look-up
        // outside the row reads key[W], which nobody writes: "no vote".
        if (active) {
            const unsigned vm = im & votem;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int x = x0 + k, d = d8.v[k];
                const int x2 = x - ((d + 8) >> 4);
                const bool ok = ((vm >> k) & 1) && (unsigned)x2 < (unsigned)W;
                atomicMin(&key[ok ? x2 : W + 1], ((uint32_t)(uint16_t)c8.v[k] << 16) | ((uint32_t)(d + 0x8000) & 0xffffu));
            }
        }
        __syncthreads();
        if (active) {
            const unsigned chk = im & votem & keepm;                  // columns whose two matches are looked up
            unsigned kill = im & ~keepm;                              // outside the valid rectangle: always dropped
            // |d2 - d| > M  <=>  (unsigned)(d2 - d + M) > 2 M; unchecked columns are masked once, at the end
            const unsigned M2 = 2u * (unsigned)maxDiff16;
            const int dofs = maxDiff16 - 0x8000;
            unsigned bad0m = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int x = x0 + k, d = d8.v[k];
                const uint32_t q = key[min((unsigned)(x - (d >> 4)), (unsigned)W)];
                bad0m |= ((unsigned)(q != ~0u) & (unsigned)((unsigned)((int)(q & 0xffffu) - d + dofs) > M2)) << k;
            }
            bad0m &= chk;
            // a pixel dies only if BOTH matches disagree: the second look-ups are needed only where the first ones did
            // (consistent regions: by none of the wave's lanes)
            if (__builtin_amdgcn_ballot_w64(bad0m != 0) != 0) {
                unsigned bad1m = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int x = x0 + k, d = d8.v[k];
                    const uint32_t q = key[min((unsigned)(x - ((d + 15) >> 4)), (unsigned)W)];
                    bad1m |= ((unsigned)(q != ~0u) & (unsigned)((unsigned)((int)(q & 0xffffu) - d + dofs) > M2)) << k;
                }
                kill |= bad0m & bad1m;
            }
            if (kill) {
#pragma unroll
                for (int k = 0; k < 8; ++k) if ((kill >> k) & 1) d8.v[k] = (int16_t)INV;
                *(Short8*)(row + x0) = d8;
                im &= ~kill;
            }
            if (SPK) *(Short8*)(fin + x0) = d8;
        }
        if (!SPK) return;                                             // (NIT == 1)
        __syncthreads();
        // ---- speckle init of the finished row (what spk_row_init does, on per-thread aggregates) ----
        int left = INV, right = INV;
        if (active) { if (x0 > 0) left = fin[x0 - 1]; if (x0 + 8 < W) right = fin[x0 + 8]; }
        // cb bit k (k = 0..8): columns x0+k-1 and x0+k are connected (both disparities, close enough)
        unsigned cb = 0;
        if (active) {
            cb |= (unsigned)conn(left, d8.v[0], INV, spkDiff);
#pragma unroll
            for (int k = 1; k < 8; ++k) cb |= (unsigned)(abs((int)d8.v[k] - (int)d8.v[k - 1]) <= spkDiff) << k;
            cb &= (im & (im << 1)) | 1u;                              // bits 1..7 need both columns to be disparities
            cb |= (unsigned)conn(d8.v[7], right, INV, spkDiff) << 8;
        }
        const unsigned hm = im & ~cb & 0xffu;                         // run heads
        unsigned lm = im & ~(cb >> 1) & 0xffu;                        // run ends
        // ---- contacts between this thread's row (B, registers) and the row above it (A, LDS) ----
        unsigned cand = 0;
        if constexpr (MERGE) {
            const uint32_t mine = active ? (hm | (im << 8)) : 0u;
            const uint32_t give = half ? prev_mine : mine;            // what the partner lane (same chunk, other half) wants to see
            const auto sw = __builtin_amdgcn_permlane32_swap(give, give, false, false);   // {lower lanes' value, upper lanes' value}, in every lane
            const uint32_t above = half ? sw[0] : sw[1];              // upper half: the lower half's row; lower half: the upper half's previous row
            prev_mine = mine;
            if (active && (half == 1 || it > 0)) {
                const int16_t* afin = ofin0 + (size_t)(half == 1 ? cur : cur ^ 1) * Wp;
                const Short8 a8 = *(const Short8*)(afin + x0);
                const unsigned startA = above & 0xffu, ima = (above >> 8) & 0xffu;
                unsigned cm = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) cm |= (unsigned)(abs((int)a8.v[k] - (int)d8.v[k]) <= spkDiff) << k;
                cm &= ima & im;
                // a contact repeats the union of the contact to its left iff neither pixel of the pair starts a run
                const unsigned leftc = (cm & 1u) && x0 > 0 ? (unsigned)conn(afin[x0 - 1], left, INV, spkDiff) : 0u;
                cand = cm & ~(((cm << 1) | leftc) & ~startA & ~hm);
            }
        }
        const int agg = hm ? ((__builtin_popcount(hm) << 16) | (x0 + (31 - __builtin_clz(hm)) + 1)) : 0;
        // inclusive wave scan with DPP row shifts / row broadcasts (a __shfl_up chain is six dependent LDS-crossbar round
        // trips); lanes without a source get the identity 0
        int t = agg;
#define RTDM_SCAN(ctrl, rmask) t = OpHead::f(t, __builtin_amdgcn_update_dpp(0, t, ctrl, rmask, 0xf, false))
        RTDM_SCAN(0x111, 0xf); RTDM_SCAN(0x112, 0xf); RTDM_SCAN(0x114, 0xf); RTDM_SCAN(0x118, 0xf);   // row_shr:1,2,4,8
        RTDM_SCAN(0x142, 0xa);                                                                          // row_bcast:15
        if constexpr (!TWO) { RTDM_SCAN(0x143, 0xc); }                                                  // row_bcast:31 (whole waves only)
#undef RTDM_SCAN
        if (hl == (TWO ? 31 : 63)) wsum[half][wv] = t;
        __syncthreads();
        int run = __builtin_amdgcn_update_dpp(0, t, 0x138, 0xf, 0xf, false);                           // wave_shr:1
        if (hl == 0) run = 0;
        for (int q = 0; q < wv; ++q) run = OpHead::f(run, wsum[half][q]);
        if (!active) { if (MERGE) continue; else return; }
        const int base = (f * g.H + y) * g.Ws;
        const int hin = (run & 0xffff) - 1, cin = run >> 16;         // head and run count carried in from the left
        // head record of the chunk, one dword instead of eight head columns: (runs that start left of the chunk) | starts << 16
        // | cand << 24.  The union-find node of a run is its INDEX in the row (dense: a row's labels, sizes and run list are a few
        // contiguous lines instead of one touched sector per run head -- the count / apply passes and this kernel's own stores
        // used to scatter over the whole plane): the pixel at bit k belongs to run cin + popcount(starts at or left of k),
        // 1-based; the merge kernels rebuild the few nodes they need from that.
        ((uint32_t*)headmap)[(size_t)(f * g.H + y) * (g.Ws >> 3) + chunk] = (uint32_t)cin | (hm << 16) | (cand << 24);
        while (lm) {                                                  // one trip per run that ends in this chunk
            const int k = __builtin_ctz(lm);
            lm &= lm - 1;
            const unsigned hb = hm & ((2u << k) - 1u);                // heads at or left of the end
            const int h = hb ? x0 + (31 - __builtin_clz(hb)) : hin;
            const int node = base + cin + __builtin_popcount(hb) - 1; // (1-based index of this run in the row) - 1
            const int len = x0 + k - h + 1;
            label[node] = node;
            size[node] = len;
            runs[node] = (uint32_t)h | ((uint32_t)len << 16);
        }
        if (x0 + 8 >= W) rowcnt[f * g.H + y] = cin + __builtin_popcount(hm);
    }
}

// The unions for the contacts k_lrcheck_vec<.., NIT > 1> has found: row pairs (y - 1, y) inside its blocks of `blk` rows
// (counted from vy0).  One thread = one chunk of row y; all it reads is the row's head records -- most have cand == 0 --
// and, for the others, the record of the chunk above.  Unions go through the same LDS queue as in k_spk_merge_strip.
__global__ __launch_bounds__(256) void k_spk_merge_rec(int32_t* label, const uint32_t* heads, int Ws, int H, int vy0, int nrows, int blk, int nch,
                                                       int32_t* size, int maxSize)
{
    constexpr int QCAP = 1024;
    __shared__ int2 queue[QCAP];
    __shared__ int qn;
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    // rows r = 1 .. nrows-1 from vy0 with r % blk != 0, numbered densely: j -> r = j + j / (blk - 1) + 1
    const int inblk = blk - 1, npairs = (nrows / blk) * inblk + max(nrows % blk - 1, 0);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < npairs * nch) {
        const int j = idx / nch, chunk = idx - j * nch;
        const int y = vy0 + j + j / inblk + 1;
        const int f = blockIdx.y;
        const size_t rrow = (size_t)(f * H + y) * (Ws >> 3);
        const uint32_t cb = heads[rrow + chunk];
        unsigned cand = cb >> 24;
        if (cand) {
            const uint32_t ca = heads[rrow - (Ws >> 3) + chunk];
            const unsigned startA = (ca >> 16) & 0xffu, startB = (cb >> 16) & 0xffu;
            const int base = (f * H + y) * Ws;
            while (cand) {
                const int k = __builtin_ctz(cand);
                cand &= cand - 1;
                const unsigned msk = (2u << k) - 1u;
                const int na = base - Ws + (int)(ca & 0xffffu) + __builtin_popcount(startA & msk) - 1;
                const int nb = base + (int)(cb & 0xffffu) + __builtin_popcount(startB & msk) - 1;
                const int slot = atomicAdd(&qn, 1);
                if (slot < QCAP) queue[slot] = make_int2(na, nb);
                else uf_union(label, na, nb);
            }
        }
    }
    __syncthreads();
    const int total = min(qn, QCAP);
    for (int i = threadIdx.x; i < total; i += 256) {
        const int a = queue[i].x, b = queue[i].y;
        if (spk_large_contact(size, a, b, maxSize)) continue;
        uf_union(label, a, b);
    }
}

// ---------------------------------------------------------------------------------------------
// k_lrcheck_pk (round 3): k_lrcheck_vec<SPK, true, 1> with TWO COLUMNS PER INSTRUCTION.  k_lrcheck_vec unpacks its eight
// columns and spends ~73 VALU instructions per pixel on per-column arithmetic and on building bit masks out of compares
// (v_cmp + v_cndmask + v_or per column and test) -- it is VALU bound.  Here the row stays packed as it arrives (four dwords
// of two int16 columns): key-slot ADDRESSES are formed in packed u16 arithmetic (LDS byte addresses fit 16 bits), votes and
// look-ups take one v_perm / one shift per column to split them, the two-sided consistency test is a packed subtraction whose
// SIGN bits are the verdict, the kill is a v_bfi on the packed row, and only what the run scan needs as bit masks (the
// validity of the final row, the connected-to-the-left flags) is extracted -- eight flags at a time with two v_perm and two
// v_dot4_u32_u8.  Measured: 354 instead of 509 VALU instructions on the always-taken path, but packed ops and v_perm issue at
// 4.4 cycles where most of k_lrcheck_vec's v_and / v_or / v_add / v_sub issue at 2.6: 1.39 -> 1.32 ms per 1024 720p pairs,
// 0.527 -> 0.487 at 640x480.  NIT > 1 (thread constants formed once for NIT row pairs; RTDM_LR_PK_PAIRS) is SLOWER, with or
// without the next trip's rows requested a trip ahead (1.51-1.77 ms): 74+ VGPRs instead of 44, and what bounds the kernel is
// how many short barrier-chained workgroups a CU holds, not its instruction count (also with barriers that do not wait
// for global memory: 1.79 ms).  Timing-only ablations (LRPK_ABL, profiles/r03_lrcheck_ablation.txt): loading the two planes and
// resetting the keys alone takes 0.59 ms -- 3.8 GB at 6.4 TB/s, the HBM floor; the speckle init costs 0.25 ms, the votes 0.09.  Key slots: key[W] takes the votes nobody may see, key[W + 1] is never written
// ("no vote").  Needs: costs < 32768 (a slot's cost half is negative only in the empty slot), minD .. minD + D inside
// int16 / 16, the workgroup's LDS below 64 KB.  Same bytes as k_lrcheck_vec.
// ---------------------------------------------------------------------------------------------
typedef short lr_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short lr_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint32_t lr_lds_u32;
__device__ __forceinline__ lr_s2 lr_s(uint32_t v) { return __builtin_bit_cast(lr_s2, v); }
__device__ __forceinline__ lr_u2 lr_u(uint32_t v) { return __builtin_bit_cast(lr_u2, v); }
__device__ __forceinline__ uint32_t lr_w(lr_s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t lr_w(lr_u2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return lr_w(lr_u(a) + lr_u(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return lr_w(lr_u(a) - lr_u(b)); }
__device__ __forceinline__ uint32_t pk_subsat_u(uint32_t a, uint32_t b) { return lr_w(__builtin_elementwise_sub_sat(lr_u(a), lr_u(b))); }
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) { return lr_w(__builtin_elementwise_min(lr_u(a), lr_u(b))); }
__device__ __forceinline__ uint32_t pk_max_i(uint32_t a, uint32_t b) { return lr_w(__builtin_elementwise_max(lr_s(a), lr_s(b))); }
template <int N> __device__ __forceinline__ uint32_t pk_ashr(uint32_t a) { return lr_w(lr_s(a) >> (short)N); }
template <int N> __device__ __forceinline__ uint32_t pk_shl(uint32_t a) { return lr_w(lr_u(a) << (unsigned short)N); }
// 1 in every half of x that is zero, else 0 -- as ONE saturating packed subtraction (written as asm: from min(x, 1) or a
// compare the compiler builds two v_cmp, two v_cndmask and a v_perm)
__device__ __forceinline__ uint32_t pk_is_zero(uint32_t x)
{ uint32_t r; asm("v_pk_sub_u16 %0, 1, %1 op_sel_hi:[0,1] clamp" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t lr_bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }   // mask ? a : b (v_bfi_b32)
__device__ __forceinline__ void lr_lds_min(uint32_t addr, uint32_t v)
{ __hip_atomic_fetch_min((lr_lds_u32*)(uintptr_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lr_lds_ld(uint32_t addr) { return *(const lr_lds_u32*)(uintptr_t)addr; }
// Barrier for threads that talk through LDS only: waits for the wave's LDS operations, NOT for its global loads and stores
// (__syncthreads() is a workgroup-scope fence + s_barrier: s_waitcnt vmcnt(0) -- every barrier behind the row's write-back
// would wait for that store to reach memory, and a trip of k_lrcheck_pk is a chain of four barriers).
__device__ __forceinline__ void lr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// eight flags, one per 16-bit half of h[0..3] (each half 0 or 1), as bits 0..7 in column order
__device__ __forceinline__ unsigned lr_bits8(const uint32_t (&h)[4])
{
    const uint32_t g0 = __builtin_amdgcn_perm(h[1], h[0], 0x06040200u), g1 = __builtin_amdgcn_perm(h[3], h[2], 0x06040200u);
    return __builtin_amdgcn_udot4(g1, 0x80402010u, __builtin_amdgcn_udot4(g0, 0x08040201u, 0u, false), false);
}

#ifndef LRPK_ABL          // timing-only ablations of k_lrcheck_pk (variant builds, wrong results): 1 no votes, 2 no look-ups, 3 no speckle
#define LRPK_ABL 0        // init, 4 no run records, 5 no row loads, 6 nothing behind the loads
#endif
template <bool SPK, int NIT>
__global__ __launch_bounds__(512) void k_lrcheck_pk(Plane16W disp, const uint16_t* cost, BMGeom g, int maxDiff16,
                                                    int32_t* label, int32_t* size, uint32_t* runs, int32_t* rowcnt,
                                                    int16_t* headmap, int spkDiff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = g.W, INV = g.filtered;
    const int Wp = (W + 7) & ~7;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, hl = lane & 31;                       // half-wave = row of the pair (see k_lrcheck_vec<.., TWO>)
    const int chunk = wv * 32 + hl;
    const size_t per_half = (size_t)Wp * 4 + 16 + (size_t)Wp * 2;
    uint32_t* key = (uint32_t*)(smem + (size_t)half * per_half);      // Wp + 4 keys (two used behind the row)
    int16_t* fin = (int16_t*)(key + Wp + 4);                          // the row after the check (SPK)
    __shared__ int wsum[2][8];
    const int x0 = chunk * 8;
    const int f = blockIdx.z;
    const int minX1 = max(g.minD + g.D, 0), maxX1 = W + min(g.minD, 0);
    const uint32_t INVpk = (uint32_t)(INV & 0xffff) * 0x00010001u;
    // ---- thread constants ----
    const uint32_t KB = (uint32_t)(uintptr_t)(lr_lds_u32*)key * 0x00010001u;   // LDS byte address of key[0], in both halves
    const uint32_t TRASH = (uint32_t)W * 0x00010001u, NONE = TRASH + 0x00010001u;   // slot indices
    uint32_t X[4], VOTEM[4], KEEPM[4];                                // the columns; halves masks (0xffff / 0)
    const auto in_range = [](int x, int lo, int hi) -> uint32_t { return (x >= lo && x < hi) ? 0xffffu : 0u; };
    // lo <= x < hi  <=>  (u16)(x - lo) < hi - lo  <=>  (hi - lo) -sat (u16)(x - lo) != 0: four packed instructions per pair of
    // columns (as scalar compares and selects the masks were a third of the kernel's prologue -- which a thread pays per row)
    const uint32_t vlo = (uint32_t)(minX1 & 0xffff) * 0x00010001u, vhl = (uint32_t)max(maxX1 - minX1, 0) * 0x00010001u;
    const uint32_t klo = (uint32_t)(g.vx0 & 0xffff) * 0x00010001u, khl = (uint32_t)max(g.vx1 - g.vx0, 0) * 0x00010001u;
    const uint32_t X0 = (uint32_t)x0 * 0x00010001u + 0x00010000u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        X[k] = X0 + (uint32_t)k * 0x00020002u;
        VOTEM[k] = pk_sub(pk_is_zero(pk_subsat_u(vhl, pk_sub(X[k], vlo))), 0x00010001u);
        KEEPM[k] = pk_sub(pk_is_zero(pk_subsat_u(khl, pk_sub(X[k], klo))), 0x00010001u);
    }
    // slot index (a negative one wraps to a large u16) -> LDS byte address, anything outside the row -> the slot `lim`
    const auto slot_addr = [&](uint32_t idx, uint32_t lim) -> uint32_t { return pk_add(pk_shl<2>(pk_min_u(idx, lim)), KB); };
    const uint32_t Mpk = (uint32_t)maxDiff16 * 0x00010001u;
    const uint32_t Spk = (uint32_t)spkDiff * 0x00010001u, S2pk = Spk + Spk;
    if (chunk == 0) { key[Wp] = ~0u; key[Wp + 1] = ~0u; }            // (W == Wp: the two slots lie behind the chunks)
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
        const int ypair = 2 * ((int)blockIdx.y * NIT + it);           // first row of this trip, from vy0
        if (NIT > 1 && g.vy0 + ypair >= g.vy1) break;                 // (uniform)
        const int y0 = g.vy0 + ypair, y = y0 + half;                  // (row addresses: wave-uniform part + the half's row step)
        const bool active = x0 < W && y < g.vy1;
        int16_t* row = disp.base + ((size_t)f * disp.frame_e + (size_t)y0 * disp.pitch_e) + (half ? (uint32_t)disp.pitch_e : 0u);
        [[maybe_unused]] const uint16_t* crow = cost + ((size_t)f * g.H + y0) * g.Ws + (half ? (uint32_t)g.Ws : 0u);
        uint32_t D[4] = {INVpk, INVpk, INVpk, INVpk}, C[4] = {0, 0, 0, 0};
        if (active) {
#if LRPK_ABL == 5
            const uint4 dq = make_uint4(0x00200020u + lane, 0x00300030u, 0x00400040u + it, 0x00200020u), cq = make_uint4(lane, 5, 6, 7);
#else
            const uint4 dq = *(const uint4*)(row + x0), cq = *(const uint4*)(crow + x0);
#endif
            D[0] = dq.x; D[1] = dq.y; D[2] = dq.z; D[3] = dq.w; C[0] = cq.x; C[1] = cq.y; C[2] = cq.z; C[3] = cq.w;
            if (x0 + 8 > W) {                                         // ragged last chunk: padding columns do not exist
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t m = in_range(x0 + 2 * k, 0, W) | (in_range(x0 + 2 * k + 1, 0, W) << 16);
                    D[k] = lr_bfi(m, D[k], INVpk);
                }
            }
            const uint4 none = make_uint4(~0u, ~0u, ~0u, ~0u);
            ((uint4*)(key + x0))[0] = none; ((uint4*)(key + x0))[1] = none;
        }
        lr_lds_barrier();
#if LRPK_ABL == 6
        if (active && (D[0] ^ C[1]) == 0x12345678u) row[x0] = 1;
        continue;
#endif
        uint32_t V[4], Dx[4];                                         // halves: is a disparity (0xffff / 0); d + 0x8000
        if (active) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                V[k] = pk_sub(pk_is_zero(D[k] ^ INVpk), 0x00010001u);
                Dx[k] = D[k] ^ 0x80008000u;
                // vote into the slot of x2 = x - ((d + 8) >> 4) with the key cost << 16 | d + 0x8000 (see k_lrcheck_vec).  A column
                // that may not vote votes with the cost 0xffff: such a key only ever replaces the empty slot's, and reads as empty
                // (negative cost half) -- no address select; targets outside the row (none inside the vote range) go to key[W]
                const uint32_t cv = C[k] | ~(V[k] & VOTEM[k]);
                const uint32_t a = slot_addr(pk_sub(X[k], pk_ashr<4>(pk_add(D[k], 0x00080008u))), TRASH);
#if LRPK_ABL == 1
                if ((a ^ cv) == 0x12345678u) key[0] = a;
#else
                lr_lds_min(a & 0xffffu, __builtin_amdgcn_perm(cv, Dx[k], 0x05040100u));
                lr_lds_min(a >> 16, __builtin_amdgcn_perm(cv, Dx[k], 0x07060302u));
#endif
            }
        }
        lr_lds_barrier();
        if (active) {
            // look-up at x - (d >> 4) (and, where that disagrees, at x - ((d + 15) >> 4)): the slot's voter disagrees iff
            // |its d - d| > M: the sign of M - |difference|; an empty slot (cost half 0xffff: negative) never disagrees
            uint32_t bad0[4], any0 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t a = slot_addr(pk_sub(X[k], pk_ashr<4>(D[k])), NONE);
#if LRPK_ABL == 2
                const uint32_t qe = a, qo = ~a;
#else
                const uint32_t qe = lr_lds_ld(a & 0xffffu), qo = lr_lds_ld(a >> 16);
#endif
                const uint32_t df = pk_sub(__builtin_amdgcn_perm(qo, qe, 0x05040100u), Dx[k]);
                const uint32_t r = pk_sub(Mpk, pk_max_i(df, pk_sub(0u, df)));
                bad0[k] = r & ~__builtin_amdgcn_perm(qo, qe, 0x07060302u) & V[k] & VOTEM[k] & KEEPM[k];
                any0 |= bad0[k];
            }
            uint32_t Dn[4] = {D[0], D[1], D[2], D[3]};
            if (__builtin_amdgcn_ballot_w64((any0 & 0x80008000u) != 0) != 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t a = slot_addr(pk_sub(X[k], pk_ashr<4>(pk_add(D[k], 0x000f000fu))), NONE);
                    const uint32_t qe = lr_lds_ld(a & 0xffffu), qo = lr_lds_ld(a >> 16);
                    const uint32_t df = pk_sub(__builtin_amdgcn_perm(qo, qe, 0x05040100u), Dx[k]);
                    const uint32_t r = pk_sub(Mpk, pk_max_i(df, pk_sub(0u, df)));
                    const uint32_t killed = pk_ashr<15>(bad0[k] & r & ~__builtin_amdgcn_perm(qo, qe, 0x07060302u));   // 0xffff where both disagree
                    Dn[k] = lr_bfi(killed, INVpk, D[k]);
                }
            }
            uint32_t chg = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Dn[k] = lr_bfi(KEEPM[k], Dn[k], INVpk);               // outside the valid rectangle: always dropped
                chg |= Dn[k] ^ D[k];
                D[k] = Dn[k];
            }
            if (chg) *(uint4*)(row + x0) = make_uint4(D[0], D[1], D[2], D[3]);
            if (SPK) *(uint4*)(fin + x0) = make_uint4(D[0], D[1], D[2], D[3]);
        }
        if (!SPK) { if (NIT > 1) { lr_lds_barrier(); continue; } else return; }
        lr_lds_barrier();
        // ---- speckle init of the finished row (as in k_lrcheck_vec) ----
        unsigned im = 0, cb = 0;
        if (active && LRPK_ABL != 3) {
            const int left = x0 > 0 ? (int)fin[x0 - 1] : INV, right = x0 + 8 < W ? (int)fin[x0 + 8] : INV;
            uint32_t iz[4], cl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                iz[k] = pk_is_zero(D[k] ^ INVpk);                     // 1 = NOT a disparity
                // columns (2k - 1, 2k) and (2k, 2k + 1): |a - b| <= S  <=>  (u16)(a - b + S) <= 2 S
                const uint32_t prev = __builtin_amdgcn_alignbit(D[k], k ? D[k - 1] : ((uint32_t)left << 16), 16);
                cl[k] = pk_is_zero(pk_subsat_u(pk_add(pk_sub(D[k], prev), Spk), S2pk));   // 1 = close
            }
            im = ~lr_bits8(iz) & 0xffu;
            // cb bit k (k = 0..8): columns x0+k-1 and x0+k are connected (both disparities, close enough)
            cb = lr_bits8(cl) & (im & ((im << 1) | (unsigned)(left != INV)));
            cb |= (unsigned)conn((int)(int16_t)(D[3] >> 16), right, INV, spkDiff) << 8;
        }
        const unsigned hm = im & ~cb & 0xffu;                         // run heads
        unsigned lm = im & ~(cb >> 1) & 0xffu;                        // run ends
        const int agg = hm ? ((__builtin_popcount(hm) << 16) | (x0 + (31 - __builtin_clz(hm)) + 1)) : 0;
        int t = agg;
#define RTDM_SCAN(ctrl, rmask) t = OpHead::f(t, __builtin_amdgcn_update_dpp(0, t, ctrl, rmask, 0xf, false))
        RTDM_SCAN(0x111, 0xf); RTDM_SCAN(0x112, 0xf); RTDM_SCAN(0x114, 0xf); RTDM_SCAN(0x118, 0xf);   // row_shr:1,2,4,8
        RTDM_SCAN(0x142, 0xa);                                                                          // row_bcast:15
#undef RTDM_SCAN
        if (hl == 31) wsum[half][wv] = t;
        lr_lds_barrier();
        int run = __builtin_amdgcn_update_dpp(0, t, 0x138, 0xf, 0xf, false);                           // wave_shr:1
        if (hl == 0) run = 0;
        for (int q = 0; q < wv; ++q) run = OpHead::f(run, wsum[half][q]);
        if (active) {
            const int base = (f * g.H + y) * g.Ws;
            const int hin = (run & 0xffff) - 1, cin = run >> 16;     // head and run count carried in from the left
            ((uint32_t*)headmap)[(size_t)(f * g.H + y) * (g.Ws >> 3) + chunk] = (uint32_t)cin | (hm << 16);
            while (lm && LRPK_ABL != 4) {                             // one trip per run that ends in this chunk
                const int k = __builtin_ctz(lm);
                lm &= lm - 1;
                const unsigned hb = hm & ((2u << k) - 1u);            // heads at or left of the end
                const int h = hb ? x0 + (31 - __builtin_clz(hb)) : hin;
                const int node = base + cin + __builtin_popcount(hb) - 1;
                const int len = x0 + k - h + 1;
                label[node] = node;
                size[node] = len;
                runs[node] = (uint32_t)h | ((uint32_t)len << 16);
            }
            if (x0 + 8 >= W) rowcnt[f * g.H + y] = cin + __builtin_popcount(hm);
        }
        // (no barrier here: the next trip's first writes to the keys, the row and wsum lie behind barriers every thread only
        //  passes after its reads of this trip)
    }
}

static bool two_ok_for_pk(const BMGeom& g, int md, int spkDiff)
{
    return 2L * g.cap * g.w * g.w < 32768 && md >= 0 && md <= 8192 && spkDiff >= 0 && spkDiff <= 4096 &&
           g.minD >= -1024 && g.minD + g.D <= 1024 && g.W + 2 < 16384;
}

// Returns 0 if the head map was written per pixel (int16 head columns, nodes = head positions), else (k_lrcheck_vec: one
// record per chunk, nodes = run indices) the number of consecutive rows, counted from g.vy0, whose pairs the kernel has
// already merged (1: none).
int launch_lrcheck(Plane16W disp, const void* cost, const BMGeom& g, int disp12MaxDiff, int n,
                   hipStream_t stream, int32_t* label, int32_t* size, uint32_t* runs, int32_t* rowcnt,
                   int16_t* headmap, int spkDiff)
{
    const int md = disp12MaxDiff * 16;
    const int nrows = g.vy1 - g.vy0;
    dim3 block(256);
    const bool k32 = g.cost16 && g.W < 65536;
    const size_t kb = k32 ? 4 : 8;
#define RTDM_LR(SPK, CT, KT, RR)                                                                                     \
    hipLaunchKernelGGL((k_lrcheck<SPK, CT, KT, RR>), dim3(1, (nrows + RR - 1) / RR, n), block,                        \
                       (size_t)g.W * (kb + 2 + (SPK ? 4 * RR : 2 * RR)), stream, disp, (const CT*)cost, g, md, label, size, runs, \
                       rowcnt, headmap, spkDiff)
    const int Wp = (g.W + 7) & ~7;                      // a ragged last chunk is masked in registers; it needs padding columns
    const bool vec = k32 && (g.Ws & 7) == 0 && g.W <= 4096 && lr_rows() == 1 && disp.pitch_e >= (size_t)Wp &&   // that belong to the plane
                     (((size_t)disp.base | (disp.pitch_e * 2) | (disp.frame_e * 2) | (size_t)cost | (size_t)headmap) & 15) == 0;
    if (vec) {
        const int chunks = Wp >> 3;
        const auto per_half = [&](int nb) { return (size_t)Wp * 4 + 16 + (size_t)nb * Wp * 2; };
        static const int two_env = env_int("RTDM_LR_TWO_ROWS", 1);   // A/B: 0 = one row per workgroup (round 2)
        // row pairs a workgroup walks and merges (1: none -- every pair is left to k_spk_merge_strip, round 3's first form)
        // Two row pairs per workgroup with the vertical contacts inside the block found right here (k_lrcheck_vec<.., NIT = 2> +
        // k_spk_merge_rec): a frame's contacts are chains of dependent L2 round trips, and with half of them settled from the head
        // records the merge of one 720p frame takes 12 us instead of 29 (the whole frame 82 -> 66 us of kernels).  Measured against
        // the packed one-pair form (k_lrcheck_pk + k_spk_merge_strip<4>) over batch sizes (profiles/r03_lr_pairs_ab.txt): frames
        // up to 1024 wide -- faster or level at every batch size (640x480: -32 % at 12 pairs, -5 % at 128, level at 512); 1280
        // wide -- faster up to ~16 pairs per call (-9 % for one frame, -23 % at 4), 2-5 % slower beyond, where k_lrcheck_pk
        // streams and the merge is HBM bound.  RTDM_LR_PAIRS=1 / 2 / 4 / 8 fixes the choice (A/B).
        static const int pairs_raw = env_int("RTDM_LR_PAIRS", 0);
        static const long pairs_rows = env_int("RTDM_LR_PAIRS_ROWS", 11520);
        const int pairs_env = (pairs_raw == 1 || pairs_raw == 2 || pairs_raw == 4 || pairs_raw == 8) ? pairs_raw
                              : ((g.W <= 1024 || (long)n * nrows <= pairs_rows) ? 2 : 1);
        // two rows per workgroup where the half-wave form wastes fewer lanes than the whole-wave form and fits 512 threads
        const int waves1 = (chunks + 63) / 64, waves2 = (chunks + 31) / 32;
        static const int two_eq = env_int("RTDM_LR_TWO_EQ", 1);      // A/B: 1 = the half-wave form also where it only ties on lanes (W = 320)
        const bool two = two_env && waves2 <= 8 && nrows >= 2 && (waves2 < 2 * waves1 || (two_eq && waves2 == 2 * waves1));
        // packed form (k_lrcheck_pk): costs below 32768, every quantity of the consistency / closeness tests inside int16,
        // LDS byte addresses inside 16 bits
        static const int pk_env = env_int("RTDM_LR_PACKED", 1);      // A/B: 0 = k_lrcheck_vec; NIT from RTDM_LR_PK_PAIRS
        static const int pk_pairs = [] { const int v = env_int("RTDM_LR_PK_PAIRS", 1); return (v == 2 || v == 4 || v == 8) ? v : 1; }();
        const bool pk_ok = pk_env && two_ok_for_pk(g, md, spkDiff) && 2 * per_half(1) + 1024 < 65536;
        const bool contacts_here = two && label && nrows >= 4 && pairs_env > 1;   // k_lrcheck_vec<.., NIT > 1> below
        if (two && pk_ok && !contacts_here) {
            const dim3 vblock((unsigned)(waves2 * 64));
            const int nit = nrows >= 16 ? pk_pairs : 1;
#define RTDM_LRP(SPK, NIT) hipLaunchKernelGGL((k_lrcheck_pk<SPK, NIT>), dim3(1, (nrows + 2 * NIT - 1) / (2 * NIT), n), vblock, 2 * per_half(1), stream, disp, \
                               (const uint16_t*)cost, g, md, label, size, runs, rowcnt, headmap, spkDiff)
#define RTDM_LRP2(NIT) do { if (label) RTDM_LRP(true, NIT); else RTDM_LRP(false, NIT); } while (0)
            if (nit == 8) RTDM_LRP2(8); else if (nit == 4) RTDM_LRP2(4); else if (nit == 2) RTDM_LRP2(2); else RTDM_LRP2(1);
#undef RTDM_LRP2
#undef RTDM_LRP
            return label ? 1 : 0;
        }
        if (two) {
            const dim3 vblock((unsigned)(waves2 * 64));
            const int nit = (label && nrows >= 4) ? pairs_env : 1;
#define RTDM_LRV(SPK, NIT) hipLaunchKernelGGL((k_lrcheck_vec<SPK, true, NIT>), dim3(1, (nrows + 2 * NIT - 1) / (2 * NIT), n), vblock, \
                               2 * per_half(NIT > 1 ? 2 : 1), stream, disp, (const uint16_t*)cost, g, md, label, size, runs, rowcnt, headmap, spkDiff)
            if (!label) RTDM_LRV(false, 1);
            else if (nit == 8) RTDM_LRV(true, 8);
            else if (nit == 4) RTDM_LRV(true, 4);
            else if (nit == 2) RTDM_LRV(true, 2);
            else RTDM_LRV(true, 1);
#undef RTDM_LRV
            return label ? (nit > 1 ? 2 * nit : 1) : 0;
        }
        const dim3 vblock((unsigned)(waves1 * 64));
        const size_t lds = per_half(1);
        if (label) hipLaunchKernelGGL((k_lrcheck_vec<true, false, 1>), dim3(1, nrows, n), vblock, lds, stream, disp, (const uint16_t*)cost, g, md, label, size, runs, rowcnt, headmap, spkDiff);
        else       hipLaunchKernelGGL((k_lrcheck_vec<false, false, 1>), dim3(1, nrows, n), vblock, lds, stream, disp, (const uint16_t*)cost, g, md, label, size, runs, rowcnt, headmap, spkDiff);
        return label ? 1 : 0;
    } else if (label) {
        const int rr = lr_rows();
        if (k32) { if (rr == 4) RTDM_LR(true, uint16_t, uint32_t, 4); else if (rr == 2) RTDM_LR(true, uint16_t, uint32_t, 2); else RTDM_LR(true, uint16_t, uint32_t, 1); }
        else if (g.cost16) { if (rr == 4) RTDM_LR(true, uint16_t, unsigned long long, 4); else if (rr == 2) RTDM_LR(true, uint16_t, unsigned long long, 2); else RTDM_LR(true, uint16_t, unsigned long long, 1); }
        else { if (rr == 4) RTDM_LR(true, int32_t, unsigned long long, 4); else if (rr == 2) RTDM_LR(true, int32_t, unsigned long long, 2); else RTDM_LR(true, int32_t, unsigned long long, 1); }
    } else {
        if (k32) RTDM_LR(false, uint16_t, uint32_t, 1);
        else if (g.cost16) RTDM_LR(false, uint16_t, unsigned long long, 1);
        else RTDM_LR(false, int32_t, unsigned long long, 1);
    }
#undef RTDM_LR
    return 0;
}

int lrcheck_rows_per_block() { return lr_rows(); }

// ---------------------------------------------------------------------------------------------
// K4 speckle filter (cv::filterSpeckles as called by cv::StereoBM::compute, SURVEY.md Appendix
// A.5): 4-connected components of pixels != newVal under |a-b| <= maxDiff; components with
// size <= maxSize become newVal.  Run-based union-find:
//   init   (fused into k_lrcheck, or k_spk_init when the left-right check is off) per row: a
//          max-scan turns the "connected to my left neighbour" flags into run heads; every pixel of
//          a horizontal run is represented by its head, which starts as its own parent and carries
//          the run length; the row's runs are also listed compactly (x | len << 16).
//   merge  one workgroup per block of rows: ONE union per vertical contact segment between two
//          runs (a pixel is skipped when its left neighbour already linked the same two runs).
//   count  one wave per row, lanes over its runs: every non-root head adds its run length to its
//          root (skipped once the root is known to be large: only "<= maxSize" matters).
//   apply  one wave per row, lanes over its runs: a run whose root is small is overwritten.
// Parent pointers only ever move to smaller indices of the same component and every hook is a
// device-scope atomicMin on a root, so stale reads are still ancestors and the set of small
// components is independent of scheduling.  Components never span frames.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_spk_init(Plane16W disp, int32_t* label, int32_t* size, uint32_t* runs,
                                                  int32_t* rowcnt, int16_t* headmap, int W, int Ws, int H, int newVal, int maxDiff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* sc = (int*)smem;                 // W
    int16_t* d = (int16_t*)(sc + W);      // W
    __shared__ int wsum[4];
    const int y = blockIdx.y, f = blockIdx.z;
    const int16_t* row = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    for (int x = threadIdx.x; x < W; x += 256) d[x] = row[x];
    __syncthreads();
    spk_row_init(d, sc, wsum, W, (f * H + y) * Ws, label, size, runs, rowcnt + (f * H + y), headmap, newVal, maxDiff);
}

// merge: one thread = 8 consecutive pixels of a row pair (y, y+1); no LDS, no scans: the heads come
// from the head map.  A pixel is skipped when its left neighbour already linked the same two runs.
// VEC: rows are 16-byte aligned, so the 8 pixels of each row come in as one 128-bit load.

template <bool VEC>
__global__ __launch_bounds__(256) void k_spk_merge(Plane16W disp, int32_t* label, const int16_t* headmap, int W, int Ws, int H,
                                                   int y_lo, int npairs, int ystep, int newVal, int maxDiff)
{
    // pairs (y, y+1) for y = y_lo + k * ystep, k < npairs.  The kernel is bound by the number of load instructions
    // (a 2-byte-per-lane load costs the address unit as much as a 16-byte one), so the left-neighbour state comes
    // from lane-1's registers; only lane 0 of a wave fetches it from memory.
    const int nxb = (W + 7) / 8;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool inb = idx < nxb * npairs;
    const int cidx = inb ? idx : 0;
    const int y = y_lo + (cidx / nxb) * ystep, x0 = (cidx % nxb) * 8;
    const int f = blockIdx.y;
    const int16_t* d0 = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    const int16_t* d1 = d0 + disp.pitch_e;
    const int base0 = (f * H + y) * Ws, base1 = base0 + Ws;
    const int16_t* h0 = headmap + base0;
    const int16_t* h1 = headmap + base1;
    Short8 a8, b8, ha8, hb8;
    const bool full = VEC && x0 + 8 <= W;
    unsigned cm = 0;
    if (inb) {
        if (full) {
            a8 = *(const Short8*)(d0 + x0); b8 = *(const Short8*)(d1 + x0);
        } else {
            for (int k = 0; k < 8; ++k) { const int x = min(x0 + k, W - 1); a8.v[k] = d0[x]; b8.v[k] = d1[x]; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) cm |= (unsigned)((x0 + k < W) && conn(a8.v[k], b8.v[k], newVal, maxDiff)) << k;
    }
    ha8.v[7] = hb8.v[7] = 0;
    if (cm) {
        if (full) {
            ha8 = *(const Short8*)(h0 + x0); hb8 = *(const Short8*)(h1 + x0);
        } else {
            for (int k = 0; k < 8; ++k) { const int x = min(x0 + k, W - 1); ha8.v[k] = h0[x]; hb8.v[k] = h1[x]; }
        }
    }
    // state of the pixel left of x0: lane-1 holds it in element 7 (same row whenever x0 > 0)
    const int packed = (int)(cm >> 7) | ((int)(uint16_t)ha8.v[7] << 1) | ((int)(uint16_t)hb8.v[7] << 17);
    const int fromLeft = __shfl_up(packed, 1);
    if (!cm) return;
    bool pc = false;
    int ph0 = -1, ph1 = -1;
    if (x0 > 0) {
        if ((threadIdx.x & 63) != 0) {
            pc = fromLeft & 1; ph0 = (fromLeft >> 1) & 0xffff; ph1 = (fromLeft >> 17) & 0x7fff;
        } else {
            pc = conn(d0[x0 - 1], d1[x0 - 1], newVal, maxDiff);
            if (pc) { ph0 = h0[x0 - 1]; ph1 = h1[x0 - 1]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool c = (cm >> k) & 1;
        const int ha = ha8.v[k], hb = hb8.v[k];
        if (c && !(pc && ph0 == ha && ph1 == hb)) uf_union(label, base0 + ha, base1 + hb);
        pc = c; ph0 = ha; ph1 = hb;
    }
}

// Strip form of the merge for aligned rows: one thread walks RS consecutive row pairs of its 8 columns, so every
// row of disparities / heads is loaded once instead of twice (as the lower row of one pair and the upper row of the
// next).  Same unions as k_spk_merge<true> with ystep = 1.
// size / maxSize: a contact between two runs that are EACH longer than maxSize needs no union -- both components are
// "large" whatever else they touch, and only "size <= maxSize" is ever asked (exact; it removes most unions: disparity
// maps are made of long runs).  size[] still holds the run lengths here (k_spk_count runs afterwards).
// COMPACT: the heads come as one record per 8-column chunk and row (k_lrcheck_vec: carried head + 1 | run starts << 16)
// instead of one int16 per pixel: a quarter of the head bytes, and "same two runs as the pixel to the left" becomes bit
// arithmetic (neither pixel of the pair starts a run).
// RS == 1: the pairs are (y, y + 1) for y = y_lo + k * ystep -- what is left when k_lrcheck_vec has merged the pairs inside
// blocks of ystep rows itself.
// REC: the first ra.blocks workgroups do k_spk_merge_rec's work instead (the contacts k_lrcheck_vec<.., NIT > 1> left in the head
// records: row pairs inside its blocks of rows) -- same queue, same drain, one launch less for a single frame.
struct MergeRecArgs { int blocks, vy0, nrows, blk, nch; };
template <int RS, bool COMPACT, bool REC = false>
__global__ __launch_bounds__(256) void k_spk_merge_strip(Plane16W disp, int32_t* label, const int16_t* headmap, int W, int Ws, int H,
                                                         int y_lo, int npairs, int newVal, int maxDiff, int32_t* size, int maxSize,
                                                         int ystep, MergeRecArgs ra)
{
    // contacts found by the 256 threads are queued in LDS and united afterwards by the first threads, one union per
    // lane: a union is a chain of dependent global accesses, and a wave with a single busy lane stalls as long as a full one
    constexpr int QCAP = 1024;
    __shared__ int2 queue[QCAP];
    __shared__ int qn;
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const bool rec_block = REC && (int)blockIdx.x < ra.blocks;            // workgroup-uniform
    if (rec_block) {
        // (k_spk_merge_rec's collection) rows r = 1 .. nrows-1 from vy0 with r % blk != 0, numbered densely: j -> r = j + j / (blk - 1) + 1
        const int inblk = ra.blk - 1, nprs = (ra.nrows / ra.blk) * inblk + max(ra.nrows % ra.blk - 1, 0);
        const int ridx = blockIdx.x * 256 + threadIdx.x;
        if (ridx < nprs * ra.nch) {
            const int j = ridx / ra.nch, chunk = ridx - j * ra.nch;
            const int y = ra.vy0 + j + j / inblk + 1;
            const int f = blockIdx.y;
            const uint32_t* heads = (const uint32_t*)headmap;
            const size_t rrow = (size_t)(f * H + y) * (Ws >> 3);
            const uint32_t cb = heads[rrow + chunk];
            unsigned cand = cb >> 24;
            if (cand) {
                const uint32_t ca = heads[rrow - (Ws >> 3) + chunk];
                const unsigned startA = (ca >> 16) & 0xffu, startB = (cb >> 16) & 0xffu;
                const int base = (f * H + y) * Ws;
                while (cand) {
                    const int k = __builtin_ctz(cand);
                    cand &= cand - 1;
                    const unsigned msk = (2u << k) - 1u;
                    const int na = base - Ws + (int)(ca & 0xffffu) + __builtin_popcount(startA & msk) - 1;
                    const int nb = base + (int)(cb & 0xffffu) + __builtin_popcount(startB & msk) - 1;
                    const int slot = atomicAdd(&qn, 1);
                    if (slot < QCAP) queue[slot] = make_int2(na, nb);
                    else uf_union(label, na, nb);
                }
            }
        }
    }
    const int nxb = (W + 7) >> 3;                         // a ragged last chunk reads the plane's padding columns and masks them
    const int nstrips = (npairs + RS - 1) / RS;
    const int idx = ((int)blockIdx.x - (REC ? ra.blocks : 0)) * 256 + threadIdx.x;
    const bool inb = !rec_block && idx < nxb * nstrips;
    const int cidx = inb ? idx : 0;
    const int strip = cidx / nxb, x0 = (cidx % nxb) * 8;
    const int y = y_lo + strip * (RS == 1 ? ystep : RS);
    const int nr = inb ? min(RS, npairs - strip * RS) : 0;
    const int f = blockIdx.y;
    const int16_t* d = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e + x0;
    int base = (f * H + y) * Ws;
    const int16_t* h = headmap + base + x0;
    const unsigned colmask = x0 + 8 <= W ? 0xffu : (0xffu >> (x0 + 8 - W));
    Short8 a8, b8, ha8, hb8;
    if (inb) a8 = *(const Short8*)d;
    bool ha_loaded = false;
    const uint32_t* hc = (const uint32_t*)headmap + (size_t)(f * H + y) * (Ws >> 3) + (x0 >> 3);   // COMPACT: this chunk's records
    uint32_t ca = (COMPACT && inb) ? hc[0] : 0u;
    // COMPACT: the strip's RS further rows and head records are requested up front (with a load, a wait and the union
    // queue's atomics per row, a wave kept one row in flight); rows past the strip's end repeat its last one and are not used
    Short8 rows[COMPACT ? RS : 1];
    uint32_t heads[COMPACT ? RS : 1];
    if constexpr (COMPACT) {
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int rr = r < nr ? r + 1 : nr;
            rows[r] = a8; heads[r] = 0u;
            if (inb) { rows[r] = *(const Short8*)(d + (size_t)rr * disp.pitch_e); heads[r] = hc[(size_t)rr * (Ws >> 3)]; }
        }
    }
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        unsigned cm = 0;
        if (r < nr) {
            if constexpr (COMPACT) b8 = rows[r]; else b8 = *(const Short8*)(d + disp.pitch_e);
#pragma unroll
            for (int k = 0; k < 8; ++k) cm |= (unsigned)conn(a8.v[k], b8.v[k], newVal, maxDiff) << k;
            cm &= colmask;
        }
        if constexpr (COMPACT) {
            const uint32_t cb = r < nr ? heads[r] : 0u;
            // contact bit of the pixel left of the chunk: lane-1's bit 7, or (first lane of a wave) from memory
            unsigned leftc = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(cm >> 7), 0x138, 0xf, 0xf, false);   // wave_shr:1
            if ((threadIdx.x & 63) == 0) leftc = (cm && x0 > 0) ? (unsigned)conn(d[-1], d[disp.pitch_e - 1], newVal, maxDiff) : 0u;
            if (x0 == 0) leftc = 0;                             // (lane-1 belongs to another strip there)
            const unsigned startA = (ca >> 16) & 0xffu, startB = (cb >> 16) & 0xffu;
            // a contact repeats the union of the contact to its left iff neither pixel of the pair starts a run
            unsigned cand = cm & ~(((cm << 1) | (leftc & 1u)) & ~startA & ~startB);
            while (cand) {
                const int k = __builtin_ctz(cand);
                cand &= cand - 1;
                // nodes = run indices (k_lrcheck_vec): runs that start left of the chunk + starts at or left of the pixel, - 1
                const unsigned ma = startA & ((2u << k) - 1u), mb = startB & ((2u << k) - 1u);
                const int ha = (int)(ca & 0xffffu) + __builtin_popcount(ma) - 1;
                const int hb = (int)(cb & 0xffffu) + __builtin_popcount(mb) - 1;
                const int slot = atomicAdd(&qn, 1);
                if (slot < QCAP) queue[slot] = make_int2(base + ha, base + Ws + hb);
                else uf_union(label, base + ha, base + Ws + hb);
            }
            ca = cb;
        } else {
        hb8.v[7] = 0;
        if (cm) {
            if (!ha_loaded) ha8 = *(const Short8*)h;
            hb8 = *(const Short8*)(h + Ws);
        } else ha8.v[7] = 0;
        const int packed = (int)(cm >> 7) | ((int)(uint16_t)ha8.v[7] << 1) | ((int)(uint16_t)hb8.v[7] << 17);
        const int fromLeft = __shfl_up(packed, 1);
        if (cm) {
            bool pc = false;
            int ph0 = -1, ph1 = -1;
            if (x0 > 0) {
                if ((threadIdx.x & 63) != 0) {
                    pc = fromLeft & 1; ph0 = (fromLeft >> 1) & 0xffff; ph1 = (fromLeft >> 17) & 0x7fff;
                } else {
                    pc = conn(d[-1], d[disp.pitch_e - 1], newVal, maxDiff);
                    if (pc) { ph0 = h[-1]; ph1 = h[Ws - 1]; }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool c = (cm >> k) & 1;
                const int ha = ha8.v[k], hb = hb8.v[k];
                if (c && !(pc && ph0 == ha && ph1 == hb)) {
                    const int slot = atomicAdd(&qn, 1);
                    if (slot < QCAP) queue[slot] = make_int2(base + ha, base + Ws + hb);
                    else uf_union(label, base + ha, base + Ws + hb);
                }
                pc = c; ph0 = ha; ph1 = hb;
            }
        }
        ha8 = hb8; ha_loaded = cm != 0;
        }
        a8 = b8;
        d += disp.pitch_e; h += Ws; base += Ws;
    }
    __syncthreads();
#ifndef MERGE_ABL          // timing-only ablations (variant builds): 1 no unions at all, 2 marks but no unions, 3 no queue either
#define MERGE_ABL 0
#endif
    const int total = MERGE_ABL == 1 || MERGE_ABL == 3 ? 0 : min(qn, QCAP);
    for (int i = threadIdx.x; i < total; i += 256) {
        const int a = queue[i].x, b = queue[i].y;
        if (MERGE_ABL == 2) { spk_large_contact(size, a, b, maxSize); continue; }
        uf_union_contact(label, size, a, b, maxSize);
    }
}

// DENSE: the node of a run is base + its index in the row (k_lrcheck_vec), else base + the x of its head (spk_row_init).
template <bool DENSE>
__global__ __launch_bounds__(256) void k_spk_count(int32_t* label, int32_t* size, const uint32_t* runs,
                                                   const int32_t* rowcnt, int Ws, int nrows, int maxSize)
{
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);        // 16 lanes per row of the batch: rows have a few dozen runs
    if (row >= nrows) return;                                   // and every lane's work is a chain of dependent loads
    const int cnt = rowcnt[row], base = row * Ws;
    for (int i = threadIdx.x & 15; i < cnt; i += 16) {
        const uint32_t rn = runs[base + i];
        const int idx = base + (DENSE ? i : (int)(rn & 0xffffu));
        const int root = uf_find(label, idx);
        if (root == idx) continue;
        st_relaxed(&label[idx], root);             // roots are final in this launch
        // a non-root's size is its run length, or maxSize + 1 if the merge marked it (spk_large_contact): nobody adds to it
        if (ld_relaxed(&size[root]) <= maxSize) atomicAdd(&size[root], ld_relaxed(&size[idx]));
    }
}

template <bool DENSE>
__global__ __launch_bounds__(256) void k_spk_apply(Plane16W disp, const int32_t* label, const int32_t* size,
                                                   const uint32_t* runs, const int32_t* rowcnt, int Ws, int H, int nrows,
                                                   int newVal, int maxSize)
{
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);        // 16 lanes per row, as in k_spk_count
    if (row >= nrows) return;
    const int cnt = rowcnt[row], base = row * Ws;
    const int f = row / H, y = row - f * H;
    int16_t* drow = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    for (int i = threadIdx.x & 15; i < cnt; i += 16) {
        const uint32_t rn = runs[base + i];
        const int x = (int)(rn & 0xffffu), len = (int)(rn >> 16);
        if (len > maxSize) continue;               // a run longer than the limit is in a large component by itself
        // after k_spk_count a head is at most a couple of hops from its root (a late path-halving
        // store of another thread may have left an ancestor instead of the root), so chase it
        int root = base + (DENSE ? i : x);
        for (int p = label[root]; p != root; p = label[root]) root = p;
        if (size[root] <= maxSize)
            for (int k = 0; k < len; ++k) drow[x + k] = (int16_t)newVal;
    }
}

// label/size/runs/headmap: n*W*H elements each; rowcnt: n*H.  If init_done, the rows [y_lo, y_hi) were
// initialised by k_lrcheck<true> (rowcnt was zeroed before it) and no other row holds a valid pixel.
void launch_speckle(Plane16W disp, int32_t* label, int32_t* size, uint32_t* runs, int32_t* rowcnt, int16_t* headmap,
                    int W, int Ws, int H, int n, int newVal, int maxSize, int maxDiff, bool init_done, int premerged_rows,
                    int y_lo, int y_hi, hipStream_t stream, bool compact_heads)
{
    dim3 block(256);
    if (!init_done) {
        y_lo = 0; y_hi = H; premerged_rows = 1;
        hipLaunchKernelGGL(k_spk_init, dim3(1, H, n), block, (size_t)W * 6, stream, disp, label, size, runs, rowcnt, headmap,
                           W, Ws, H, newVal, maxDiff);
    }
    // row pairs still to merge: (y, y+1), y = first + k*step.  The init pass may already have merged the pairs
    // inside blocks of premerged_rows rows (k_lrcheck<SPK>); then only the pairs across blocks remain.
    const int step = premerged_rows > 1 ? premerged_rows : 1;
    const int first = y_lo + step - 1;
    const int last = min(y_hi, H) - 2;               // last y with y+1 initialised
    const int npairs = last >= first ? (last - first) / step + 1 : 0;
    MergeRecArgs rec{};
    static const int fuse_rec = env_int("RTDM_MERGE_REC_FUSED", 1);      // A/B: 0 = k_spk_merge_rec as a launch of its own
    if (compact_heads && step > 1) {               // the contacts inside blocks of `step` rows are in the head records (k_lrcheck_vec<.., NIT > 1>)
        const int nr = min(y_hi, H) - y_lo, inside = (nr / step) * (step - 1) + max(nr % step - 1, 0), nxb = (W + 7) / 8;
        if (inside > 0) {
            if (fuse_rec && npairs > 0) rec = MergeRecArgs{(inside * nxb + 255) / 256, y_lo, nr, step, nxb};   // rides in k_spk_merge_strip<1, true, true> below
            else hipLaunchKernelGGL(k_spk_merge_rec, dim3((inside * nxb + 255) / 256, n), block, 0, stream, label, (const uint32_t*)headmap, Ws, H, y_lo, nr, step, nxb,
                                    size, maxSize);
        }
    }
    if (npairs > 0) {
        const int nxb = (W + 7) / 8;
        const bool vec = (((size_t)disp.base | (disp.pitch_e * 2) | (disp.frame_e * 2)) & 15) == 0 && (Ws & 7) == 0 &&
                         disp.pitch_e >= (size_t)((W + 7) & ~7);   // a ragged last chunk reads (never writes) padding columns
        dim3 grid((nxb * npairs + 255) / 256, n);
        static const int rs = env_int("RTDM_MERGE_STRIP", 4);
        if (compact_heads && step > 1) {           // k_lrcheck_vec<.., NIT > 1> has found the contacts inside blocks of `step` rows
            dim3 sgrid((nxb * npairs + 255) / 256 + rec.blocks, n);
            if (rec.blocks) hipLaunchKernelGGL((k_spk_merge_strip<1, true, true>), sgrid, block, 0, stream, disp, label, headmap, W, Ws, H, first, npairs, newVal, maxDiff, size, maxSize, step, rec);
            else hipLaunchKernelGGL((k_spk_merge_strip<1, true>), sgrid, block, 0, stream, disp, label, headmap, W, Ws, H, first, npairs, newVal, maxDiff, size, maxSize, step, MergeRecArgs{});
        } else
        if (compact_heads) {                       // written by k_lrcheck_vec, whose alignment conditions imply `vec`
            // Strips of four row pairs read every row 1.25 times instead of twice -- what a batch wants (the kernel streams the
            // plane) -- but a single frame is 111 workgroups whose threads each work through up to four queued unions, chains of
            // dependent L2 round trips: 32 us, more than the frame's search.  Small launches take shorter strips: more workgroups,
            // one union per thread.
            static const int rs_env = env_int("RTDM_MERGE_RS", 0);    // A/B: 1 / 2 / 4 fixes the strip length
            int rsc = 4;
            while (rsc > 1 && (long)((nxb * ((npairs + rsc - 1) / rsc) + 255) / 256) * n < 1024) rsc >>= 1;
            if (rs_env == 1 || rs_env == 2 || rs_env == 4) rsc = rs_env;
            dim3 sgrid((nxb * ((npairs + rsc - 1) / rsc) + 255) / 256, n);
            if (rsc == 4)      hipLaunchKernelGGL((k_spk_merge_strip<4, true>), sgrid, block, 0, stream, disp, label, headmap, W, Ws, H, first, npairs, newVal, maxDiff, size, maxSize, 1, MergeRecArgs{});
            else if (rsc == 2) hipLaunchKernelGGL((k_spk_merge_strip<2, true>), sgrid, block, 0, stream, disp, label, headmap, W, Ws, H, first, npairs, newVal, maxDiff, size, maxSize, 1, MergeRecArgs{});
            else               hipLaunchKernelGGL((k_spk_merge_strip<1, true>), sgrid, block, 0, stream, disp, label, headmap, W, Ws, H, first, npairs, newVal, maxDiff, size, maxSize, 1, MergeRecArgs{});
        } else
        if (vec && step == 1 && rs > 1) {
            const int RSV = rs >= 8 ? 8 : 4;
            dim3 sgrid((nxb * ((npairs + RSV - 1) / RSV) + 255) / 256, n);
            if (RSV == 8) hipLaunchKernelGGL((k_spk_merge_strip<8, false>), sgrid, block, 0, stream, disp, label, headmap, W, Ws, H, first, npairs, newVal, maxDiff, size, maxSize, 1, MergeRecArgs{});
            else          hipLaunchKernelGGL((k_spk_merge_strip<4, false>), sgrid, block, 0, stream, disp, label, headmap, W, Ws, H, first, npairs, newVal, maxDiff, size, maxSize, 1, MergeRecArgs{});
        } else
        if (vec) hipLaunchKernelGGL(k_spk_merge<true>, grid, block, 0, stream, disp, label, headmap, W, Ws, H, first, npairs, step, newVal, maxDiff);
        else     hipLaunchKernelGGL(k_spk_merge<false>, grid, block, 0, stream, disp, label, headmap, W, Ws, H, first, npairs, step, newVal, maxDiff);
    }
    const int nrows = n * H;
    if (compact_heads) {                          // k_lrcheck_vec: nodes are run indices
        hipLaunchKernelGGL(k_spk_count<true>, dim3((nrows + 15) / 16), block, 0, stream, label, size, runs, rowcnt, Ws, nrows, maxSize);
        hipLaunchKernelGGL(k_spk_apply<true>, dim3((nrows + 15) / 16), block, 0, stream, disp, label, size, runs, rowcnt, Ws, H, nrows, newVal, maxSize);
    } else {
        hipLaunchKernelGGL(k_spk_count<false>, dim3((nrows + 15) / 16), block, 0, stream, label, size, runs, rowcnt, Ws, nrows, maxSize);
        hipLaunchKernelGGL(k_spk_apply<false>, dim3((nrows + 15) / 16), block, 0, stream, disp, label, size, runs, rowcnt, Ws, H, nrows, newVal, maxSize);
    }
}

}  // namespace rtdm
