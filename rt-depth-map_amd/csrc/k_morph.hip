// k_morph.hip -- K5: morphological open + close with the 10x10 MORPH_ELLIPSE element, the device
// side of SWMorphologicalFilter::run (/root/reference/filter/mf-sw.cpp:19-28; element size from
// /root/reference/include/filter/mf-sw.h:11-12).  Semantics: SURVEY.md Appendix B; oracle:
// oracle/morph_oracle.c.
//
// ONE launch for all four passes (erode, dilate, dilate, erode).  A workgroup owns a 64x32 output
// tile; the source tile with a (20 up/left, 16 down/right) halo is staged in LDS once and each
// pass shrinks the live region by the element's reach (5 up/left, 4 down/right), so nothing but
// the input and the final output touches HBM (algorithmic bytes: 1 read + 1 write per pixel).
// The element's ten rows are runs of length 1, 7, 9, 10 x5, 9, 7 (Appendix B), so each pass first
// builds horizontal running minima (maxima) of length 7, 9 and 10 per row, then combines ten values
// vertically.  Threads work on four pixels at a time: aligned dword LDS accesses (≈6 per pixel and pass instead of the 84
// byte reads of the direct form), bytes widened to (u16, u16) pairs by v_perm so that one v_pk_min/max_u16 serves two pixels.
// Out-of-image samples never win: they are staged as the neutral value of the pass that reads them
// (255 before an erosion, 0 before a dilation), which is OpenCV's default constant border.
#include "rtdm_kernels.h"

namespace rtdm {

static constexpr int TW = 64, TH = 32;          // output tile
static constexpr int RL = 5, RR = 4;            // reach of one pass: left/up, right/down
static constexpr int SW = TW + 4 * (RL + RR);   // 100 staged columns
static constexpr int SH = TH + 4 * (RL + RR);   // 68 staged rows
static constexpr int SP = 104;                  // row pitch of the LDS planes

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));

// elementwise min / max of two (u16, u16) pairs: v_pk_min_u16 / v_pk_max_u16
template <bool DILATE> __device__ __forceinline__ uint32_t pm(uint32_t a, uint32_t b)
{
    const us2_t x = __builtin_bit_cast(us2_t, a), y = __builtin_bit_cast(us2_t, b);
    return __builtin_bit_cast(uint32_t, DILATE ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y));
}
// bytes i and i+1 (0 <= i <= 6) of the 8 bytes hi:lo as a (u16, u16) pair
__device__ __forceinline__ uint32_t pair_at(uint32_t hi, uint32_t lo, int i)
{ return __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)i | ((uint32_t)(i + 1) << 16)); }
// (m0, m1), (m2, m3) -> bytes m0 m1 m2 m3
__device__ __forceinline__ uint32_t pack4(uint32_t p01, uint32_t p23) { return __builtin_amdgcn_perm(p23, p01, 0x06040200u); }

// One pass over the live region: input plane `in` valid on columns [c0, c0+cw) x rows [r0, r0+rh)
// (tile-local coordinates), output written to `out` on the region shrunk by the reach.
// Every thread produces FOUR consecutive pixels at a time from aligned dword accesses; the bytes are widened to
// (u16, u16) pairs with v_perm so that one v_pk_min/max_u16 serves two pixels.  Groups that straddle the edge of the
// live region also compute a few dead pixels: nothing reads those.
template <bool DILATE>
__device__ __forceinline__ void morph_pass(const uint8_t* in, uint8_t* out, uint8_t* h7, uint8_t* h9, uint8_t* h10,
                                           int c0, int r0, int cw, int rh, int gx0, int gy0, int W, int H, int next_neutral,
                                           bool interior)
{
    // horizontal running extrema; h7[x] covers [x, x+7), h9 [x, x+9), h10 [x, x+10)
    constexpr int NG = SP / 4;                                  // 4-pixel groups per row
    for (int i = threadIdx.x; i < rh * NG; i += 256) {
        const int y = r0 + i / NG, x0 = 4 * (i % NG);
        const uint32_t* p = (const uint32_t*)(in + y * SP + x0);
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
        // V[j] = (b_j, b_j+1), j = 0..11
        const uint32_t v0 = pair_at(d1, d0, 0), v1 = pair_at(d1, d0, 1), v2 = pair_at(d1, d0, 2), v3 = pair_at(d1, d0, 3);
        const uint32_t v4 = pair_at(d2, d1, 0), v5 = pair_at(d2, d1, 1), v6 = pair_at(d2, d1, 2), v7 = pair_at(d2, d1, 3);
        const uint32_t v8 = pair_at(d3, d2, 0), v9 = pair_at(d3, d2, 1), v10 = pair_at(d3, d2, 2), v11 = pair_at(d3, d2, 3);
        const uint32_t c = pm<DILATE>(pm<DILATE>(v2, v3), pm<DILATE>(pm<DILATE>(v4, v5), v6));          // b2..b7 over both lanes
        const uint32_t a7 = pm<DILATE>(c, pm<DILATE>(v0, v1));                                            // (m7[0], m7[1])
        const uint32_t b7 = pm<DILATE>(c, pm<DILATE>(v7, v8));                                            // (m7[2], m7[3])
        const uint32_t a9 = pm<DILATE>(a7, pm<DILATE>(v7, v8)), b9 = pm<DILATE>(b7, pm<DILATE>(v9, v10));
        const uint32_t a10 = pm<DILATE>(a9, v9), b10 = pm<DILATE>(b9, v11);
        *(uint32_t*)(h7 + y * SP + x0) = pack4(a7, b7);
        *(uint32_t*)(h9 + y * SP + x0) = pack4(a9, b9);
        *(uint32_t*)(h10 + y * SP + x0) = pack4(a10, b10);
    }
    __syncthreads();
    // output (y, x) for x in [c0+5, c0+cw-4), y in [r0+5, r0+rh-4):  element row i reads source row
    // y+i-5, columns x+j-5 for j in its run (Appendix B): 5 | 2..8 | 1..9 | 0..9 x5 | 1..9 | 2..8
    const int ow = cw - (RL + RR), oh = rh - (RL + RR);
    const int xg0 = (c0 + RL) & ~3;                               // first aligned group that holds an output column
    const int ng = (c0 + RL + ow + 3 - xg0) >> 2;
    for (int i = threadIdx.x; i < ng * oh; i += 256) {
        const int y = r0 + RL + i / ng, x0 = xg0 + 4 * (i % ng);
        const auto row2 = [&](const uint8_t* plane, int yy, int back, uint32_t& lo, uint32_t& hi) {
            // pixels x0-back .. x0-back+3 of row yy as two pairs; back in {0, 3, 4, 5}
            const uint32_t* q = (const uint32_t*)(plane + yy * SP + x0);
            if (back == 0) { const uint32_t d = q[0]; lo = pair_at(0, d, 0); hi = pair_at(0, d, 2); }
            else if (back == 4) { const uint32_t d = q[-1]; lo = pair_at(0, d, 0); hi = pair_at(0, d, 2); }
            else if (back == 3) { const uint32_t dl = q[-1], dh = q[0]; lo = pair_at(dh, dl, 1); hi = pair_at(dh, dl, 3); }
            else { const uint32_t dl = q[-2], dh = q[-1]; lo = pair_at(dh, dl, 3); hi = pair_at(dh, dl, 5); }
        };
        uint32_t m01, m23, t01, t23;
        row2(in, y - 5, 0, m01, m23);
        row2(h7, y - 4, 3, t01, t23);  m01 = pm<DILATE>(m01, t01); m23 = pm<DILATE>(m23, t23);
        row2(h9, y - 3, 4, t01, t23);  m01 = pm<DILATE>(m01, t01); m23 = pm<DILATE>(m23, t23);
#pragma unroll
        for (int k = -2; k <= 2; ++k) { row2(h10, y + k, 5, t01, t23); m01 = pm<DILATE>(m01, t01); m23 = pm<DILATE>(m23, t23); }
        row2(h9, y + 3, 4, t01, t23);  m01 = pm<DILATE>(m01, t01); m23 = pm<DILATE>(m23, t23);
        row2(h7, y + 4, 3, t01, t23);  m01 = pm<DILATE>(m01, t01); m23 = pm<DILATE>(m23, t23);
        uint32_t res = pack4(m01, m23);
        if (!interior) {                                          // outside the image: never wins the next pass
            const int gy = gy0 + y;
            const bool rowout = gy < 0 || gy >= H;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int gx = gx0 + x0 + k;
                if (rowout || gx < 0 || gx >= W) res = (res & ~(0xffu << (8 * k))) | ((uint32_t)next_neutral << (8 * k));
            }
        }
        *(uint32_t*)(out + y * SP + x0) = res;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_morph_open_close(Plane8 in, Plane8W out, int W, int H)
{
    // 16 bytes of slack behind every plane: the last group of the last row reads one dword too far
    __shared__ __attribute__((aligned(16))) uint8_t A[SH * SP + 16], B[SH * SP + 16], h7[SH * SP + 16], h9[SH * SP + 16], h10[SH * SP + 16];
    const int f = blockIdx.z;
    const int gx0 = blockIdx.x * TW - 4 * RL, gy0 = blockIdx.y * TH - 4 * RL;   // image coords of tile-local (0,0)
    const bool interior = gx0 >= 0 && gy0 >= 0 && gx0 + SW <= W && gy0 + SH <= H;   // the whole staged region is inside the image
    const uint8_t* src = in.base + (size_t)f * in.frame;
    const bool al_in = (((size_t)src | in.pitch) & 3) == 0;       // gx0 is a multiple of 4 by construction
    if (interior && al_in) {                                       // whole dwords: 25 per staged row
        for (int i = threadIdx.x; i < SH * (SW / 4); i += 256) {
            const int y = i / (SW / 4), xq = i - y * (SW / 4);
            *(uint32_t*)(A + y * SP + 4 * xq) = *(const uint32_t*)(src + (size_t)(gy0 + y) * in.pitch + gx0 + 4 * xq);
        }
    } else {
        for (int i = threadIdx.x; i < SH * SW; i += 256) {
            const int y = i / SW, x = i - y * SW;
            const int gx = gx0 + x, gy = gy0 + y;
            A[y * SP + x] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? src[(size_t)gy * in.pitch + gx] : (uint8_t)255;
        }
    }
    __syncthreads();
    const int S = RL + RR;
    morph_pass<false>(A, B, h7, h9, h10, 0, 0, SW, SH, gx0, gy0, W, H, 0, interior);                               // erode  -> dilate next
    morph_pass<true>(B, A, h7, h9, h10, RL, RL, SW - S, SH - S, gx0, gy0, W, H, 0, interior);                      // dilate -> dilate next
    morph_pass<true>(A, B, h7, h9, h10, 2 * RL, 2 * RL, SW - 2 * S, SH - 2 * S, gx0, gy0, W, H, 255, interior);    // dilate -> erode next
    morph_pass<false>(B, A, h7, h9, h10, 3 * RL, 3 * RL, SW - 3 * S, SH - 3 * S, gx0, gy0, W, H, 0, interior);     // erode  -> final
    uint8_t* dst = out.base + (size_t)f * out.frame;
    const bool al_out = (((size_t)dst | out.pitch) & 3) == 0;
    if (al_out && (int)(blockIdx.x + 1) * TW <= W && (int)(blockIdx.y + 1) * TH <= H) {     // full tile: 16 dwords per row
        for (int i = threadIdx.x; i < (TW / 4) * TH; i += 256) {
            const int y = i / (TW / 4), xq = i - y * (TW / 4);
            *(uint32_t*)(dst + (size_t)(blockIdx.y * TH + y) * out.pitch + blockIdx.x * TW + 4 * xq) =
                *(const uint32_t*)(A + (y + 4 * RL) * SP + 4 * RL + 4 * xq);
        }
    } else {
        for (int i = threadIdx.x; i < TW * TH; i += 256) {
            const int y = i / TW, x = i - y * TW;
            const int gx = blockIdx.x * TW + x, gy = blockIdx.y * TH + y;
            if (gx < W && gy < H) dst[(size_t)gy * out.pitch + gx] = A[(y + 4 * RL) * SP + x + 4 * RL];
        }
    }
}

void launch_morph_open_close(Plane8 in, Plane8W out, uint8_t*, uint8_t*, int W, int H, int n, hipStream_t stream)
{
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, n);
    hipLaunchKernelGGL(k_morph_open_close, grid, dim3(256), 0, stream, in, out, W, H);
}

}  // namespace rtdm
