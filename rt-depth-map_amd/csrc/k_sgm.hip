// k_sgm.hip -- the device counterpart of the reference's SWSemiGlobalMatcher (/root/reference/stereo-matcher/
// sgbm-sw.cpp:12-37 -> cv::StereoSGBM, P1 = 600, P2 = 2400): paths = 5 is the library's MODE_SGBM (what sgbm-sw.cpp:15
// creates), paths = 8 its MODE_HH (BASELINE config 5, "8-path").  The algorithm is the restatement in
// oracle/sgm_oracle.c (rules R1-R12 there; integer arithmetic, tolerance 0 against that oracle; parity against a real
// cv::StereoSGBM is unpinned).  Cost volumes live on the column domain [x0, x1) = [minD+D, W+min(minD,0)) and are laid
// out [frame][y][x - x0][d] with d fastest, so a wavefront's lanes = consecutive disparities = one coalesced line per pixel.
//
//   k_sgm_bounds x-Sobel (vertical edge replication) clipped to +-15, + 15, and the BT bounds of it and of the
//                intensity (border columns overwritten with 15, R1), per pixel          (HBM bound)
//   k_sgm_pixbox Birchfield-Tomasi pixel cost (gradient + (intensity >> 2), u8, kept in LDS) and the blockSize x blockSize sum
//                with clamped coordinates -> C (u16), windows <= 7; larger windows: k_sgm_pix (u8 volume) + k_sgm_box / _any
//   k_sgm_path_h one HALF-WAVE per path line, packed u16 recurrence: L_r, S = min(S + L_r, 32767) (R5); the two horizontal
//                directions in one launch (-> S, <- S2)
//   k_sgm_sweep  the three directions that advance a row per step in one row-synchronous pass (adds S2; the last sweep decides
//                the winners: wave minimum, uniqueness, quadratic sub-pixel) -> 8 bytes per pixel
//   k_sgm_lrfinal one workgroup per row: the votes of the integer winners (LDS, right-most voter wins ties: R7) and the
//                always-on left-right check (R9)
//   k_sgm_median 3x3 median with clamped coordinates (R10) + the speckle filter's per-row init
//   (round-2 forms kept for A/B: k_sgm_path / k_sgm_path_w, one workgroup / one wave per line and direction; k_sgm_select)
#include "rtdm_kernels.h"
#include "rtdm_device.h"

#include <cstdlib>
#include <mutex>

namespace rtdm {

static constexpr int FTZ = 15;

// Per pixel and image, once: the Birchfield-Tomasi bounds (value, min and max against the half-way points to the two
// neighbours) of the clipped x-gradient and of the raw intensity, packed as two uchar4 -- the pixel-cost kernel then
// needs one 8-byte load per (pixel, image) instead of six byte loads per (pixel, disparity, image).
__device__ __forceinline__ int sgm_grad(const uint8_t* r0, const uint8_t* r1, const uint8_t* r2, int x, int W)
{
    if (x <= 0 || x >= W - 1) return FTZ;
    const int g = ((int)r1[x + 1] - (int)r1[x - 1]) * 2 + ((int)r0[x + 1] - (int)r0[x - 1]) + ((int)r2[x + 1] - (int)r2[x - 1]);
    return min(max(g, -FTZ), FTZ) + FTZ;
}
__device__ __forceinline__ uint32_t bt_pack(int v, int m, int p, bool has_m, bool has_p)
{
    const int l = has_m ? (v + m) / 2 : v, r = has_p ? (v + p) / 2 : v;
    return (uint32_t)v | ((uint32_t)min(min(l, r), v) << 8) | ((uint32_t)max(max(l, r), v) << 16);
}

__global__ __launch_bounds__(256) void k_sgm_bounds(Plane8 L, Plane8 R, uint2* bl, uint2* br, int W, int H, int n)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y = blockIdx.y;
    int f = blockIdx.z;
    const bool right = f >= n;
    if (right) f -= n;
    const Plane8 S = right ? R : L;
    const uint8_t* img = S.base + (size_t)f * S.frame;
    const uint8_t* r1 = img + (size_t)y * S.pitch;
    const uint8_t* r0 = img + (size_t)(y > 0 ? y - 1 : y) * S.pitch;
    const uint8_t* r2 = img + (size_t)(y < H - 1 ? y + 1 : y) * S.pitch;
    const bool hm = x > 0, hp = x < W - 1;
    const uint32_t gb = bt_pack(sgm_grad(r0, r1, r2, x, W), hm ? sgm_grad(r0, r1, r2, x - 1, W) : 0,
                                hp ? sgm_grad(r0, r1, r2, x + 1, W) : 0, hm, hp);
    // R1: the library overwrites columns 0 and W-1 of the raw-intensity row with ftzero as well, before the bounds are taken
    const auto raw = [&](int i) -> int { return (i <= 0 || i >= W - 1) ? FTZ : (int)r1[i]; };
    const uint32_t rb = bt_pack(raw(x), hm ? raw(x - 1) : 0, hp ? raw(x + 1) : 0, hm, hp);
    (right ? br : bl)[((size_t)f * H + y) * W + x] = make_uint2(gb, rb);
}

__device__ __forceinline__ int bt_cost(uint32_t a, uint32_t b)
{
    const int u = a & 0xff, u0 = (a >> 8) & 0xff, u1 = (a >> 16) & 0xff;
    const int v = b & 0xff, v0 = (b >> 8) & 0xff, v1 = (b >> 16) & 0xff;
    const int c0 = max(0, max(u - v1, v0 - u));
    const int c1 = max(0, max(v - u1, u0 - v));
    return min(c0, c1);
}

// pixel cost, u8: one thread per (x, four consecutive d); d fastest.  Round 3: two disparities per instruction in packed u16 --
// max(0, u - v1, v0 - u) is max(u -sat v1, v0 -sat u) -- 11 VALU per (x, d) instead of ~30 (the kernel was VALU bound at 1.9 TB/s).
typedef unsigned short sgm_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sgm_subs(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(sgm_us2, a), __builtin_bit_cast(sgm_us2, b))); }
__device__ __forceinline__ uint32_t sgm_max2(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(sgm_us2, a), __builtin_bit_cast(sgm_us2, b))); }
__device__ __forceinline__ uint32_t sgm_min2(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(sgm_us2, a), __builtin_bit_cast(sgm_us2, b))); }
// wrapping / saturating packed u16 sums and differences (the path recurrence, the block sums)
typedef unsigned short sgm_us2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sgm_add2(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, (sgm_us2w)(__builtin_bit_cast(sgm_us2w, a) + __builtin_bit_cast(sgm_us2w, b))); }
__device__ __forceinline__ uint32_t sgm_sub2(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, (sgm_us2w)(__builtin_bit_cast(sgm_us2w, a) - __builtin_bit_cast(sgm_us2w, b))); }
__device__ __forceinline__ uint32_t sgm_adds2(uint32_t a, uint32_t b)
{ return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(sgm_us2w, a), __builtin_bit_cast(sgm_us2w, b))); }

// bt_cost of one left pixel (u, u0, u1 replicated into both halves) against two right pixels (low / high half)
__device__ __forceinline__ uint32_t bt_cost2(uint32_t U, uint32_t U0, uint32_t U1, uint32_t V, uint32_t V0, uint32_t V1)
{ return sgm_min2(sgm_max2(sgm_subs(U, V1), sgm_subs(V0, U)), sgm_max2(sgm_subs(V, U1), sgm_subs(U0, V))); }

__global__ __launch_bounds__(256) void k_sgm_pix(const uint2* bl, const uint2* br, uint8_t* pix, SGMGeom g)
{
    const int dq = g.D >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over W1 * D/4
    if (idx >= (size_t)g.W1 * dq) return;
    const int d = (int)(idx % dq) * 4, xi = (int)(idx / dq);
    const int y = blockIdx.y, f = blockIdx.z;
    const int x = g.x0 + xi, xr = x - (d + g.minD);                     // element j pairs x with xr - j
    const size_t row = ((size_t)f * g.H + y) * g.W;
    const uint2 a = bl[row + x];
    uint2 b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = br[row + xr - j];
    const auto rep = [](uint32_t w, int k) -> uint32_t { return ((w >> (8 * k)) & 0xffu) * 0x00010001u; };
    const uint32_t Ug = rep(a.x, 0), Ug0 = rep(a.x, 1), Ug1 = rep(a.x, 2), Ur = rep(a.y, 0), Ur0 = rep(a.y, 1), Ur1 = rep(a.y, 2);
    uint32_t c[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint2 lo = b[2 * p], hi = b[2 * p + 1];                   // low half: element 2p, high half: element 2p + 1
        const uint32_t cg = bt_cost2(Ug, Ug0, Ug1, __builtin_amdgcn_perm(hi.x, lo.x, 0x0C040C00u), __builtin_amdgcn_perm(hi.x, lo.x, 0x0C050C01u),
                                     __builtin_amdgcn_perm(hi.x, lo.x, 0x0C060C02u));
        const uint32_t cr = bt_cost2(Ur, Ur0, Ur1, __builtin_amdgcn_perm(hi.y, lo.y, 0x0C040C00u), __builtin_amdgcn_perm(hi.y, lo.y, 0x0C050C01u),
                                     __builtin_amdgcn_perm(hi.y, lo.y, 0x0C060C02u));
        c[p] = cg + ((cr >> 2) & 0x003f003fu);                          // both <= 63 + 30: no carry between the halves
    }
    *(uint32_t*)(pix + (((size_t)f * g.H + y) * g.W1 + xi) * g.D + d) = __builtin_amdgcn_perm(c[1], c[0], 0x06040200u);
}

// block cost: thread = (x, four consecutive d); walks down a strip of rows keeping the last 2R+1 horizontal sums in
// registers, so every pixel-cost element is read (2R+1) times instead of (2R+1)^2 times
template <int R>
__global__ __launch_bounds__(256) void k_sgm_box(const uint8_t* pix, uint16_t* C, SGMGeom g, int rows_per_strip, int cost_limit, int32_t* ovf)
{
    const int dq = g.D >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over W1 * D/4
    if (idx >= (size_t)g.W1 * dq) return;
    const int d = (int)(idx % dq) * 4, xi = (int)(idx / dq);
    const int f = blockIdx.z;
    const int y0 = blockIdx.y * rows_per_strip, y1 = min(y0 + rows_per_strip, g.H);
    const uint8_t* base = pix + (size_t)f * g.H * g.W1 * g.D + d;
    int xs[2 * R + 1];
#pragma unroll
    for (int k = 0; k <= 2 * R; ++k) xs[k] = min(max(xi + k - R, 0), g.W1 - 1) * g.D;
    struct Sum4 { int v[4]; };
    const auto hsum = [&](int y) -> Sum4 {
        const uint8_t* row = base + (size_t)min(max(y, 0), g.H - 1) * g.W1 * g.D;
        Sum4 s = {{0, 0, 0, 0}};
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            const uint32_t w = *(const uint32_t*)(row + xs[k]);
            s.v[0] += w & 0xff; s.v[1] += (w >> 8) & 0xff; s.v[2] += (w >> 16) & 0xff; s.v[3] += w >> 24;
        }
        return s;
    };
    Sum4 ring[2 * R + 1];
    int sum[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k <= 2 * R; ++k) {
        ring[k] = hsum(y0 - R + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) sum[j] += ring[k].v[j];
    }
    uint16_t* out = C + (((size_t)f * g.H) * g.W1 + xi) * g.D + d;
    for (int y = y0; y < y1; y += 2 * R + 1) {
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) {
            if (y + k < y1) {
                *(uint2*)(out + (size_t)(y + k) * g.W1 * g.D) =
                    make_uint2((uint32_t)sum[0] | ((uint32_t)sum[1] << 16), (uint32_t)sum[2] | ((uint32_t)sum[3] << 16));
                if (cost_limit > 0 && max(max(sum[0], sum[1]), max(sum[2], sum[3])) > cost_limit) *ovf = 1;
                const Sum4 h = hsum(y + k + R + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) { sum[j] += h.v[j] - ring[k].v[j]; }
                ring[k] = h;
            }
        }
    }
}

// Pixel cost and block sum in ONE kernel for the small windows (R <= 3; D = 16, 32, 64, 128, 256): a workgroup owns a tile of
// TX = 4 * CG output columns (CG = 256 / (D / 4) columns per pass of its threads) and walks a strip of rows; per source row every
// thread computes the Birchfield-Tomasi costs of ~4.5 (column, four disparities) items of the tile + halo into LDS (k_sgm_pix's
// arithmetic, u8), and after one barrier (the tile is double-buffered) sums 2R + 1 of them from LDS for each of its four output
// columns and slides the vertical window in registers (k_sgm_box's ring, packed u16).  The u8 volume is neither written nor
// read: 280 MB of HBM traffic per 720p D = 128 pair, and k_sgm_box's 2R + 1 trips to L2 per output are LDS reads.
template <int R, int DQ>
__global__ __launch_bounds__(256) void k_sgm_pixbox(const uint2* bl, const uint2* br, uint16_t* C, SGMGeom g, int rows_per_strip)
{
    constexpr int CG = 256 / DQ, TX = 4 * CG, TW = TX + 2 * R, NP = (TW + CG - 1) / CG, W1R = 2 * R + 1;
    __shared__ uint32_t tile[2][TW][DQ];                    // [row parity][tile column][disparity quad]: four u8 pixel costs
    const int dqi = threadIdx.x % DQ, cg = threadIdx.x / DQ, d = dqi * 4;
    const int xt0 = blockIdx.x * TX;                        // first output column of the tile (W1 domain)
    const int f = blockIdx.z;
    const int y0 = blockIdx.y * rows_per_strip, y1 = min(y0 + rows_per_strip, g.H);
    const auto rep = [](uint32_t w, int k) -> uint32_t { return ((w >> (8 * k)) & 0xffu) * 0x00010001u; };
    uint32_t ring[4][W1R][2], sum[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { sum[i][0] = sum[i][1] = 0u; for (int k = 0; k < W1R; ++k) ring[i][k][0] = ring[i][k][1] = 0u; }
    const int nsrc = (y1 - y0) + 2 * R;                     // source rows y0 - R .. y1 - 1 + R (clamped into the frame)
    for (int base = 0; base < nsrc; base += W1R) {
#pragma unroll
        for (int k = 0; k < W1R; ++k) {
            const int t = base + k;
            if (t < nsrc) {                                 // workgroup-uniform
                const int par = t & 1;
                const int ysrc = min(max(y0 - R + t, 0), g.H - 1);
                const size_t row = ((size_t)f * g.H + ysrc) * g.W;
                // pixel costs of this row's tile columns (+ halo), clamped into [0, W1)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int tc = cg + p * CG;
                    if (tc < TW) {
                        const int xi = min(max(xt0 - R + tc, 0), g.W1 - 1);
                        const int x = g.x0 + xi, xr = x - (d + g.minD);
                        const uint2 a = bl[row + x];
                        uint2 bb[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) bb[j] = br[row + xr - j];
                        const uint32_t Ug = rep(a.x, 0), Ug0 = rep(a.x, 1), Ug1 = rep(a.x, 2), Ur = rep(a.y, 0), Ur0 = rep(a.y, 1), Ur1 = rep(a.y, 2);
                        uint32_t c2[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const uint2 lo = bb[2 * q], hi = bb[2 * q + 1];
                            const uint32_t cgr = bt_cost2(Ug, Ug0, Ug1, __builtin_amdgcn_perm(hi.x, lo.x, 0x0C040C00u), __builtin_amdgcn_perm(hi.x, lo.x, 0x0C050C01u),
                                                          __builtin_amdgcn_perm(hi.x, lo.x, 0x0C060C02u));
                            const uint32_t cin = bt_cost2(Ur, Ur0, Ur1, __builtin_amdgcn_perm(hi.y, lo.y, 0x0C040C00u), __builtin_amdgcn_perm(hi.y, lo.y, 0x0C050C01u),
                                                          __builtin_amdgcn_perm(hi.y, lo.y, 0x0C060C02u));
                            c2[q] = cgr + ((cin >> 2) & 0x003f003fu);
                        }
                        tile[par][tc][dqi] = __builtin_amdgcn_perm(c2[1], c2[0], 0x06040200u);
                    }
                }
                __syncthreads();
                const int yo = y0 - 2 * R + t;               // the output row whose window this source row completes
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tcol = cg + i * CG;            // output column xt0 + tcol: tile columns tcol .. tcol + 2R
                    uint32_t h0 = 0u, h1 = 0u;
#pragma unroll
                    for (int q = 0; q < W1R; ++q) {
                        const uint32_t w = tile[par][tcol + q][dqi];
                        h0 = sgm_add2(h0, __builtin_amdgcn_perm(0u, w, 0x0C010C00u));    // (d, d + 1) as u16
                        h1 = sgm_add2(h1, __builtin_amdgcn_perm(0u, w, 0x0C030C02u));    // (d + 2, d + 3)
                    }
                    sum[i][0] = sgm_sub2(sgm_add2(sum[i][0], h0), ring[i][k][0]);
                    sum[i][1] = sgm_sub2(sgm_add2(sum[i][1], h1), ring[i][k][1]);
                    ring[i][k][0] = h0; ring[i][k][1] = h1;
                    const int xo = xt0 + tcol;
                    if (yo >= y0 && xo < g.W1)
                        *(uint2*)(C + (((size_t)f * g.H + yo) * g.W1 + xo) * g.D + d) = make_uint2(sum[i][0], sum[i][1]);
                }
            }
        }
    }
}

// Any window (R > 8: the register ring of k_sgm_box<R> would not fit): the running vertical sum gains the entering row's
// horizontal sum and loses the leaving row's, both recomputed -- 2 (2R + 1) loads per output instead of 2R + 1.  Sums are
// 32-bit; a block cost above cost_limit (> 0) sets *ovf and is stored truncated (the caller refuses the frame).
__global__ __launch_bounds__(256) void k_sgm_box_any(const uint8_t* pix, uint16_t* C, SGMGeom g, int R, int rows_per_strip, int cost_limit, int32_t* ovf)
{
    const int dq = g.D >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over W1 * D/4
    if (idx >= (size_t)g.W1 * dq) return;
    const int d = (int)(idx % dq) * 4, xi = (int)(idx / dq);
    const int f = blockIdx.z;
    const int y0 = blockIdx.y * rows_per_strip, y1 = min(y0 + rows_per_strip, g.H);
    const uint8_t* base = pix + (size_t)f * g.H * g.W1 * g.D + d;
    struct Sum4 { int v[4]; };
    const auto hsum = [&](int y) -> Sum4 {
        const uint8_t* row = base + (size_t)min(max(y, 0), g.H - 1) * g.W1 * g.D;
        Sum4 s = {{0, 0, 0, 0}};
        for (int k = -R; k <= R; ++k) {
            const uint32_t w = *(const uint32_t*)(row + (size_t)min(max(xi + k, 0), g.W1 - 1) * g.D);
            s.v[0] += w & 0xff; s.v[1] += (w >> 8) & 0xff; s.v[2] += (w >> 16) & 0xff; s.v[3] += w >> 24;
        }
        return s;
    };
    int sum[4] = {0, 0, 0, 0};
    for (int k = -R; k <= R; ++k) { const Sum4 h = hsum(y0 + k); for (int j = 0; j < 4; ++j) sum[j] += h.v[j]; }
    uint16_t* out = C + (((size_t)f * g.H) * g.W1 + xi) * g.D + d;
    for (int y = y0; y < y1; ++y) {
        *(uint2*)(out + (size_t)y * g.W1 * g.D) =
            make_uint2((uint32_t)(sum[0] & 0xffff) | ((uint32_t)sum[1] << 16), (uint32_t)(sum[2] & 0xffff) | ((uint32_t)sum[3] << 16));
        if (cost_limit > 0 && max(max(sum[0], sum[1]), max(sum[2], sum[3])) > cost_limit) *ovf = 1;
        const Sum4 a = hsum(y + R + 1), b = hsum(y - R);
        for (int j = 0; j < 4; ++j) sum[j] += a.v[j] - b.v[j];
    }
}

// wave-wide minimum (DPP), uniform result
__device__ __forceinline__ int wave_min_i32(int v)
{
#define RTDM_DPP_MIN(ctrl, rmask) v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, ctrl, rmask, 0xf, false))
    RTDM_DPP_MIN(0xB1, 0xf); RTDM_DPP_MIN(0x4E, 0xf); RTDM_DPP_MIN(0x141, 0xf); RTDM_DPP_MIN(0x140, 0xf);
    RTDM_DPP_MIN(0x142, 0xa); RTDM_DPP_MIN(0x143, 0xc);
#undef RTDM_DPP_MIN
    return __builtin_amdgcn_readlane(v, 63);
}

// One workgroup per path line of direction (dx, dy); thread t = disparity t (blockDim = D rounded
// up to a multiple of 64).  S (+)= L_r.
__global__ __launch_bounds__(256) void k_sgm_path(const uint16_t* C, uint16_t* S, SGMGeom g, int dx, int dy, int P1, int P2,
                                                  int first_dir)
{
    __shared__ int lbuf[2][256 + 2];      // L_r of the previous pixel, padded at d = -1 and d = D
    __shared__ int wmin[2][4];
    const int d = threadIdx.x, D = g.D, W1 = g.W1, H = g.H;
    const int nw = (blockDim.x + 63) >> 6, wv = threadIdx.x >> 6;
    const int f = blockIdx.y;
    // start pixel of this line
    int sx, sy;
    const int line = blockIdx.x;
    if (dy == 0) { sy = line; sx = dx > 0 ? 0 : W1 - 1; }
    else if (dx == 0) { sx = line; sy = dy > 0 ? 0 : H - 1; }
    else if (line < W1) { sx = line; sy = dy > 0 ? 0 : H - 1; }
    else { const int k = line - W1 + 1; sx = dx > 0 ? 0 : W1 - 1; sy = dy > 0 ? k : H - 1 - k; }
    const size_t fbase = (size_t)f * H * W1 * D;
    const bool live = d < D;
    const int BIG = 1 << 28;
    int x = sx, y = sy, step = 0;
    int cnext = (live && x >= 0 && x < W1 && y >= 0 && y < H) ? C[fbase + ((size_t)y * W1 + x) * D + d] : 0;
    while (x >= 0 && x < W1 && y >= 0 && y < H) {
        const size_t off = fbase + ((size_t)y * W1 + x) * D + d;
        const int c = cnext;
        const int nx = x + dx, ny = y + dy;
        if (live && nx >= 0 && nx < W1 && ny >= 0 && ny < H) cnext = C[fbase + ((size_t)ny * W1 + nx) * D + d];   // prefetch
        int l;
        if (step == 0) {
            l = c;
        } else {
            const int* pb = lbuf[(step - 1) & 1];
            int mprev = wmin[(step - 1) & 1][0];
            for (int q = 1; q < nw; ++q) mprev = min(mprev, wmin[(step - 1) & 1][q]);
            const int best = min(min(pb[d + 1], mprev + P2), min(pb[d], pb[d + 2]) + P1);
            l = c + best - mprev;
        }
        if (live) {
            if (first_dir) S[off] = (uint16_t)l; else S[off] = (uint16_t)min((int)S[off] + l, 32767);   // R5
        }
        int* cb = lbuf[step & 1];
        cb[d + 1] = live ? l : BIG;
        if (d == 0) { cb[0] = BIG; cb[D + 1] = BIG; }
        const int m = wave_min_i32(live ? l : BIG);
        if ((threadIdx.x & 63) == 0) wmin[step & 1][wv] = m;
        __syncthreads();
        x = nx; y = ny; ++step;
    }
}

// Wave-per-line form of k_sgm_path: one WAVE walks one path line, lane l holds the NPL consecutive disparities
// l*NPL .. l*NPL+NPL-1 (NPL divides D), so the recurrence needs no LDS and no barrier: d-1 / d+1 of the lane's end
// elements come from the neighbouring lanes by DPP wave shifts, the line minimum by a DPP reduction.  The walk is a
// serial chain of W1 (or H) steps whose loads would each cost a full memory round trip, so C and S are fetched PF
// steps ahead through a register ring.  S (+)= L_r, same values as k_sgm_path.
template <int NPL> struct PackU16 { uint16_t v[NPL]; };

template <int NPL>
__device__ __forceinline__ PackU16<NPL> ld_pack(const uint16_t* p)
{
    PackU16<NPL> r;
    if constexpr (NPL == 2) { const uint32_t w = *(const uint32_t*)p; r.v[0] = (uint16_t)w; r.v[1] = (uint16_t)(w >> 16); }
    else if constexpr (NPL == 4) { const uint2 w = *(const uint2*)p; r.v[0] = (uint16_t)w.x; r.v[1] = (uint16_t)(w.x >> 16); r.v[2] = (uint16_t)w.y; r.v[3] = (uint16_t)(w.y >> 16); }
    else { for (int j = 0; j < NPL; ++j) r.v[j] = p[j]; }
    return r;
}
template <int NPL>
__device__ __forceinline__ void st_pack(uint16_t* p, const int* l)
{
    if constexpr (NPL == 2) { *(uint32_t*)p = (uint32_t)(l[0] & 0xffff) | ((uint32_t)l[1] << 16); }
    else if constexpr (NPL == 4) { *(uint2*)p = make_uint2((uint32_t)(l[0] & 0xffff) | ((uint32_t)l[1] << 16), (uint32_t)(l[2] & 0xffff) | ((uint32_t)l[3] << 16)); }
    else { for (int j = 0; j < NPL; ++j) p[j] = (uint16_t)l[j]; }
}

// LAST (the last direction of a frame): the aggregated costs min(S + L_r, 32767) of a pixel are complete the moment this
// wave has them in its lanes, so the winner-take-all step of k_sgm_select runs right here -- wave minimum of S << 8 | d,
// uniqueness vote, S[d* +- 1] by v_readlane, quadratic sub-pixel in lane 0 -- and S is neither written back nor read again
// (424 MB of HBM traffic per 720p D = 128 pair, and the select kernel's launch).  What leaves is 8 bytes per pixel:
// {x16 disparity or INV, integer winner + minD or minD - 1, minimum cost, 0} for k_sgm_lrfinal.
struct SgmWin { int16_t d16, bd; uint16_t mins, pad; };

template <int NPL, int PF, bool LAST>
__global__ __launch_bounds__(256) void k_sgm_path_w(const uint16_t* C, uint16_t* S, SGMGeom g, int dx, int dy, int P1, int P2,
                                                    int first_dir, int nlines, SgmWin* win, int uniq)
{
    const int lane = threadIdx.x & 63;
    const int line = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= nlines) return;                                   // whole waves only
    const int D = g.D, W1 = g.W1, H = g.H;
    int sx, sy;
    if (dy == 0) { sy = line; sx = dx > 0 ? 0 : W1 - 1; }
    else if (dx == 0) { sx = line; sy = dy > 0 ? 0 : H - 1; }
    else if (line < W1) { sx = line; sy = dy > 0 ? 0 : H - 1; }
    else { const int k = line - W1 + 1; sx = dx > 0 ? 0 : W1 - 1; sy = dy > 0 ? k : H - 1 - k; }
    const int nx = dx > 0 ? W1 - sx : (dx < 0 ? sx + 1 : 0x7fffffff);
    const int ny = dy > 0 ? H - sy : (dy < 0 ? sy + 1 : 0x7fffffff);
    const int nsteps = min(nx, ny);
    const int d0 = lane * NPL;
    const bool live = d0 < D;
    const int BIG = 1 << 28;
    const long stride = ((long)dy * W1 + dx) * D;
    const size_t off0 = (size_t)blockIdx.y * H * W1 * D + ((size_t)sy * W1 + sx) * D + (live ? d0 : 0);
    const uint16_t* cp = C + off0;
    uint16_t* sp = S + off0;
    PackU16<NPL> cr[PF], sr[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        if (k < nsteps) {
            cr[k] = ld_pack<NPL>(cp + (long)k * stride);
            if (!first_dir) sr[k] = ld_pack<NPL>(sp + (long)k * stride);
        }
    }
    int l[NPL];
    int mprev = 0;
    for (int base = 0; base < nsteps; base += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int step = base + k;
            if (step >= nsteps) break;
            const PackU16<NPL> c = cr[k], sv = sr[k];
            if (step + PF < nsteps) {
                cr[k] = ld_pack<NPL>(cp + (long)(step + PF) * stride);
                if (!first_dir) sr[k] = ld_pack<NPL>(sp + (long)(step + PF) * stride);
            }
            if (step == 0) {
#pragma unroll
                for (int j = 0; j < NPL; ++j) l[j] = live ? (int)c.v[j] : BIG;
            } else {
                // neighbours of the lane's end elements: lane-1's last, lane+1's first (BIG outside the wave)
                const int lo = __builtin_amdgcn_update_dpp(BIG, l[NPL - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
                const int hi = __builtin_amdgcn_update_dpp(BIG, l[0], 0x130, 0xf, 0xf, false);         // wave_shl:1
                int nl[NPL];
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const int dn = j ? l[j - 1] : lo, up = j + 1 < NPL ? l[j + 1] : hi;
                    const int best = min(min(l[j], mprev + P2), min(dn, up) + P1);
                    nl[j] = live ? (int)c.v[j] + best - mprev : BIG;
                }
#pragma unroll
                for (int j = 0; j < NPL; ++j) l[j] = nl[j];
            }
            int o[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) o[j] = first_dir ? l[j] : min((int)sv.v[j] + l[j], 32767);   // R5: saturating sum
            if constexpr (!LAST) {
                if (live) st_pack<NPL>(sp + (long)step * stride, o);
            } else {
                // winner-take-all on the finished pixel (k_sgm_select's first half, lanes = NPL consecutive disparities each)
                unsigned key = 0x7fffffffu;
#pragma unroll
                for (int j = 0; j < NPL; ++j) if (live) key = min(key, ((unsigned)o[j] << 8) | (unsigned)(d0 + j));
                key = (unsigned)wave_min_i32((int)key);              // keys are < 2^24: the signed minimum is fine
                const int mins = (int)(key >> 8), bd = (int)(key & 0xffu);
                bool hit = false;
                const int lim = mins * 100;
#pragma unroll
                for (int j = 0; j < NPL; ++j) hit |= live && abs(d0 + j - bd) > 1 && o[j] * (100 - uniq) < lim;
                const int xi = sx + step * dx, yy = sy + step * dy;
                SgmWin wv;
                wv.d16 = (int16_t)((g.minD - 1) * 16); wv.bd = (int16_t)(g.minD - 1); wv.mins = 0; wv.pad = 0;
                if (!__any(hit) && mins < 32767) {                    // wave-uniform (mins = 32767: every cost saturated, the library finds no winner)
                    const int ip = min(bd + 1, D - 1), in = max(bd - 1, 0);
                    int s_p = 0, s_n = 0;
#pragma unroll
                    for (int j = 0; j < NPL; ++j) {
                        if (ip % NPL == j) s_p = __builtin_amdgcn_readlane(o[j], ip / NPL);
                        if (in % NPL == j) s_n = __builtin_amdgcn_readlane(o[j], in / NPL);
                    }
                    int d16 = bd * 16;
                    if (bd > 0 && bd < D - 1) {
                        const int den = max(s_n + s_p - 2 * mins, 1);
                        d16 += div_trunc_rcp((s_n - s_p) * 16 + den, den * 2);        // |numerator| < 2^21
                    }
                    wv.d16 = (int16_t)(d16 + g.minD * 16); wv.bd = (int16_t)(bd + g.minD); wv.mins = (uint16_t)mins;
                }
                if (lane == 0) win[((size_t)blockIdx.y * H + yy) * W1 + xi] = wv;
            }
            int m = l[0];
#pragma unroll
            for (int j = 1; j < NPL; ++j) m = min(m, l[j]);
            mprev = wave_min_i32(m);
        }
    }
}

// Half-wave form of the path pass (round 3, second half): one HALF-WAVE per path line, two neighbouring lines per wave, and
// the whole recurrence in packed 16-bit arithmetic.  A lane holds 2 * NP2 consecutive disparities as NP2 u16 pairs
// (D = 64 * NP2 fills the 32 lanes; a smaller D leaves the upper lanes dead), so
//   * a step is v_pk_min/add/sub_u16 on pairs -- L_r <= block cost + P2 <= 32767 (rtdm_sgm_create) and the "no neighbour"
//     value is 0xffff under a saturating + P1 --, d - 1 / d + 1 of a pair are two v_alignbit over (previous, own, next)
//     pair, the pairs at the lane's ends come from the neighbouring lanes by DPP wave shifts (replaced by 0xffff at the
//     half-wave's ends), and S is added as it was loaded: no unpacking, no packing;
//   * the line minimum is four DPP steps inside the rows of 16 and one v_permlane16_swap between the two rows of a half, for
//     both lines at once;
//   * the two lines of a wave are neighbours in memory for every direction but the horizontal ones (columns x and x + 1 of
//     one row: 2 * 2 D bytes in one piece), which halves the number of separate pieces the pass asks HBM for.
// Same values as k_sgm_path_w / k_sgm_path (tests: every D, both modes, against the oracle).
template <int NP2> struct PackW { uint32_t w[NP2]; };
template <int NP2>
__device__ __forceinline__ PackW<NP2> ld_w(const uint16_t* p)
{
    PackW<NP2> r;
    if constexpr (NP2 == 1) { r.w[0] = *(const uint32_t*)p; }
    else if constexpr (NP2 == 2) { const uint2 v = *(const uint2*)p; r.w[0] = v.x; r.w[1] = v.y; }
    else { const uint4 v = *(const uint4*)p; r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w; }
    return r;
}
template <int NP2>
__device__ __forceinline__ void st_w(uint16_t* p, const uint32_t* o)
{
    if constexpr (NP2 == 1) { *(uint32_t*)p = o[0]; }
    else if constexpr (NP2 == 2) { *(uint2*)p = make_uint2(o[0], o[1]); }
    else { *(uint4*)p = make_uint4(o[0], o[1], o[2], o[3]); }
}

// minimum over the lane's half-wave, in every lane of that half (values < 2^31)
__device__ __forceinline__ int half_min_i32(int v)
{
#define RTDM_DPP_MIN(ctrl) v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, ctrl, 0xf, 0xf, false))
    RTDM_DPP_MIN(0xB1); RTDM_DPP_MIN(0x4E); RTDM_DPP_MIN(0x141); RTDM_DPP_MIN(0x140);      // the row's minimum in each of its lanes
#undef RTDM_DPP_MIN
    const auto s = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);   // {rows 0 0 2 2, rows 1 1 3 3}
    return min((int)s[0], (int)s[1]);
}

// Winner-take-all on the finished pixels of a wave's two half-waves (k_sgm_select's first half; as in k_sgm_path_w<.., true>,
// but every quantity is a per-half VECTOR value: both pixels are decided by the same instructions).  o = the aggregated
// costs of the lane's 2 * NP2 disparities d0 .. as u16 pairs.  Every lane of a half returns that half's record.
template <int NP2>
__device__ __forceinline__ SgmWin sgm_wta_half(const uint32_t* o, bool live, int lane, int d0, int D, int uniq, int minD)
{
    int v[2 * NP2];
    unsigned key = 0x7fffffffu;
#pragma unroll
    for (int r = 0; r < NP2; ++r) {
        v[2 * r] = (int)(o[r] & 0xffffu); v[2 * r + 1] = (int)(o[r] >> 16);
        key = min(key, ((unsigned)v[2 * r] << 8) | (unsigned)(d0 + 2 * r));
        key = min(key, ((unsigned)v[2 * r + 1] << 8) | (unsigned)(d0 + 2 * r + 1));
    }
    key = (unsigned)half_min_i32(live ? (int)key : 0x7fffffff);     // keys are < 2^24
    const int mins = (int)(key >> 8), bd = (int)(key & 0xffu);
    bool hit = false;
    const int lim = mins * 100;
#pragma unroll
    for (int j = 0; j < 2 * NP2; ++j) hit |= (unsigned)(d0 + j - bd + 1) > 2u && v[j] * (100 - uniq) < lim;
    const unsigned long long hits = __ballot(hit && live);
    // every aggregated cost saturated (R5) at 32767: the library's search for a cost BELOW its initial SHRT_MAX finds none, its
    // bestDisp stays -1 and what it writes is the invalid value -- no winner, no vote (reachable with a large P2 and 8 paths)
    const bool rejected = ((lane & 32) ? (uint32_t)(hits >> 32) : (uint32_t)hits) != 0u || mins >= 32767;
    // S[d* +- 1]: the pair that holds it, from the lane that holds it (ds_bpermute, no LDS memory involved)
    const int ip = min(bd + 1, D - 1), in = max(bd - 1, 0);
    constexpr int LG = NP2 == 1 ? 1 : (NP2 == 2 ? 2 : 3);         // log2 of the disparities per lane
    const int ap = ((lane & 32) + (ip >> LG)) << 2, an = ((lane & 32) + (in >> LG)) << 2;
    uint32_t wp = 0, wn = 0;
#pragma unroll
    for (int r = 0; r < NP2; ++r) {
        const uint32_t tp = (uint32_t)__builtin_amdgcn_ds_bpermute(ap, (int)o[r]);
        const uint32_t tn = (uint32_t)__builtin_amdgcn_ds_bpermute(an, (int)o[r]);
        if (((ip >> 1) & (NP2 - 1)) == r) wp = tp;
        if (((in >> 1) & (NP2 - 1)) == r) wn = tn;
    }
    const int s_p = (int)((wp >> ((ip & 1) << 4)) & 0xffffu), s_n = (int)((wn >> ((in & 1) << 4)) & 0xffffu);
    int d16 = bd * 16;
    if (bd > 0 && bd < D - 1) {
        const int den = max(s_n + s_p - 2 * mins, 1);
        d16 += div_trunc_rcp((s_n - s_p) * 16 + den, den * 2);            // |numerator| < 2^21
    }
    SgmWin wv;
    wv.d16 = (int16_t)((minD - 1) * 16); wv.bd = (int16_t)(minD - 1); wv.mins = 0; wv.pad = 0;
    if (!rejected) { wv.d16 = (int16_t)(d16 + minD * 16); wv.bd = (int16_t)(bd + minD); wv.mins = (uint16_t)mins; }
    return wv;
}

template <int NP2, int PF, bool LAST>
__global__ __launch_bounds__(256) void k_sgm_path_h(const uint16_t* C, uint16_t* S, SGMGeom g, int dx_, int dy, int P1, int P2,
                                                    int first_dir, int nlines, SgmWin* win, int uniq, uint16_t* S2)
{
    // S2 != null (the two horizontal directions side by side, first_dir = 1): lines [0, nlines) run (dx, 0) and write S, lines
    // [nlines, 2 nlines) run (-dx, 0) and write S2 -- the first sweep adds the two up.  Same bytes moved as one pass after the
    // other (the second one's read of S against the sweep's read of S2), but twice the lines in flight: a single pair's 720 rows
    // are 360 waves on 1024 SIMDs, each a serial chain of W1 steps.
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int line0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    const int nall = S2 ? 2 * nlines : nlines;
    if (line0 >= nall) return;                                    // whole waves only
    const int gline = min(line0 + half, nall - 1);
    const bool line_ok = line0 + half < nall;
    const bool second = gline >= nlines;                          // (only with S2)
    const int line = second ? gline - nlines : gline;
    const int dx = second ? -dx_ : dx_;
    const int D = g.D, W1 = g.W1, H = g.H;
    int sx, sy;
    if (dy == 0) { sy = line; sx = dx > 0 ? 0 : W1 - 1; }
    else if (dx == 0) { sx = line; sy = dy > 0 ? 0 : H - 1; }
    else if (line < W1) { sx = line; sy = dy > 0 ? 0 : H - 1; }
    else { const int k = line - W1 + 1; sx = dx > 0 ? 0 : W1 - 1; sy = dy > 0 ? k : H - 1 - k; }
    const int nx = dx > 0 ? W1 - sx : (dx < 0 ? sx + 1 : 0x7fffffff);
    const int ny = dy > 0 ? H - sy : (dy < 0 ? sy + 1 : 0x7fffffff);
    const int nsteps = line_ok ? min(nx, ny) : 0;                 // of this half's line
    const int nmax = max(__builtin_amdgcn_readlane(nsteps, 0), __builtin_amdgcn_readlane(nsteps, 32));
    const int d0 = hl * 2 * NP2;
    const bool live = d0 < D;                                     // D is a multiple of 16 = of 2 * NP2
    const uint32_t NONE = 0xffffffffu;
    const long stride = ((long)dy * W1 + dx) * D;
    const size_t off0 = (size_t)blockIdx.y * H * W1 * D + ((size_t)sy * W1 + sx) * D + (live ? d0 : 0);
    const uint16_t* cp = C + off0;
    uint16_t* sp = (second ? S2 : S) + off0;
    const uint32_t P1s = (uint32_t)P1 * 0x10001u, P2s = (uint32_t)P2 * 0x10001u;
    PackW<NP2> cr[PF], sr[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        if (k < nsteps) {
            cr[k] = ld_w<NP2>(cp + (long)k * stride);
            if (!first_dir) sr[k] = ld_w<NP2>(sp + (long)k * stride);
        }
    }
    uint32_t l[NP2];
    uint32_t mps = 0, mpP2 = 0;                                   // previous pixel's line minimum, and that + P2, in both halves
    for (int base = 0; base < nmax; base += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int step = base + k;
            if (step >= nmax) break;
            const PackW<NP2> c = cr[k], sv = sr[k];
            if (step + PF < nsteps) {
                cr[k] = ld_w<NP2>(cp + (long)(step + PF) * stride);
                if (!first_dir) sr[k] = ld_w<NP2>(sp + (long)(step + PF) * stride);
            }
            if (step == 0) {
#pragma unroll
                for (int r = 0; r < NP2; ++r) l[r] = live ? c.w[r] : NONE;
            } else {
                // the pairs next to the lane's own: lane - 1's last, lane + 1's first; none outside the half-wave
                uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)NONE, (int)l[NP2 - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
                uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)NONE, (int)l[0], 0x130, 0xf, 0xf, false);         // wave_shl:1
                lo = hl == 0 ? NONE : lo;
                hi = hl == 31 ? NONE : hi;
                uint32_t nl[NP2];
#pragma unroll
                for (int r = 0; r < NP2; ++r) {
                    const uint32_t prev = r ? l[r - 1] : lo, next = r + 1 < NP2 ? l[r + 1] : hi;
                    const uint32_t dn = __builtin_amdgcn_alignbit(l[r], prev, 16);      // {prev.hi, own.lo}: d - 1 of both elements
                    const uint32_t up = __builtin_amdgcn_alignbit(next, l[r], 16);      // {own.hi, next.lo}: d + 1
                    const uint32_t best = sgm_min2(sgm_min2(l[r], mpP2), sgm_adds2(sgm_min2(dn, up), P1s));
                    nl[r] = sgm_sub2(sgm_add2(c.w[r], best), mps);
                }
#pragma unroll
                for (int r = 0; r < NP2; ++r) l[r] = live ? nl[r] : NONE;
            }
            uint32_t o[NP2];
#pragma unroll
            for (int r = 0; r < NP2; ++r) o[r] = first_dir ? l[r] : sgm_min2(sgm_add2(sv.w[r], l[r]), 0x7fff7fffu);   // R5
            if constexpr (!LAST) {
                if (live && step < nsteps) st_w<NP2>(sp + (long)step * stride, o);
            } else {
                const SgmWin wv = sgm_wta_half<NP2>(o, live, lane, d0, D, uniq, g.minD);
                if (hl == 0 && step < nsteps) {
                    const int xi = sx + step * dx, yy = sy + step * dy;
                    win[((size_t)blockIdx.y * H + yy) * W1 + xi] = wv;
                }
            }
            uint32_t mm = l[0];
#pragma unroll
            for (int r = 1; r < NP2; ++r) mm = sgm_min2(mm, l[r]);
            const int m = half_min_i32((int)min(mm & 0xffffu, mm >> 16));
            mps = (uint32_t)m * 0x10001u;
            mpP2 = sgm_add2(mps, P2s);
        }
    }
}

// Row-synchronous sweep (round 3): the three directions that advance one row per step -- (0, dy), (+1, dy), (-1, dy) -- in ONE
// pass, so C is read once and S read-modified-written once for the three of them (separate passes: three reads of C, three
// read-modify-writes of S; with the last sweep deciding the winners, S is not written at all).  A workgroup owns a strip of
// 8 * CPH columns of one frame and walks its rows; a half-wave owns CPH (4, 2 or 1) neighbouring columns and keeps the previous
// row's L_r of its 3 * CPH (column, direction) lines in registers.  A diagonal line changes column every row: inside a half-wave that is a
// register rename (the columns are processed in the order that makes the update in-place), between the half-waves of a
// workgroup the edge line goes through LDS (double-buffered, one barrier per row), and between neighbouring STRIPS through a
// small ring in global memory whose 64-bit words carry their own tag (epoch << 16 | row + 1 in the high half, the u16 pair in
// the low half: single-copy atomic, so a word that shows the expected tag is the expected data -- no fence, no L2 write-back).
// Every strip needs its neighbours' edge of the PREVIOUS row, which they publish at the start of that row: the strips of a
// frame advance in lockstep within a row of each other, and a wait is normally already satisfied.  All workgroups of the
// launch must be resident at once (grid <= what the device holds, one sweep at a time per process: launch_sweep_c); as a second
// line of defence a wait gives up after ~1 s, sets *abortf and the pass runs to its end without waiting (the host then
// reports the call as failed and the handle falls back to one pass per direction).
static constexpr int SWEEP_RING = 4;                      // rows of edge data kept per (strip, side)
__device__ __forceinline__ unsigned long long ld_u64_relaxed(const unsigned long long* p)
{ return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u64_relaxed(unsigned long long* p, unsigned long long v)
{ __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one step of one line for both half-waves of the wave: L <- C + min(Lp[d], Lp[d -+ 1] + P1, min Lp + P2) - min Lp, or C where
// the line starts; mps <- the new line minimum in both halves of a dword.  L may alias Lp.
template <int NP2, bool MAY_START>
__device__ __forceinline__ void sgm_line_step(uint32_t* L, uint32_t& mps, const uint32_t* Lp, uint32_t mpsp, const uint32_t* c,
                                              bool start, bool live, int hl, uint32_t P1s, uint32_t P2s)
{
    const uint32_t NONE = 0xffffffffu;
    uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)NONE, (int)Lp[NP2 - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
    uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)NONE, (int)Lp[0], 0x130, 0xf, 0xf, false);         // wave_shl:1
    lo = hl == 0 ? NONE : lo;
    hi = hl == 31 ? NONE : hi;
    const uint32_t mpP2 = sgm_add2(mpsp, P2s);
    uint32_t nl[NP2];
#pragma unroll
    for (int r = 0; r < NP2; ++r) {
        const uint32_t prev = r ? Lp[r - 1] : lo, next = r + 1 < NP2 ? Lp[r + 1] : hi;
        const uint32_t dn = __builtin_amdgcn_alignbit(Lp[r], prev, 16), up = __builtin_amdgcn_alignbit(next, Lp[r], 16);
        const uint32_t best = sgm_min2(sgm_min2(Lp[r], mpP2), sgm_adds2(sgm_min2(dn, up), P1s));
        nl[r] = sgm_sub2(sgm_add2(c[r], best), mpsp);
        if (MAY_START) nl[r] = start ? c[r] : nl[r];
    }
    uint32_t mm = NONE;
#pragma unroll
    for (int r = 0; r < NP2; ++r) { L[r] = live ? nl[r] : NONE; mm = sgm_min2(mm, L[r]); }
    mps = (uint32_t)half_min_i32((int)min(mm & 0xffffu, mm >> 16)) * 0x10001u;
}

template <int NP2, bool LAST, int CPH, bool ADD2>
__global__ __launch_bounds__(256) void k_sgm_sweep(const uint16_t* C, uint16_t* S, SGMGeom g, int dy, int P1, int P2, int strips,
                                                   int items, unsigned long long* ring, int32_t* abortf, uint32_t epoch, SgmWin* win,
                                                   int uniq, int mute_strip, const uint16_t* S2)
{
    __shared__ uint32_t xch[2][8][2][NP2 + 1][32];             // [row parity][half-wave][0: (+1, dy) edge, 1: (-1, dy) edge][pairs, minimum][lane]
    const int lane = threadIdx.x & 63, hl = lane & 31, hw = threadIdx.x >> 5;
    const int D = g.D, W1 = g.W1, H = g.H;
    const int d0 = hl * 2 * NP2;
    const bool live = d0 < D;
    const uint32_t NONE = 0xffffffffu;
    const uint32_t P1s = (uint32_t)P1 * 0x10001u, P2s = (uint32_t)P2 * 0x10001u;
    const size_t rowstride = (size_t)W1 * D;                   // elements
    bool gave_up = false;                                      // (per lane; only the polling lanes ever set it)
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int f = item / strips, s = item - f * strips;
        const int xb = s * (8 * CPH) + hw * CPH;               // the half-wave's first column
        int xc[CPH]; bool okc[CPH];
#pragma unroll
        for (int c = 0; c < CPH; ++c) { okc[c] = xb + c < W1; xc[c] = min(xb + c, W1 - 1); }
        const size_t fbase = (size_t)f * H * rowstride + (live ? d0 : 0);
        const bool has_left = s > 0, has_right = (s + 1) * (8 * CPH) < W1;
        unsigned long long* ring_me = ring + (size_t)(f * strips + s) * 2 * SWEEP_RING * 32 * NP2;
        const unsigned long long* ring_l = ring_me - (size_t)2 * SWEEP_RING * 32 * NP2;                                  // strip s - 1, side 0
        const unsigned long long* ring_r = ring_me + (size_t)2 * SWEEP_RING * 32 * NP2 + (size_t)SWEEP_RING * 32 * NP2;  // strip s + 1, side 1
        uint32_t L0[CPH][NP2], L1[CPH][NP2], L2[CPH][NP2], m0[CPH], m1[CPH], m2[CPH];
#pragma unroll
        for (int c = 0; c < CPH; ++c) { m0[c] = m1[c] = m2[c] = 0; for (int r = 0; r < NP2; ++r) L0[c][r] = L1[c][r] = L2[c][r] = NONE; }
        // this row's and the next row's costs: C and S of the half-wave's four columns, requested a row ahead
        PackW<NP2> cn[CPH], sn[CPH], tn[ADD2 ? CPH : 1];                // (two rows ahead measured slower: 1.16 -> 1.25 ms per pair at 4 pairs per call)
        {
            const int y = dy > 0 ? 0 : H - 1;
#pragma unroll
            for (int c = 0; c < CPH; ++c) {
                cn[c] = ld_w<NP2>(C + fbase + (size_t)y * rowstride + (size_t)xc[c] * D);
                sn[c] = ld_w<NP2>(S + fbase + (size_t)y * rowstride + (size_t)xc[c] * D);
                if constexpr (ADD2) tn[c] = ld_w<NP2>(S2 + fbase + (size_t)y * rowstride + (size_t)xc[c] * D);   // the other horizontal direction's L_r (k_sgm_path_h)
            }
        }
        for (int t = 0; t < H; ++t) {
            const int y = dy > 0 ? t : H - 1 - t, par = t & 1;
            PackW<NP2> cc[CPH], sc[CPH];
#pragma unroll
            for (int c = 0; c < CPH; ++c) {
                cc[c] = cn[c]; sc[c] = sn[c];
                if constexpr (ADD2) { for (int r = 0; r < NP2; ++r) sc[c].w[r] = sgm_min2(sgm_add2(sc[c].w[r], tn[c].w[r]), 0x7fff7fffu); }   // R5
            }
            if (t + 1 < H) {
                const int yn = y + dy;
#pragma unroll
                for (int c = 0; c < CPH; ++c) {
                    cn[c] = ld_w<NP2>(C + fbase + (size_t)yn * rowstride + (size_t)xc[c] * D);
                    sn[c] = ld_w<NP2>(S + fbase + (size_t)yn * rowstride + (size_t)xc[c] * D);
                    if constexpr (ADD2) tn[c] = ld_w<NP2>(S2 + fbase + (size_t)yn * rowstride + (size_t)xc[c] * D);
                }
            }
            const bool first_row = t == 0;
            uint32_t acc[CPH][NP2];
            const auto add_to = [&](int c, const uint32_t* L) {
#pragma unroll
                for (int r = 0; r < NP2; ++r) acc[c][r] = sgm_min2(sgm_add2(acc[c][r], L[r]), 0x7fff7fffu);   // R5
            };
#pragma unroll
            for (int c = 0; c < CPH; ++c) for (int r = 0; r < NP2; ++r) acc[c][r] = sc[c].w[r];
            // the neighbouring strip's edge of the previous row was published a row ago: ask for it now, look at it after the
            // row's other lines (a load from the ring is a round trip to memory, ~1 us: as long as a whole row of a lone wave)
            const bool poll_l = hw == 0 && has_left && !first_row, poll_r = hw == 7 && has_right && !first_row;
            const unsigned long long* ring_src = (poll_l ? ring_l : ring_r) + ((size_t)((t - 1) & (SWEEP_RING - 1)) * 32 + hl) * NP2;
            unsigned long long w[NP2];
#pragma unroll
            for (int r = 0; r < NP2; ++r) w[r] = 0ull;
            if (poll_l || poll_r) {
#pragma unroll
                for (int r = 0; r < NP2; ++r) w[r] = ld_u64_relaxed(ring_src + r);
            }
            // (+1, dy): columns CPH - 1 ... 1 take the line of their left neighbour's previous row -- in place in that order
#pragma unroll
            for (int c = CPH - 1; c >= 1; --c) { sgm_line_step<NP2, true>(L1[c], m1[c], L1[c - 1], m1[c - 1], cc[c].w, first_row, live, hl, P1s, P2s); add_to(c, L1[c]); }
            // (-1, dy): columns 0 ... CPH - 2 take their right neighbour's; a line starts at the frame's last column
#pragma unroll
            for (int c = 0; c <= CPH - 2; ++c) { sgm_line_step<NP2, true>(L2[c], m2[c], L2[c + 1], m2[c + 1], cc[c].w, first_row || xb + c == W1 - 1, live, hl, P1s, P2s); add_to(c, L2[c]); }
            // the two lines that enter the half-wave's columns from outside: from the neighbouring half-wave (LDS, written in the
            // previous row) or, at the strip's ends, from the neighbouring strip (the ring)
            uint32_t inL[NP2], inR[NP2], inLm = 0, inRm = 0;
#pragma unroll
            for (int r = 0; r < NP2; ++r) { inL[r] = NONE; inR[r] = NONE; }
            if (!first_row) {
                if (hw > 0) { for (int r = 0; r < NP2; ++r) inL[r] = xch[par ^ 1][hw - 1][0][r][hl]; inLm = xch[par ^ 1][hw - 1][0][NP2][hl]; }
                if (hw < 7) { for (int r = 0; r < NP2; ++r) inR[r] = xch[par ^ 1][hw + 1][1][r][hl]; inRm = xch[par ^ 1][hw + 1][1][NP2][hl]; }
            }
            const int wv = threadIdx.x >> 6;
            const bool ring_l_wave = wv == 0 && has_left && !first_row, ring_r_wave = wv == 3 && has_right && !first_row;   // wave-uniform
            // the entering lines that do not come through the ring, now: what a strip hands to its neighbours (the (+1, dy) line of
            // its last column, the (-1, dy) line of its first) never depends on what it is still waiting for from them
            if (!ring_l_wave) { sgm_line_step<NP2, true>(L1[0], m1[0], inL, inLm, cc[0].w, first_row || xb == 0, live, hl, P1s, P2s); add_to(0, L1[0]); }
            if (!ring_r_wave) {
                sgm_line_step<NP2, true>(L2[CPH - 1], m2[CPH - 1], inR, inRm, cc[CPH - 1].w, first_row || xb + CPH - 1 >= W1 - 1, live, hl, P1s, P2s);
                add_to(CPH - 1, L2[CPH - 1]);
            }
            const unsigned long long tag = ((unsigned long long)((epoch << 16) | (uint32_t)(t + 1))) << 32;
            if (hw == 7 && has_right && s != mute_strip) {          // (mute_strip >= 0: the test of the give-up path -- that strip never publishes)
#pragma unroll
                for (int r = 0; r < NP2; ++r) st_u64_relaxed(ring_me + ((size_t)(t & (SWEEP_RING - 1)) * 32 + hl) * NP2 + r, tag | L1[CPH - 1][r]);
            }
            if (hw == 0 && has_left) {
#pragma unroll
                for (int r = 0; r < NP2; ++r)
                    st_u64_relaxed(ring_me + (size_t)SWEEP_RING * 32 * NP2 + ((size_t)(t & (SWEEP_RING - 1)) * 32 + hl) * NP2 + r, tag | L2[0][r]);
            }
            // (0, dy)
#pragma unroll
            for (int c = 0; c < CPH; ++c) { sgm_line_step<NP2, true>(L0[c], m0[c], L0[c], m0[c], cc[c].w, first_row, live, hl, P1s, P2s); add_to(c, L0[c]); }
            if (ring_l_wave || ring_r_wave) {
                const uint32_t want = (epoch << 16) | (uint32_t)t;                  // the previous row's tag
                bool done = !(poll_l || poll_r) || gave_up;
                for (int spin = 0;; ++spin) {
                    if (!done) {
                        bool all = true;
#pragma unroll
                        for (int r = 0; r < NP2; ++r) { if (spin) w[r] = ld_u64_relaxed(ring_src + r); all &= (uint32_t)(w[r] >> 32) == want; }
                        done = all;
                    }
                    if (__all(done)) break;
                    if ((spin & 63) == 63 && (spin > (1 << 20) || __hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        if (!done) { gave_up = true; __hip_atomic_store(abortf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (poll_l) { for (int r = 0; r < NP2; ++r) inL[r] = (uint32_t)w[r]; }
                if (poll_r) { for (int r = 0; r < NP2; ++r) inR[r] = (uint32_t)w[r]; }
                // the minimum of a line that came through the ring is not sent along: take it here (both halves do, one needs it)
                uint32_t mmL = NONE, mmR = NONE;
#pragma unroll
                for (int r = 0; r < NP2; ++r) { mmL = sgm_min2(mmL, live ? inL[r] : NONE); mmR = sgm_min2(mmR, live ? inR[r] : NONE); }
                const uint32_t hmL = (uint32_t)half_min_i32((int)min(mmL & 0xffffu, mmL >> 16)) * 0x10001u;
                const uint32_t hmR = (uint32_t)half_min_i32((int)min(mmR & 0xffffu, mmR >> 16)) * 0x10001u;
                if (poll_l) inLm = hmL;
                if (poll_r) inRm = hmR;
                if (ring_l_wave) { sgm_line_step<NP2, true>(L1[0], m1[0], inL, inLm, cc[0].w, xb == 0, live, hl, P1s, P2s); add_to(0, L1[0]); }
                if (ring_r_wave) {
                    sgm_line_step<NP2, true>(L2[CPH - 1], m2[CPH - 1], inR, inRm, cc[CPH - 1].w, xb + CPH - 1 >= W1 - 1, live, hl, P1s, P2s);
                    add_to(CPH - 1, L2[CPH - 1]);
                }
            }
            // the edge lines for the neighbouring half-waves' next row (final only now when one of them came through the ring)
#pragma unroll
            for (int r = 0; r < NP2; ++r) { xch[par][hw][0][r][hl] = L1[CPH - 1][r]; xch[par][hw][1][r][hl] = L2[0][r]; }
            xch[par][hw][0][NP2][hl] = m1[CPH - 1]; xch[par][hw][1][NP2][hl] = m2[0];
            // S (or the winners)
#pragma unroll
            for (int c = 0; c < CPH; ++c) {
                if constexpr (!LAST) {
                    if (live && okc[c]) st_w<NP2>(S + fbase + (size_t)y * rowstride + (size_t)xc[c] * D, acc[c]);
                } else {
                    const SgmWin wv = sgm_wta_half<NP2>(acc[c], live, lane, d0, D, uniq, g.minD);
                    if (hl == 0 && okc[c]) win[((size_t)f * H + y) * W1 + xc[c]] = wv;
                }
            }
            __syncthreads();
        }
    }
}

// winner-take-all + uniqueness + sub-pixel + left-right check; one workgroup per row.  Each WAVE takes
// every 4th pixel of the row with its LANES on the disparities (coalesced 2*D-byte reads of S): wave-min of
// the key S << 8 | d (first minimum), __any() for the uniqueness test, readlane for S[d* +- 1].
template <bool SPK, int NCH>
__global__ __launch_bounds__(256) void k_sgm_select(const uint16_t* S, Plane16W disp, SGMGeom g, int uniq, int disp12MaxDiff,
                                                    int32_t* label, int32_t* size, uint32_t* runs, int32_t* rowcnt,
                                                    int16_t* headmap, int spkDiff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* key = (unsigned long long*)smem;     // W : (minS << 32 | x) votes per right column
    int16_t* bdv = (int16_t*)(key + g.W);                    // W : integer winner + minD of column x (or minD-1)
    int16_t* row = bdv + g.W;                                // W : disparity row
    __shared__ int wsum[4];
    const int y = blockIdx.y, f = blockIdx.z;
    const int W = g.W, D = g.D, minD = g.minD, INV = (minD - 1) * 16;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int x = threadIdx.x; x < W; x += 256) { key[x] = ~0ull; bdv[x] = (int16_t)(minD - 1); row[x] = (int16_t)INV; }
    __syncthreads();
    const uint16_t* srow = S + (((size_t)f * g.H + y) * g.W1) * D;
    // the wave's pixels are a serial chain (reduce, test, vote): their S values are fetched PF pixels ahead, otherwise every
    // pixel would cost a full memory round trip
    constexpr int PF = 6;
    uint16_t pre[PF][NCH];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int xp = wv + 4 * p;
#pragma unroll
        for (int c = 0; c < NCH; ++c) { const int d = lane + 64 * c; pre[p][c] = (xp < g.W1 && d < D) ? srow[(size_t)xp * D + d] : (uint16_t)0x7fff; }
    }
    for (int xb = wv; xb < g.W1; xb += 4 * PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int xi = xb + 4 * p;
        if (xi >= g.W1) break;
        int v[NCH];
        unsigned k = 0x7fffffffu;                            // (reduced as signed: keep the sentinel positive)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = lane + 64 * c;
            v[c] = d < D ? (int)pre[p][c] : 0x7fff;
            if (d < D) k = min(k, ((unsigned)v[c] << 8) | (unsigned)d);
        }
        {
            const int xn = xi + 4 * PF;                      // refill this ring slot
#pragma unroll
            for (int c = 0; c < NCH; ++c) { const int d = lane + 64 * c; if (xn < g.W1 && d < D) pre[p][c] = srow[(size_t)xn * D + d]; }
        }
        k = (unsigned)wave_min_i32((int)k);                  // keys are < 2^24: signed min is fine
        const int mins = (int)(k >> 8), bd = (int)(k & 0xffu);
        bool hit = false;
        const int lim = mins * 100;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = lane + 64 * c;
            hit |= d < D && abs(d - bd) > 1 && v[c] * (100 - uniq) < lim;
        }
        if (__any(hit) || mins >= 32767) continue;            // wave-uniform (mins = 32767: every cost saturated, the library finds no winner)
        const int ip = min(bd + 1, D - 1), in = max(bd - 1, 0);
        int sp = 0, sn = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if ((ip >> 6) == c) sp = __builtin_amdgcn_readlane(v[c], ip & 63);
            if ((in >> 6) == c) sn = __builtin_amdgcn_readlane(v[c], in & 63);
        }
        if (lane == 0) {
            const int x = g.x0 + xi;
            const int x2 = x - (bd + minD);
            // R7: the library walks the row from right to left and replaces a vote only by a strictly smaller cost, so among
            // equal costs the right-most voter stays: the key's low word grows to the left
            if (x2 >= 0 && x2 < W) atomicMin(&key[x2], ((unsigned long long)(unsigned)mins << 32) | (unsigned)(0xffff - x));
            bdv[x] = (int16_t)(bd + minD);
            int d16 = bd * 16;
            if (bd > 0 && bd < D - 1) {
                const int den = max(sn + sp - 2 * mins, 1);
                d16 += div_trunc_rcp((sn - sp) * 16 + den, den * 2);        // |numerator| < 2^21
            }
            row[x] = (int16_t)(d16 + minD * 16);
        }
      }
    }
    __syncthreads();
    int16_t* out = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    int16_t* fin = (int16_t*)(row + W);                      // W : final row (for the speckle init)
    for (int x = threadIdx.x; x < W; x += 256) {
        int d1 = row[x];
        if (d1 != INV) {                                      // R9: always on (the host passes disp12MaxDiff > 0 ? it : 1)
            const int da = d1 >> 4, db = (d1 + 15) >> 4;
            const int xa = x - da, xb = x - db;
            // a column nobody voted for holds the library's initial value, the SCALED invalid disparity, which passes its
            // ">= minD" test for minD >= 2 (restated, not repaired)
            const auto vote = [&](int xv) -> int { return key[xv] != ~0ull ? (int)bdv[0xffff - (unsigned)(key[xv] & 0xffffu)] : INV; };
            bool ba = false, bb = false;
            if (xa >= 0 && xa < W) { const int v = vote(xa); ba = v >= minD && abs(v - da) > disp12MaxDiff; }
            if (xb >= 0 && xb < W) { const int v = vote(xb); bb = v >= minD && abs(v - db) > disp12MaxDiff; }
            if (ba && bb) d1 = INV;
        }
        out[x] = (int16_t)d1;
        if (SPK) fin[x] = (int16_t)d1;
    }
    if (SPK) {
        __syncthreads();
        spk_row_init(fin, (int*)key, wsum, W, (f * g.H + y) * W, label, size, runs, rowcnt + (f * g.H + y), headmap, INV, spkDiff);
    }
}

// The second half of k_sgm_select for winners that were found inside the last path pass (k_sgm_path_w<.., LAST>): the votes
// of a row (R7), the always-on left-right check (R9) and the row's x16 disparities.  One workgroup per row.
__global__ __launch_bounds__(256) void k_sgm_lrfinal(const SgmWin* win, Plane16W disp, SGMGeom g, int disp12MaxDiff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* key = (unsigned long long*)smem;     // W : (minS << 32 | 0xffff - x) votes per right column
    int16_t* bdv = (int16_t*)(key + g.W);                    // W : integer winner + minD of column x (or minD-1)
    int16_t* row = bdv + g.W;                                // W : disparity row
    const int y = blockIdx.y, f = blockIdx.z;
    const int W = g.W, minD = g.minD, INV = (minD - 1) * 16;
    for (int x = threadIdx.x; x < W; x += 256) { key[x] = ~0ull; bdv[x] = (int16_t)(minD - 1); row[x] = (int16_t)INV; }
    __syncthreads();
    const SgmWin* wrow = win + ((size_t)f * g.H + y) * g.W1;
    for (int xi = threadIdx.x; xi < g.W1; xi += 256) {
        const SgmWin w = wrow[xi];
        if (w.bd < minD) continue;                            // rejected by the uniqueness test: no vote, INV
        const int x = g.x0 + xi, x2 = x - (int)w.bd;
        if (x2 >= 0 && x2 < W) atomicMin(&key[x2], ((unsigned long long)w.mins << 32) | (unsigned)(0xffff - x));
        bdv[x] = w.bd;
        row[x] = w.d16;
    }
    __syncthreads();
    int16_t* out = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    for (int x = threadIdx.x; x < W; x += 256) {
        int d1 = row[x];
        if (d1 != INV) {                                      // R9: always on (the host passes disp12MaxDiff > 0 ? it : 1)
            const int da = d1 >> 4, db = (d1 + 15) >> 4;
            const int xa = x - da, xb = x - db;
            const auto vote = [&](int xv) -> int { return key[xv] != ~0ull ? (int)bdv[0xffff - (unsigned)(key[xv] & 0xffffu)] : INV; };
            bool ba = false, bb = false;
            if (xa >= 0 && xa < W) { const int v = vote(xa); ba = v >= minD && abs(v - da) > disp12MaxDiff; }
            if (xb >= 0 && xb < W) { const int v = vote(xb); bb = v >= minD && abs(v - db) > disp12MaxDiff; }
            if (ba && bb) d1 = INV;
        }
        out[x] = (int16_t)d1;
    }
}

// R10: medianBlur(disp, disp, 3) -- 3x3 median of the int16 map with clamped coordinates -- followed by the speckle
// filter's per-row init on the filtered row.  One workgroup per row; the three source rows stream through L2.
__device__ __forceinline__ void mnmx(int& a, int& b) { const int t = min(a, b); b = max(a, b); a = t; }
template <bool SPK>
__global__ __launch_bounds__(256) void k_sgm_median(const int16_t* src, Plane16W disp, int W, int H, int INV, int32_t* label,
                                                    int32_t* size, uint32_t* runs, int32_t* rowcnt, int16_t* headmap, int spkDiff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* sc = (int*)smem;                                   // W : scan scratch of the speckle init
    int16_t* fin = (int16_t*)(sc + W);                      // W : the filtered row
    __shared__ int wsum[4];
    const int y = blockIdx.y, f = blockIdx.z;
    const int16_t* base = src + (size_t)f * H * W;
    const int16_t* r0 = base + (size_t)max(y - 1, 0) * W;
    const int16_t* r1 = base + (size_t)y * W;
    const int16_t* r2 = base + (size_t)min(y + 1, H - 1) * W;
    int16_t* out = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    for (int x = threadIdx.x; x < W; x += 256) {
        const int xm = max(x - 1, 0), xp = min(x + 1, W - 1);
        int p0 = r0[xm], p1 = r0[x], p2 = r0[xp], p3 = r1[xm], p4 = r1[x], p5 = r1[xp], p6 = r2[xm], p7 = r2[x], p8 = r2[xp];
        // median-of-nine exchange network (19 exchanges)
        mnmx(p1, p2); mnmx(p4, p5); mnmx(p7, p8); mnmx(p0, p1); mnmx(p3, p4); mnmx(p6, p7); mnmx(p1, p2); mnmx(p4, p5);
        mnmx(p7, p8); mnmx(p0, p3); mnmx(p5, p8); mnmx(p4, p7); mnmx(p3, p6); mnmx(p1, p4); mnmx(p2, p5); mnmx(p4, p7);
        mnmx(p4, p2); mnmx(p6, p4); mnmx(p4, p2);
        out[x] = (int16_t)p4;
        if (SPK) fin[x] = (int16_t)p4;
    }
    if (SPK) {
        __syncthreads();
        spk_row_init(fin, sc, wsum, W, (f * H + y) * W, label, size, runs, rowcnt + (f * H + y), headmap, INV, spkDiff);
    }
}

template <bool SPK>
static void launch_select(int nch, dim3 grid, size_t lds, hipStream_t stream, const uint16_t* S, Plane16W disp, const SGMGeom& g,
                          int uniq, int md, const SGMBuffers& b, int spkDiff)
{
    dim3 blk(256);
#define RTDM_SEL(N) hipLaunchKernelGGL((k_sgm_select<SPK, N>), grid, blk, lds, stream, S, disp, g, uniq, md, b.label, b.size, b.runs, b.rowcnt, b.headmap, spkDiff)
    switch (nch) { case 1: RTDM_SEL(1); break; case 2: RTDM_SEL(2); break; case 3: RTDM_SEL(3); break; default: RTDM_SEL(4); break; }
#undef RTDM_SEL
}

// S = min(S + S2, 32767) (R5), two elements per thread: only where the two horizontal directions ran side by side into S and S2
// and the sweep that was to add them up could not be launched after all
__global__ __launch_bounds__(256) void k_sgm_add_s2(uint32_t* S, const uint32_t* S2, size_t npairs)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < npairs) S[i] = sgm_min2(sgm_add2(S[i], S2[i]), 0x7fff7fffu);
}

static inline int sgm_np2(int D) { return D <= 64 ? 1 : (D <= 128 ? 2 : 4); }
size_t sgm_ring_words(int maxW, int D, int max_batch)                  // sized for the narrowest strips (8 columns)
{ return (size_t)max_batch * ((size_t)(maxW + 7) / 8) * 2 * SWEEP_RING * 32 * sgm_np2(D); }

// One row-synchronous pass over (0, dy), (+1, dy), (-1, dy).  false = not launched (the caller runs the three passes).
template <int NP2, bool LAST, int CPH, bool ADD2>
static int sweep_capacity(const SGMBuffers& b)
{
    int& cap = b.sweep_cap[(((NP2 == 1 ? 0 : (NP2 == 2 ? 1 : 2)) * 2 + (LAST ? 1 : 0)) * 3 + (CPH == 4 ? 2 : CPH - 1)) * 2 + (ADD2 ? 1 : 0)];
    if (cap == 0) {
        int dev = 0, cus = 0, per_cu = 0;
        cap = -1;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sgm_sweep<NP2, LAST, CPH, ADD2>, 256, 0) == hipSuccess && per_cu > 0)
            cap = per_cu * cus;
        (void)hipGetLastError();
    }
    return cap;
}
// The workgroups of a sweep wait for each other, so all of them must be resident at once: the grid is no larger than what the
// device holds, and the sweeps of ONE PROCESS never run side by side -- they all go through one stream per device (two sweeps
// half resident each would wait for workgroups that cannot start).  The caller's stream and the sweep stream are tied
// together by the handle's two events; kernels of other streams may share the device with a sweep (they do not wait for it,
// so they finish and make room).  (hipLaunchCooperativeKernel would promise the residency, but a process that has used it
// from a thread other than its main one dies in the runtime's exit handlers on ROCm 7.2: tools/sgm_two_threads.py.)
struct SweepLane { std::mutex mu; hipStream_t s = nullptr; };
static SweepLane& sweep_lane(int dev) { static SweepLane lanes[64]; return lanes[dev & 63]; }

template <int NP2, bool LAST, int CPH, bool ADD2>
static bool launch_sweep_c(const SGMGeom& g, const SGMBuffers& b, int dy, int P1, int P2, int n, SgmWin* win, int uniq, hipStream_t stream, const uint16_t* S2in, bool probe)
{
    const int cap = sweep_capacity<NP2, LAST, CPH, ADD2>(b);
    const int strips = (g.W1 + 8 * CPH - 1) / (8 * CPH);
    if (cap < strips || !b.ev_in || !b.ev_out) return false;
    const int items = n * strips;
    if ((size_t)items * 2 * SWEEP_RING * 32 * NP2 > b.ring_words) return false;
    if (probe) return true;                                              // (would be launched: the caller plans its passes on that)
    const int grid = items <= cap ? items : cap / strips * strips;       // the strips of a frame run in the same round
    const uint32_t epoch = (*b.epoch + 1) & 0xffffu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    SweepLane& lane = sweep_lane(dev);
    std::lock_guard<std::mutex> lk(lane.mu);
    if (!lane.s && hipStreamCreateWithFlags(&lane.s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); lane.s = nullptr; return false; }
    if (hipEventRecord((hipEvent_t)b.ev_in, stream) != hipSuccess || hipStreamWaitEvent(lane.s, (hipEvent_t)b.ev_in, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    // RTDM_SGM_SWEEP_TEST_GIVEUP=1 (tests only): strip 0 never publishes its edge, so its neighbour's wait must run into its bound
    static const int mute = env_int("RTDM_SGM_SWEEP_TEST_GIVEUP", 0) ? 0 : -1;
    hipLaunchKernelGGL((k_sgm_sweep<NP2, LAST, CPH, ADD2>), dim3(grid), dim3(256), 0, lane.s, b.C, b.S, g, dy, P1, P2, strips, items, b.ring, b.abortf, epoch, win, uniq, mute, S2in);
    // (from here on the caller's stream has to wait for the sweep stream whatever happens, or it would run ahead of it)
    (void)hipEventRecord((hipEvent_t)b.ev_out, lane.s);
    (void)hipStreamWaitEvent(stream, (hipEvent_t)b.ev_out, 0);
    ++*b.epoch;
    return true;
}
template <int NP2, bool LAST>
static bool launch_sweep_t(const SGMGeom& g, const SGMBuffers& b, int dy, int P1, int P2, int n, SgmWin* win, int uniq, hipStream_t stream, const uint16_t* S2in, bool probe)
{
    // the narrowest strips whose workgroups all fit the device at once: 8 columns (one per half-wave), 16, 32 -- a frame's rows
    // are a serial chain, so the pass is latency bound until every SIMD holds several waves, and the fewer lines a wave carries
    // the shorter its row; wider strips pay the per-row overhead (barrier, edges, addresses) less often
    // (RTDM_SGM_SWEEP_COLS=1 / 2 / 4 fixes the choice: A/B)
    static const int cols_env = env_int("RTDM_SGM_SWEEP_COLS", 0);
    // (the instantiation that also adds S2 -- the first sweep after the side-by-side horizontal directions -- holds one more
    // volume's row in registers: a template parameter, so that the other sweep keeps its occupancy)
    const bool add2 = S2in != nullptr || (probe && b.S2 != nullptr);
    const int cap1 = add2 ? sweep_capacity<NP2, LAST, 1, true>(b) : sweep_capacity<NP2, LAST, 1, false>(b);
    const int cap2 = add2 ? sweep_capacity<NP2, LAST, 2, true>(b) : sweep_capacity<NP2, LAST, 2, false>(b);
    const int pick = cols_env ? cols_env : (n * ((g.W1 + 7) / 8) <= cap1 ? 1 : (n * ((g.W1 + 15) / 16) <= cap2 ? 2 : 4));
#define RTDM_SWC(CC) (add2 ? launch_sweep_c<NP2, LAST, CC, true>(g, b, dy, P1, P2, n, win, uniq, stream, S2in, probe) \
                           : launch_sweep_c<NP2, LAST, CC, false>(g, b, dy, P1, P2, n, win, uniq, stream, S2in, probe))
    if (pick == 1 && RTDM_SWC(1)) return true;
    if (pick <= 2 && RTDM_SWC(2)) return true;
    return RTDM_SWC(4);
#undef RTDM_SWC
}
static bool launch_sweep(bool last, const SGMGeom& g, const SGMBuffers& b, int dy, int P1, int P2, int n, SgmWin* win, int uniq, hipStream_t stream,
                         const uint16_t* S2in = nullptr, bool probe = false)
{
    switch (sgm_np2(g.D) * 2 + (last ? 1 : 0)) {
        case 2: return launch_sweep_t<1, false>(g, b, dy, P1, P2, n, win, uniq, stream, S2in, probe);
        case 3: return launch_sweep_t<1, true>(g, b, dy, P1, P2, n, win, uniq, stream, S2in, probe);
        case 4: return launch_sweep_t<2, false>(g, b, dy, P1, P2, n, win, uniq, stream, S2in, probe);
        case 5: return launch_sweep_t<2, true>(g, b, dy, P1, P2, n, win, uniq, stream, S2in, probe);
        case 8: return launch_sweep_t<4, false>(g, b, dy, P1, P2, n, win, uniq, stream, S2in, probe);
        default: return launch_sweep_t<4, true>(g, b, dy, P1, P2, n, win, uniq, stream, S2in, probe);
    }
}

void launch_sgm(Plane8 L, Plane8 R, Plane16W disp, const SGMGeom& g, const SGMBuffers& b, int blockSize, int P1, int P2,
                int uniq, int disp12MaxDiff, int speckleWindowSize, int speckleRange, int paths, int n, hipStream_t stream,
                int cost_limit)
{
    dim3 blk(256);
    hipLaunchKernelGGL(k_sgm_bounds, dim3((g.W + 255) / 256, g.H, 2 * n), blk, 0, stream, L, R, (uint2*)b.gl, (uint2*)b.gr, g.W, g.H, n);
    const unsigned nxd = (unsigned)(((size_t)g.W1 * (g.D / 4) + 255) / 256);       // D is a multiple of 16
    // RTDM_SGM_PIXBOX=0 (A/B): pixel cost and block sum as two kernels with the u8 volume between them, for every window
    static const int pixbox_env = env_int("RTDM_SGM_PIXBOX", 1);
    const int Rw = blockSize / 2, dq = g.D / 4;
    const bool pixbox = pixbox_env && Rw <= 3 && (dq == 4 || dq == 8 || dq == 16 || dq == 32 || dq == 64) && !cost_limit;
    if (pixbox) {
        const int rps = 48, strips = (g.H + rps - 1) / rps, tx = 4 * (256 / dq);
        const dim3 pgrid((g.W1 + tx - 1) / tx, strips, n);
#define RTDM_PB(RR, QQ) hipLaunchKernelGGL((k_sgm_pixbox<RR, QQ>), pgrid, blk, 0, stream, (const uint2*)b.gl, (const uint2*)b.gr, b.C, g, rps)
#define RTDM_PBR(RR) do { switch (dq) { case 4: RTDM_PB(RR, 4); break; case 8: RTDM_PB(RR, 8); break; case 16: RTDM_PB(RR, 16); break; \
                                       case 32: RTDM_PB(RR, 32); break; default: RTDM_PB(RR, 64); break; } } while (0)
        switch (Rw) { case 0: RTDM_PBR(0); break; case 1: RTDM_PBR(1); break; case 2: RTDM_PBR(2); break; default: RTDM_PBR(3); break; }
#undef RTDM_PBR
#undef RTDM_PB
    } else {
    hipLaunchKernelGGL(k_sgm_pix, dim3(nxd, g.H, n), blk, 0, stream, (const uint2*)b.gl, (const uint2*)b.gr, b.pix, g);
    {
        const int rps = 48, strips = (g.H + rps - 1) / rps;
        const dim3 bgrid(nxd, strips, n);
        switch (blockSize / 2) {                      // windows <= 17: the row sums of a strip stay in registers
#define RTDM_BOX(RR) case RR: hipLaunchKernelGGL(k_sgm_box<RR>, bgrid, blk, 0, stream, b.pix, b.C, g, rps, cost_limit, b.ovf); break;
            RTDM_BOX(0) RTDM_BOX(1) RTDM_BOX(2) RTDM_BOX(3) RTDM_BOX(4) RTDM_BOX(5) RTDM_BOX(6) RTDM_BOX(7) RTDM_BOX(8)
            default: hipLaunchKernelGGL(k_sgm_box_any, bgrid, blk, 0, stream, b.pix, b.C, g, blockSize / 2, rps, cost_limit, b.ovf); break;
#undef RTDM_BOX
        }
    }
    }
    static const int dirs[8][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {-1, 1}, {1, -1}, {-1, -1}};
    const int threads = (g.D + 63) & ~63;
    int npl = (g.D + 63) / 64;                       // disparities per lane of the wave-per-line kernel: must divide D
    if (g.D % npl) npl = 4;
    static const int wave_paths = env_int("RTDM_SGM_WAVE_PATHS", 1);
    const bool aligned = (((size_t)b.C | (size_t)b.S) & 7) == 0;
    const bool aligned16 = (((size_t)b.C | (size_t)b.S) & 15) == 0;
    // RTDM_SGM_HALF=0 (A/B): every path pass on k_sgm_path_w (one wave per line, 32-bit arithmetic)
    static const int half_paths = env_int("RTDM_SGM_HALF", 2);
    // RTDM_SGM_FUSE_SELECT=0 (A/B): the last direction writes S like the others and k_sgm_select reads it back
    static const int fuse_env = env_int("RTDM_SGM_FUSE_SELECT", 1);
    const bool fuse_select = fuse_env && wave_paths && aligned && g.D <= 256;
    const int last_dir = paths == 5 ? 5 : 7;
    SgmWin* win = (SgmWin*)b.gr;                     // the right image's bounds are dead once the pixel costs exist: 8 bytes per pixel
    // RTDM_SGM_SWEEP=0 (A/B): one pass per direction for the six that advance a row per step as well
    static const int sweep_env = env_int("RTDM_SGM_SWEEP", 1);
    bool sweep = sweep_env && half_paths && wave_paths && aligned16 && fuse_select && b.ring && b.abortf && *b.abortf == 0;
    bool swept_down = false, swept_up = false;
    // RTDM_SGM_DUAL=0 (A/B): the two horizontal directions one after the other (the second adds to S) instead of side by side
    static const int dual_env = env_int("RTDM_SGM_DUAL", 1);
    bool s2_pending = false;                         // S2 holds the (-1, 0) direction's L_r and has not been added to S yet
    for (int k = 0; k < 8; ++k) {
        const int dx = dirs[k][0], dy = dirs[k][1];
        if (paths == 5 && dy < 0) continue;          // MODE_SGBM's five directions: nothing runs upwards
        if (k == 0 && sweep && dual_env && b.S2 && (((size_t)b.S2) & 15) == 0 &&
            launch_sweep(paths == 5, g, b, 1, P1, P2, n, win, uniq, stream, nullptr, true)) {
            // both horizontal directions in one launch: (1, 0) -> S, (-1, 0) -> S2; the downward sweep adds the two up
            const dim3 hgrid((2 * g.H + 7) / 8, n);
#define RTDM_PATHD(N) hipLaunchKernelGGL((k_sgm_path_h<N, 8, false>), hgrid, blk, 0, stream, b.C, b.S, g, 1, 0, P1, P2, 1, g.H, win, uniq, b.S2)
            if (g.D <= 64) RTDM_PATHD(1); else if (g.D <= 128) RTDM_PATHD(2); else RTDM_PATHD(4);
#undef RTDM_PATHD
            s2_pending = true;
            continue;
        }
        if (k == 1 && s2_pending) continue;
        if (dy != 0) {
            // (0, dy), (+1, dy), (-1, dy) in one row-synchronous pass: -> <- down [up]; the last sweep decides the winners
            bool& done = dy > 0 ? swept_down : swept_up;
            if (done) continue;
            if (sweep && (k == 2 || k == 3)) {
                const bool last_sweep = paths == 5 || dy < 0;
                if (launch_sweep(last_sweep, g, b, dy, P1, P2, n, win, uniq, stream, s2_pending ? b.S2 : nullptr)) { done = true; s2_pending = false; continue; }
                sweep = false;                       // not launched: this and the remaining directions run as passes of their own
            }
            if (s2_pending) {                        // (the sweep that was to add S2 could not be launched)
                const size_t npairs = (size_t)n * g.H * g.W1 * g.D / 2;
                hipLaunchKernelGGL(k_sgm_add_s2, dim3((unsigned)((npairs + 255) / 256)), blk, 0, stream, (uint32_t*)b.S, (const uint32_t*)b.S2, npairs);
                s2_pending = false;
            }
        }
        const int lines = dy == 0 ? g.H : (dx == 0 ? g.W1 : g.W1 + g.H - 1);
        const bool last = fuse_select && k == last_dir;
        if (half_paths && wave_paths && aligned16 && g.D <= 256 && (!last || half_paths > 1)) {
            // half-wave lines, packed arithmetic: eight lines per workgroup (RTDM_SGM_HALF=1: all passes but the last)
            const dim3 hgrid((lines + 7) / 8, n);
            const int first = k == 0 ? 1 : 0;
#define RTDM_PATHH(N, P) do { if (last) hipLaunchKernelGGL((k_sgm_path_h<N, P, true>), hgrid, blk, 0, stream, b.C, b.S, g, dx, dy, P1, P2, first, lines, win, uniq, (uint16_t*)nullptr); \
                              else hipLaunchKernelGGL((k_sgm_path_h<N, P, false>), hgrid, blk, 0, stream, b.C, b.S, g, dx, dy, P1, P2, first, lines, win, uniq, (uint16_t*)nullptr); } while (0)
            if (g.D <= 64) RTDM_PATHH(1, 8); else if (g.D <= 128) RTDM_PATHH(2, 8); else RTDM_PATHH(4, 8);
#undef RTDM_PATHH
        } else if (wave_paths && aligned && g.D <= 256) {
            const dim3 wgrid((lines + 3) / 4, n);
            const int first = k == 0 ? 1 : 0;
#define RTDM_PATHW(N, P) do { if (last) hipLaunchKernelGGL((k_sgm_path_w<N, P, true>), wgrid, blk, 0, stream, b.C, b.S, g, dx, dy, P1, P2, first, lines, win, uniq); \
                              else hipLaunchKernelGGL((k_sgm_path_w<N, P, false>), wgrid, blk, 0, stream, b.C, b.S, g, dx, dy, P1, P2, first, lines, win, uniq); } while (0)
            switch (npl) {
                case 1: RTDM_PATHW(1, 8); break;
                case 2: RTDM_PATHW(2, 8); break;
                case 3: RTDM_PATHW(3, 8); break;
                default: RTDM_PATHW(4, 8); break;
            }
#undef RTDM_PATHW
        } else {
            hipLaunchKernelGGL(k_sgm_path, dim3(lines, n), dim3(threads), 0, stream, b.C, b.S, g, dx, dy, P1, P2, k == 0 ? 1 : 0);
        }
    }
    const bool speckle = speckleWindowSize > 0;                           // R11
    const size_t lds = (size_t)g.W * (8 + 2 + 2 + 2);
    // select -> a temporary plane (the bounds buffer of the left image is free again), median -> the caller's plane
    int16_t* tmp = (int16_t*)b.gl;
    const Plane16W tplane{tmp, (size_t)g.W, (size_t)g.W * g.H};
    if (fuse_select) hipLaunchKernelGGL(k_sgm_lrfinal, dim3(1, g.H, n), blk, lds, stream, win, tplane, g, disp12MaxDiff);
    else launch_select<false>((g.D + 63) / 64, dim3(1, g.H, n), lds, stream, b.S, tplane, g, uniq, disp12MaxDiff, b, 0);
    const size_t mlds = (size_t)g.W * 6;
    const int INV = (g.minD - 1) * 16;
    if (speckle) {
        hipLaunchKernelGGL((k_sgm_median<true>), dim3(1, g.H, n), blk, mlds, stream, tmp, disp, g.W, g.H, INV, b.label, b.size, b.runs,
                           b.rowcnt, b.headmap, 16 * speckleRange);
        launch_speckle(disp, b.label, b.size, b.runs, b.rowcnt, b.headmap, g.W, g.W, g.H, n, INV, speckleWindowSize,
                       16 * speckleRange, true, 1, 0, g.H, stream);
    } else {
        hipLaunchKernelGGL((k_sgm_median<false>), dim3(1, g.H, n), blk, mlds, stream, tmp, disp, g.W, g.H, INV, b.label, b.size, b.runs,
                           b.rowcnt, b.headmap, 0);
    }
}

}  // namespace rtdm
