// k_basic.hip -- the HBM-bound per-pixel / per-row stages of the pipeline for gfx950:
//   K1 x-Sobel prefilter, FILTERED fill, K3 left-right check, K4 speckle filter.
// Semantics: SURVEY.md Appendix A.3a / A.4 / A.5 (what cv::StereoBM does behind
// /root/reference/stereo-matcher/bm-sw.cpp:35); oracle: oracle/bm_oracle.c.
#include "rtdm_kernels.h"

namespace rtdm {

// ---------------------------------------------------------------------------------------------
// K1 prefilter: one thread = 4 consecutive output bytes of one row (dword store).
// Rows come in pairs; a trailing odd row is all `cap`; row -1 mirrors to 1, row H to H-2.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prefilter(Plane8 L, Plane8 R, Plane8W Lp, Plane8W Rp,
                                                   int W, int H, int cap, int n)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int y = blockIdx.y;
    int f = blockIdx.z;
    if (x0 >= W) return;
    const bool right = f >= n;
    if (right) f -= n;
    const Plane8 S = right ? R : L;
    const Plane8W O = right ? Rp : Lp;
    const uint8_t* src = S.base + (size_t)f * S.frame;
    uint8_t* dst = O.base + (size_t)f * O.frame + (size_t)y * O.pitch;
    const int npair = (H >= 2) ? (H & ~1) : 0;
    uint8_t out[4];
    if (y >= npair) {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = (uint8_t)cap;
    } else {
        const int ya = (y > 0) ? y - 1 : 1;
        const int yb = (y < H - 1) ? y + 1 : H - 2;
        const uint8_t* ra = src + (size_t)ya * S.pitch;
        const uint8_t* rc = src + (size_t)y * S.pitch;
        const uint8_t* rb = src + (size_t)yb * S.pitch;
        // column sums s(x) = a + 2c + b for x0-1 .. x0+4 (clamped reads; clamped columns are
        // only consumed by outputs that are forced to `cap` anyway)
        int s[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int x = x0 - 1 + k;
            x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
            s[k] = (int)ra[x] + 2 * (int)rc[x] + (int)rb[x];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + k;
            int g = s[k + 2] - s[k];
            g = g < -cap ? -cap : (g > cap ? cap : g);
            out[k] = (x == 0 || x >= W - 1) ? (uint8_t)cap : (uint8_t)(g + cap);
        }
    }
    if (x0 + 3 < W && ((O.pitch & 3) == 0)) {
        *(uint32_t*)(dst + x0) = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) |
                                 ((uint32_t)out[3] << 24);
    } else {
        for (int k = 0; k < 4 && x0 + k < W; ++k) dst[x0 + k] = out[k];
    }
}

void launch_prefilter(Plane8 L, Plane8 R, Plane8W Lp, Plane8W Rp, int W, int H, int cap, int n,
                      hipStream_t stream)
{
    dim3 grid((W + 1023) / 1024, H, 2 * n);
    hipLaunchKernelGGL(k_prefilter, grid, dim3(256), 0, stream, L, R, Lp, Rp, W, H, cap, n);
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill16(Plane16W d, int W, int H, int value)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    d.base[(size_t)blockIdx.z * d.frame_e + (size_t)blockIdx.y * d.pitch_e + x] = (int16_t)value;
}

void launch_fill16(Plane16W disp, int W, int H, int n, int value, hipStream_t stream)
{
    hipLaunchKernelGGL(k_fill16, dim3((W + 255) / 256, H, n), dim3(256), 0, stream, disp, W, H, value);
}

// ---------------------------------------------------------------------------------------------
// K3 left-right check: one workgroup per (valid row, frame).  LDS holds a snapshot of the row and
// one 64-bit key per column: (cost << 32 | x); ds_min_u64 reproduces validateDisparity's pass 1
// ("strictly smaller cost wins, first x wins ties").  Pass 2 reads the snapshot, so the in-place
// update cannot race.  Columns outside the valid rectangle are masked in the same pass.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lrcheck(Plane16W disp, const int32_t* cost, BMGeom g, int maxDiff16)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* key = (unsigned long long*)smem;            // W
    int16_t* snap = (int16_t*)(smem + (size_t)g.W * 8);             // W
    const int y = g.vy0 + blockIdx.y;
    const int f = blockIdx.z;
    int16_t* row = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    const int32_t* crow = cost + ((size_t)f * g.H + y) * g.W;
    const int W = g.W, INV = g.filtered;
    for (int x = threadIdx.x; x < W; x += 256) { key[x] = ~0ull; snap[x] = row[x]; }
    __syncthreads();
    const int minX1 = max(g.minD + g.D, 0), maxX1 = W + min(g.minD, 0);
    for (int x = minX1 + threadIdx.x; x < maxX1; x += 256) {
        const int d = snap[x];
        if (d == INV) continue;
        const int x2 = x - ((d + 8) >> 4);
        if (x2 < 0 || x2 >= W) continue;
        const unsigned long long k = ((unsigned long long)(unsigned)crow[x] << 32) | (unsigned)x;
        atomicMin(&key[x2], k);
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        int d = snap[x];
        bool kill = (x < g.vx0 || x >= g.vx1);
        if (!kill && d != INV && x >= minX1 && x < maxX1) {
            const int x0 = x - (d >> 4), x1 = x - ((d + 15) >> 4);
            bool bad0 = false, bad1 = false;
            if (x0 >= 0 && x0 < W && key[x0] != ~0ull) {
                const int d2 = snap[(unsigned)(key[x0] & 0xffffffffu)];
                bad0 = abs(d2 - d) > maxDiff16;
            }
            if (x1 >= 0 && x1 < W && key[x1] != ~0ull) {
                const int d2 = snap[(unsigned)(key[x1] & 0xffffffffu)];
                bad1 = abs(d2 - d) > maxDiff16;
            }
            kill = bad0 && bad1;
        }
        if (kill && d != INV) row[x] = (int16_t)INV;
    }
}

void launch_lrcheck(Plane16W disp, const int32_t* cost, const BMGeom& g, int disp12MaxDiff, int n,
                    hipStream_t stream)
{
    const size_t lds = (size_t)g.W * 10;
    hipLaunchKernelGGL(k_lrcheck, dim3(1, g.vy1 - g.vy0, n), dim3(256), lds, stream, disp, cost, g,
                       disp12MaxDiff * 16);
}

// ---------------------------------------------------------------------------------------------
// K4 speckle filter: union-find over the pixel graph (4-neighbour edges with |a-b| <= maxDiff
// between pixels != newVal).  Parent pointers only ever decrease, all updates are device-scope
// atomicMin, so stale reads are still ancestors and the result (which components are small) is
// independent of scheduling.  Components never span frames.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(const int32_t* parent, int x)
{
    int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

__device__ __forceinline__ void uf_union(int32_t* parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }   // a > b: hang a under b
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;                                   // a was no longer a root; retry from its parent
    }
}

__global__ __launch_bounds__(256) void k_spk_init(Plane16W disp, int32_t* label, int32_t* size, int W, int H, int newVal)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y = blockIdx.y, f = blockIdx.z;
    const int idx = (f * H + y) * W + x;
    const int d = disp.base[(size_t)f * disp.frame_e + (size_t)y * disp.pitch_e + x];
    label[idx] = (d != newVal) ? idx : -1;
    size[idx] = 0;
}

__global__ __launch_bounds__(256) void k_spk_merge(Plane16W disp, int32_t* label, int W, int H, int newVal, int maxDiff)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y = blockIdx.y, f = blockIdx.z;
    const int16_t* base = disp.base + (size_t)f * disp.frame_e;
    const int d = base[(size_t)y * disp.pitch_e + x];
    if (d == newVal) return;
    const int idx = (f * H + y) * W + x;
    if (x + 1 < W) {
        const int e = base[(size_t)y * disp.pitch_e + x + 1];
        if (e != newVal && abs(d - e) <= maxDiff) uf_union(label, idx, idx + 1);
    }
    if (y + 1 < H) {
        const int e = base[(size_t)(y + 1) * disp.pitch_e + x];
        if (e != newVal && abs(d - e) <= maxDiff) uf_union(label, idx, idx + W);
    }
}

__global__ __launch_bounds__(256) void k_spk_count(int32_t* label, int32_t* size, int total)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    if (label[idx] < 0) return;
    const int root = uf_find(label, idx);
    label[idx] = root;            // roots are final here (no unions in this launch)
    atomicAdd(&size[root], 1);
}

__global__ __launch_bounds__(256) void k_spk_apply(Plane16W disp, const int32_t* label, const int32_t* size,
                                                   int W, int H, int newVal, int maxSize)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y = blockIdx.y, f = blockIdx.z;
    const int idx = (f * H + y) * W + x;
    const int root = label[idx];
    if (root < 0) return;
    if (size[root] <= maxSize)
        disp.base[(size_t)f * disp.frame_e + (size_t)y * disp.pitch_e + x] = (int16_t)newVal;
}

void launch_speckle(Plane16W disp, int32_t* label, int32_t* size, int W, int H, int n, int newVal,
                    int maxSize, int maxDiff, hipStream_t stream)
{
    dim3 grid((W + 255) / 256, H, n), block(256);
    const int total = n * W * H;
    hipLaunchKernelGGL(k_spk_init, grid, block, 0, stream, disp, label, size, W, H, newVal);
    hipLaunchKernelGGL(k_spk_merge, grid, block, 0, stream, disp, label, W, H, newVal, maxDiff);
    hipLaunchKernelGGL(k_spk_count, dim3((total + 255) / 256), block, 0, stream, label, size, total);
    hipLaunchKernelGGL(k_spk_apply, grid, block, 0, stream, disp, label, size, W, H, newVal, maxSize);
}

}  // namespace rtdm
