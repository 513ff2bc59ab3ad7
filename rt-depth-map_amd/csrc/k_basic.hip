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
// K4 speckle filter (cv::filterSpeckles as called by cv::StereoBM::compute, SURVEY.md Appendix
// A.5): 4-connected components of pixels != newVal under |a-b| <= maxDiff; components with
// size <= maxSize become newVal.  Run-based union-find:
//   init   one workgroup per row: a max-scan turns the "connected to my left neighbour" flags into
//          run heads; every pixel of a horizontal run is represented by its head, which starts as
//          its own parent and carries the run length.
//   merge  one workgroup per row pair: ONE union per vertical contact segment between two runs
//          (a pixel is skipped when its left neighbour already linked the same two runs).
//   count  every non-root head adds its run length to its root (skipped once the root is known to
//          be large: only "<= maxSize" matters) and is re-pointed straight at the root.
//   apply  one workgroup per row: heads chase to their root, look its size up once, pixels read the flag.
// Parent pointers only ever move to smaller indices of the same component and every hook is a
// device-scope atomicMin on a root, so stale reads are still ancestors and the set of small
// components is independent of scheduling.  Components never span frames.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ld_relaxed(const int32_t* p)
{ return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_relaxed(int32_t* p, int v)
{ __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(int32_t* parent, int x)   // with path halving
{
    for (;;) {
        const int p = ld_relaxed(&parent[x]);
        if (p == x) return x;
        const int gp = ld_relaxed(&parent[p]);
        if (gp == p) return p;
        st_relaxed(&parent[x], gp);
        x = gp;
    }
}

__device__ __forceinline__ void uf_union(int32_t* parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }   // a > b: hang a under b
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;                                         // a was no longer a root; retry from its parent
    }
}

// Inclusive max-scan over hp[0..W) (LDS, int16: x at run heads, -1 elsewhere), 256 threads,
// contiguous chunks per thread.  Must be called by the whole workgroup; ends with a barrier.
__device__ __forceinline__ void head_scan(int16_t* hp, int W, int* wsum)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int CH = (W + 255) >> 8;
    const int x0 = tid * CH;
    int run = -1;
    for (int k = 0; k < CH; ++k) {
        const int x = x0 + k;
        if (x < W) { run = max(run, (int)hp[x]); hp[x] = (int16_t)run; }
    }
    int v = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o);
        if (lane >= o) v = max(v, u);
    }
    if (lane == 63) wsum[wv] = v;
    __syncthreads();
    int excl = __shfl_up(v, 1);
    if (lane == 0) excl = -1;
    for (int q = 0; q < wv; ++q) excl = max(excl, wsum[q]);
    for (int k = 0; k < CH; ++k) {
        const int x = x0 + k;
        if (x < W) hp[x] = (int16_t)max((int)hp[x], excl);
    }
    __syncthreads();
}

__device__ __forceinline__ bool conn(int a, int b, int newVal, int maxDiff)
{ return a != newVal && b != newVal && abs(a - b) <= maxDiff; }

__global__ __launch_bounds__(256) void k_spk_init(Plane16W disp, int32_t* label, int32_t* size, int W, int H,
                                                  int newVal, int maxDiff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int16_t* d = (int16_t*)smem;          // W
    int16_t* hp = d + W;                  // W
    __shared__ int wsum[4];
    const int y = blockIdx.y, f = blockIdx.z;
    const int16_t* row = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    const int base = (f * H + y) * W;
    for (int x = threadIdx.x; x < W; x += 256) d[x] = row[x];
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        const int v = d[x];
        const bool head = v != newVal && !(x > 0 && conn(v, d[x - 1], newVal, maxDiff));
        hp[x] = head ? (int16_t)x : (int16_t)-1;
    }
    __syncthreads();
    head_scan(hp, W, wsum);
    for (int x = threadIdx.x; x < W; x += 256) {
        const int v = d[x];
        if (v == newVal) continue;
        const int h = hp[x];
        if (h == x) label[base + x] = base + x;
        const bool last = (x == W - 1) || !conn(v, d[x + 1], newVal, maxDiff);
        if (last) size[base + h] = x - h + 1;
    }
}

__global__ __launch_bounds__(256) void k_spk_merge(Plane16W disp, int32_t* label, int W, int H, int newVal, int maxDiff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int16_t* d0 = (int16_t*)smem;
    int16_t* d1 = d0 + W;
    int16_t* h0 = d1 + W;
    int16_t* h1 = h0 + W;
    __shared__ int wsum[4];
    const int y = blockIdx.y, f = blockIdx.z;      // rows y and y+1
    const int16_t* r0 = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    const int16_t* r1 = r0 + disp.pitch_e;
    const int base0 = (f * H + y) * W, base1 = base0 + W;
    for (int x = threadIdx.x; x < W; x += 256) { d0[x] = r0[x]; d1[x] = r1[x]; }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        const int a = d0[x], b = d1[x];
        h0[x] = (a != newVal && !(x > 0 && conn(a, d0[x - 1], newVal, maxDiff))) ? (int16_t)x : (int16_t)-1;
        h1[x] = (b != newVal && !(x > 0 && conn(b, d1[x - 1], newVal, maxDiff))) ? (int16_t)x : (int16_t)-1;
    }
    __syncthreads();
    head_scan(h0, W, wsum);
    head_scan(h1, W, wsum);
    for (int x = threadIdx.x; x < W; x += 256) {
        if (!conn(d0[x], d1[x], newVal, maxDiff)) continue;
        const bool dup = x > 0 && conn(d0[x - 1], d1[x - 1], newVal, maxDiff) && h0[x - 1] == h0[x] && h1[x - 1] == h1[x];
        if (!dup) uf_union(label, base0 + h0[x], base1 + h1[x]);
    }
}

__global__ __launch_bounds__(256) void k_spk_count(Plane16W disp, int32_t* label, int32_t* size, int W, int H,
                                                   int newVal, int maxDiff, int maxSize)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y = blockIdx.y, f = blockIdx.z;
    const int16_t* row = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    const int v = row[x];
    if (v == newVal) return;
    if (x > 0 && conn(v, row[x - 1], newVal, maxDiff)) return;      // not a run head
    const int idx = (f * H + y) * W + x;
    const int root = uf_find(label, idx);
    if (root == idx) return;
    st_relaxed(&label[idx], root);             // roots are final in this launch
    if (ld_relaxed(&size[root]) <= maxSize) atomicAdd(&size[root], size[idx]);
}

__global__ __launch_bounds__(256) void k_spk_apply(Plane16W disp, const int32_t* label, const int32_t* size,
                                                   int W, int H, int newVal, int maxDiff, int maxSize)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int16_t* d = (int16_t*)smem;          // W
    int16_t* hp = d + W;                  // W
    uint8_t* small = (uint8_t*)(hp + W);  // W
    __shared__ int wsum[4];
    const int y = blockIdx.y, f = blockIdx.z;
    int16_t* row = disp.base + (size_t)f * disp.frame_e + (size_t)y * disp.pitch_e;
    const int base = (f * H + y) * W;
    for (int x = threadIdx.x; x < W; x += 256) d[x] = row[x];
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        const int v = d[x];
        const bool head = v != newVal && !(x > 0 && conn(v, d[x - 1], newVal, maxDiff));
        hp[x] = head ? (int16_t)x : (int16_t)-1;
        if (head) {
            // after k_spk_count a head is at most a couple of hops from its root (a late path-halving
            // store of another thread may have left an ancestor instead of the root), so chase it
            int root = base + x;
            for (int p = label[root]; p != root; p = label[root]) root = p;
            small[x] = size[root] <= maxSize;
        }
    }
    __syncthreads();
    head_scan(hp, W, wsum);
    for (int x = threadIdx.x; x < W; x += 256)
        if (d[x] != newVal && small[hp[x]]) row[x] = (int16_t)newVal;
}

void launch_speckle(Plane16W disp, int32_t* label, int32_t* size, int W, int H, int n, int newVal,
                    int maxSize, int maxDiff, hipStream_t stream)
{
    dim3 rows(1, H, n), block(256);
    hipLaunchKernelGGL(k_spk_init, rows, block, (size_t)W * 4, stream, disp, label, size, W, H, newVal, maxDiff);
    if (H > 1)
        hipLaunchKernelGGL(k_spk_merge, dim3(1, H - 1, n), block, (size_t)W * 8, stream, disp, label, W, H, newVal, maxDiff);
    hipLaunchKernelGGL(k_spk_count, dim3((W + 255) / 256, H, n), block, 0, stream, disp, label, size, W, H, newVal, maxDiff, maxSize);
    hipLaunchKernelGGL(k_spk_apply, rows, block, (size_t)W * 5, stream, disp, label, size, W, H, newVal, maxDiff, maxSize);
}

}  // namespace rtdm
