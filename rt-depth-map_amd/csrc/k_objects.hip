// k_objects.hip -- the object detection that produces the matcher's ROI, on the device:
//     cvtColor(BGR2HSV) + inRange                     (/root/reference/estimator.cpp:40-43)   -> k_hsv_inrange
//     [morphFilter->run: k_morph.hip]                 (estimator.cpp:45)
//     findContours(RETR_EXTERNAL) + boundingRect      (estimator.cpp:47, 167-174)             -> k_cc_*
// Semantics: oracle/objects_oracle.c.  Connected components by union-find over one label plane: foreground pixels
// are united 8-connected, background pixels 4-connected, background on the image border with a virtual frame node
// (index W*H).  Roots are the smallest index of a set = the component's first pixel in raster order, which is what
// decides both the contour order and (through the background left of it) whether the component is external.
#include <mutex>
#include "rtdm_kernels.h"
#include "rtdm_device.h"

namespace rtdm {

struct HsvTabs { int sdiv[256], hdiv[256]; };
static __device__ HsvTabs g_hsv;

__global__ void k_hsv_tabs()
{
    const int i = threadIdx.x;
    // round(255*4096 / i), round(180*4096 / (6 i)): exact in integers (no quotient is a tie)
    g_hsv.sdiv[i] = i ? ((255 << 12) * 2 + i) / (2 * i) : 0;
    g_hsv.hdiv[i] = i ? ((180 << 12) * 2 + 6 * i) / (12 * i) : 0;
}

__global__ __launch_bounds__(256) void k_hsv_inrange(const uint8_t* rgb, size_t pitch, int W, int H, int lh, int ls, int lv,
                                                     int hh, int hs, int hv, uint8_t* mask, size_t mpitch)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W * H) return;
    const int y = idx / W, x = idx - y * W;
    const uint8_t* p = rgb + (size_t)y * pitch + (size_t)x * 3;
    const int r = p[0], g = p[1], b = p[2];
    const int v = max(r, max(g, b)), vmin = min(r, min(g, b)), diff = v - vmin;
    const int s = (diff * g_hsv.sdiv[v] + (1 << 11)) >> 12;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * g_hsv.hdiv[diff] + (1 << 11)) >> 12;
    if (h < 0) h += 180;
    h = min(max(h, 0), 255);
    const bool ok = h >= lh && h <= hh && s >= ls && s <= hs && v >= lv && v <= hv;
    mask[(size_t)y * mpitch + x] = ok ? 255 : 0;
}

__device__ __forceinline__ bool cc_fg(const uint8_t* mask, size_t mpitch, int W, int H, int x, int y, int zero_border)
{
    if (zero_border && (x == 0 || y == 0 || x == W - 1 || y == H - 1)) return false;
    return mask[(size_t)y * mpitch + x] != 0;
}

struct OpMax {
    static __device__ int id() { return 0; }
    static __device__ int f(int a, int b) { return max(a, b); }
};

// One workgroup per row: every pixel's parent becomes the first pixel of its horizontal run of same-kind pixels
// (max-scan of "run starts here" flags), so no union is ever needed along a row.
__global__ __launch_bounds__(256) void k_cc_rows(const uint8_t* mask, size_t mpitch, int32_t* label, int4* box, int W, int H,
                                                 int zero_border, int* count)
{
    extern __shared__ int sc[];
    __shared__ int wsum[4];
    const int y = blockIdx.x;
    for (int x = threadIdx.x; x < W; x += 256) {
        const bool v = cc_fg(mask, mpitch, W, H, x, y, zero_border);
        const bool start = x == 0 || cc_fg(mask, mpitch, W, H, x - 1, y, zero_border) != v;
        sc[x] = start ? x + 1 : 0;
    }
    __syncthreads();
    row_scan<OpMax>(sc, W, wsum);
    for (int x = threadIdx.x; x < W; x += 256) {
        label[y * W + x] = y * W + sc[x] - 1;
        box[y * W + x] = make_int4(x, y, x, y);
    }
    if (y == 0 && threadIdx.x == 0) { label[W * H] = W * H; *count = 0; }
}

__global__ __launch_bounds__(256) void k_cc_merge(const uint8_t* mask, size_t mpitch, int32_t* label, int W, int H, int zero_border)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= W * H) return;
    const int y = p / W, x = p - y * W;
    const bool v = cc_fg(mask, mpitch, W, H, x, y, zero_border);
    const bool hasL = x > 0, hasU = y > 0;
    const bool l = hasL && cc_fg(mask, mpitch, W, H, x - 1, y, zero_border) == v;            // left neighbour of my kind
    const bool u = hasU && cc_fg(mask, mpitch, W, H, x, y - 1, zero_border) == v;            // upper neighbour of my kind
    const bool ul = hasL && hasU && cc_fg(mask, mpitch, W, H, x - 1, y - 1, zero_border) == v;
    // a vertical contact needs one union per segment: if my left neighbour and the pixel above it are of my kind too,
    // they made the link for both runs (rows are already joined by k_cc_rows)
    if (u && !(l && ul)) uf_union(label, p, p - W);
    if (v) {
        // foreground is 8-connected; the diagonals matter only when the pixel above is background
        if (!u && ul) uf_union(label, p, p - W - 1);
        if (!u && hasU && x + 1 < W && cc_fg(mask, mpitch, W, H, x + 1, y - 1, zero_border)) uf_union(label, p, p - W + 1);
    } else if ((x == 0 || y == 0 || x == W - 1 || y == H - 1) && !(l && (y == 0 || y == H - 1)) && !(u && (x == 0 || x == W - 1))) {
        uf_union(label, p, W * H);                       // background on the image border hangs on the frame node (once per border run)
    }
}

// Boxes from the horizontal runs only: a run's first pixel reports (min x, min y, max y), its last pixel (max x).
// A blob of 100k pixels would otherwise serialise 400k atomics on the four words of its root.
__global__ __launch_bounds__(256) void k_cc_bbox(const uint8_t* mask, size_t mpitch, int32_t* label, int4* box, int W, int H, int zero_border)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= W * H) return;
    const int y = p / W, x = p - y * W;
    if (!cc_fg(mask, mpitch, W, H, x, y, zero_border)) return;
    const bool first = x == 0 || !cc_fg(mask, mpitch, W, H, x - 1, y, zero_border);
    const bool last = x == W - 1 || !cc_fg(mask, mpitch, W, H, x + 1, y, zero_border);
    if (!first && !last) return;
    const int r = uf_find(label, p);
    if (r == p) return;                                  // the root's own coordinates are its initial box
    int* b = (int*)&box[r];
    if (first) { atomicMin(b + 0, x); atomicMin(b + 1, y); atomicMax(b + 3, y); }
    if (last) atomicMax(b + 2, x);
}

// one record per foreground root: first pixel, box, external flag
__global__ __launch_bounds__(256) void k_cc_collect(const uint8_t* mask, size_t mpitch, int32_t* label, const int4* box, int W, int H,
                                                    int zero_border, int* count, int* records, int max_records)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= W * H) return;
    const int y = p / W, x = p - y * W;
    if (!cc_fg(mask, mpitch, W, H, x, y, zero_border) || ld_relaxed(&label[p]) != p) return;
    const bool external = x == 0 || uf_find(label, p - 1) == uf_find(label, W * H);
    const int slot = atomicAdd(count, 1);
    if (slot >= max_records) return;
    const int4 b = box[p];
    int* r = records + 6 * slot;
    r[0] = p; r[1] = b.x; r[2] = b.y; r[3] = b.z - b.x + 1; r[4] = b.w - b.y + 1; r[5] = external ? 1 : 0;
}

void launch_hsv_inrange(const uint8_t* rgb, size_t pitch, int W, int H, const int lo[3], const int hi[3], uint8_t* mask,
                        size_t mpitch, hipStream_t stream)
{
    // the tables live in device globals: filled once per device, and finished before any thread's launch can use them
    // (std::call_once holds later callers back until the first has synchronised)
    static std::once_flag tabs[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto fill = [&] {
        hipLaunchKernelGGL(k_hsv_tabs, dim3(1), dim3(256), 0, stream);
        (void)hipStreamSynchronize(stream);
    };
    if (dev >= 0 && dev < 64) std::call_once(tabs[dev], fill);
    else fill();
    hipLaunchKernelGGL(k_hsv_inrange, dim3((W * H + 255) / 256), dim3(256), 0, stream, rgb, pitch, W, H, lo[0], lo[1], lo[2],
                       hi[0], hi[1], hi[2], mask, mpitch);
}

size_t cc_scratch_bytes(int W, int H, int max_records)
{ return ((size_t)W * H + 1) * 4 + 64 + (size_t)W * H * 16 + 64 + (size_t)max_records * 24 + 64; }

// label: W*H+1 ints, box: W*H int4, count: 1 int, records: max_records x 6 ints -- all carved out of `scratch`.
// On return the kernels are enqueued; *d_count / *d_records point at the results inside scratch.
void launch_cc_boxes(const uint8_t* mask, size_t mpitch, int W, int H, int zero_border, void* scratch, int max_records,
                     int** d_count, int** d_records, hipStream_t stream)
{
    const int N = W * H;
    uint8_t* s = (uint8_t*)scratch;
    int4* box = (int4*)s;                         s += (size_t)N * 16 + 64;
    int32_t* label = (int32_t*)s;                 s += (((size_t)N + 1) * 4 + 63) & ~(size_t)63;
    int* records = (int*)s;                       s += (size_t)max_records * 24;
    int* count = (int*)s;
    const dim3 grid((N + 1 + 255) / 256), block(256);
    hipLaunchKernelGGL(k_cc_rows, dim3(H), block, (size_t)W * sizeof(int), stream, mask, mpitch, label, box, W, H, zero_border, count);
    hipLaunchKernelGGL(k_cc_merge, grid, block, 0, stream, mask, mpitch, label, W, H, zero_border);
    hipLaunchKernelGGL(k_cc_bbox, grid, block, 0, stream, mask, mpitch, label, box, W, H, zero_border);
    hipLaunchKernelGGL(k_cc_collect, grid, block, 0, stream, mask, mpitch, label, box, W, H, zero_border, count, records, max_records);
    *d_count = count; *d_records = records;
}

}  // namespace rtdm
