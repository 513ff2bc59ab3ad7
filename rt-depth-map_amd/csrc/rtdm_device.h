// rtdm_device.h -- device-side helpers shared by the row kernels (k_basic.hip, k_sgm.hip):
// workgroup scans over a row held in LDS, union-find primitives, the speckle filter's per-row init.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtdm {

// Scan element = (number of run heads so far) << 16 | (x of the nearest head to the left + 1):
// one pass yields both the head of every pixel and the compact index of every run.
struct OpHead {
    static __device__ int id() { return 0; }
    static __device__ int f(int a, int b) { return (int)(((unsigned)a & 0xffff0000u) + ((unsigned)b & 0xffff0000u)) | max(a & 0xffff, b & 0xffff); }
};

// Inclusive scan of v[0..W) (int32 in LDS) in place.  Whole workgroup (any multiple of 64 threads up to 256);
// ends with a barrier.
template <typename Op>
__device__ __forceinline__ void row_scan(int* v, int W, int* wsum)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nt = blockDim.x;
    const int CH = (W + nt - 1) / nt;
    const int x0 = tid * CH;
    int run = Op::id();
    for (int k = 0; k < CH; ++k) {
        const int x = x0 + k;
        if (x < W) { run = Op::f(run, v[x]); v[x] = run; }
    }
    int t = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(t, o);
        if (lane >= o) t = Op::f(t, u);
    }
    __syncthreads();                 // previous users of wsum are done
    if (lane == 63) wsum[wv] = t;
    __syncthreads();
    int excl = __shfl_up(t, 1);
    if (lane == 0) excl = Op::id();
    for (int q = 0; q < wv; ++q) excl = Op::f(excl, wsum[q]);
    for (int k = 0; k < CH; ++k) {
        const int x = x0 + k;
        if (x < W) v[x] = Op::f(v[x], excl);
    }
    __syncthreads();
}

// truncating num / den for den > 0, |num| < 2^24, without the ~30-instruction integer division sequence
__device__ __forceinline__ int div_trunc_rcp(int num, int den)
{
    const unsigned an = (unsigned)(num < 0 ? -num : num);
    unsigned q = (unsigned)((float)an * __builtin_amdgcn_rcpf((float)den));
    int rem = (int)an - (int)(q * (unsigned)den);
    if (rem < 0) { --q; rem += den; }
    if (rem < 0) { --q; rem += den; }
    if (rem >= den) { ++q; rem -= den; }
    if (rem >= den) { ++q; }
    return num < 0 ? -(int)q : (int)q;
}

__device__ __forceinline__ bool conn(int a, int b, int newVal, int maxDiff)
{ return a != newVal && b != newVal && abs(a - b) <= maxDiff; }

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

// uf_union for a contact whose two nodes are usually still roots (the speckle merge: every node starts as one), with the
// "large run" shortcut of the speckle filter folded in -- all four first reads (both parents, both sizes) are issued
// together.  Agent-scope accesses cost a trip to memory each (~1 us), and a merge kernel on ONE frame is nothing but such
// chains: sizes -> find a -> find b -> hook was four trips per contact, this is two.
// Returns without uniting if at least one run is longer than maxSize; the other one is then marked (see spk_large_contact).
__device__ __forceinline__ void uf_union_contact(int32_t* parent, int32_t* size, int a, int b, int maxSize)
{
    int pa = ld_relaxed(&parent[a]), pb = ld_relaxed(&parent[b]);
    const bool la = ld_relaxed(&size[a]) > maxSize, lb = ld_relaxed(&size[b]) > maxSize;
    if (la || lb) {
        if (la != lb) atomicMax(&size[la ? b : a], maxSize + 1);
        return;
    }
    for (;;) {
        // walk both up to their roots (path halving as in uf_find; pa / pb are the parents already read)
        while (pa != a) { const int gp = ld_relaxed(&parent[pa]); if (gp == pa) { a = pa; break; } st_relaxed(&parent[a], gp); a = gp; pa = ld_relaxed(&parent[a]); }
        while (pb != b) { const int gp = ld_relaxed(&parent[pb]); if (gp == pb) { b = pb; break; } st_relaxed(&parent[b], gp); b = gp; pb = ld_relaxed(&parent[b]); }
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }    // a > b: hang a under b
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        pa = old; pb = b;                                // a was no longer a root: go on from its parent; b is (was) a root
    }
}

// Speckle "init" for one row held in LDS (d[0..W)): finds the horizontal runs, makes every run head
// its own parent with the run length as its size, appends (x | len << 16) to the row's run list and
// writes the per-pixel head map (x of the run head, int16) that the merge step reads.
// sc is a W-element int32 scratch array.  Whole workgroup.
// hm_lds (optional): the head map is also kept in LDS; hm_global = false skips the global head map
// (rows whose both neighbours are merged by the same workgroup never need it).
__device__ __forceinline__ void spk_row_init(const int16_t* d, int* sc, int* wsum, int W, int base,
                                             int32_t* label, int32_t* size, uint32_t* runs, int32_t* rowcnt,
                                             int16_t* headmap, int newVal, int maxDiff,
                                             int16_t* hm_lds = nullptr, bool hm_global = true)
{
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const int v = d[x];
        const bool head = v != newVal && !(x > 0 && conn(v, d[x - 1], newVal, maxDiff));
        sc[x] = head ? ((1 << 16) | (x + 1)) : 0;
    }
    __syncthreads();
    row_scan<OpHead>(sc, W, wsum);
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const int v = d[x];
        const int h = (sc[x] & 0xffff) - 1;
        if (hm_global) headmap[base + x] = (int16_t)h;
        if (hm_lds) hm_lds[x] = (int16_t)h;
        if (v == newVal) continue;
        const bool last = (x == W - 1) || !conn(v, d[x + 1], newVal, maxDiff);
        if (!last) continue;
        const int len = x - h + 1;
        label[base + h] = base + h;
        size[base + h] = len;
        runs[base + (sc[x] >> 16) - 1] = (uint32_t)h | ((uint32_t)len << 16);
    }
    if (threadIdx.x == 0) *rowcnt = sc[W - 1] >> 16;
}

}  // namespace rtdm
