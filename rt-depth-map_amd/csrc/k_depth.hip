// k_depth.hip -- the step right after the matcher, kept on the device:
//     left_disp /= 16.;  reprojectImageTo3D(left_disp, xyz, Q, true, CV_32F);  calc_depth(...)
// (/root/reference/estimator.cpp:75-77 and 206-263).  Only the per-object mean Z leaves the GPU.
// Semantics and the floating-point contract: oracle/depth_oracle.c (Z = (float)(Zh/Wh) from double
// arithmetic, mean = double sum / count).  The sum order here is fixed (tree per row, rows in order), so the
// result is reproducible run to run; against the oracle's row-major order it agrees to ~1e-15 relative.
#include "rtdm_kernels.h"

namespace rtdm {

__device__ __forceinline__ int rhe_div16(int d)     // d/16, ties to even (Mat /= 16. on CV_16S)
{
    int q = d >> 4;
    const int r = d & 15;
    if (r > 8 || (r == 8 && (q & 1))) ++q;
    return q;
}

__global__ __launch_bounds__(256) void k_depth_min(const int16_t* disp, size_t pitch_e, int W, int H, int* minval)
{
    __shared__ int red[256];
    int m = 0x7fffffff;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < W * H; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        m = min(m, rhe_div16(disp[(size_t)y * pitch_e + x]));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] = min(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
    if (threadIdx.x == 0) atomicMin(minval, red[0]);
}

// one workgroup = one row of one region: partial (sum of Z, count)
__global__ __launch_bounds__(256) void k_depth_rows(const int16_t* disp, size_t pitch_e, const uint8_t* mask, size_t mpitch,
                                                    DepthQ q, const int* regions, const int* minval, int maxH,
                                                    double* psum, int* pcnt)
{
    __shared__ double rs[256];
    __shared__ int rc[256];
    const int i = blockIdx.y, r = blockIdx.x;
    const int rx = regions[4 * i], ry = regions[4 * i + 1], rw = regions[4 * i + 2], rh = regions[4 * i + 3];
    double s = 0.0;
    int c = 0;
    if (r < rh) {
        const int y = ry + r, mind = *minval;
        for (int x = rx + threadIdx.x; x < rx + rw; x += 256) {
            const int d = rhe_div16(disp[(size_t)y * pitch_e + x]);
            const double Zh = q.q[8] * x + q.q[9] * y + q.q[10] * d + q.q[11];
            const double Wh = q.q[12] * x + q.q[13] * y + q.q[14] * d + q.q[15];
            float z = (float)(Zh / Wh);
            if (d == mind) z = 10000.0f;
            const double zd = (double)z;
            if (fabs(zd - 10000.0) < 1.1920928955078125e-07 || fabs(zd) > 10000.0 || mask[(size_t)y * mpitch + x] == 0) continue;
            s += zd; ++c;
        }
    }
    rs[threadIdx.x] = s; rc[threadIdx.x] = c;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) { rs[threadIdx.x] += rs[threadIdx.x + k]; rc[threadIdx.x] += rc[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { psum[(size_t)i * maxH + r] = rs[0]; pcnt[(size_t)i * maxH + r] = rc[0]; }
}

__global__ void k_depth_final(const double* psum, const int* pcnt, const int* regions, int n, int maxH, double unit,
                              double* mean, int* counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int rh = regions[4 * i + 3];
    double s = 0.0;
    int c = 0;
    for (int r = 0; r < rh; ++r) { s += psum[(size_t)i * maxH + r]; c += pcnt[(size_t)i * maxH + r]; }
    counts[i] = c;
    mean[i] = c > 0 ? (s / c) * unit / 10.0 : 0.0;
}

size_t depth_scratch_bytes(int max_regions, int maxH)
{ return 16 + (size_t)max_regions * 16 + (size_t)max_regions * maxH * 12 + (size_t)max_regions * 12 + 64; }

// scratch layout: [minval | regions n*4 int | psum n*maxH double | pcnt n*maxH int | mean n double | counts n int]
void launch_depth_stats(const int16_t* disp, size_t pitch_e, int W, int H, const DepthQ& q, const uint8_t* mask, size_t mpitch,
                        const int* h_regions, int n, int maxH, double unit, void* scratch, double* h_mean, int* h_counts,
                        hipStream_t stream)
{
    unsigned char* p = (unsigned char*)scratch;
    int* minval = (int*)p; p += 16;
    int* dreg = (int*)p; p += (size_t)n * 16;
    p = (unsigned char*)(((size_t)p + 7) & ~(size_t)7);
    double* psum = (double*)p; p += (size_t)n * maxH * 8;
    double* mean = (double*)p; p += (size_t)n * 8;
    int* pcnt = (int*)p; p += (size_t)n * maxH * 4;
    int* counts = (int*)p;
    // initial minimum written on the device (0x7f7f7f7f is above any int16): an async copy from a stack local would
    // only be safe if the runtime staged pageable sources at enqueue time
    (void)hipMemsetAsync(minval, 0x7f, 4, stream);
    (void)hipMemcpyAsync(dreg, h_regions, (size_t)n * 16, hipMemcpyHostToDevice, stream);
    hipLaunchKernelGGL(k_depth_min, dim3(256), dim3(256), 0, stream, disp, pitch_e, W, H, minval);
    if (n > 0) {
        hipLaunchKernelGGL(k_depth_rows, dim3(maxH, n), dim3(256), 0, stream, disp, pitch_e, mask, mpitch, q, dreg, minval, maxH, psum, pcnt);
        hipLaunchKernelGGL(k_depth_final, dim3((n + 63) / 64), dim3(64), 0, stream, psum, pcnt, dreg, n, maxH, unit, mean, counts);
        (void)hipMemcpyAsync(h_mean, mean, (size_t)n * 8, hipMemcpyDeviceToHost, stream);
        (void)hipMemcpyAsync(h_counts, counts, (size_t)n * 4, hipMemcpyDeviceToHost, stream);
    }
}

}  // namespace rtdm
