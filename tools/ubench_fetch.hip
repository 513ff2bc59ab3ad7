// ubench_fetch.hip -- calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths that
// k_search_fast uses (4-byte-per-lane loads, 2-byte-per-lane strided stores) on a known byte count.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o tools/ubench_fetch
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/ubench_fetch     (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read4(const uint32_t* in, uint32_t* out, size_t n)      // 4 B per lane
{ size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; uint32_t a = 0; for (; i < n; i += (size_t)gridDim.x * 256) a ^= in[i]; if (a == 0x12345) out[0] = a; }
__global__ __launch_bounds__(256) void k_read16(const uint4* in, uint32_t* out, size_t n)        // 16 B per lane
{ size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; uint32_t a = 0; for (; i < n; i += (size_t)gridDim.x * 256) { uint4 v = in[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; } if (a == 0x12345) out[0] = a; }
__global__ __launch_bounds__(256) void k_write2s(uint16_t* out, size_t n)                        // 2 B per lane, 8 B stride, 4 phases
{ size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * 256) { size_t t = i & 255, b = i & ~(size_t)255; out[b + (t >> 6) + 4 * (t & 63)] = (uint16_t)i; } }
__global__ __launch_bounds__(256) void k_write16(uint4* out, size_t n)                           // 16 B per lane
{ size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * 256) out[i] = make_uint4(i, 1, 2, 3); }

int main()
{
    const size_t bytes = (size_t)1 << 30;   // 1 GiB, far beyond the 256 MiB Infinity Cache
    void *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 0, bytes));
    hipLaunchKernelGGL(k_read4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, bytes / 4);
    hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const uint4*)a, (uint32_t*)b, bytes / 16);
    hipLaunchKernelGGL(k_write2s, dim3(4096), dim3(256), 0, 0, (uint16_t*)b, bytes / 2);
    hipLaunchKernelGGL(k_write16, dim3(4096), dim3(256), 0, 0, (uint4*)b, bytes / 16);
    CHECK(hipDeviceSynchronize());
    printf("each kernel touches %zu bytes\n", bytes);
    return 0;
}
