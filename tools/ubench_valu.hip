// ubench_valu.hip -- issue cost of the VALU instructions the search kernels are made of, on gfx950 (run on the GPU box).
// Each kernel runs 16 independent copies of ONE instruction per loop trip, 8 waves per SIMD, and reports SIMD cycles per
// wave-instruction at the clock the chip held (s_memtime / s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048, UNROLL = 16;

#define KERNEL(NAME, ASM)                                                                                   \
__global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)                                   \
{                                                                                                           \
    uint32_t a[UNROLL], b = seed ^ threadIdx.x, c = seed * 3u + threadIdx.x;                                \
    uint64_t w = ((uint64_t)b << 32) | c, q[UNROLL];                                                        \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { a[u] = u + threadIdx.x; q[u] = w + u; }            \
    for (int it = 0; it < ITERS; ++it) {                                                                    \
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { ASM; }                                         \
    }                                                                                                       \
    uint32_t r = 0;                                                                                         \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) r ^= a[u] ^ (uint32_t)q[u] ^ (uint32_t)(q[u] >> 32); \
    out[blockIdx.x * 256 + threadIdx.x] = r;                                                                \
}

KERNEL(k_add,      asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[u]) : "v"(b)))
KERNEL(k_and,      asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[u]) : "v"(b)))
KERNEL(k_min,      asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(a[u]) : "v"(b)))
KERNEL(k_lshl,     asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[u])))
KERNEL(k_mov,      asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[u]) : "v"(b)))
KERNEL(k_cnd32,    asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(a[u]) : "v"(b) : "vcc"))
KERNEL(k_cnd64,    asm volatile("v_cndmask_b32_e64 %0, %1, %0, s[4:5]" : "+v"(a[u]) : "v"(b)))
KERNEL(k_cmp,      asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" :: "v"(a[u]), "v"(b) : "vcc"))
KERNEL(k_cmp64,    asm volatile("v_cmp_lt_u32_e64 s[6:7], %0, %1" :: "v"(a[u]), "v"(b) : "s6", "s7"))
KERNEL(k_min3,     asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_perm,     asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_align,    asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[u]) : "v"(b)))
KERNEL(k_add3,     asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_lshladd,  asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[u]) : "v"(b)))
KERNEL(k_bfe,      asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[u])))
KERNEL(k_andor,    asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_pkadd,    asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[u]) : "v"(b)))
KERNEL(k_pksub,    asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[u]) : "v"(b)))
KERNEL(k_pksubsat, asm volatile("v_pk_sub_u16 %0, %1, %0 clamp" : "+v"(a[u]) : "v"(b)))
KERNEL(k_pkmin,    asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[u]) : "v"(b)))
KERNEL(k_sad,      asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_msad,     asm volatile("v_msad_u8 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_qsad,     asm volatile("v_qsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(q[u]) : "v"(b)))
KERNEL(k_qsad2,    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[u]) : "v"(w), "v"(b)))
KERNEL(k_mqsad,    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[u]) : "v"(w), "v"(b)))
KERNEL(k_swap,     asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(a[u]), "+v"(b)))
KERNEL(k_dppmov,   asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[u]) : "v"(b)))
KERNEL(k_dppmin,   asm volatile("v_min_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[u]) : "v"(b)))
KERNEL(k_mullo,    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b)))
KERNEL(k_mul24,    asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(a[u]) : "v"(b)))
KERNEL(k_mad24,    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_dot2,     asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_dot4,     asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_pkmax,    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[u]) : "v"(b)))
KERNEL(k_sadu16,   asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_pkmad,    asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_readlane, asm volatile("v_readlane_b32 s8, %0, 3" :: "v"(a[u]) : "s8"))
KERNEL(k_max3,     asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_med3,     asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_sub,      asm volatile("v_sub_u32_e32 %0, %1, %0" : "+v"(a[u]) : "v"(b)))
KERNEL(k_xor,      asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[u]) : "v"(b)))
KERNEL(k_or3,      asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)))
KERNEL(k_max,      asm volatile("v_max_u32_e32 %0, %1, %0" : "+v"(a[u]) : "v"(b)))

template <typename K> static int run(const char* name, K kern)
{
    uint32_t* d; CHECK(hipMalloc(&d, 256 * 2048 * 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, 12345u);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    const double winstr = 2048.0 * 4 * ITERS * UNROLL;          // wave-instructions
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / winstr;         // SIMD cycles per wave-instruction at 2.4 GHz
    printf("%-28s %7.3f ms  %5.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, ms, cyc);
    CHECK(hipFree(d));
    return 0;
}
#define RUN(k) run(#k, k)
int main()
{
    RUN(k_add); RUN(k_sub); RUN(k_and); RUN(k_xor); RUN(k_min); RUN(k_max); RUN(k_lshl); RUN(k_mov); RUN(k_cnd32); RUN(k_cnd64); RUN(k_cmp); RUN(k_cmp64);
    RUN(k_min3); RUN(k_max3); RUN(k_med3); RUN(k_perm); RUN(k_align); RUN(k_add3); RUN(k_or3); RUN(k_lshladd); RUN(k_bfe); RUN(k_andor);
    RUN(k_pkadd); RUN(k_pksub); RUN(k_pksubsat); RUN(k_pkmin); RUN(k_pkmax); RUN(k_pkmad);
    RUN(k_sad); RUN(k_msad); RUN(k_sadu16); RUN(k_qsad); RUN(k_qsad2); RUN(k_mqsad); RUN(k_dot2); RUN(k_dot4);
    RUN(k_swap); RUN(k_dppmov); RUN(k_dppmin); RUN(k_readlane); RUN(k_mullo); RUN(k_mul24); RUN(k_mad24);
    return 0;
}
