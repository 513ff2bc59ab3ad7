// ubench_sad.hip -- throughput + semantics probe of the packed-u8 SAD instructions on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_sad.hip -o tools/ubench_sad ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096, UNROLL = 16;

template <int OP>
__global__ __launch_bounds__(256) void k_tp(uint64_t* out, uint32_t seed)
{
    uint64_t acc[UNROLL];
    uint32_t a = seed * (threadIdx.x + 1), b = seed ^ (threadIdx.x * 2654435761u);
    uint64_t w = ((uint64_t)a << 32) | b;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc[u] = u;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == 0) acc[u] = __builtin_amdgcn_qsad_pk_u16_u8(w + u, b, acc[u]);
            if (OP == 1) acc[u] = __builtin_amdgcn_mqsad_pk_u16_u8(w + u, b, acc[u]);
            if (OP == 2) acc[u] = __builtin_amdgcn_sad_u8((uint32_t)acc[u] + a, b, (uint32_t)acc[u]);
            if (OP == 3) acc[u] = __builtin_amdgcn_msad_u8((uint32_t)acc[u] + a, b, (uint32_t)acc[u]);
            if (OP == 4) { typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                us2 x = __builtin_bit_cast(us2, (uint32_t)acc[u]), y = __builtin_bit_cast(us2, b);
                x = x + y; acc[u] = __builtin_bit_cast(uint32_t, x); }
            if (OP == 5) { typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                us2 x = __builtin_bit_cast(us2, (uint32_t)acc[u]), y = __builtin_bit_cast(us2, b);
                x = __builtin_elementwise_min(x, y); acc[u] = __builtin_bit_cast(uint32_t, x) + u; }
            if (OP == 6) acc[u] = __builtin_amdgcn_perm((uint32_t)acc[u], a, b);
            if (OP == 7) acc[u] = (uint32_t)acc[u] + a;
            if (OP == 8) acc[u] = __builtin_amdgcn_sad_u16((uint32_t)acc[u] + a, b, (uint32_t)acc[u]);
            if (OP == 9) acc[u] = __builtin_amdgcn_alignbyte((uint32_t)acc[u], a, b & 3);
            if (OP == 10) acc[u] = min((uint32_t)acc[u] + 1u, min(a, b + u));   // v_min3_u32
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) r ^= acc[u];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ void k_sem(uint64_t* out)
{
    // S0 = bytes 0..7 = {10,20,30,40,50,60,70,80}; S1 = {11,22,33,44}; acc halves = {1,2,3,4}
    const uint64_t s0 = 0x5046'3C32'281E'140Aull; const uint32_t s1 = 0x2C21160Bu; const uint64_t acc = 0x0004000300020001ull;
    out[0] = __builtin_amdgcn_qsad_pk_u16_u8(s0, s1, acc);
    out[1] = __builtin_amdgcn_mqsad_pk_u16_u8(s0, 0x2C00160Bu, acc);   // reference byte 2 = 0 -> masked?
    out[2] = __builtin_amdgcn_msad_u8(0x281E140Au, 0x2C00160Bu, 100u);
    out[3] = __builtin_amdgcn_msad_u8(0x2C00160Bu, 0x281E140Au, 100u);
    out[4] = __builtin_amdgcn_sad_u8(0x281E140Au, 0x2C21160Bu, 100u);
    out[5] = __builtin_amdgcn_sad_u16(0x00140005u, 0x000A0009u, 7u);
    out[6] = __builtin_amdgcn_perm(0x33221100u, 0x77665544u, 0x05010400u);
    out[7] = __builtin_amdgcn_alignbyte(0x77665544u, 0x33221100u, 1);
}

template <int OP> static int run(const char* name, int lanes_per_op)
{
    uint64_t* d; CHECK(hipMalloc(&d, 256 * 2048 * 8));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_tp<OP>, dim3(2048), dim3(256), 0, 0, d, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_tp<OP>, dim3(2048), dim3(256), 0, 0, d, 12345u);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double ops = 2048.0 * 256 * ITERS * UNROLL;
    printf("%-22s %8.3f ms  %8.2f Tlane-op/s  (%.2f wave-instr/clk/CU @2.4GHz)\n", name, ms, ops / ms / 1e9,
           ops / 64 / (ms * 1e-3) / 256 / 2.4e9);
    (void)lanes_per_op;
    CHECK(hipFree(d));
    return 0;
}

int main()
{
    uint64_t* d; CHECK(hipMalloc(&d, 64));
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(1), 0, 0, d);
    uint64_t h[8]; CHECK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
    // expected qsad: shift0: |10-11|+|20-22|+|30-33|+|40-44| = 10 (+1) ; shift1: |20-11|+|30-22|+|40-33|+|50-44| = 30 (+2)
    //                shift2: 50 (+3) ; shift3: 70 (+4)
    printf("qsad  = %016llx (expect 004a 0035 0020 000b)\n", (unsigned long long)h[0]);
    printf("mqsad = %016llx (ref byte2=0 masked -> shift0: 1+2+4=7(+1)=8, shift1: 9+8+6=23(+2), ...)\n", (unsigned long long)h[1]);
    printf("msad(S0=data,S1=ref with zero byte2) = %llu (100+1+2+4=107 if S1 is the masking reference)\n", (unsigned long long)h[2]);
    printf("msad(swapped)                        = %llu\n", (unsigned long long)h[3]);
    printf("sad_u8 = %llu (expect 110)\n", (unsigned long long)h[4]);
    printf("sad_u16 = %llu (expect 7+4+10=21)\n", (unsigned long long)h[5]);
    printf("perm = %08llx, alignbyte = %08llx (expect 44332211)\n", (unsigned long long)h[6], (unsigned long long)h[7]);
    run<7>("v_add_u32 (baseline)", 1);
    run<0>("v_qsad_pk_u16_u8", 1);
    run<1>("v_mqsad_pk_u16_u8", 1);
    run<2>("v_sad_u8 (+add)", 1);
    run<3>("v_msad_u8 (+add)", 1);
    run<8>("v_sad_u16 (+add)", 1);
    run<4>("v_pk_add_u16", 1);
    run<5>("v_pk_min_u16 (+add)", 1);
    run<6>("v_perm_b32", 1);
    run<9>("v_alignbyte_b32", 1);
    run<10>("v_min3_u32 (+add)", 1);
    return 0;
}
