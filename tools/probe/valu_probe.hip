// Issue rate of the integer vector instructions the kernels lean on: cycles per wave64
// instruction per SIMD with 8 waves per SIMD, 16 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_op(unsigned* out, int iters, unsigned seed) {
    unsigned v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 2654435761u + i * seed;
    const unsigned c = seed | 1u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 2) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(0x03020100u));
            if (OP == 4) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 7) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 8) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(v[i]) : "v"(c));
            if (OP == 9) asm volatile("v_max3_i32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 10) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 11) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c));
            if (OP == 13) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 14) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 15) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 17) asm volatile("v_med3_i32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 18) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 19) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 20) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(v[i]), "v"(c) : "vcc");
            if (OP == 21) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(unsigned long long*)&v[i & ~1]) : "v"(*(unsigned long long*)&v[(i & ~1)]));
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= v[i];
    if (acc == 0x12345u) out[0] = acc;
}

template <int OP>
void run(const char* name, unsigned* out, double clockGHz) {
    const int iters = 2000;
    const int blocks = 256 * 8 * 4;  // 8 blocks of 4 waves per CU, 4 rounds
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_op<OP>), dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_op<OP>), dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double waveInstr = (double)blocks * 4 * iters * 16;
    const double perSimdPerSec = waveInstr / 1024.0 / (ms * 1e-3);
    printf("%-18s %.3f ms  %.2f cycles per wave64 instruction per SIMD at %.2f GHz\n", name, ms, clockGHz * 1e9 / perSimdPerSec, clockGHz);
}

int main() {
    unsigned* out;
    CK(hipMalloc(&out, 256));
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz / 1e6;
    run<0>("v_add_u32", out, ghz);
    run<1>("v_xor_b32", out, ghz);
    run<2>("v_bcnt_u32_b32", out, ghz);
    run<3>("v_perm_b32", out, ghz);
    run<4>("v_dot4_u32_u8", out, ghz);
    run<5>("v_mad_u32_u24", out, ghz);
    run<6>("v_mul_lo_u32", out, ghz);
    run<7>("v_fma_f32", out, ghz);
    run<8>("v_alignbyte_b32", out, ghz);
    run<9>("v_max3_i32", out, ghz);
    run<10>("v_min_u32", out, ghz);
    run<11>("v_lshl_or_b32", out, ghz);
    run<12>("v_cndmask_b32", out, ghz);
    run<13>("v_and_or_b32", out, ghz);
    run<14>("v_sad_u8", out, ghz);
    run<15>("v_pk_add_u16", out, ghz);
    run<16>("v_pk_max_i16", out, ghz);
    run<17>("v_med3_i32", out, ghz);
    run<18>("v_mul_u32_u24", out, ghz);
    run<19>("v_sub_u32", out, ghz);
    run<20>("v_cmp_gt_i32", out, ghz);
    run<21>("v_pk_fma_f32", out, ghz);
    return 0;
}
