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
    const unsigned long long mask = 0x5555aaaa3333ccccull ^ seed;
    unsigned sacc = 0;
    if (OP == 12) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
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
            if (OP == 22) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "s"(mask));
            if (OP == 23) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 24) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[i]));
            if (OP == 25) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(v[i]));
            if (OP == 26) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 27) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 28) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(unsigned long long*)&v[i & ~1]) : "v"(*(unsigned long long*)&v[(i & ~1)]));
            if (OP == 29) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
            if (OP == 30) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(v[i]) : "v"(c & 0xfcu));
            if (OP == 31) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : "vcc");
            if (OP == 32) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sacc) : "v"(v[i]));
            if (OP == 33) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 34) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(v[i]));
            if (OP == 35) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 36) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(v[i]));
            if (OP == 37) asm volatile("v_or_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 38) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 39) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 40) asm volatile("v_min3_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 41) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 42) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 43) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 44) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 45) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 46) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 47) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 48) asm volatile("v_max_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 49) asm volatile("v_max_u16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 50) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 51) asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0xc8" : "+v"(v[i]) : "v"(c));
            if (OP == 52) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 53) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 54) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 55) asm volatile("v_add_u16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 56) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[i]));
            if (OP == 57) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 58) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
            if (OP == 59) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(v[i]) : "v"(c));
            if (OP == 60) asm volatile("v_min_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "+v"(v[i]) : "v"(c));
            if (OP == 61) asm volatile("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(v[i]) : "v"(c));
            if (OP == 62) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(v[i]));
            if (OP == 63) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 64) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 65) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(unsigned long long*)&v[i & ~1]) : "v"(*(unsigned long long*)&v[(i & ~1)]));
            if (OP == 66) asm volatile("v_med3_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
            if (OP == 67) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(v[i]) : "v"(c) : "vcc");
            if (OP == 68) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 69) asm volatile("v_xnor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (OP == 70) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[i]) : "s"(c));
            if (OP == 71) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(v[i]));
            if (OP == 72) asm volatile("v_and_b32 %0, 15, %0" : "+v"(v[i]));
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= v[i];
    if (acc == 0x12345u) out[0] = acc + sacc;
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
    printf("%-26s %.3f ms  %.2f cycles per wave64 instruction per SIMD at %.2f GHz\n", name, ms, clockGHz * 1e9 / perSimdPerSec, clockGHz);
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
    run<22>("v_cndmask_b32 (sgpr mask)", out, ghz);
    run<23>("v_bfi_b32", out, ghz);
    run<24>("v_rndne_f32", out, ghz);
    run<25>("v_cvt_i32_f32", out, ghz);
    run<26>("v_mul_f32", out, ghz);
    run<27>("v_add_f32", out, ghz);
    run<28>("v_pk_mul_f32", out, ghz);
    run<29>("v_mov_b32 dpp row_shr", out, ghz);
    run<30>("ds_bpermute_b32 + wait", out, ghz);
    run<31>("v_cmp + v_cndmask (pair)", out, ghz);
    run<32>("v_readlane_b32", out, ghz);
    run<33>("v_add3_u32", out, ghz);
    run<34>("v_lshlrev_b32", out, ghz);
    run<35>("v_and_b32", out, ghz);
    run<36>("v_cvt_f32_i32", out, ghz);
    run<37>("v_or_b32", out, ghz);
    run<38>("v_min_f32", out, ghz);
    run<39>("v_max_f32", out, ghz);
    run<40>("v_min3_f32", out, ghz);
    run<41>("v_pk_min_f16", out, ghz);
    run<42>("v_pk_max_f16", out, ghz);
    run<43>("v_pk_add_f16", out, ghz);
    run<44>("v_pk_mul_f16", out, ghz);
    run<45>("v_pk_fma_f16", out, ghz);
    run<46>("v_pk_minimum3_f16", out, ghz);
    run<47>("v_pk_maximum3_f16", out, ghz);
    run<48>("v_max_f16", out, ghz);
    run<49>("v_max_u16", out, ghz);
    run<50>("v_max_i32", out, ghz);
    run<51>("v_bitop3_b32", out, ghz);
    run<52>("v_mov_b32", out, ghz);
    run<53>("v_pk_sub_i16", out, ghz);
    run<54>("v_pk_min_u16", out, ghz);
    run<55>("v_add_u16", out, ghz);
    run<56>("v_lshrrev_b32", out, ghz);
    run<57>("v_mbcnt_lo_u32_b32", out, ghz);
    run<58>("v_add_u32 dpp row_shr", out, ghz);
    run<59>("v_and_b32 sdwa byte1", out, ghz);
    run<60>("v_min_u16 sdwa bytes", out, ghz);
    run<61>("v_add_f32 sdwa", out, ghz);
    run<62>("v_cvt_f32_ubyte1", out, ghz);
    run<63>("v_sub_f32", out, ghz);
    run<64>("v_fmac_f32", out, ghz);
    run<65>("v_pk_add_f32", out, ghz);
    run<66>("v_med3_f32", out, ghz);
    run<67>("v_add_co_u32", out, ghz);
    run<68>("v_subrev_u32", out, ghz);
    run<69>("v_xnor_b32", out, ghz);
    run<70>("v_add_u32 (sgpr src)", out, ghz);
    run<71>("v_add_u32 (literal)", out, ghz);
    run<72>("v_and_b32 (inline const)", out, ghz);
    return 0;
}
