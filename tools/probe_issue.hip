// Probe (not product): issue cost in cycles of the vector instructions the fp64 codec kernels are made of, one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_issue.bin tools/probe_issue.hip ; tools/probe_issue.bin
// Each test runs 16 independent chains of one instruction, 64 instructions per loop trip, and reports cycles per instruction
// (s_memtime, 100 MHz constant clock on this part -> scaled by the measured shader clock through a v_add_u32 reference).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define BODY(ASM, C) \
    for (int it = 0; it < iters; it++) { \
        REP4(asm volatile(ASM : "+v"(r[0]) : C); asm volatile(ASM : "+v"(r[1]) : C); asm volatile(ASM : "+v"(r[2]) : C); asm volatile(ASM : "+v"(r[3]) : C); \
             asm volatile(ASM : "+v"(r[4]) : C); asm volatile(ASM : "+v"(r[5]) : C); asm volatile(ASM : "+v"(r[6]) : C); asm volatile(ASM : "+v"(r[7]) : C); \
             asm volatile(ASM : "+v"(r[8]) : C); asm volatile(ASM : "+v"(r[9]) : C); asm volatile(ASM : "+v"(r[10]) : C); asm volatile(ASM : "+v"(r[11]) : C); \
             asm volatile(ASM : "+v"(r[12]) : C); asm volatile(ASM : "+v"(r[13]) : C); asm volatile(ASM : "+v"(r[14]) : C); asm volatile(ASM : "+v"(r[15]) : C);) }

template <int T> __global__ void k(long long* out, int iters, double c, int ci)
{
    double r[16]; int ri[16]; 
    for (int j = 0; j < 16; j++) { r[j] = threadIdx.x * 0.5 + j; ri[j] = threadIdx.x + j; }
    __syncthreads();
    const long long t0 = clock64();
    if (T == 0) { int* r = ri; BODY("v_add_u32 %0, %0, %1", "v"(ci)) }
    if (T == 1) BODY("v_add_f64 %0, %0, %1", "v"(c))
    if (T == 2) BODY("v_mul_f64 %0, %0, %1", "v"(c))
    if (T == 3) BODY("v_fma_f64 %0, %0, %1, %1", "v"(c))
    if (T == 4) { for (int it = 0; it < iters; it++) { REP4(for (int j = 0; j < 16; j++) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r[j]) : "v"(ri[j]));) } }
    if (T == 5) { for (int it = 0; it < iters; it++) { REP4(for (int j = 0; j < 16; j++) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ri[j]) : "v"(r[j]));) } }
    if (T == 6) { int* r = ri; BODY("v_cndmask_b32 %0, %0, %1, vcc", "v"(ci)) }
    if (T == 7) { int* r = ri; BODY("v_mul_lo_u32 %0, %0, %1", "v"(ci)) }
    if (T == 8) { int* r = ri; BODY("v_mul_hi_i32 %0, %0, %1", "v"(ci)) }
    if (T == 9) { int* r = ri; BODY("v_sad_u8 %0, %0, %1, %0", "v"(ci)) }
    if (T == 10) { int* r = ri; BODY("v_med3_i32 %0, %0, %1, %1", "v"(ci)) }
    if (T == 11) { int* r = ri; BODY("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v"(ci)) }
    if (T == 12) { int* r = ri; BODY("v_mul_i32_i24 %0, %0, %1", "v"(ci)) }
    if (T == 13) { int* r = ri; BODY("v_bfe_u32 %0, %0, 8, 8", "v"(ci)) }
    if (T == 14) { int* r = ri; BODY("v_perm_b32 %0, %0, %1, %1", "v"(ci)) }
    if (T == 15) { int* r = ri; BODY("v_lshl_add_u32 %0, %0, 3, %1", "v"(ci)) }
    if (T == 16) { for (int it = 0; it < iters; it++) { REP4(for (int j = 0; j < 16; j++) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(ri[j]) : "v"(ri[(j + 1) & 15]));) } }
    if (T == 17) { for (int it = 0; it < iters; it++) { REP4(for (int j = 0; j < 16; j++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[j]) : "v"(ri[j]));) } }
    if (T == 18) { for (int it = 0; it < iters; it++) { REP4(for (int j = 0; j < 16; j++) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(r[j]) : "v"(ri[j]));) } }
    if (T == 19) BODY("v_max_f64 %0, %0, %1", "v"(c))
    if (T == 20) BODY("v_ldexp_f64 %0, %0, %1", "v"(ci))
    if (T == 21) { int* r = ri; BODY("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2", "v"(ci)) }
    if (T == 22) BODY("v_mov_b64 %0, %1", "v"(c))
    if (T == 23) BODY("v_rndne_f64 %0, %0", "v"(c))
    if (T == 24) BODY("v_trunc_f64 %0, %0", "v"(c))
    if (T == 25) { int* r = ri; BODY("v_cndmask_b32_e64 %0, %0, %1, s[10:11]", "v"(ci)) }
    if (T == 26) { int* r = ri; BODY("v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", "v"(ci)) }
    if (T == 27) { int* r = ri; BODY("v_cmp_gt_i32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]", "v"(ci)) }
    if (T == 28) { int* r = ri; BODY("v_cmp_gt_i32 vcc, %0, %1", "v"(ci)) }
    if (T == 29) { int* r = ri; BODY("s_nop 0\n v_cndmask_b32 %0, %0, %1, vcc", "v"(ci)) }
    if (T == 30) { int* r = ri; BODY("v_xor_b32 %0, %0, %1\n v_and_b32 %0, %0, %1", "v"(ci)) }
    const long long t1 = clock64();
    double s = 0; int si = 0;
    for (int j = 0; j < 16; j++) { s += r[j]; si += ri[j]; }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.678 && si == 77) out[1] = 1;
}
template <int T> double run(long long* d, int iters)
{
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(256), 0, 0, d, iters, 1.0000001, 3);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(256), 0, 0, d, iters, 1.0000001, 3);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 / ((double)iters * 64);      // seconds per instruction (one wave per SIMD)
}
int main()
{
    long long* d; hipMalloc(&d, 64);
    const int iters = 20000;
    const char* names[] = { "v_add_u32", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64_i32", "v_cvt_i32_f64", "v_cndmask_b32", "v_mul_lo_u32", "v_mul_hi_i32",
                            "v_sad_u8", "v_med3_i32", "v_mov_b32_dpp", "v_mul_i32_i24", "v_bfe_u32", "v_perm_b32", "v_lshl_add_u32", "v_cvt_f32_ubyte1", "v_cvt_f64_f32",
                            "v_cvt_f64_u32", "v_max_f64", "v_ldexp_f64", "v_sub_u32_sdwa", "v_mov_b64", "v_rndne_f64", "v_trunc_f64",
                            "v_cndmask_e64 sgpr", "v_cmp+v_cndmask vcc (2)", "v_cmp_e64+v_cndmask_e64 (2)", "v_cmp vcc", "s_nop+v_cndmask (2)", "v_xor+v_and (2)" };
    double t[31];
    t[0] = run<0>(d, iters); t[1] = run<1>(d, iters); t[2] = run<2>(d, iters); t[3] = run<3>(d, iters); t[4] = run<4>(d, iters); t[5] = run<5>(d, iters);
    t[6] = run<6>(d, iters); t[7] = run<7>(d, iters); t[8] = run<8>(d, iters); t[9] = run<9>(d, iters); t[10] = run<10>(d, iters); t[11] = run<11>(d, iters);
    t[12] = run<12>(d, iters); t[13] = run<13>(d, iters); t[14] = run<14>(d, iters); t[15] = run<15>(d, iters); t[16] = run<16>(d, iters); t[17] = run<17>(d, iters);
    t[18] = run<18>(d, iters); t[19] = run<19>(d, iters); t[20] = run<20>(d, iters); t[21] = run<21>(d, iters); t[22] = run<22>(d, iters); t[23] = run<23>(d, iters); t[24] = run<24>(d, iters);
    t[25] = run<25>(d, iters); t[26] = run<26>(d, iters); t[27] = run<27>(d, iters); t[28] = run<28>(d, iters); t[29] = run<29>(d, iters); t[30] = run<30>(d, iters);
    printf("reference: v_add_u32 %.3f ns per instruction = 4 cycles -> %.2f GHz\n", t[0] * 1e9, 4.0 / (t[0] * 1e9));
    for (int i = 0; i < 31; i++) printf("%-18s %6.2f cycles\n", names[i], t[i] / t[0] * 4.0);
    return 0;
}
