// Probe (not product): what a vector instruction costs the SIMD's issue port with 1, 2, 3 and 4 waves on the SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_issue_waves.bin tools/probe_issue_waves.hip ; tools/probe_issue_waves.bin
// One workgroup per CU (84 KB of dynamic LDS each: two cannot share a CU), 256 * w threads = w waves on each of the four SIMDs,
// every wave 16 independent chains of ONE instruction, 64 instructions per loop trip.  Timed by WALL CLOCK (HIP events around the
// launch; 256 workgroups, one per CU) and, inside wave 0 of workgroup 0, by s_memtime and by s_memrealtime (100 MHz); nothing is
// normalised to an assumed cost.  Reported per instruction class and waves per SIMD:
//   ns per wave-instruction on ONE SIMD  = wall time / (trips * 64 * w)
//   cycles at the clock s_memtime counted = the same in s_memtime ticks (the tick is the shader cycle, MI355X_MICROARCH.md constants)
// If 32-bit instructions cost the port 2 cycles, the w = 2 column is half the w = 1 column; if 4, they are equal.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define REP4(x) x x x x
#define BODY(ASM, C) \
    for (int it = 0; it < iters; it++) { \
        REP4(asm volatile(ASM : "+v"(r[0]) : C); asm volatile(ASM : "+v"(r[1]) : C); asm volatile(ASM : "+v"(r[2]) : C); asm volatile(ASM : "+v"(r[3]) : C); \
             asm volatile(ASM : "+v"(r[4]) : C); asm volatile(ASM : "+v"(r[5]) : C); asm volatile(ASM : "+v"(r[6]) : C); asm volatile(ASM : "+v"(r[7]) : C); \
             asm volatile(ASM : "+v"(r[8]) : C); asm volatile(ASM : "+v"(r[9]) : C); asm volatile(ASM : "+v"(r[10]) : C); asm volatile(ASM : "+v"(r[11]) : C); \
             asm volatile(ASM : "+v"(r[12]) : C); asm volatile(ASM : "+v"(r[13]) : C); asm volatile(ASM : "+v"(r[14]) : C); asm volatile(ASM : "+v"(r[15]) : C);) }
// two instructions alternating, eight chains each: 64 instructions per trip, half of each kind
#define MIX(ASMA, ASMB, C) \
    for (int it = 0; it < iters; it++) { \
        REP4(asm volatile(ASMA : "+v"(r[0]) : C); asm volatile(ASMB : "+v"(ri[0]) : "v"(ci)); asm volatile(ASMA : "+v"(r[1]) : C); asm volatile(ASMB : "+v"(ri[1]) : "v"(ci)); \
             asm volatile(ASMA : "+v"(r[2]) : C); asm volatile(ASMB : "+v"(ri[2]) : "v"(ci)); asm volatile(ASMA : "+v"(r[3]) : C); asm volatile(ASMB : "+v"(ri[3]) : "v"(ci)); \
             asm volatile(ASMA : "+v"(r[4]) : C); asm volatile(ASMB : "+v"(ri[4]) : "v"(ci)); asm volatile(ASMA : "+v"(r[5]) : C); asm volatile(ASMB : "+v"(ri[5]) : "v"(ci)); \
             asm volatile(ASMA : "+v"(r[6]) : C); asm volatile(ASMB : "+v"(ri[6]) : "v"(ci)); asm volatile(ASMA : "+v"(r[7]) : C); asm volatile(ASMB : "+v"(ri[7]) : "v"(ci));) }

constexpr int NT = 21;
#define OPD(ASM, j) asm volatile(ASM : "+v"(r[j]) : "v"(c));
#define OPI(ASM, j) asm volatile(ASM : "+v"(ri[j]) : "v"(ci));
#define AF "v_add_f64 %0, %0, %1"
#define AI "v_add_u32 %0, %0, %1"
template <int T> __global__ void k(long long* out, int iters, double c, int ci, float cf)
{
    extern __shared__ unsigned char lds[];
    double r[16]; int ri[16]; float rf[16];
    for (int j = 0; j < 16; j++) { r[j] = threadIdx.x * 0.5 + j; ri[j] = threadIdx.x + j; rf[j] = threadIdx.x * 0.25f + j; }
    if (ci == 12345) lds[threadIdx.x] = 1;            // (keeps the LDS reservation alive)
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    if (T == 0) { int* r = ri; BODY("v_add_u32 %0, %0, %1", "v"(ci)) }
    if (T == 1) { int* r = ri; BODY("v_and_b32 %0, %0, %1", "v"(ci)) }
    if (T == 2) { int* r = ri; BODY("v_cndmask_b32 %0, %0, %1, vcc", "v"(ci)) }
    if (T == 3) { int* r = ri; BODY("v_mul_i32_i24 %0, %0, %1", "v"(ci)) }
    if (T == 4) BODY("v_add_f64 %0, %0, %1", "v"(c))
    if (T == 5) BODY("v_mul_f64 %0, %0, %1", "v"(c))
    if (T == 6) BODY("v_fma_f64 %0, %0, %1, %1", "v"(c))
    if (T == 7) { for (int it = 0; it < iters; it++) { REP4(for (int j = 0; j < 16; j++) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r[j]) : "v"(ri[j]));) } }
    if (T == 8) { for (int it = 0; it < iters; it++) { REP4(for (int j = 0; j < 16; j++) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ri[j]) : "v"(r[j]));) } }
    if (T == 9) MIX("v_add_f64 %0, %0, %1", "v_add_u32 %0, %0, %1", "v"(c))
    if (T == 10) MIX("v_fma_f64 %0, %0, %1, %1", "v_and_b32 %0, %0, %1", "v"(c))
    if (T == 11) { float* r = rf; BODY("v_fma_f32 %0, %0, %1, %1", "v"(cf)) }
    if (T == 12) { int* r = ri; BODY("v_sad_u8 %0, %0, %1, %0", "v"(ci)) }
    if (T == 13) { int* r = ri; BODY("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v"(ci)) }
    if (T == 14) { int* r = ri; BODY("v_med3_i32 %0, %0, %1, %1", "v"(ci)) }
    if (T == 15) { int* r = ri; BODY("v_mul_hi_u32 %0, %0, %1", "v"(ci)) }
    if (T == 16) { int* r = ri; BODY("v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", "v"(ci)) }      // two instructions per slot
    // simple integer instructions of four kinds in turn
    if (T == 17) { for (int it = 0; it < iters; it++) { REP4(OPI(AI, 0) OPI("v_and_b32 %0, %0, %1", 1) OPI("v_lshlrev_b32 %0, 1, %0", 2) OPI("v_xor_b32 %0, %0, %1", 3)
                   OPI(AI, 4) OPI("v_and_b32 %0, %0, %1", 5) OPI("v_lshlrev_b32 %0, 1, %0", 6) OPI("v_xor_b32 %0, %0, %1", 7) OPI(AI, 8) OPI("v_and_b32 %0, %0, %1", 9)
                   OPI("v_lshlrev_b32 %0, 1, %0", 10) OPI("v_xor_b32 %0, %0, %1", 11) OPI(AI, 12) OPI("v_and_b32 %0, %0, %1", 13) OPI("v_lshlrev_b32 %0, 1, %0", 14) OPI("v_xor_b32 %0, %0, %1", 15)) } }
    // v_add_u32 and v_sad_u8 in turn
    if (T == 18) { for (int it = 0; it < iters; it++) { REP4(OPI(AI, 0) OPI("v_sad_u8 %0, %0, %1, %0", 1) OPI(AI, 2) OPI("v_sad_u8 %0, %0, %1, %0", 3) OPI(AI, 4) OPI("v_sad_u8 %0, %0, %1, %0", 5)
                   OPI(AI, 6) OPI("v_sad_u8 %0, %0, %1, %0", 7) OPI(AI, 8) OPI("v_sad_u8 %0, %0, %1, %0", 9) OPI(AI, 10) OPI("v_sad_u8 %0, %0, %1, %0", 11) OPI(AI, 12) OPI("v_sad_u8 %0, %0, %1, %0", 13)
                   OPI(AI, 14) OPI("v_sad_u8 %0, %0, %1, %0", 15)) } }
    // one fp64 add in four (three v_add_u32 between)
    if (T == 19) { for (int it = 0; it < iters; it++) { REP4(OPD(AF, 0) OPI(AI, 1) OPI(AI, 2) OPI(AI, 3) OPD(AF, 4) OPI(AI, 5) OPI(AI, 6) OPI(AI, 7) OPD(AF, 8) OPI(AI, 9) OPI(AI, 10) OPI(AI, 11)
                   OPD(AF, 12) OPI(AI, 13) OPI(AI, 14) OPI(AI, 15)) } }
    // three fp64 adds in four
    if (T == 20) { for (int it = 0; it < iters; it++) { REP4(OPD(AF, 0) OPD(AF, 1) OPD(AF, 2) OPI(AI, 3) OPD(AF, 4) OPD(AF, 5) OPD(AF, 6) OPI(AI, 7) OPD(AF, 8) OPD(AF, 9) OPD(AF, 10) OPI(AI, 11)
                   OPD(AF, 12) OPD(AF, 13) OPD(AF, 14) OPI(AI, 15)) } }
    const long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    double s = 0; int si = 0; float sf = 0;
    for (int j = 0; j < 16; j++) { s += r[j]; si += ri[j]; sf += rf[j]; }
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (s == 12345.678 && si == 77 && sf == 3.f) out[2] = 1;
}
struct R { double wall_s, ticks, real_s; };
template <int T> R run(long long* d, long long* h, int iters, int w)
{
    const size_t lds = 84 * 1024;
    hipFuncSetAttribute((const void*)k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<T>, dim3(256), dim3(256 * w), lds, 0, d, iters, 1.0000001, 3, 1.0001f);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    double best = 1e9; R r{};
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k<T>, dim3(256), dim3(256 * w), lds, 0, d, iters, 1.0000001, 3, 1.0001f);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        if (ms * 1e-3 < best) { best = ms * 1e-3; r = R{ ms * 1e-3, (double)h[0], (double)h[1] * 1e-8 }; }
    }
    hipEventDestroy(a); hipEventDestroy(b);
    if (hipGetLastError() != hipSuccess) { printf("launch failed (T=%d w=%d)\n", T, w); exit(1); }
    return r;
}
template <int T> void all(long long* d, long long* h, const char* name, int per_slot)
{
    const int iters = 20000;
    printf("%-34s", name);
    double cyc[4];
    for (int w = 1; w <= 4; w++) {
        const R r = run<T>(d, h, iters, w);
        const double n = (double)iters * 64 * per_slot * w;           // wave-instructions issued on one SIMD
        cyc[w - 1] = r.ticks / n;
        printf("  w=%d: %6.3f ns  %5.2f cyc (%4.2f GHz)", w, r.wall_s * 1e9 / n, r.ticks / n, r.ticks / r.real_s * 1e-9);
    }
    printf("   [1 -> 2 waves: x%.2f]\n", cyc[0] / cyc[1]);
}
int main()
{
    long long *d, *h = (long long*)malloc(64); hipMalloc(&d, 64);
    printf("per WAVE-INSTRUCTION ON ONE SIMD: wall ns (HIP events, 256 workgroups, one per CU) = what the SIMD sustains | s_memtime ticks of WAVE 0 alone\n"
           "(the oldest wave of its SIMD keeps its own rate whatever runs beside it: issue goes by age) (clock = ticks / s_memrealtime)\n");
    all<0>(d, h, "v_add_u32", 1); all<1>(d, h, "v_and_b32", 1); all<2>(d, h, "v_cndmask_b32 (vcc)", 1); all<3>(d, h, "v_mul_i32_i24", 1);
    all<11>(d, h, "v_fma_f32", 1); all<12>(d, h, "v_sad_u8", 1); all<13>(d, h, "v_mov_b32_dpp row_shr", 1); all<14>(d, h, "v_med3_i32", 1);
    all<15>(d, h, "v_mul_hi_u32", 1); all<16>(d, h, "v_cmp + v_cndmask (per instruction)", 2);
    all<4>(d, h, "v_add_f64", 1); all<5>(d, h, "v_mul_f64", 1); all<6>(d, h, "v_fma_f64", 1); all<7>(d, h, "v_cvt_f64_i32", 1); all<8>(d, h, "v_cvt_i32_f64", 1);
    all<9>(d, h, "mix 50/50 v_add_f64 + v_add_u32", 1); all<10>(d, h, "mix 50/50 v_fma_f64 + v_and_b32", 1);
    all<19>(d, h, "mix 25/75 v_add_f64 + v_add_u32", 1); all<20>(d, h, "mix 75/25 v_add_f64 + v_add_u32", 1);
    all<17>(d, h, "simple ints: add, and, lshl, xor", 1); all<18>(d, h, "mix 50/50 v_add_u32 + v_sad_u8", 1);
    return 0;
}
