// Probe (not product): what does a DEPENDENT fp64 instruction cost a lone wave?  The transform passes of the codec are chains
// acc = acc + x[k] * c[k]; hipcc schedules them depth-first (mul, add, mul, add, ... of ONE chain: every instruction waits for
// the one before).  One wave per SIMD; NCH independent chains interleaved; cycles per instruction by s_memtime against a
// v_add_u32 reference for the clock.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_dep.bin tools/probe_dep.hip ; tools/probe_dep.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NCH, int KIND> __global__ void k(long long* out, int iters, double c)
{
    double acc[8], x[8], tmp[8];
    for (int j = 0; j < 8; j++) { acc[j] = threadIdx.x * 0.5 + j; x[j] = threadIdx.x * 0.25 + j; tmp[j] = 0; }
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 16; rep++) {
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                if (KIND == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[j]) : "v"(c));
                if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[j]) : "v"(c));
                if (KIND == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x[j]), "v"(c));
                if (KIND == 3) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(tmp[j]) : "v"(x[j]), "v"(c));       // the kernel's pattern: mul then
            }
            if (KIND == 3) {
#pragma unroll
                for (int j = 0; j < NCH; j++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[j]) : "v"(tmp[j]));   // dependent add
            }
        }
    }
    const long long t1 = clock64();
    double s = 0; for (int j = 0; j < 8; j++) s += acc[j] + tmp[j];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (long long)s; }
}
__global__ void kref(long long* out, int iters, int ci)
{
    int r[16]; for (int j = 0; j < 16; j++) r[j] = threadIdx.x + j;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[j]) : "v"(ci));
    const long long t1 = clock64();
    int s = 0; for (int j = 0; j < 16; j++) s += r[j];
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s; }
}
template <int NCH, int KIND> double run(long long* d, double ref)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<NCH, KIND>), dim3(1), dim3(64), 0, 0, d, iters, 1.0000001);
    long long h[2]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * NCH * (KIND == 3 ? 2 : 1);
    return h[0] / n / ref * 4.0;
}
int main()
{
    long long* d; (void)hipMalloc(&d, 4096);
    hipLaunchKernelGGL(kref, dim3(1), dim3(64), 0, 0, d, 2000, 3);
    long long h[2]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double ref = h[0] / (2000.0 * 64);       // timer ticks per independent v_add_u32 (= 4 shader cycles)
    printf("reference: %.4f timer ticks per independent v_add_u32 (taken as 4 cycles)\n", ref);
    const char* kn[4] = {"v_add_f64 acc,acc,c", "v_mul_f64 acc,acc,c", "v_fma_f64 acc,x,c,acc", "v_mul_f64 t,x,c ; v_add_f64 acc,acc,t"};
#define ROW(K) printf("%-40s chains 1: %5.2f  2: %5.2f  4: %5.2f  8: %5.2f  cycles per instruction\n", kn[K], run<1, K>(d, ref), run<2, K>(d, ref), run<4, K>(d, ref), run<8, K>(d, ref));
    ROW(0) ROW(1) ROW(2) ROW(3)
    return 0;
}
