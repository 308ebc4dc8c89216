// Probe (not product): issue cost of v_qsad_pk_u16_u8 (four byte-offset SADs of a 64-bit window against a dword, packed u16
// accumulators) against the v_alignbyte_b32 + v_sad_u8 pair it could replace in k_me.  One wave per SIMD, 16 independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int T> __global__ __launch_bounds__(64) void k(long long* out, int iters, unsigned w0, unsigned w1, unsigned cur, int sh)
{
    unsigned long long acc[16]; unsigned a32[16];
    for (int j = 0; j < 16; j++) { acc[j] = threadIdx.x + j; a32[j] = threadIdx.x + j; }
    const unsigned long long win = ((unsigned long long)w1 << 32) | (w0 + threadIdx.x);
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (T == 0) acc[j] = __builtin_amdgcn_qsad_pk_u16_u8(win + j, cur, acc[j]);
                if (T == 1) a32[j] = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w1 + j, w0 + threadIdx.x, sh), cur, a32[j]);
                if (T == 2) a32[j] = __builtin_amdgcn_sad_u8(w0 + j + threadIdx.x, cur, a32[j]);
            }
    }
    const long long t1 = clock64();
    unsigned long long s = 0; for (int j = 0; j < 16; j++) s += acc[j] + a32[j];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (long long)s; }
}
int main()
{
    long long* d; (void)hipMalloc(&d, 2048 * 16);
    long long h[2048];
    const int iters = 2000;
    const char* names[3] = {"v_qsad_pk_u16_u8", "v_alignbyte_b32 + v_sad_u8", "v_sad_u8"};
    for (int t = 0; t < 3; t++) {
        if (t == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, d, iters, 0x01020304u, 0x05060708u, 0x0a0b0c0du, 2);
        if (t == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, d, iters, 0x01020304u, 0x05060708u, 0x0a0b0c0du, 2);
        if (t == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 0, 0, d, iters, 0x01020304u, 0x05060708u, 0x0a0b0c0du, 2);
        (void)hipMemcpy(h, d, 1024 * 16, hipMemcpyDeviceToHost);
        long long best = 1LL << 60; for (int b = 0; b < 1024; b++) if (h[b * 2] < best) best = h[b * 2];
        printf("%-30s %.2f ticks per source-level op (64 per trip)\n", names[t], (double)best / iters / 64.0);
    }
    return 0;
}
