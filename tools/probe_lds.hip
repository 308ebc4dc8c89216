// Probe (not product): LDS cost of blk8_chain's five transposes as a function of the tile stride (dwords per block tile) and of a
// per-row pad, one wave per SIMD:  hipcc --offload-arch=gfx950 -O2 -o tools/probe_lds.bin tools/probe_lds.hip ; tools/probe_lds.bin
// Lane (jb = lane >> 3, i = lane & 7) works on tile jb; patterns as in icsp_blk8.hip.inc:
//   X  xpose_d      8 x ds_write_b64 t[k*RS + i]        then 8 doubles read t[i*RS + k]     (RS = 8 + pad doubles per row)
//   Z  zig-zag      8 x ds_write_b16 t16[zz(v, i)]      then one 16-byte read ((uint4*)tile)[i]
//   Q  dequantised  8 x ds_write_b32 t32[v*8 + i]       then 8 ints read t32[i*8 + u]
//   P  pixels       8 x ds_write_b8  t8[y*8 + i]        then one 8-byte read ((uint2*)tile)[i]
// Reported: shader cycles per pattern (s_memtime ticks scaled by a v_add reference is not needed: relative numbers matter).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__constant__ uint8_t c_zz[64];

template <int PAD> __global__ __launch_bounds__(64) void k_lds(long long* out, int stride_dw, int iters)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int l = threadIdx.x & 63, i = l & 7, jb = l >> 3;
    uint32_t* tile = lds + jb * stride_dw;
    constexpr int RS = 8 + PAD;                                // doubles per row of the double transposes (compile time, as in the product)
    double a[8], b[8];
    int q[8], r[8];
    for (int k = 0; k < 8; k++) { a[k] = l * 0.25 + k; q[k] = l + k; }
    unsigned long long zz8 = 0;
    for (int v = 0; v < 8; v++) zz8 |= (unsigned long long)c_zz[v * 8 + i] << (8 * v);
    long long acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        long long t0 = clock64();
        {   double* t = (double*)tile;
            for (int k = 0; k < 8; k++) t[k * RS + i] = a[k];
            __builtin_amdgcn_wave_barrier();
            for (int k = 0; k < 8; k++) b[k] = t[i * RS + k];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int k = 0; k < 8; k++) a[k] = b[k] + 1.0; }
        long long t1 = clock64();
        {   int16_t* t16 = (int16_t*)tile;
            for (int v = 0; v < 8; v++) t16[(zz8 >> (8 * v)) & 0xff] = (int16_t)q[v];
            __builtin_amdgcn_wave_barrier();
            const uint4 x = ((const uint4*)tile)[i];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            q[0] += (int)x.x; q[1] += (int)x.y; q[2] += (int)x.z; q[3] += (int)x.w; }
        long long t2 = clock64();
        {   int* t32 = (int*)tile;
            for (int v = 0; v < 8; v++) t32[v * 8 + i] = q[v];
            __builtin_amdgcn_wave_barrier();
            for (int u = 0; u < 8; u++) r[u] = t32[i * 8 + u];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int u = 0; u < 8; u++) q[u] = r[u] + 1; }
        long long t3 = clock64();
        {   uint8_t* t8 = (uint8_t*)tile;
            for (int y = 0; y < 8; y++) t8[y * 8 + i] = (uint8_t)q[y];
            __builtin_amdgcn_wave_barrier();
            const uint2 x = ((const uint2*)tile)[i];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            q[4] += (int)x.x; q[5] += (int)x.y; }
        long long t4 = clock64();
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3;
    }
    double s = 0; for (int k = 0; k < 8; k++) s += a[k] + q[k];
    if (l == 0) { for (int k = 0; k < 4; k++) out[blockIdx.x * 5 + k] = acc[k]; out[blockIdx.x * 5 + 4] = (long long)s; }
}

int main()
{
    uint8_t zz[64]; int zk = 0;
    for (int s = 0; s < 15; s++) {
        if (s & 1) { for (int r = (s < 8 ? 0 : s - 7); r <= (s < 8 ? s : 7); r++) zz[r * 8 + (s - r)] = (uint8_t)zk++; }
        else       { for (int r = (s < 8 ? s : 7); r >= (s < 8 ? 0 : s - 7); r--) zz[r * 8 + (s - r)] = (uint8_t)zk++; }
    }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_zz), zz, 64);
    long long* d; (void)hipMalloc(&d, 4096 * 5 * 8);
    long long h[5 * 1024];
    const int iters = 2000, nb = 1024;                         // 1024 single-wave workgroups: one per SIMD
    printf("stride_dw pad   X(double transpose) Z(zig-zag) Q(int transpose) P(pixels)   sum   [clock64 ticks per iteration, median over workgroups]\n");
    for (int pad = 0; pad <= 1; pad++)
        for (int st = 128 + pad * 16; st <= 176; st += 2) {
            if (st < 8 * (8 + pad) * 2) continue;
            if (pad) hipLaunchKernelGGL(k_lds<1>, dim3(nb), dim3(64), 8 * st * 4 + 64, 0, d, st, iters);
            else     hipLaunchKernelGGL(k_lds<0>, dim3(nb), dim3(64), 8 * st * 4 + 64, 0, d, st, iters);
            (void)hipMemcpy(h, d, nb * 5 * 8, hipMemcpyDeviceToHost);
            double v[4];
            for (int c = 0; c < 4; c++) { long long t[1024]; for (int b = 0; b < nb; b++) t[b] = h[b * 5 + c];
                qsort(t, nb, 8, [](const void* x, const void* y) { long long a = *(const long long*)x, b = *(const long long*)y; return a < b ? -1 : a > b; });
                v[c] = (double)t[nb / 2] / iters; }
            printf("%8d %3d   %10.1f %14.1f %12.1f %14.1f %9.1f\n", st, pad, v[0], v[1], v[2], v[3], v[0] + v[1] + v[2] + v[3]);
        }
    return 0;
}
