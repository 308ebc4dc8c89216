// Probe (not product): latency of the first global loads of a kernel, stamped in-kernel with s_memtime (wave 0 of block 0 and of the last block).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_rd(const uint4* a, uint4* b, unsigned long long* stamps, int rep)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint4 v = a[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint4 w = a[(i + 64 * 1024 * 1024 / 16) ^ (v.x & 1)];          // second, dependent load from another region
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    uint4 x = a[i ^ (w.x & 1)];                                      // third: the first region again (L1/L2-warm now)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (x.x == 0x12345678u) b[i] = x;
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
        unsigned long long* s = stamps + (rep * 2 + (blockIdx.x != 0)) * 4;
        s[0] = t1 - t0; s[1] = t2 - t1; s[2] = t3 - t2; s[3] = t0;
    }
}
__global__ void k_wr(uint4* b) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; b[i] = make_uint4(i, 1, 2, 3); }
int main()
{
    uint4 *a, *b; (void)hipMalloc(&a, 256 << 20); (void)hipMalloc(&b, 256 << 20); (void)hipMemset(a, 1, 256 << 20);
    unsigned long long* st; (void)hipMalloc(&st, 1 << 20); (void)hipMemset(st, 0, 1 << 20);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int wgs : {1, 64, 1100, 16384}) {
        for (int mode = 0; mode < 2; mode++) {
            const int reps = 40;
            for (int r = 0; r < reps; r++) {
                if (mode == 1) hipLaunchKernelGGL(k_wr, dim3(wgs), dim3(256), 0, s, (uint4*)a);       // previous kernel rewrites what is read next
                hipLaunchKernelGGL(k_rd, dim3(wgs), dim3(256), 0, s, a, b, st, r);
            }
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h(reps * 8);
            (void)hipMemcpy(h.data(), st, reps * 64, hipMemcpyDeviceToHost);
            double m[6] = {0, 0, 0, 0, 0, 0};
            for (int r = 10; r < reps; r++) for (int k = 0; k < 2; k++) for (int j = 0; j < 3; j++) m[k * 3 + j] += h[(r * 2 + k) * 4 + j] / (double)(reps - 10);
            printf("wgs %5d %s: block0 load1 %6.0f load2 %6.0f load3 %6.0f cyc | last block %6.0f %6.0f %6.0f  (100 MHz ticks? see below)\n", wgs,
                   mode ? "after a kernel that rewrote the data" : "data untouched                      ", m[0], m[1], m[2], m[3], m[4], m[5]);
        }
    }
    return 0;
}
