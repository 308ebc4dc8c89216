// Probe (not product): N streams of one process, each a chain of short dependent kernels -- do the chains run side by side?
// A P-step chain of the codec is kernels of about 10 us on 1500 workgroups; two chains per context run side by side, a fourth
// busy stream has cost 0.3-0.9x in the codec.  This isolates the runtime's part.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_streams.bin tools/probe_streams.hip
//   probe_streams.bin [wgs per kernel] [alu iterations] [kernels per chain]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
__global__ void k_alu(int* p, int iters)
{
    int v = threadIdx.x + p[0];
    for (int i = 0; i < iters; i++) { v = v * 1664525 + 1013904223; asm volatile("" : "+v"(v)); }
    if (v == 123456789) p[1] = v;
}
int main(int argc, char** argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 1500, K = argc > 3 ? atoi(argv[3]) : 200;
    int* d; (void)hipMalloc(&d, 1 << 20); (void)hipMemset(d, 0, 1 << 20);
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("wgs %d x 256 threads, %d iterations, %d kernels per chain; priorities lo %d hi %d; GPU_MAX_HW_QUEUES=%s\n", wgs, iters, K, lo, hi,
           getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");
    for (int mode = 0; mode < 3; mode++) {           // 0: all default priority; 1: all high; 2: stream 1 low, the rest high (the codec's mix)
        for (int N = 1; N <= 8; N++) {
            std::vector<hipStream_t> st(N);
            for (int s = 0; s < N; s++) {
                const int pr = mode == 0 ? 0 : (mode == 1 ? hi : (s == 1 ? lo : hi));
                (void)hipStreamCreateWithPriority(&st[s], hipStreamNonBlocking, pr);
            }
            auto pass = [&](int k) { for (int i = 0; i < k; i++) for (int s = 0; s < N; s++) hipLaunchKernelGGL(k_alu, dim3(wgs), dim3(256), 0, st[s], d, iters); };
            pass(20);
            for (int s = 0; s < N; s++) (void)hipStreamSynchronize(st[s]);
            auto t0 = std::chrono::steady_clock::now();
            pass(K);
            auto t1 = std::chrono::steady_clock::now();
            for (int s = 0; s < N; s++) (void)hipStreamSynchronize(st[s]);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            const double en = std::chrono::duration<double, std::micro>(t1 - t0).count();
            printf("mode %d  %d streams: %8.2f us per chain step (all chains together), %6.2f us per kernel of work, host enqueue %5.2f us per launch\n",
                   mode, N, us / K, us / K / N, en / K / N);
            for (int s = 0; s < N; s++) (void)hipStreamDestroy(st[s]);
        }
    }
    return 0;
}
