// Probe (not product): what does a dependent kernel boundary cost on this box, as a function of what the kernels do?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_launch.bin tools/probe_launch.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_empty(int* p) {}
__global__ void k_rd(const uint4* a, uint4* b) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; uint4 v = a[i]; if (v.x == 0x12345678u) b[i] = v; }
__global__ void k_wr(const uint4* a, uint4* b) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; b[i] = make_uint4(i, 1, 2, 3); }
__global__ void k_rw(const uint4* a, uint4* b) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; uint4 v = a[i]; v.x += 1; b[i] = v; }
__global__ void k_alu(int* p, int iters) { int v = threadIdx.x + p[0]; for (int i = 0; i < iters; i++) { v = v * 1664525 + 1013904223; asm volatile("" : "+v"(v)); } if (v == 123456789) p[1] = v; }
int main()
{
    int* d; (void)hipMalloc(&d, 64 << 20); (void)hipMemset(d, 0, 64 << 20);
    uint4 *a, *b; (void)hipMalloc(&a, 256 << 20); (void)hipMalloc(&b, 256 << 20); (void)hipMemset(a, 1, 256 << 20);
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto&& launch) {
        for (int i = 0; i < 50; i++) launch(i);
        (void)hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < 200; i++) launch(i);
        (void)hipEventRecord(e1, st);
        (void)hipStreamSynchronize(st);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s host %7.2f us  device %7.2f us per launch\n", name, us, ms * 1000 / 200);
    };
    run("empty 1485 x 256", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1485), dim3(256), 0, st, d); });
    run("alu chain 2000 iters, 1485 x 256", [&](int) { hipLaunchKernelGGL(k_alu, dim3(1485), dim3(256), 0, st, d, 2000); });
    for (int wgs : {64, 1100, 2700, 16384}) {
        char nm[96];
        snprintf(nm, 96, "read %5.1f MB (same buffer each time)", wgs * 4096 / 1e6);
        run(nm, [&](int) { hipLaunchKernelGGL(k_rd, dim3(wgs), dim3(256), 0, st, a, b); });
        snprintf(nm, 96, "write %5.1f MB (same buffer each time)", wgs * 4096 / 1e6);
        run(nm, [&](int) { hipLaunchKernelGGL(k_wr, dim3(wgs), dim3(256), 0, st, a, b); });
        snprintf(nm, 96, "read+write %5.1f MB each, same buffers", wgs * 4096 / 1e6);
        run(nm, [&](int) { hipLaunchKernelGGL(k_rw, dim3(wgs), dim3(256), 0, st, a, b); });
        snprintf(nm, 96, "ping-pong a->b, b->a %5.1f MB", wgs * 4096 / 1e6);
        run(nm, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_rw, dim3(wgs), dim3(256), 0, st, b, a); else hipLaunchKernelGGL(k_rw, dim3(wgs), dim3(256), 0, st, a, b); });
    }
    return 0;
}
