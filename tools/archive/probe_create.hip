// Probe (not product): what costs time when a context is set up?  hipcc -O2 -o tools/probe_create.bin tools/probe_create.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int* p) { if (p) p[0] = 1; }
int main()
{
    double t = now();
    int n = 0; (void)hipGetDeviceCount(&n); (void)hipSetDevice(0);
    printf("runtime init + device count %.1f ms\n", (now() - t) * 1e3); t = now();
    void* p; (void)hipMalloc(&p, 1 << 20);
    printf("first hipMalloc 1 MB        %.2f ms\n", (now() - t) * 1e3); t = now();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (int*)p); (void)hipDeviceSynchronize();
    printf("first launch (module load)  %.2f ms\n", (now() - t) * 1e3);
    for (int rep = 0; rep < 2; rep++) {
        t = now();
        void* q[16];
        for (int i = 0; i < 16; i++) (void)hipMalloc(&q[i], (size_t)8 << 20);
        printf("16 x hipMalloc 8 MB         %.2f ms\n", (now() - t) * 1e3); t = now();
        void* big; (void)hipMalloc(&big, (size_t)128 << 20);
        printf("1 x hipMalloc 128 MB        %.2f ms\n", (now() - t) * 1e3); t = now();
        void* h[3];
        for (int i = 0; i < 3; i++) (void)hipHostMalloc(&h[i], (size_t)8 << 20, hipHostMallocDefault);
        printf("3 x hipHostMalloc 8 MB      %.2f ms\n", (now() - t) * 1e3); t = now();
        void* hb; (void)hipHostMalloc(&hb, (size_t)24 << 20, hipHostMallocDefault);
        printf("1 x hipHostMalloc 24 MB     %.2f ms\n", (now() - t) * 1e3); t = now();
        hipStream_t s1, s2, s3; int lo, hi; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        (void)hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi);
        printf("stream create (high prio)   %.2f ms\n", (now() - t) * 1e3); t = now();
        (void)hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo);
        printf("stream create (low prio)    %.2f ms\n", (now() - t) * 1e3); t = now();
        (void)hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
        printf("stream create (default)     %.2f ms\n", (now() - t) * 1e3); t = now();
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, (int*)p); (void)hipStreamSynchronize(s1);
        printf("first launch on s1          %.2f ms\n", (now() - t) * 1e3); t = now();
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s2, (int*)p); (void)hipStreamSynchronize(s2);
        printf("first launch on s2          %.2f ms\n", (now() - t) * 1e3); t = now();
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s3, (int*)p); (void)hipStreamSynchronize(s3);
        printf("first launch on s3          %.2f ms\n", (now() - t) * 1e3); t = now();
        hipEvent_t e[4]; for (int i = 0; i < 4; i++) (void)hipEventCreateWithFlags(&e[i], hipEventDisableTiming);
        printf("4 events                    %.2f ms\n", (now() - t) * 1e3); t = now();
        (void)hipMemsetAsync(big, 0, 64 << 20, s1); (void)hipStreamSynchronize(s1);
        printf("memset 64 MB + sync         %.2f ms\n", (now() - t) * 1e3); t = now();
        (void)hipMemcpyAsync(big, hb, 24 << 20, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1);
        printf("H2D 24 MB pinned            %.2f ms\n", (now() - t) * 1e3); t = now();
        (void)hipMemcpyAsync(hb, big, 24 << 20, hipMemcpyDeviceToHost, s1); (void)hipStreamSynchronize(s1);
        printf("D2H 24 MB pinned            %.2f ms\n", (now() - t) * 1e3); t = now();
        for (int i = 0; i < 16; i++) (void)hipFree(q[i]);
        (void)hipFree(big); for (int i = 0; i < 3; i++) (void)hipHostFree(h[i]); (void)hipHostFree(hb);
        (void)hipStreamDestroy(s1); (void)hipStreamDestroy(s2); (void)hipStreamDestroy(s3);
        printf("free everything             %.2f ms\n---\n", (now() - t) * 1e3);
    }
    return 0;
}
