// Probe (not product): what does the HOST pay to enqueue a kernel on this runtime, and which ways around it exist?
//   hipcc --offload-arch=gfx950 -O2 -pthread -o tools/probe_enqueue.bin tools/probe_enqueue.hip
// Kernels take ~300 bytes of by-value arguments like the product's (Geo + FrameSel + DevBufs) and do nothing.
//   1. one thread, one stream;  2. one thread, three streams in turn + an event record / wait pair every ten launches (a pass's shape);
//   3. two threads, a stream each (do launches from two threads run side by side?);  4. a captured graph of 30 launches on one stream;
//   5. a captured graph of 3 streams' fork / join shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>
struct Args { long long a[38]; };
__global__ void k_nop(Args x, int* p) { if (x.a[0] == 0x12345 && threadIdx.x == 999) p[0] = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    (void)hipSetDevice(0);
    int* d; (void)hipMalloc(&d, 64);
    hipStream_t s[4];
    for (auto& x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    hipEvent_t ev[4];
    for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    Args a{};
    const int N = 3000;
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_nop, dim3(64), dim3(256), 0, s[i & 3], a, d);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        double t = now();
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_nop, dim3(64), dim3(256), 0, s[0], a, d);
        double te = now(); (void)hipDeviceSynchronize();
        printf("1 thread, 1 stream:            %.2f us per launch to enqueue (%.2f with the drain)\n", (te - t) / N * 1e6, (now() - t) / N * 1e6);
        t = now();
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(k_nop, dim3(64), dim3(256), 0, s[i % 3], a, d);
            if (i % 10 == 9) { (void)hipEventRecord(ev[0], s[i % 3]); (void)hipStreamWaitEvent(s[(i + 1) % 3], ev[0], 0); }
        }
        te = now(); (void)hipDeviceSynchronize();
        printf("1 thread, 3 streams + events:  %.2f us per launch to enqueue (%.2f with the drain)\n", (te - t) / N * 1e6, (now() - t) / N * 1e6);
        t = now();
        {
            std::vector<std::thread> th;
            for (int w = 0; w < 2; w++) th.emplace_back([&, w] { (void)hipSetDevice(0); for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_nop, dim3(64), dim3(256), 0, s[w], a, d); });
            for (auto& x : th) x.join();
        }
        te = now(); (void)hipDeviceSynchronize();
        printf("2 threads, a stream each:      %.2f us per launch to enqueue, both threads' launches counted (%.2f with the drain)\n", (te - t) / (2 * N) * 1e6, (now() - t) / (2 * N) * 1e6);
        t = now();
        {
            std::vector<std::thread> th;
            for (int w = 0; w < 3; w++) th.emplace_back([&, w] { (void)hipSetDevice(0); for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_nop, dim3(64), dim3(256), 0, s[w], a, d); });
            for (auto& x : th) x.join();
        }
        te = now(); (void)hipDeviceSynchronize();
        printf("3 threads, a stream each:      %.2f us per launch to enqueue (%.2f with the drain)\n", (te - t) / (3 * N) * 1e6, (now() - t) / (3 * N) * 1e6);
    }
    // graphs
    {
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 30; i++) hipLaunchKernelGGL(k_nop, dim3(64), dim3(256), 0, s[0], a, d);
        if (hipStreamEndCapture(s[0], &g) != hipSuccess || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("graph capture failed: %s\n", hipGetErrorString(hipGetLastError())); return 0; }
        for (int i = 0; i < 20; i++) (void)hipGraphLaunch(ge, s[0]);
        (void)hipDeviceSynchronize();
        double t = now();
        for (int i = 0; i < 200; i++) (void)hipGraphLaunch(ge, s[0]);
        double te = now(); (void)hipDeviceSynchronize();
        printf("graph of 30 launches, 1 chain: %.2f us per graph launch to enqueue = %.2f per kernel (%.2f per kernel with the drain)\n", (te - t) / 200 * 1e6, (te - t) / 6000 * 1e6, (now() - t) / 6000 * 1e6);
        // two graphs on two streams in turn
        t = now();
        for (int i = 0; i < 200; i++) (void)hipGraphLaunch(ge, s[i & 1]);
        te = now(); (void)hipDeviceSynchronize();
        printf("  the same graph on two streams in turn: %.2f us per graph launch (%.2f per kernel with the drain)\n", (te - t) / 200 * 1e6, (now() - t) / 6000 * 1e6);
    }
    {
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal);
        (void)hipEventRecord(ev[1], s[0]); (void)hipStreamWaitEvent(s[1], ev[1], 0);
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_nop, dim3(64), dim3(256), 0, s[1], a, d);
        (void)hipEventRecord(ev[2], s[1]); (void)hipStreamWaitEvent(s[0], ev[2], 0);
        for (int i = 0; i < 27; i++) hipLaunchKernelGGL(k_nop, dim3(64), dim3(256), 0, s[0], a, d);
        if (hipStreamEndCapture(s[0], &g) != hipSuccess || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("graph capture (fork/join) failed: %s\n", hipGetErrorString(hipGetLastError())); return 0; }
        for (int i = 0; i < 20; i++) (void)hipGraphLaunch(ge, s[0]);
        (void)hipDeviceSynchronize();
        double t = now();
        for (int i = 0; i < 200; i++) (void)hipGraphLaunch(ge, s[i & 1 ? 2 : 0]);
        double te = now(); (void)hipDeviceSynchronize();
        printf("graph of 3 + 27 launches with a fork and a join, two streams in turn: %.2f us per graph launch to enqueue (%.2f per kernel with the drain)\n", (te - t) / 200 * 1e6, (now() - t) / 6000 * 1e6);
    }
    return 0;
}
