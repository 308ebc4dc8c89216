// Probe (not product): do stream creations of one process run side by side when made from several threads?
// hipcc -O2 -pthread -o tools/probe_streams_mt.bin tools/probe_streams_mt.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    double t = now();
    int n = 0; (void)hipGetDeviceCount(&n); (void)hipSetDevice(0);
    printf("runtime init %.1f ms\n", (now() - t) * 1e3);
    for (int threads : { 1, 3, 1, 3 }) {
        hipStream_t s[3];
        t = now();
        if (threads == 1) { for (int i = 0; i < 3; i++) (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); }
        else {
            std::vector<std::thread> th;
            for (int i = 0; i < 3; i++) th.emplace_back([&, i] { (void)hipSetDevice(0); double t0 = now(); (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); printf("   thread %d: %.1f ms\n", i, (now() - t0) * 1e3); });
            for (auto& x : th) x.join();
        }
        printf("3 streams from %d thread(s): %.1f ms\n", threads, (now() - t) * 1e3);
        for (int i = 0; i < 3; i++) (void)hipStreamDestroy(s[i]);
    }
    return 0;
}
