// probe_duplex.hip -- do host-to-device and device-to-host DMA overlap when the host side is a registered file mapping, in the
// chunked, several-thread shape icsp_enc uses?  hipcc --offload-arch=gfx950 -O2 -o tools/probe_duplex.bin tools/probe_duplex.hip -lpthread
// usage: probe_duplex.bin [dir=/dev/shm] [MB=456]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
__global__ void k_touch(char* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = (char)(p[0] + 1); }
// device -> pinned host by a kernel instead of a DMA engine: 16 bytes per lane, grid-stride
__global__ void k_copy_out(uint4* dst, const uint4* src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t MB = argc > 2 ? (size_t)atol(argv[2]) : 456, N = MB << 20;
    const std::string fin = dir + "/probe_duplex_in.bin", fout = dir + "/probe_duplex_out.bin";
    { std::vector<char> z(N, 7); int fd = open(fin.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644); if (write(fd, z.data(), N) != (ssize_t)N) return 1; close(fd); }
    char *d0, *d1; (void)hipMalloc((void**)&d0, N); (void)hipMalloc((void**)&d1, N); (void)hipMemset(d1, 3, N);
    int fi = open(fin.c_str(), O_RDONLY);
    char* mi = (char*)mmap(nullptr, N, PROT_READ, MAP_SHARED | MAP_POPULATE, fi, 0);
    int fo = open(fout.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644); if (ftruncate(fo, N)) return 1;
    char* mo = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fo, 0);
    printf("register in: %s, out: %s\n", hipGetErrorName(hipHostRegister(mi, N, hipHostRegisterPortable | hipHostRegisterReadOnly)),
           hipGetErrorName(hipHostRegister(mo, N, hipHostRegisterPortable)));
    char *hp, *hq; (void)hipHostMalloc((void**)&hp, N, hipHostMallocDefault); (void)hipHostMalloc((void**)&hq, N, hipHostMallocDefault); memset(hp, 1, N); memset(hq, 2, N);
    hipStream_t s[8]; for (auto& x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    auto both = [&](const char* what, char* in, char* out) {
        for (int rep = 0; rep < 3; rep++) {
            double t = now(); (void)hipMemcpyAsync(d0, in, N, hipMemcpyHostToDevice, s[0]); (void)hipStreamSynchronize(s[0]); const double th = now() - t;
            t = now(); (void)hipMemcpyAsync(out, d1, N, hipMemcpyDeviceToHost, s[1]); (void)hipStreamSynchronize(s[1]); const double td = now() - t;
            t = now(); (void)hipMemcpyAsync(d0, in, N, hipMemcpyHostToDevice, s[0]); (void)hipMemcpyAsync(out, d1, N, hipMemcpyDeviceToHost, s[1]);
            (void)hipStreamSynchronize(s[0]); (void)hipStreamSynchronize(s[1]);
            printf("%-28s whole: H2D %.1f ms, D2H %.1f ms, both at once %.1f ms\n", what, th * 1e3, td * 1e3, (now() - t) * 1e3);
        }
    };
    both("hipHostMalloc", hp, hq);
    both("registered mappings", mi, mo);
    // chunked, one thread per direction
    for (int nch : {6, 12, 24}) for (int reg = 0; reg < 2; reg++) {
        char* in = reg ? mi : hp; char* out = reg ? mo : hq;
        const size_t c = N / nch;
        const double t = now();
        std::thread a([&] { for (int k = 0; k < nch; k++) { (void)hipMemcpyAsync(d0 + k * c, in + k * c, c, hipMemcpyHostToDevice, s[0]); (void)hipStreamSynchronize(s[0]); } });
        std::thread b([&] { for (int k = 0; k < nch; k++) { (void)hipMemcpyAsync(out + k * c, d1 + k * c, c, hipMemcpyDeviceToHost, s[1]); (void)hipStreamSynchronize(s[1]); } });
        a.join(); b.join();
        printf("%-20s %2d chunks, one thread per direction: %.1f ms\n", reg ? "registered mappings" : "hipHostMalloc", nch, (now() - t) * 1e3);
    }
    // icsp_enc's shape: W workers, each on its own stream: H2D chunk (async), then D2H of the same chunk (sync); chunks handed out in order
    for (int W : {1, 2, 3, 4}) for (int gate = 0; gate < 2; gate++) for (int reg = 0; reg < 2; reg++) {
        char* in = reg ? mi : hp; char* out = reg ? mo : hq;
        const int nch = 6; const size_t c = N / nch;
        std::mutex m, up; int next = 0;
        const double t = now();
        std::vector<std::thread> th;
        for (int w = 0; w < W; w++) th.emplace_back([&, w] {
            for (;;) {
                int k; { std::lock_guard<std::mutex> l(m); k = next++; }
                if (k >= nch) break;
                if (gate) { std::lock_guard<std::mutex> l(up); (void)hipMemcpyAsync(d0 + k * c, in + k * c, c, hipMemcpyHostToDevice, s[2 + w]); (void)hipStreamSynchronize(s[2 + w]); }
                else (void)hipMemcpyAsync(d0 + k * c, in + k * c, c, hipMemcpyHostToDevice, s[2 + w]);
                (void)hipMemcpyAsync(out + k * c, d1 + k * c, c, hipMemcpyDeviceToHost, s[2 + w]); (void)hipStreamSynchronize(s[2 + w]);
            }
        });
        for (auto& x : th) x.join();
        printf("%-20s %d workers, 6 chunks, up then down on the worker's stream%s: %.1f ms\n", reg ? "registered mappings" : "hipHostMalloc", W, gate ? ", uploads take turns" : "", (now() - t) * 1e3);
    }
    // the same with the download on a second stream of the worker (event-ordered), so that the worker's next upload can start under it
    for (int W : {1, 2, 3}) for (int reg = 0; reg < 2; reg++) {
        char* in = reg ? mi : hp; char* out = reg ? mo : hq;
        const int nch = 6; const size_t c = N / nch;
        hipStream_t up_s, dn_s; (void)hipStreamCreateWithFlags(&up_s, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&dn_s, hipStreamNonBlocking);
        std::vector<hipEvent_t> ev(nch); for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        const double t = now();
        for (int k = 0; k < nch; k++) {   // one host thread: all uploads on one stream, all downloads on another, download k after upload k
            (void)hipMemcpyAsync(d0 + k * c, in + k * c, c, hipMemcpyHostToDevice, up_s); (void)hipEventRecord(ev[k], up_s);
            (void)hipStreamWaitEvent(dn_s, ev[k], 0); (void)hipMemcpyAsync(out + k * c, d1 + k * c, c, hipMemcpyDeviceToHost, dn_s);
        }
        (void)hipStreamSynchronize(up_s); (void)hipStreamSynchronize(dn_s);
        if (W == 1) printf("%-20s one upload stream + one download stream, download k after upload k: %.1f ms\n", reg ? "registered mappings" : "hipHostMalloc", (now() - t) * 1e3);
        for (auto& e : ev) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(up_s); (void)hipStreamDestroy(dn_s);
    }
    // what makes icsp_enc's copies take turns on one engine?  streams with a priority / a kernel in front of every copy
    for (int prio = 0; prio < 2; prio++) for (int kern = 0; kern < 3; kern++) {
        hipStream_t q[3];
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        for (auto& x : q) { if (prio) (void)hipStreamCreateWithPriority(&x, hipStreamNonBlocking, hi); else (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking); }
        const int nch = 6; const size_t c = N / nch;
        for (int rep = 0; rep < 2; rep++) {
            std::mutex m, up; int next = 0;
            const double t = now();
            std::vector<std::thread> th;
            for (int w = 0; w < 3; w++) th.emplace_back([&, w] {
                for (;;) {
                    int k; { std::lock_guard<std::mutex> l(m); k = next++; }
                    if (k >= nch) break;
                    { std::lock_guard<std::mutex> l(up);
                      if (kern == 2) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, q[w], d1 + k * c);
                      (void)hipMemcpyAsync(d0 + k * c, mi + k * c, c, hipMemcpyHostToDevice, q[w]); (void)hipStreamSynchronize(q[w]); }
                    if (kern >= 1) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, q[w], d1 + k * c);
                    (void)hipMemcpyAsync(mo + k * c, d1 + k * c, c, hipMemcpyDeviceToHost, q[w]); (void)hipStreamSynchronize(q[w]);
                }
            });
            for (auto& x : th) x.join();
            printf("3 workers, uploads take turns, %s streams, %s: %.1f ms\n", prio ? "high-priority" : "plain",
                   kern == 0 ? "copies only" : kern == 1 ? "a kernel before every download" : "a kernel before every upload and download", (now() - t) * 1e3);
        }
        for (auto& x : q) (void)hipStreamDestroy(x);
    }
    // does a stream keep the DMA engine of its first copy?  bind: 0 = every stream's first copy issued while nothing else is in
    // flight (what sequential set-up does), 1 = while the earlier streams are kept busy with queued copies
    for (int bind = 0; bind < 2; bind++) {
        hipStream_t q[3];
        for (auto& x : q) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
        const double tb = now();
        for (int w = 0; w < 3; w++) {
            (void)hipMemcpyAsync(d0, hp, 64 << 20, hipMemcpyHostToDevice, q[w]);
            if (bind) for (int r = 0; r < 24; r++) (void)hipMemcpyAsync(d0 + ((size_t)w << 26), hp, 64 << 20, hipMemcpyHostToDevice, q[w]);
            else (void)hipStreamSynchronize(q[w]);
        }
        for (auto& x : q) (void)hipStreamSynchronize(x);
        printf("first copies %s: %.1f ms\n", bind ? "under the earlier streams' copies" : "one after the other", (now() - tb) * 1e3);
        const int nch = 6; const size_t c = N / nch;
        for (int rep = 0; rep < 3; rep++) {
            std::mutex m, up; int next = 0;
            const double t = now();
            std::vector<std::thread> th;
            for (int w = 0; w < 3; w++) th.emplace_back([&, w] {
                for (;;) {
                    int k; { std::lock_guard<std::mutex> l(m); k = next++; }
                    if (k >= nch) break;
                    { std::lock_guard<std::mutex> l(up); (void)hipMemcpyAsync(d0 + k * c, mi + k * c, c, hipMemcpyHostToDevice, q[w]); (void)hipStreamSynchronize(q[w]); }
                    (void)hipMemcpyAsync(mo + k * c, d1 + k * c, c, hipMemcpyDeviceToHost, q[w]); (void)hipStreamSynchronize(q[w]);
                }
            });
            for (auto& x : th) x.join();
            printf("   3 workers, uploads take turns: %.1f ms\n", (now() - t) * 1e3);
        }
        for (auto& x : q) (void)hipStreamDestroy(x);
    }
    // downloads by a copy kernel, uploads by the DMA engine: can never share an engine
    for (int wgs : {16, 32, 64, 128, 256}) for (int reg = 0; reg < 2; reg++) {
        char* out = reg ? mo : hq; char* in = reg ? mi : hp;
        for (int rep = 0; rep < 2; rep++) {
            double t = now();
            hipLaunchKernelGGL(k_copy_out, dim3(wgs), dim3(256), 0, s[1], (uint4*)out, (const uint4*)d1, N / 16); (void)hipStreamSynchronize(s[1]);
            const double td = now() - t;
            t = now();
            (void)hipMemcpyAsync(d0, in, N, hipMemcpyHostToDevice, s[0]);
            hipLaunchKernelGGL(k_copy_out, dim3(wgs), dim3(256), 0, s[1], (uint4*)out, (const uint4*)d1, N / 16);
            (void)hipStreamSynchronize(s[0]); (void)hipStreamSynchronize(s[1]);
            if (rep) printf("copy kernel D2H, %3d workgroups, %-20s: alone %.1f ms (%.1f GB/s), beside a DMA upload %.1f ms\n", wgs, reg ? "registered mappings" : "hipHostMalloc", td * 1e3, N / td / 1e9, (now() - t) * 1e3);
        }
    }
    (void)hipHostUnregister(mi); (void)hipHostUnregister(mo); munmap(mi, N); munmap(mo, N); close(fi); close(fo);
    unlink(fin.c_str()); unlink(fout.c_str());
    return 0;
}
