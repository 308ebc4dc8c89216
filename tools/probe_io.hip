// Probe (not product): what can the host side of a streaming encoder reach on this box?
//   hipcc -O2 -o tools/probe_io.bin tools/probe_io.hip -lpthread ; tools/probe_io.bin [dir] [MB]
// Measures, for a file of the given size in the given directory (default /dev/shm, 456 MB = 3000 CIF frames):
//   page-cache reads (pread) with 1..16 threads; writes into a fresh file by pwrite and by mmap + memcpy with 1..16 threads;
//   hipHostMalloc / hipHostRegister cost by size; H2D, D2H and both at once from pinned memory; H2D straight from a
//   registered mapping of the input file.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double par(int nt, F f)
{
    const double t = now();
    std::vector<std::thread> th;
    for (int i = 0; i < nt; i++) th.emplace_back(f, i);
    for (auto& x : th) x.join();
    return now() - t;
}
int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t MB = argc > 2 ? (size_t)atol(argv[2]) : 456;
    const size_t N = MB << 20;
    const std::string fin = dir + "/probe_io_in.bin", fout = dir + "/probe_io_out.bin";
    uint8_t* src = (uint8_t*)malloc(N);
    for (size_t i = 0; i < N; i += 4096) src[i] = (uint8_t)i;
    { int fd = open(fin.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); size_t o = 0; while (o < N) { ssize_t k = write(fd, src + o, N - o); if (k <= 0) return 1; o += k; } close(fd); }
    uint8_t* dst = (uint8_t*)malloc(N);
    memset(dst, 1, N);
    printf("dir %s, %zu MB, %u hardware threads\n", dir.c_str(), MB, std::thread::hardware_concurrency());
    const int nts[] = {1, 2, 4, 8, 16, 32};
    for (int nt : nts) {
        int fd = open(fin.c_str(), O_RDONLY);
        const double t = par(nt, [&](int i) { size_t a = N / nt * i, b = (i == nt - 1) ? N : N / nt * (i + 1); while (a < b) { ssize_t k = pread(fd, dst + a, std::min<size_t>(b - a, 8 << 20), a); if (k <= 0) break; a += k; } });
        close(fd);
        printf("pread        %2d threads: %6.1f ms  %5.1f GB/s\n", nt, t * 1e3, N / t / 1e9);
    }
    for (int nt : nts) {
        int fd = open(fin.c_str(), O_RDONLY);
        const uint8_t* m = (const uint8_t*)mmap(nullptr, N, PROT_READ, MAP_SHARED, fd, 0);
        const double t = par(nt, [&](int i) { size_t a = N / nt * i, b = (i == nt - 1) ? N : N / nt * (i + 1); memcpy(dst + a, m + a, b - a); });
        munmap((void*)m, N); close(fd);
        printf("mmap read    %2d threads: %6.1f ms  %5.1f GB/s\n", nt, t * 1e3, N / t / 1e9);
    }
    for (int nt : nts) {
        unlink(fout.c_str());
        int fd = open(fout.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (ftruncate(fd, N)) return 1;
        const double t = par(nt, [&](int i) { size_t a = N / nt * i, b = (i == nt - 1) ? N : N / nt * (i + 1); while (a < b) { ssize_t k = pwrite(fd, src + a, std::min<size_t>(b - a, 8 << 20), a); if (k <= 0) break; a += k; } });
        close(fd);
        printf("pwrite fresh %2d threads: %6.1f ms  %5.1f GB/s\n", nt, t * 1e3, N / t / 1e9);
    }
    for (int nt : nts) {
        unlink(fout.c_str());
        int fd = open(fout.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (ftruncate(fd, N)) return 1;
        const double t0 = now();
        uint8_t* m = (uint8_t*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        const double t = par(nt, [&](int i) { size_t a = N / nt * i, b = (i == nt - 1) ? N : N / nt * (i + 1); memcpy(m + a, src + a, b - a); });
        const double t1 = now();
        munmap(m, N); close(fd);
        printf("mmap write   %2d threads: %6.1f ms  %5.1f GB/s   (mmap..copy %.1f ms, munmap+close %.1f ms)\n", nt, t * 1e3, N / t / 1e9, (t1 - t0) * 1e3, (now() - t1) * 1e3);
    }
    {   // overwrite an existing file (pages already there)
        int fd = open(fout.c_str(), O_WRONLY);
        const double t = par(8, [&](int i) { size_t a = N / 8 * i, b = (i == 7) ? N : N / 8 * (i + 1); while (a < b) { ssize_t k = pwrite(fd, src + a, std::min<size_t>(b - a, 8 << 20), a); if (k <= 0) break; a += k; } });
        close(fd);
        printf("pwrite over existing pages, 8 threads: %6.1f ms  %5.1f GB/s\n", t * 1e3, N / t / 1e9);
    }
    // ---- device side
    double t = now();
    int n = 0; (void)hipGetDeviceCount(&n);
    if (n <= 0) { printf("no device\n"); return 0; }
    (void)hipSetDevice(0);
    void* d0; void* d1; (void)hipMalloc(&d0, N); (void)hipMalloc(&d1, N);
    printf("runtime init + 2 x hipMalloc %zu MB: %.1f ms\n", MB, (now() - t) * 1e3);
    for (size_t mb : {8, 32, 128, 456}) {
        if (mb > MB) break;
        t = now(); void* h; (void)hipHostMalloc(&h, mb << 20, hipHostMallocDefault); const double ta = now() - t;
        t = now(); memset(h, 0, mb << 20); const double tm = now() - t;
        t = now(); (void)hipHostFree(h); const double tf = now() - t;
        void* r = aligned_alloc(4096, mb << 20); memset(r, 1, mb << 20);
        t = now(); const hipError_t e = hipHostRegister(r, mb << 20, hipHostRegisterDefault); const double tr = now() - t;
        t = now(); if (e == hipSuccess) (void)hipHostUnregister(r); const double tu = now() - t;
        free(r);
        printf("pinned %3zu MB: hipHostMalloc %.2f ms (first touch %.2f ms, free %.2f ms); hipHostRegister of touched memory %.2f ms (%s), unregister %.2f ms\n",
               mb, ta * 1e3, tm * 1e3, tf * 1e3, tr * 1e3, hipGetErrorName(e), tu * 1e3);
    }
    void* hp; void* hq; (void)hipHostMalloc(&hp, N, hipHostMallocDefault); (void)hipHostMalloc(&hq, N, hipHostMallocDefault);
    memcpy(hp, src, N);
    hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; rep++) {
        t = now(); (void)hipMemcpyAsync(d0, hp, N, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1);
        const double th = now() - t;
        t = now(); (void)hipMemcpyAsync(hq, d1, N, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2);
        const double td = now() - t;
        t = now(); (void)hipMemcpyAsync(d0, hp, N, hipMemcpyHostToDevice, s1); (void)hipMemcpyAsync(hq, d1, N, hipMemcpyDeviceToHost, s2);
        (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2);
        const double tb = now() - t;
        printf("pinned copies of %zu MB: H2D %.1f ms (%.1f GB/s), D2H %.1f ms (%.1f GB/s), both at once %.1f ms\n", MB, th * 1e3, N / th / 1e9, td * 1e3, N / td / 1e9, tb * 1e3);
    }
    {   // pageable source / destination
        t = now(); (void)hipMemcpy(d0, src, N, hipMemcpyHostToDevice); const double th = now() - t;
        t = now(); (void)hipMemcpy(dst, d1, N, hipMemcpyDeviceToHost); const double td = now() - t;
        printf("pageable copies: H2D %.1f ms (%.1f GB/s), D2H %.1f ms (%.1f GB/s)\n", th * 1e3, N / th / 1e9, td * 1e3, N / td / 1e9);
    }
    {   // the input file mapped and registered: H2D without a staging copy?
        int fd = open(fin.c_str(), O_RDONLY);
        for (int priv = 0; priv < 2; priv++) {
            void* m = mmap(nullptr, N, PROT_READ | (priv ? PROT_WRITE : 0), (priv ? MAP_PRIVATE : MAP_SHARED) | MAP_POPULATE, fd, 0);
            t = now(); const hipError_t e = hipHostRegister(m, N, priv ? hipHostRegisterDefault : hipHostRegisterReadOnly); const double tr = now() - t;
            printf("hipHostRegister of the %s input mapping: %s, %.1f ms\n", priv ? "private" : "shared read-only", hipGetErrorName(e), tr * 1e3);
            if (e == hipSuccess) {
                t = now(); (void)hipMemcpyAsync(d0, m, N, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1);
                printf("   H2D from it: %.1f ms (%.1f GB/s)\n", (now() - t) * 1e3, N / (now() - t) / 1e9);
                (void)hipHostUnregister(m);
            } else (void)hipGetLastError();
            munmap(m, N);
        }
        close(fd);
    }
    {   // the output file mapped and registered: D2H straight into the page cache?
        unlink(fout.c_str());
        int fd = open(fout.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (ftruncate(fd, N)) return 1;
        t = now();
        void* m = mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
        const double tp = now() - t;
        t = now(); const hipError_t e = hipHostRegister(m, N, hipHostRegisterDefault); const double tr = now() - t;
        printf("output mapping: mmap+populate %.1f ms, hipHostRegister %s %.1f ms\n", tp * 1e3, hipGetErrorName(e), tr * 1e3);
        if (e == hipSuccess) {
            t = now(); (void)hipMemcpyAsync(m, d1, N, hipMemcpyDeviceToHost, s2); (void)hipStreamSynchronize(s2);
            printf("   D2H into it: %.1f ms (%.1f GB/s)\n", (now() - t) * 1e3, N / (now() - t) / 1e9);
            (void)hipHostUnregister(m);
        } else (void)hipGetLastError();
        munmap(m, N); close(fd);
    }
    {   // first-use costs: repeated 45 MB copies from a freshly registered read-only mapping / into a partly registered one
        int fd = open(fin.c_str(), O_RDONLY);
        void* m = mmap(nullptr, N, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        (void)hipHostRegister(m, N, hipHostRegisterPortable | hipHostRegisterReadOnly);
        const size_t c = std::min<size_t>(N, (size_t)45 << 20);
        for (int rep = 0; rep < 4; rep++) {
            t = now(); (void)hipMemcpyAsync(d0, (char*)m + (rep & 1) * c, c, hipMemcpyHostToDevice, s1); const double tc = now() - t; (void)hipStreamSynchronize(s1);
            printf("H2D 45 MB from the registered input mapping, copy %d: call %.2f ms, done %.2f ms\n", rep, tc * 1e3, (now() - t) * 1e3);
        }
        (void)hipHostUnregister(m); munmap(m, N); close(fd);
        unlink(fout.c_str());
        fd = open(fout.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (ftruncate(fd, 4 * N)) return 1;
        uint8_t* o = (uint8_t*)mmap(nullptr, 4 * N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        t = now(); const int mr = madvise(o, N / 2, 23 /* MADV_POPULATE_WRITE */); const double tp = now() - t;
        t = now(); const hipError_t e = hipHostRegister(o, N / 2, hipHostRegisterPortable); const double tr = now() - t;
        printf("sparse output mapping: madvise(POPULATE_WRITE) of %zu MB rc %d %.1f ms, register %s %.1f ms\n", MB / 2, mr, tp * 1e3, hipGetErrorName(e), tr * 1e3);
        const size_t c2 = std::min<size_t>(N / 4, (size_t)13 << 20);
        for (int rep = 0; rep < 4; rep++) {
            const size_t off = rep < 2 ? 64 : 15;
            t = now(); (void)hipMemcpyAsync(o + off + rep * c2, (char*)d1 + off, c2, hipMemcpyDeviceToHost, s2); const double tc = now() - t; (void)hipStreamSynchronize(s2);
            printf("D2H 13 MB into it at offset %% 64 = %zu, copy %d: call %.2f ms, done %.2f ms\n", off % 64, rep, tc * 1e3, (now() - t) * 1e3);
        }
        for (size_t sz : {(size_t)4096, (size_t)65536, (size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20}) {
            t = now(); (void)hipMemcpyAsync(o, d1, sz, hipMemcpyDeviceToHost, s2); const double tc = now() - t; (void)hipStreamSynchronize(s2);
            printf("D2H %8zu bytes into it: call %.3f ms, done %.3f ms\n", sz, tc * 1e3, (now() - t) * 1e3);
        }
        (void)hipHostUnregister(o); munmap(o, 4 * N); close(fd);
    }
    unlink(fin.c_str()); unlink(fout.c_str());
    return 0;
}
