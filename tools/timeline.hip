// Diagnostic build (never part of the product): when does every workgroup of the LAST P step's kernels start and end?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w -DICSP_TIMELINE -Iinclude -o tools/timeline.bin tools/timeline.hip icspcodec_amd/csrc/icsp_bitstream.cpp
// One line per (kernel, GOP group): workgroups, first/last start, first/last end (us from the earliest record; s_memrealtime,
// 100 MHz), median and maximum workgroup duration.
#include "../icspcodec_amd/csrc/icsp_device.hip"
#include "../icspcodec_amd/csrc/icsp_sched.cpp"      // (the host half: contexts, scheduling, the C ABI)
#include <algorithm>
#include <vector>
int main(int argc, char** argv)
{
    const int nframes = argc > 1 ? atoi(argv[1]) : 300, period = argc > 2 ? atoi(argv[2]) : 10;
    icsp_params_t p{352, 288, 8, 8, period};
    icsp_ctx_t* ctx = nullptr;
    if (int rc = icsp_create(&ctx, &p, 0, nframes)) { printf("create: %s\n", icsp_strerror(rc)); return 1; }
    std::vector<uint8_t> clip((size_t)nframes * 152064);
    unsigned x = 12345;
    for (size_t i = 0; i < clip.size(); i++) { x = x * 1664525u + 1013904223u; clip[i] = (uint8_t)(100 + ((i / 352 + i / 152064 * 3) % 64) + ((x >> 24) % 9)); }
    icsp_upload(ctx, clip.data(), 0, nframes);
    for (int rep = 0; rep < 60; rep++) { icsp_encode_resident(ctx, 0, nframes); icsp_sync(ctx); }
    std::vector<TlEntry> e(8 * 2 * 8192);
    (void)hipMemset(nullptr, 0, 0);
    { std::vector<TlEntry> z(e.size(), TlEntry{0, 0}); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl), z.data(), z.size() * sizeof(TlEntry)); }
    icsp_encode_resident(ctx, 0, nframes); icsp_sync(ctx);
    (void)hipMemcpyFromSymbol(e.data(), HIP_SYMBOL(g_tl), e.size() * sizeof(TlEntry));
    const char* names[7] = {"?", "k_me<false>", "k_me<true>", "serial", "search-in-fused", "k_residual8(P)", "k_residual8(I)"};
    unsigned long long T0 = ~0ull;
    for (int k = 1; k < 6; k++) for (auto& r : e) if (r.t0 && r.t0 < T0) T0 = r.t0;      // (I-frame chroma excluded from the origin)
    printf("%-18s %3s %6s %9s %9s %9s %9s %8s %8s\n", "kernel", "grp", "wgs", "start0", "startN", "end0", "endN", "med_dur", "max_dur");
    for (int grp = 0; grp < 2; grp++)
        for (int k : {1, 3, 4, 5}) {
            std::vector<TlEntry> v;
            for (int i = 0; i < 8192; i++) { const TlEntry& r = e[(k * 2 + grp) * 8192 + i]; if (r.t0) v.push_back(r); }
            if (v.empty()) continue;
            std::vector<double> d;
            unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
            for (auto& r : v) { d.push_back((r.t1 - r.t0) / 100.0); s0 = std::min(s0, r.t0); s1 = std::max(s1, r.t0); e0 = std::min(e0, r.t1); e1 = std::max(e1, r.t1); }
            std::sort(d.begin(), d.end());
            printf("%-18s %3d %6zu %9.2f %9.2f %9.2f %9.2f %8.2f %8.2f\n", names[k], grp, v.size(), (s0 - T0) / 100.0, (s1 - T0) / 100.0,
                   (e0 - T0) / 100.0, (e1 - T0) / 100.0, d[d.size() / 2], d.back());
        }
    icsp_destroy(ctx);
    return 0;
}
