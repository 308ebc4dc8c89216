// Diagnostic build: cycle stamps inside k_me<false> (never part of the product).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DICSP_DIAG -DICSP_DIAG_ME -Iinclude -o tools/diag_me.bin tools/diag_me.hip icspcodec_amd/csrc/icsp_bitstream.cpp
#include "../icspcodec_amd/csrc/icsp_device.hip"
#include "../icspcodec_amd/csrc/icsp_sched.cpp"      // (the host half: contexts, scheduling, the C ABI)
#include <vector>
#include <cstdlib>
int main()
{
    const int nframes = 300;
    icsp_params_t p{352, 288, 8, 8, 10};
    icsp_ctx_t* ctx = nullptr;
    if (int rc = icsp_create(&ctx, &p, 0, nframes)) { printf("create: %s\n", icsp_strerror(rc)); return 1; }
    std::vector<uint8_t> clip((size_t)nframes * 152064);
    unsigned x = 12345;
    for (size_t i = 0; i < clip.size(); i++) { x = x * 1664525u + 1013904223u; clip[i] = (uint8_t)(100 + ((i / 352) % 64) + ((x >> 24) % 9)); }
    icsp_upload(ctx, clip.data(), 0, nframes);
    for (int rep = 0; rep < 5; rep++) { icsp_encode_resident(ctx, 0, nframes); icsp_sync(ctx); }
    unsigned long long d[16];
    hipMemcpyFromSymbol(d, HIP_SYMBOL(g_diag), sizeof(d));
    const char* names[5] = {"LDS writes + barrier", "SAD loop (64 candidates)", "resolve + block sums + stores", "entry -> kernel args + tables in registers", "issue + wait: current rows, window segments"};
    for (int i : {3, 4, 0, 1, 2}) printf("%-46s %8llu cyc  %6.2f us\n", names[i], d[i], d[i] / 2400.0);
    printf("wave 0 of workgroup 0: %llu shader cycles, %llu x10ns realtime\n", d[9], d[8]);
    icsp_destroy(ctx);
    return 0;
}
