// Diagnostic build: cycle stamps inside k_frame_serial (never part of the product).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DICSP_DIAG -DICSP_DIAG_SERIAL -Iinclude -o tools/diag_serial.bin tools/diag_serial.hip icspcodec_amd/csrc/icsp_bitstream.cpp
#include "../icspcodec_amd/csrc/icsp_device.hip"
#include "../icspcodec_amd/csrc/icsp_sched.cpp"      // (the host half: contexts, scheduling, the C ABI)
#include <vector>
#include <cstdlib>
int main()
{
    const int nframes = (getenv("DIAG_HD") ? 8 : 300);
    const int W = getenv("DIAG_HD") ? 1920 : 352, H = getenv("DIAG_HD") ? 1088 : 288;
    icsp_params_t p{W, H, 8, 8, getenv("DIAG_HD") ? 4 : 10};
    icsp_ctx_t* ctx = nullptr;
    if (int rc = icsp_create(&ctx, &p, 0, nframes)) { printf("create: %s\n", icsp_strerror(rc)); return 1; }
    std::vector<uint8_t> clip((size_t)nframes * W * H * 3 / 2);
    unsigned x = 12345;
    for (size_t i = 0; i < clip.size(); i++) { x = x * 1664525u + 1013904223u; clip[i] = (uint8_t)(100 + ((i / W) % 64) + ((x >> 24) % 9)); }
    icsp_upload(ctx, clip.data(), 0, nframes);
    for (int rep = 0; rep < 5; rep++) { icsp_encode_resident(ctx, 0, nframes); icsp_sync(ctx); }
    unsigned long long d[16];
    hipMemcpyFromSymbol(d, HIP_SYMBOL(g_diag), sizeof(d));
    const char* names[6] = {"1 states+init", "2 loads->LDS", "2b mv/mvd stores", "3 chains (wave 0 = luma)", "barrier after chains", "4 predictor stores"};
    for (int i = 0; i < 6; i++) printf("%-28s %8llu cyc  %6.2f us\n", names[i], d[i], d[i] / 2400.0);
    printf("kernel body %llu shader cycles, %llu x10ns realtime -> clock %.3f GHz\n", d[9], d[8], d[9] / (d[8] * 10.0));
    icsp_destroy(ctx);
    return 0;
}
