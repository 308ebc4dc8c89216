// Diagnostic build of the intra kernel with in-kernel cycle stamps (s_memtime) — never part of the product.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DICSP_DIAG -Iinclude -o /tmp/diag_intra tools/diag_intra.hip icspcodec_amd/csrc/icsp_bitstream.cpp
#include "../icspcodec_amd/csrc/icsp_device.hip"
#include "../icspcodec_amd/csrc/icsp_sched.cpp"      // (the host half: contexts, scheduling, the C ABI)
#include <vector>
#include <cstdlib>
int main(int argc, char** argv)
{
    int nframes = argc > 1 ? atoi(argv[1]) : 30;
    icsp_params_t p{352, 288, 16, 16, 0};
    icsp_ctx_t* ctx = nullptr;
    if (int rc = icsp_create(&ctx, &p, 0, nframes)) { printf("create: %s\n", icsp_strerror(rc)); return 1; }
    std::vector<uint8_t> clip((size_t)nframes * 152064);
    unsigned x = 12345;
    for (auto& v : clip) { x = x * 1664525u + 1013904223u; v = (uint8_t)(128 + ((x >> 24) % 40) - 20); }
    icsp_upload(ctx, clip.data(), 0, nframes);
    for (int rep = 0; rep < 5; rep++) { icsp_encode_resident(ctx, 0, nframes); icsp_sync(ctx); }
    unsigned long long d[16];
    hipMemcpyFromSymbol(d, HIP_SYMBOL(g_diag), sizeof(d));
    const char* names[7] = {"pred+SAE+mode", "residual+mpm+dcpred", "DCT", "quant+zigzag+iq-xpose", "IDCT", "recon+stores", "barrier"};
    unsigned long long tot = 0; for (int i = 0; i < 7; i++) tot += d[i];
    for (int i = 0; i < 7; i++) printf("%-24s %10llu cyc  %5.1f%%  %7.1f cyc/step\n", names[i], d[i], 100.0 * d[i] / tot, d[i] / 114.0);
    printf("loop total %llu shader cycles, %llu x10ns realtime -> clock %.3f GHz, %.2f us/step\n", d[9], d[8], d[9] / (d[8] * 10.0), d[8] * 0.01 / 114);
    icsp_destroy(ctx);
    return 0;
}
