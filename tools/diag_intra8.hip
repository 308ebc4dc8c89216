// Diagnostic build of the 8-lane intra kernel with in-kernel cycle stamps (s_memtime) -- never part of the product.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DICSP_DIAG8 -Iinclude -o tools/diag_intra8.bin tools/diag_intra8.hip icspcodec_amd/csrc/icsp_bitstream.cpp icspcodec_amd/csrc/icsp_topology.cpp
//   ICSP_INTRA_FORM=8 tools/diag_intra8.bin [frames per range] [ranges alternating]
#include "../icspcodec_amd/csrc/icsp_device.hip"
#include "../icspcodec_amd/csrc/icsp_sched.cpp"      // (the host half: contexts, scheduling, the C ABI)
#include <vector>
#include <cstdlib>
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 300, R = argc > 2 ? atoi(argv[2]) : 2;
    icsp_params_t p{352, 288, 16, 16, 0};
    icsp_ctx_t* ctx = nullptr;
    if (int rc = icsp_create(&ctx, &p, 0, n * R)) { printf("create: %s\n", icsp_strerror(rc)); return 1; }
    std::vector<uint8_t> clip((size_t)n * R * 152064);
    unsigned x = 12345;
    for (auto& v : clip) { x = x * 1664525u + 1013904223u; v = (uint8_t)(128 + ((x >> 24) % 40) - 20); }
    icsp_upload(ctx, clip.data(), 0, n * R);
    for (int rep = 0; rep < 40; rep++) icsp_encode_resident(ctx, (rep % R) * n, n);
    icsp_sync(ctx);
    unsigned long long d[20];
    (void)hipMemcpyFromSymbol(d, HIP_SYMBOL(g_diag8), sizeof(d));
    const char* names[11] = {"source row + neighbour state arrive", "SAE, mode, residual, predictors", "forward pass 1", "transpose (LDS)", "forward pass 2, quantiser",
                             "zig-zag + dequantised transpose (LDS), level store", "inverse pass 1", "transpose (LDS)", "inverse pass 2", "reconstruction, state, ring", "step barrier"};
    unsigned long long tot = 0; for (int i = 0; i < 11; i++) tot += d[i];
    for (int i = 0; i < 11; i++) printf("%-52s %10llu cyc  %5.1f%%  %7.1f cyc/step\n", names[i], d[i], 100.0 * d[i] / tot, d[i] / 114.0);
    printf("kernel %llu shader cycles, %llu x10ns realtime -> clock %.3f GHz, %.2f us/step (%d frames x %d ranges)\n", d[15], d[14], d[15] / (d[14] * 10.0), d[14] * 0.01 / 114, n, R);
    icsp_destroy(ctx);
    return 0;
}
