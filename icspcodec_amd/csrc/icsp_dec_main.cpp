// icsp_dec — host program with the reference decoder's command line, driving the HIP decode path through the C ABI.
//
// Mirrors, without copying, the host side of the reference decoder (all under /root/reference/source/decoder/):
//   main                  decode.cpp:4-27                      positional: nframes binfname QPDC QPAC intraPeriod imgfname
//   IcspCodec::init       ICSP_Codec_Decoder.h:237-283         opens output\<binfname>, reads header + body
//   IcspCodec::decoding   ICSP_Codec_Decoder.h:284-357         decodes, writes check_test_intra_yuv.yuv (header period 1)
//                                                              or check_test_inter_yuv.yuv, then appends
//                                                              "decoding time: ... PSNR: ... QPDC: ... QPAC: ... Period: ..."
//                                                              to experimental_Result_Decoding.txt (luma PSNR against data\<imgfname>)
//   checkResultFrames     ICSP_Codec_Decoder_source.cpp:4476-4527
// The reference builds its paths with a Windows separator ("output\\%s", "data\\%s"); here output/<name>, the literal
// output\<name> and <name> itself are tried in that order.  QPDC / QPAC / intraPeriod on the command line are ignored by
// the reference too (it uses the header's); they are accepted and checked against the header with a warning.
// The parse is bit-serial and stays on the host; reconstruction runs on the GPU (include/icsp_hip.h).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <string>
#include <vector>
#include "icsp_hip.h"

static FILE* open_in(const char* dir, const char* name, std::string* used)
{
    const std::string cands[3] = { std::string(dir) + "/" + name, std::string(dir) + "\\" + name, std::string(name) };
    for (const auto& c : cands) {
        FILE* f = fopen(c.c_str(), "rb");
        if (f) { if (used) *used = c; return f; }
    }
    return nullptr;
}

int main(int argc, char* argv[])
{
    if (argc < 7) { printf("usage: icsp_dec nframes binfname QPDC QPAC intraPeriod imgfname\n"); return -1; }
    const int nframes = atoi(argv[1]);
    const char* binfname = argv[2];
    const char* imgfname = argv[6];
    std::string path;
    FILE* fp = open_in("output", binfname, &path);
    if (!fp) { printf("fail to load output\\%s\n", binfname); exit(-1); }
    fseek(fp, 0, SEEK_END);
    const long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::vector<uint8_t> bin((size_t)sz);
    if (fread(bin.data(), 1, (size_t)sz, fp) != (size_t)sz) { printf("error in loadbinCIF\n"); exit(-1); }
    fclose(fp);

    icsp_params_t p;
    if (icsp_parse_header(bin.data(), bin.size(), &p) != ICSP_OK) { printf("this bin file is not icspCodec file\nerror in readHeader\n"); exit(-1); }
    // a frame needs at least 2 bits per block, so the file size bounds the frame count a header can justify
    if (nframes <= 0 || (uint64_t)nframes * (p.width / 16) * (p.height / 16) * 12 > (uint64_t)bin.size() * 8) { printf("error in readBlockData\n"); exit(-1); }
    if (atoi(argv[3]) != p.qp_dc || atoi(argv[4]) != p.qp_ac || atoi(argv[5]) != p.intra_period)
        fprintf(stderr, "[note] command-line QPDC/QPAC/intraPeriod differ from the header (%d %d %d); the header is used\n",
                p.qp_dc, p.qp_ac, p.intra_period);
    const int W = p.width, H = p.height;
    const size_t nmb = (size_t)(W / 16) * (H / 16), fsz = (size_t)W * H * 3 / 2;
    std::vector<int16_t> levels(nmb * 384 * nframes);
    std::vector<uint8_t> acflag(nmb * 6 * nframes), mpm(nmb * 4 * nframes), dec(fsz * nframes);
    std::vector<int8_t> mvd(nmb * 2 * nframes);
    int rc = icsp_parse_bitstream(bin.data(), bin.size(), nframes, levels.data(), acflag.data(), mpm.data(), mvd.data());
    if (rc) { printf("error in readBlockData\n"); exit(-1); }

    const clock_t t0 = clock();
    icsp_ctx_t* ctx = nullptr;
    rc = icsp_create(&ctx, &p, 0, nframes);
    if (rc) { printf("[ERROR] %s\n", icsp_strerror(rc)); exit(-1); }
    rc = icsp_upload_syntax(ctx, 0, nframes, levels.data(), mpm.data(), mvd.data());
    if (!rc) rc = icsp_decode_resident(ctx, 0, nframes);
    if (!rc) rc = icsp_download(ctx, 0, nframes, nullptr, nullptr, nullptr, nullptr, dec.data());
    if (rc) { printf("[ERROR] %s: %s\n", icsp_strerror(rc), icsp_last_error(ctx)); exit(-1); }
    icsp_destroy(ctx);
    const double detime = (double)(clock() - t0) / CLOCKS_PER_SEC;

    // checkResultFrames(..., INTRA|INTER, SAVE_YUV)
    // header period 0 (what the encoder writes for --intraPeriod 0) is all-intra too
    const char* outname = (p.intra_period <= 1) ? "check_test_intra_yuv.yuv" : "check_test_inter_yuv.yuv";
    FILE* out = fopen(outname, "wb");
    if (!out) { printf("fail to save yuv\n"); }
    else { fwrite(dec.data(), fsz, nframes, out); fclose(out); }

    // luma PSNR against the original (ICSP_Codec_Decoder.h:313-352)
    FILE* img = open_in("data", imgfname, nullptr);
    if (!img) { printf(" error in imgfp\n"); exit(-1); }
    std::vector<uint8_t> orig((size_t)W * H);
    double psnr = 0;
    for (int f = 0; f < nframes; f++) {
        if (fread(orig.data(), (size_t)W * H, 1, img) != 1) { printf(" error in imgfp\n"); exit(-1); }
        fseek(img, (long)(W / 2) * (H / 2) * 2, SEEK_CUR);
        double mse = 0;
        const uint8_t* d = dec.data() + f * fsz;
        for (int i = 0; i < W * H; i++) { const double v = (double)orig[i] - (double)d[i]; mse += v * v; }
        mse /= (double)(W * H);
        psnr += 20. * log10(255. / sqrt(mse));
    }
    fclose(img);
    psnr /= (double)nframes;
    FILE* txt = fopen("experimental_Result_Decoding.txt", "at");
    if (txt) {
        fprintf(txt, "decoding time: %.4lf(s) PSNR: %.4lf QPDC: %d  QPAC: %d Period: %d\n", detime, psnr, p.qp_dc, p.qp_ac, p.intra_period);
        fclose(txt);
    }
    return 0;
}
