// icsp_enc — host program with the reference encoder's command line, driving the HIP hot path through the C ABI.
//
// Mirrors, without copying, the host side of the reference (all under /root/reference/source/encoder/):
//   main                      encoder_main.cpp:4-24     option parsing, output prefix = input path up to the first '_'
//   parsing_command           ICSP_Codec_Encoder_source.cpp:94-165 (ENC)   -i -n -q --qpdc --qpac --intraPeriod --EnMultiThread -h --help
//   YCbCrLoad                 ENC:247-283               planar I420 frames read with fread
//   single_thread_encoding    ENC:217-245               I/P decision per frame, "Encoding FRAME_%03d(%c) done!" lines
//   multi_thread_encoding     ENC:179-213, ICSP_thread.cpp:39-77   closed-GOP job queue -> here: closed-GOP shards, one host thread per GPU
//   makebitstream             ENC:4849-4900             <prefix>_compCIF_<QDC>_<QAC>_<period>.bin
//   checkResultFrames         ENC:6376-6421             test_yuv.yuv (reconstruction) in the working directory
// What runs on the GPU is everything between loading the frames and writing the files, bit packing included
// (include/icsp_hip.h); the host concatenates the per-device bit strings and adds the header.
//
// Extensions use long options the reference rejects as unknown, so its own surface is unchanged:
//   --hostpack      sequential bit writer on the host instead of the device packer (same bytes; for cross-checks)
//   --gpus N        shard closed GOPs over N devices (default 1; --EnMultiThread N means the same here, N host threads = N GPUs)
//   --width W --height H   frame size (the reference hard-codes 352x288, encoder_main.cpp:20)
// Deliberate differences: the thread-pool mode also writes the .bin (the reference commented that call out,
// ICSP_thread.cpp:76); --intraPeriod 0 works with --EnMultiThread (the reference divides by zero, ICSP_thread.cpp:43);
// a trailing partial GOP is encoded (the reference never encodes the remainder frames).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>
#include "icsp_hip.h"

namespace {

enum { SUCCESS = 0, UNENOUGH_PARAM, UNCORRECT_PARAM, FAIL_MEM_ALLOC };

struct Options {
    char yuv_fname[256];
    int total_frames, qp_dc, qp_ac, intra_period, multi_thread_mode, nthreads;
    int gpus, width, height, hostpack;
};

void print_help_message()
{
    printf("usage: ./ICSPCodec [option] [values]\n");
    printf("-i : input yuv sequence\n");
    printf("-w : width\n");
    printf("-h : height\n");
    printf("-n : the number of frames(default is 1)\n");
    printf("-q : QP of DC and AC (16, 8, or 1)\n");
    printf("-h : help message\n");
    printf("--help : help message\n");
    printf("--qpdc : QP of DC (16, 8, or 1)\n");
    printf("--qpac : QP of AC (16, 8, or 1)\n");
    printf("--intraPeriod: period of intra frame(0: All intra)\n");
    printf("--EnMultiThread: enable multi threading mode, also the number of thread(0~4, 0 is disable)\n");
    printf("--gpus : [MI355X build] number of GPUs to shard closed GOPs over (default 1)\n");
    printf("--hostpack : [MI355X build] pack the bitstream on the host instead of on the device (same bytes)\n");
    printf("--width, --height : [MI355X build] frame size, multiples of 16 (default 352x288)\n");
}

[[noreturn]] void print_error_message(int err_type, const char* func_name)
{
    switch (err_type) {
    case UNENOUGH_PARAM: printf("[ERROR] unenough parameters in %s\n", func_name); break;
    case UNCORRECT_PARAM: printf("[ERROR] uncorrect parameters in %s\n", func_name); break;
    case FAIL_MEM_ALLOC: printf("[ERROR] fail memory allocation in %s\n", func_name); break;
    default: printf("[ERROR] unknown reason\n");
    }
    exit(-1);
}

// Same scan as the reference: every token starting with '-' is an option, its value is argv[i+1] and is NOT skipped
// (so a value that itself starts with '-' is parsed as an option); unknown options are an error.
int parsing_command(int argc, char* argv[], Options* cmd)
{
    if (argc < 2) return UNENOUGH_PARAM;
    for (int i = 1; i < argc; i++) {
        const char* o = argv[i];
        const char* val = (i + 1 < argc) ? argv[i + 1] : "0";
        if (o[0] != '-') continue;
        if (o[1] == '-') {
            const char* name = o + 2;
            if (!strcmp(name, "qpdc")) cmd->qp_dc = atoi(val);
            else if (!strcmp(name, "qpac")) cmd->qp_ac = atoi(val);
            else if (!strcmp(name, "intraPeriod")) cmd->intra_period = atoi(val);
            else if (!strcmp(name, "EnMultiThread")) { cmd->multi_thread_mode = atoi(val); cmd->nthreads = cmd->multi_thread_mode; }
            else if (!strcmp(name, "help")) { print_help_message(); exit(0); }
            else if (!strcmp(name, "gpus")) cmd->gpus = atoi(val);
            else if (!strcmp(name, "width")) cmd->width = atoi(val);
            else if (!strcmp(name, "height")) cmd->height = atoi(val);
            else if (!strcmp(name, "hostpack")) cmd->hostpack = 1;
            else return UNCORRECT_PARAM;
        } else {
            if (o[1] == 'i') { strncpy(cmd->yuv_fname, val, 255); cmd->yuv_fname[255] = 0; }
            else if (o[1] == 'n') cmd->total_frames = atoi(val);
            else if (o[1] == 'q') { cmd->qp_ac = atoi(val); cmd->qp_dc = atoi(val); }
            else if (o[1] == 'h') { print_help_message(); exit(0); }
            else return UNCORRECT_PARAM;
        }
    }
    return SUCCESS;
}

struct Shard { int device, first, count, rc; std::string err; std::vector<uint8_t> body; uint64_t bits; };

} // namespace

int main(int argc, char* argv[])
{
    Options opt;
    memset(&opt, 0, sizeof(opt));
    opt.total_frames = 1;          // README default; the reference leaves it uninitialised (ENC:84-91)
    opt.gpus = 0; opt.width = 352; opt.height = 288;
    int ret = parsing_command(argc, argv, &opt);
    if (ret != SUCCESS) print_error_message(ret, "parsing_command");

    // output prefix = input path up to the first '_' (encoder_main.cpp:10-17); whole name if there is none
    std::string prefix(opt.yuv_fname);
    size_t us = prefix.find('_');
    if (us != std::string::npos) prefix.resize(us);

    const int W = opt.width, H = opt.height, n = opt.total_frames;
    const size_t fsz = (size_t)W * H * 3 / 2, nmb = (size_t)(W / 16) * (H / 16);
    if (n <= 0 || opt.qp_dc <= 0 || opt.qp_ac <= 0 || W % 16 || H % 16) print_error_message(UNCORRECT_PARAM, "parsing_command");

    // YCbCrLoad (ENC:247-283)
    FILE* fp = fopen(opt.yuv_fname, "rb");
    if (!fp) { printf("fail to load cif.yuv\n error from YCbCrLoad\n"); exit(-1); }
    std::vector<uint8_t> yuv(fsz * n);
    size_t got = fread(yuv.data(), fsz, n, fp);
    fclose(fp);
    if ((int)got != n) { printf("fail to load cif.yuv\n error from YCbCrLoad\n"); exit(-1); }

    // closed-GOP shards, one host thread + one context per GPU (the analogue of encoding_thread, ENC:186-213)
    int ngpu = opt.gpus > 0 ? opt.gpus : (opt.multi_thread_mode > 0 ? opt.nthreads : 1);
    const int L = opt.intra_period > 0 ? opt.intra_period : 1;
    const int ngop = (n + L - 1) / L;
    if (ngpu > ngop) ngpu = ngop;
    // --EnMultiThread N / --gpus N ask for N shards; with fewer devices than that, shards share devices round-robin (each
    // shard is its own context and host thread), so the reference's thread option still works on a one-GPU box
    const int ndev = icsp_device_count() > 0 ? icsp_device_count() : 1;
    std::vector<Shard> shards(ngpu);
    int g0 = 0;
    for (int d = 0; d < ngpu; d++) {
        int gcount = ngop / ngpu + (d < ngop % ngpu ? 1 : 0);
        shards[d].device = d % ndev; shards[d].first = g0 * L;
        shards[d].count = std::min(n, (g0 + gcount) * L) - g0 * L;
        shards[d].rc = 0; shards[d].bits = 0;
        g0 += gcount;
    }
    icsp_params_t params{ W, H, opt.qp_dc, opt.qp_ac, opt.intra_period };
    std::vector<uint8_t> recon(fsz * n);
    // --hostpack: bring levels/flags/vectors back and run the sequential writer on the host (the round-1 path, kept for
    // cross-checking).  Default: each device packs the bits of its shard, only bits + reconstruction cross PCIe.
    std::vector<int16_t> levels;
    std::vector<uint8_t> acflag, mpm;
    std::vector<int8_t> mvd;
    if (opt.hostpack) { levels.resize(nmb * 384 * n); acflag.resize(nmb * 6 * n); mpm.resize(nmb * 4 * n); mvd.resize(nmb * 2 * n); }
    auto work = [&](Shard* s) {
        icsp_ctx_t* ctx = nullptr;
        s->rc = icsp_create(&ctx, &params, s->device, s->count);
        if (s->rc) { s->err = icsp_strerror(s->rc); return; }
        const size_t f = s->first;
        if (opt.hostpack) {
            s->rc = icsp_encode_gop(ctx, yuv.data() + f * fsz, s->count, levels.data() + f * nmb * 384, acflag.data() + f * nmb * 6,
                                    mpm.data() + f * nmb * 4, mvd.data() + f * nmb * 2, recon.data() + f * fsz);
        } else {
            s->body.resize(icsp_bitstream_bound(&params, s->count));
            s->rc = icsp_upload(ctx, yuv.data() + f * fsz, 0, s->count);
            if (!s->rc) s->rc = icsp_encode_resident(ctx, 0, s->count);
            if (!s->rc) s->rc = icsp_pack_bits(ctx, 0, s->count, s->body.data(), s->body.size(), &s->bits);
            if (!s->rc) s->rc = icsp_download(ctx, 0, s->count, nullptr, nullptr, nullptr, nullptr, recon.data() + f * fsz);
        }
        if (s->rc) s->err = std::string(icsp_strerror(s->rc)) + ": " + icsp_last_error(ctx);
        icsp_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (int d = 1; d < ngpu; d++) th.emplace_back(work, &shards[d]);
    work(&shards[0]);
    for (auto& t : th) t.join();
    for (auto& s : shards)
        if (s.rc) { printf("[ERROR] GPU %d: %s\n", s.device, s.err.c_str()); exit(-1); }

    for (int f = 0; f < n; f++)                                            // print_frame_end_message (ENC:44-48)
        printf("Encoding FRAME_%03d(%c) done!\n", f, (opt.intra_period == 0 || f % opt.intra_period == 0) ? 'I' : 'P');

    // makebitstream (ENC:4849-4900)
    size_t cap = icsp_bitstream_bound(&params, n) + 2, nbytes = 0;
    std::vector<uint8_t> bs(cap);
    int rc;
    if (opt.hostpack) {
        rc = icsp_write_bitstream(&params, n, levels.data(), acflag.data(), mpm.data(), mvd.data(), bs.data(), cap, &nbytes);
    } else {
        std::vector<const uint8_t*> pieces;
        std::vector<uint64_t> pbits;
        for (auto& s : shards) { pieces.push_back(s.body.data()); pbits.push_back(s.bits); }
        rc = icsp_bitstream_assemble(&params, (int)pieces.size(), pieces.data(), pbits.data(), bs.data(), cap, &nbytes);
    }
    if (rc) { printf("[ERROR] %s in makebitstream\n", icsp_strerror(rc)); exit(-1); }
    char name[512];
    snprintf(name, sizeof(name), "%s_compCIF_%d_%d_%d.bin", prefix.c_str(), opt.qp_dc, opt.qp_ac, opt.intra_period);
    FILE* out = fopen(name, "wb");
    if (!out) { printf("fail to open compCIF.bin\n"); exit(-1); }
    fwrite(bs.data(), nbytes, 1, out);
    fclose(out);

    // checkResultFrames(..., SAVE_YUV) (ENC:6376-6413)
    FILE* ry = fopen("test_yuv.yuv", "wb");
    if (!ry) { printf("fail to save yuv\n"); return 0; }
    fwrite(recon.data(), fsz, n, ry);
    fclose(ry);
    return 0;
}
