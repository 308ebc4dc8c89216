// icsp_enc — host program with the reference encoder's command line, driving the HIP hot path through the C ABI.
//
// Mirrors, without copying, the host side of the reference (all under /root/reference/source/encoder/):
//   main                      encoder_main.cpp:4-24     option parsing, output prefix = input path up to the first '_'
//   parsing_command           ICSP_Codec_Encoder_source.cpp:94-165 (ENC)   -i -n -q --qpdc --qpac --intraPeriod --EnMultiThread -h --help
//   YCbCrLoad                 ENC:247-283               planar I420 frames read from the input file
//   single_thread_encoding    ENC:217-245               I/P decision per frame, "Encoding FRAME_%03d(%c) done!" lines
//   multi_thread_encoding     ENC:179-213, ICSP_thread.cpp:39-77   closed-GOP job queue -> here: closed-GOP shards, one host thread each
//   makebitstream             ENC:4849-4900             <prefix>_compCIF_<QDC>_<QAC>_<period>.bin
//   checkResultFrames         ENC:6376-6421             test_yuv.yuv (reconstruction) in the working directory
// What runs on the GPU is everything between loading the frames and writing the files, bit packing included
// (include/icsp_hip.h).
//
// Streaming layout (load -> encode -> write of the reference, ENC:247-283 / 217-245 / 6376-6421, as a pipeline): the clip is
// cut into shards of whole closed GOPs; every shard has a host thread and a context of its own (several per device), and
// does  pread -> pinned buffer -> H2D -> kernels -> device bit packer -> D2H (bits + reconstruction) -> pwrite  for its
// frames.  The shards run at different phases, so one's transfers and file I/O overlap with another's kernels; nothing is
// read or written twice, and the output bytes do not depend on the number of shards.  The per-shard bit strings are placed
// into the .bin image at their bit offsets by the shard threads (icsp_bitstream_place).
//
// Extensions use long options the reference rejects as unknown, so its own surface is unchanged:
//   --hostpack      sequential bit writer on the host instead of the device packer (same bytes; for cross-checks)
//   --gpus N        devices to shard over (default 1); --EnMultiThread N asks for N shards (the reference's N worker threads)
//   --streams S     shards (contexts + host threads) per device when --EnMultiThread is not given (default: one per 150 CIF
//                   frames' worth of macroblocks, at most 4)
//   --width W --height H   frame size (the reference hard-codes 352x288, encoder_main.cpp:20)
//   --stats         one more output line at the end: "[icsp_enc]{json}" with the wall-clock split (bench.py's e2e leg)
// Deliberate differences: the thread-pool mode also writes the .bin (the reference commented that call out,
// ICSP_thread.cpp:76); --intraPeriod 0 works with --EnMultiThread (the reference divides by zero, ICSP_thread.cpp:43);
// a trailing partial GOP is encoded (the reference never encodes the remainder frames).
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "icsp_hip.h"

namespace {

enum { SUCCESS = 0, UNENOUGH_PARAM, UNCORRECT_PARAM, FAIL_MEM_ALLOC };

struct Options {
    char yuv_fname[256];
    int total_frames, qp_dc, qp_ac, intra_period, multi_thread_mode, nthreads;
    int gpus, width, height, hostpack, streams, stats;
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
    printf("--streams : [MI355X build] shards (host thread + context) per GPU (default up to 4)\n");
    printf("--hostpack : [MI355X build] pack the bitstream on the host instead of on the device (same bytes)\n");
    printf("--width, --height : [MI355X build] frame size, multiples of 16 (default 352x288)\n");
    printf("--stats : [MI355X build] print a final line with the wall-clock split\n");
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
            else if (!strcmp(name, "streams")) cmd->streams = atoi(val);
            else if (!strcmp(name, "width")) cmd->width = atoi(val);
            else if (!strcmp(name, "height")) cmd->height = atoi(val);
            else if (!strcmp(name, "hostpack")) cmd->hostpack = 1;
            else if (!strcmp(name, "stats")) cmd->stats = 1;
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

struct Piece { uint64_t bits; std::vector<uint8_t> bytes; };      // the bit string of one chunk
struct Shard {
    int device, first, count, rc;
    std::string err;
    std::vector<Piece> pieces;                             // in frame order
    uint64_t bits;                                         // sum over the pieces
    uint8_t* body; size_t body_cap; bool body_own;         // pinned staging: one chunk's bit string,
    uint8_t* in; uint8_t* recon;                           //   frames and reconstruction
    icsp_ctx_t* ctx;
    double t_setup, t_read, t_gpu, t_write;
};

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool pread_all(int fd, uint8_t* dst, size_t n, off_t off)
{
    while (n) {
        const ssize_t k = pread(fd, dst, n, off);
        if (k <= 0) return false;
        dst += k; n -= (size_t)k; off += k;
    }
    return true;
}

bool pwrite_all(int fd, const uint8_t* src, size_t n, off_t off)
{
    while (n) {
        const ssize_t k = pwrite(fd, src, n, off);
        if (k <= 0) return false;
        src += k; n -= (size_t)k; off += k;
    }
    return true;
}

} // namespace

int main(int argc, char* argv[])
{
    const double t_start = now();
    Options opt;
    memset(&opt, 0, sizeof(opt));
    opt.total_frames = 1;          // README default; the reference leaves it uninitialised (ENC:84-91)
    opt.gpus = 0; opt.width = 352; opt.height = 288; opt.streams = 0;
    int ret = parsing_command(argc, argv, &opt);
    if (ret != SUCCESS) print_error_message(ret, "parsing_command");

    // output prefix = input path up to the first '_' (encoder_main.cpp:10-17); whole name if there is none
    std::string prefix(opt.yuv_fname);
    size_t us = prefix.find('_');
    if (us != std::string::npos) prefix.resize(us);

    const int W = opt.width, H = opt.height, n = opt.total_frames;
    const size_t fsz = (size_t)W * H * 3 / 2, nmb = (size_t)(W / 16) * (H / 16);
    if (n <= 0 || opt.qp_dc <= 0 || opt.qp_ac <= 0 || W % 16 || H % 16) print_error_message(UNCORRECT_PARAM, "parsing_command");

    // YCbCrLoad (ENC:247-283): the file must hold n frames; the shards read their own parts
    const int fd_in = open(opt.yuv_fname, O_RDONLY);
    struct stat st;
    if (fd_in < 0 || fstat(fd_in, &st) != 0 || (uint64_t)st.st_size < (uint64_t)fsz * n) { printf("fail to load cif.yuv\n error from YCbCrLoad\n"); exit(-1); }

    // closed-GOP shards, one host thread + one context each (the analogue of encoding_thread, ENC:186-213).
    // --EnMultiThread N: N shards; otherwise --gpus G (default 1) devices x --streams S (default 4) shards per device.  With
    // fewer devices than asked for, shards share devices round-robin, so every option still works on a one-GPU box.
    const int L = opt.intra_period > 0 ? opt.intra_period : 1;
    const int ngop = (n + L - 1) / L;
    const int ndev_seen = icsp_device_count();                      // starts the HIP runtime
    const int ndev = ndev_seen > 0 ? ndev_seen : 1;
    int nshard;
    if (opt.multi_thread_mode > 0) nshard = opt.nthreads;
    else if (opt.gpus > 0 && opt.streams <= 0) nshard = opt.gpus;   // --gpus N alone: N shards, as before
    else {
        // setting a context up costs about as much as encoding 150 CIF frames (streams, pinned memory), so short clips get few
        int per_dev = opt.streams > 0 ? opt.streams : std::min(4, std::max(1, (int)((uint64_t)n * nmb / (150 * 396))));
        nshard = (opt.gpus > 0 ? opt.gpus : 1) * per_dev;
    }
    if (nshard > ngop) nshard = ngop;
    if (nshard > 64) nshard = 64;
    if (nshard < 1) nshard = 1;
    std::vector<Shard> shards(nshard);
    int g0 = 0;
    for (int d = 0; d < nshard; d++) {
        const int gcount = ngop / nshard + (d < ngop % nshard ? 1 : 0);
        Shard& s = shards[d];
        s.device = d % ndev; s.first = g0 * L;
        s.count = std::min(n, (g0 + gcount) * L) - g0 * L;
        s.rc = 0; s.bits = 0; s.body = nullptr; s.body_cap = 0; s.in = nullptr; s.recon = nullptr; s.ctx = nullptr; s.body_own = false;
        s.t_setup = s.t_read = s.t_gpu = s.t_write = 0;
        g0 += gcount;
    }
    icsp_params_t params{ W, H, opt.qp_dc, opt.qp_ac, opt.intra_period };

    // checkResultFrames(..., SAVE_YUV) (ENC:6376-6413): every shard writes its frames at their place in the file
    const int fd_rec = open("test_yuv.yuv", O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd_rec >= 0 && ftruncate(fd_rec, (off_t)(fsz * n)) != 0) { /* pwrite extends the file anyway */ }

    // --hostpack: bring levels/flags/vectors back and run the sequential writer on the host (kept for cross-checking).
    std::vector<int16_t> levels;
    std::vector<uint8_t> acflag, mpm;
    std::vector<int8_t> mvd;
    if (opt.hostpack) { levels.resize(nmb * 384 * n); acflag.resize(nmb * 6 * n); mpm.resize(nmb * 4 * n); mvd.resize(nmb * 2 * n); }
    // A shard streams its frames through the device in chunks of whole GOPs (about 48 frames, 7 MB of CIF): the context's
    // frame store and the pinned staging buffers are sized for one chunk, whatever the length of the clip.
    int chunk = ((48 + L - 1) / L) * L;
    double t_init_done = 0;
    auto work = [&](Shard* s) {
        double t0 = now();
        const int cmax = std::min(chunk, s->count);
        s->rc = icsp_create(&s->ctx, &params, s->device, cmax);
        if (s->rc) { s->err = icsp_strerror(s->rc); return; }
        if (s == &shards[0]) t_init_done = now();                  // runtime up, code object loaded, first context built
        // one pinned allocation (each costs milliseconds) carved into frames | reconstruction | bit string.  A chunk's bit
        // string is almost always smaller than its frames; the worst case (icsp_bitstream_bound, 7x) is only allocated if the
        // packer reports that it does not fit
        const size_t cbytes = (fsz * cmax + 255) & ~(size_t)255;
        s->in = (uint8_t*)icsp_host_alloc(cbytes * (opt.hostpack ? 2 : 3));
        if (!s->in) { s->rc = ICSP_ERR_MEM_ALLOC; s->err = "pinned host memory"; return; }
        s->recon = s->in + cbytes;
        if (!opt.hostpack) { s->body_cap = cbytes; s->body = s->recon + cbytes; s->body_own = false; }
        s->t_setup = now() - t0;
        for (int c0 = 0; c0 < s->count; c0 += cmax) {
            const int cn = std::min(cmax, s->count - c0);
            const size_t f = (size_t)s->first + c0, bytes = fsz * cn;
            t0 = now();
            if (!pread_all(fd_in, s->in, bytes, (off_t)(f * fsz))) { s->rc = ICSP_ERR_RANGE; s->err = "short read"; return; }
            s->t_read += now() - t0; t0 = now();
            if (opt.hostpack) {
                s->rc = icsp_encode_gop(s->ctx, s->in, cn, levels.data() + f * nmb * 384, acflag.data() + f * nmb * 6,
                                        mpm.data() + f * nmb * 4, mvd.data() + f * nmb * 2, s->recon);
            } else {
                uint64_t bits = 0;
                s->rc = icsp_upload(s->ctx, s->in, 0, cn);
                if (!s->rc) s->rc = icsp_encode_resident(s->ctx, 0, cn);
                if (!s->rc) s->rc = icsp_pack_bits(s->ctx, 0, cn, s->body, s->body_cap, &bits);
                if (s->rc == ICSP_ERR_RANGE) {
                    s->body_own = true;
                    s->body_cap = icsp_bitstream_bound(&params, cmax);
                    s->body = (uint8_t*)icsp_host_alloc(s->body_cap);
                    s->rc = s->body ? icsp_pack_bits(s->ctx, 0, cn, s->body, s->body_cap, &bits) : ICSP_ERR_MEM_ALLOC;
                }
                if (!s->rc) s->rc = icsp_download(s->ctx, 0, cn, nullptr, nullptr, nullptr, nullptr, s->recon);
                if (!s->rc) {
                    s->pieces.emplace_back();
                    s->pieces.back().bits = bits;
                    s->pieces.back().bytes.assign(s->body, s->body + (size_t)((bits + 7) / 8));
                    s->bits += bits;
                }
            }
            if (s->rc) { s->err = std::string(icsp_strerror(s->rc)) + ": " + icsp_last_error(s->ctx); return; }
            s->t_gpu += now() - t0; t0 = now();
            if (fd_rec >= 0) pwrite_all(fd_rec, s->recon, bytes, (off_t)(f * fsz));
            s->t_write += now() - t0;
        }
    };
    {
        std::vector<std::thread> th;
        for (int d = 1; d < nshard; d++) th.emplace_back(work, &shards[d]);
        work(&shards[0]);
        for (auto& t : th) t.join();
    }
    for (auto& s : shards)
        if (s.rc) { printf("[ERROR] GPU %d: %s\n", s.device, s.err.c_str()); exit(-1); }
    const double t_encoded = now();

    for (int f = 0; f < n; f++)                                            // print_frame_end_message (ENC:44-48)
        printf("Encoding FRAME_%03d(%c) done!\n", f, (opt.intra_period == 0 || f % opt.intra_period == 0) ? 'I' : 'P');

    // makebitstream (ENC:4849-4900)
    size_t nbytes = 0;
    std::vector<uint8_t> bs;
    int rc;
    if (opt.hostpack) {
        bs.resize(icsp_bitstream_bound(&params, n) + 2);
        rc = icsp_write_bitstream(&params, n, levels.data(), acflag.data(), mpm.data(), mvd.data(), bs.data(), bs.size(), &nbytes);
    } else {
        uint64_t total = 0;
        std::vector<uint64_t> at(nshard);
        for (int d = 0; d < nshard; d++) { at[d] = total; total += shards[d].bits; }
        bs.resize(14 + (size_t)(total / 8) + 3);
        rc = icsp_bitstream_begin(&params, total, bs.data(), bs.size(), &nbytes);
        if (!rc) {
            std::vector<std::thread> th;
            auto place = [&](int d) {
                uint64_t o = at[d];
                for (auto& pc : shards[d].pieces) {
                    if (int r = icsp_bitstream_place(bs.data(), bs.size(), o, pc.bytes.data(), pc.bits)) shards[d].rc = r;
                    o += pc.bits;
                }
            };
            for (int d = 1; d < nshard; d++) th.emplace_back(place, d);
            place(0);
            for (auto& t : th) t.join();
            for (auto& s : shards) if (s.rc) rc = s.rc;
        }
        if (!rc) rc = icsp_bitstream_end(bs.data(), total);
    }
    if (rc) { printf("[ERROR] %s in makebitstream\n", icsp_strerror(rc)); exit(-1); }
    char name[512];
    snprintf(name, sizeof(name), "%s_compCIF_%d_%d_%d.bin", prefix.c_str(), opt.qp_dc, opt.qp_ac, opt.intra_period);
    FILE* out = fopen(name, "wb");
    if (!out) { printf("fail to open compCIF.bin\n"); exit(-1); }
    fwrite(bs.data(), nbytes, 1, out);
    fclose(out);
    if (fd_rec < 0) printf("fail to save yuv\n");
    else close(fd_rec);
    close(fd_in);
    const double t_files = now();
    if (opt.stats) {
        double su = 0, rd = 0, gp = 0, wr = 0;
        for (auto& s : shards) { su = std::max(su, s.t_setup); rd = std::max(rd, s.t_read); gp = std::max(gp, s.t_gpu); wr = std::max(wr, s.t_write); }
        printf("[icsp_enc]{\"frames\": %d, \"shards\": %d, \"devices\": %d, \"chunk_frames\": %d, \"init_s\": %.4f, \"encode_s\": %.4f, "
               "\"bitstream_and_files_s\": %.4f, \"max_shard_setup_s\": %.4f, \"max_shard_read_s\": %.4f, \"max_shard_gpu_s\": %.4f, "
               "\"max_shard_write_s\": %.4f, \"e2e_fps_excl_init\": %.1f, \"e2e_fps_incl_init\": %.1f}\n",
               n, nshard, std::min(ndev, nshard), chunk, t_init_done - t_start, t_encoded - t_init_done, t_files - t_encoded, su, rd, gp, wr,
               n / (t_files - t_init_done), n / (t_files - t_start));
    }
    // contexts and pinned buffers go last: tearing the runtime down is not part of producing the files
    for (auto& s : shards) { icsp_host_free(s.in); if (s.body_own) icsp_host_free(s.body); icsp_destroy(s.ctx); }
    return 0;
}
