// icsp_enc — host program with the reference encoder's command line, driving the HIP hot path through the C ABI.
//
// Mirrors, without copying, the host side of the reference (all under /root/reference/source/encoder/):
//   main                      encoder_main.cpp:4-24     option parsing, output prefix = input path up to the first '_'
//   parsing_command           ICSP_Codec_Encoder_source.cpp:94-165 (ENC)   -i -n -q --qpdc --qpac --intraPeriod --EnMultiThread -h --help
//   YCbCrLoad                 ENC:247-283               planar I420 frames read from the input file
//   single_thread_encoding    ENC:217-245               I/P decision per frame, "Encoding FRAME_%03d(%c) done!" lines
//   multi_thread_encoding     ENC:179-213, ICSP_thread.cpp:39-77   closed-GOP job queue -> here: a queue of closed-GOP chunks, N worker threads
//   makebitstream             ENC:4849-4900             <prefix>_compCIF_<QDC>_<QAC>_<period>.bin
//   checkResultFrames         ENC:6376-6421             test_yuv.yuv (reconstruction) in the working directory
// What runs on the GPU is everything between loading the frames and writing the files, bit packing included
// (include/icsp_hip.h).
//
// Streaming layout (load -> encode -> write of the reference, ENC:247-283 / 217-245 / 6376-6421, as a pipeline): the clip is
// a queue of chunks of whole closed GOPs; workers -- a host thread and a context each -- take chunks in order and move them
// through the device:  H2D -> kernels -> device bit packer -> D2H (bits + reconstruction).  The workers are at different
// phases, so one's transfers overlap with another's kernels; all uploads of a device are made by one uploader thread on one
// stream of the device, all downloads run on another (icsp_copy_streams), and a worker's next chunk goes up while its
// current one is packed and downloaded.
//   * Mapped mode (default): the input file and test_yuv.yuv are mmap'ed and the mappings pinned (icsp_host_register), so the
//     uploads read the page cache and the downloads write it by DMA -- the host copies nothing.  A helper thread maps the
//     files and allocates the output's pages (MAP_POPULATE: the slowest host step, ~6 GB/s on tmpfs whatever the thread
//     count) while the HIP runtime starts.
//   * Staged mode (--staged, clips above the mapping cap, or when mapping / pinning is refused): pread -> pinned buffer ->
//     device and device -> pinned buffer -> pwrite, per chunk.
// Nothing is read or written twice and the output bytes do not depend on the mode, the chunk size or the number of workers.
// The chunks' bit strings are placed into the (mmap'ed) .bin at their bit offsets by several threads (icsp_bitstream_place).
//
// Extensions use long options the reference rejects as unknown, so its own surface is unchanged:
//   --hostpack      sequential bit writer on the host instead of the device packer (same bytes; for cross-checks)
//   --gpus N        devices to spread the workers over (default 1); --EnMultiThread N asks for N workers (the reference's N threads)
//   --streams S     workers (contexts + host threads) per device when --EnMultiThread is not given (default: one per
//                   chunk, at most 2)
//   --chunk F       frames per chunk (rounded up to whole GOPs; default: 512 CIF frames' worth of macroblocks)
//   --staged        staging buffers instead of pinned file mappings
//   --nopin         the files are mapped but not pinned, as when the runtime refuses the ranges (tests the fall-back: staging
//                   buffers for frames and reconstruction, host placement of the bits into the unpinned .bin mapping)
//   --binest B      bytes of the .bin mapping to populate and pin (default 30 % of the input + 1 MB); tests use it to push
//                   chunks onto the host-placement path
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
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "icsp_hip.h"

namespace {

enum { SUCCESS = 0, UNENOUGH_PARAM, UNCORRECT_PARAM, FAIL_MEM_ALLOC };

struct Options {
    char yuv_fname[256];
    int total_frames, qp_dc, qp_ac, intra_period, multi_thread_mode, nthreads;
    int gpus, width, height, hostpack, streams, stats, staged, chunk, nopin;
    long long binest;
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
    printf("--streams : [MI355X build] workers (host thread + context) per GPU (default up to 2)\n");
    printf("--chunk : [MI355X build] frames per chunk moved through the device (default 512 CIF frames' worth)\n");
    printf("--staged : [MI355X build] staging buffers instead of pinned mappings of the input and output files\n");
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
            else if (!strcmp(name, "staged")) cmd->staged = 1;
            else if (!strcmp(name, "nopin")) cmd->nopin = 1;
            else if (!strcmp(name, "chunk")) cmd->chunk = atoi(val);
            else if (!strcmp(name, "binest")) cmd->binest = atoll(val);
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

// A chunk: whole GOPs that go through the device together.  Its bit string lands either straight in the .bin mapping
// (direct) or in `bytes`, to be placed at the end.
struct Chunk {
    int first, count;
    uint64_t bits, at;                                     // length and bit offset in the body (known when its turn has come)
    bool direct;
    std::vector<uint8_t> bytes;
};
// A worker: host thread + context (+ pinned buffers) that takes chunks from the queue.  --EnMultiThread N = N workers.
struct Worker {
    int device, rc, chunks;
    std::string err;
    icsp_ctx_t* ctx;
    uint8_t* body; size_t body_cap;                        // pinned: one chunk's bit string, when it cannot go direct
    uint8_t* stage;                                        // pinned staging for frames / reconstruction (staged mode only)
    uint8_t* stage_in;                                     //   its input half (null when the input file is mapped and pinned)
    int up_chunk, up_rc; bool up_done;                     // the upload this worker has handed to its device's uploader thread
    double t_setup, t_read, t_write, t_up, t_enc, t_count, t_turn, t_pack, t_down;
    double t_create, t_prepare, t_copystreams, t_mapwait, t_stage, t_warm;     // the parts of t_setup
    int node, bound;                                       // NUMA node of the worker's device (-1 unknown), whether the thread was bound to it
};

// One-shot gate: the helper thread opens it when the file mappings are settled, the main thread when the runtime is up.
struct Gate {
    std::mutex m; std::condition_variable cv; bool open = false;
    void set() { { std::lock_guard<std::mutex> l(m); open = true; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return open; }); }
};
// All workers start streaming together once every context and buffer exists: "init" ends when the last one arrives.
struct Barrier {
    std::mutex m; std::condition_variable cv; int waiting = 0, total = 0; double t_last = 0;
};
// The body's bit cursor: chunk c learns its offset when every chunk before it has reported its length.
struct Cursor {
    std::mutex m; std::condition_variable cv; int turn = 0; uint64_t bits = 0; bool failed = false;
};

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

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

// On a box whose page cache does not hold the HIP runtime yet (a fresh node: the first run of this program there) the runtime's
// first stream creation took 162 ms instead of 25: page faults into libamdhip64.so from a cold disk, one small read after the
// other (tools/cold_cache.sh: with the libraries read beforehand the same first run takes 26 ms; reading them with `cat` takes
// 0.4 s, so it is latency, not volume).  This asks the kernel to read the runtime's libraries ahead -- whole files, many requests
// in flight -- from a thread of its own while main() parses the command line and hsa_init runs.  Costs nothing when they are cached.
void readahead_runtime_libraries()
{
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return;
    char line[1024];
    std::vector<std::string> seen;
    while (fgets(line, sizeof(line), f)) {
        const char* path = strchr(line, '/');
        if (!path) continue;
        if (!strstr(path, "libamdhip64") && !strstr(path, "libhsa-runtime64") && !strstr(path, "libicsp_hip") && !strstr(path, "librocprofiler-register")) continue;
        std::string ps(path);
        while (!ps.empty() && (ps.back() == '\n' || ps.back() == ' ')) ps.pop_back();
        if (std::find(seen.begin(), seen.end(), ps) != seen.end()) continue;
        seen.push_back(ps);
        const int fd = open(ps.c_str(), O_RDONLY);
        if (fd < 0) continue;
        struct stat st;
        if (fstat(fd, &st) == 0) {
            // (WILLNEED only queues what fits the device's read-ahead window per call: walk the file in 2 MB steps)
            for (off_t o = 0; o < st.st_size; o += (off_t)2 << 20) (void)posix_fadvise(fd, o, (off_t)2 << 20, POSIX_FADV_WILLNEED);
        }
        close(fd);
    }
    fclose(f);
}

} // namespace

int main(int argc, char* argv[])
{
    const double t_start = now();
    if (!getenv("ICSP_NO_READAHEAD")) std::thread(readahead_runtime_libraries).detach();
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

    // YCbCrLoad (ENC:247-283): the file must hold n frames; the workers read their own parts
    const int fd_in = open(opt.yuv_fname, O_RDONLY);
    struct stat st;
    if (fd_in < 0 || fstat(fd_in, &st) != 0 || (uint64_t)st.st_size < (uint64_t)fsz * n) { printf("fail to load cif.yuv\n error from YCbCrLoad\n"); exit(-1); }
    // checkResultFrames(..., SAVE_YUV) (ENC:6376-6413): every chunk's frames are written at their place in the file
    const size_t total_bytes = fsz * (size_t)n;
    const int fd_rec = open("test_yuv.yuv", O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd_rec >= 0 && ftruncate(fd_rec, (off_t)total_bytes) != 0) { /* pwrite extends the file anyway */ }
    // makebitstream (ENC:4849-4900)
    char name[512];
    snprintf(name, sizeof(name), "%s_compCIF_%d_%d_%d.bin", prefix.c_str(), opt.qp_dc, opt.qp_ac, opt.intra_period);
    const int fd_bin = open(name, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd_bin < 0) { printf("fail to open compCIF.bin\n"); exit(-1); }
    icsp_params_t params{ W, H, opt.qp_dc, opt.qp_ac, opt.intra_period };

    // Mapped mode: a helper maps the three files while the runtime starts -- allocating the pages of test_yuv.yuv and of the
    // part of the .bin the stream is expected to need (the file is sized for the worst case, sparse) -- then pins them.
    const size_t kMapCap = (size_t)8 << 30;                         // input + output above this are streamed through staging buffers
    const bool want_map = !opt.staged && 2 * total_bytes <= kMapCap;
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    const size_t bin_cap = icsp_bitstream_bound(&params, n) + 2;    // worst case, header included
    // (the part populated and pinned: 30 % of the input + 1 MB for all-intra, 24 % with P frames -- what QP >= 8 needs on
    //  camera-like content, and every populated page the stream does not use costs time twice, allocated and freed again;
    //  strings that end beyond it come back through a pinned buffer and are placed by the host.  --binest BYTES overrides)
    const size_t bin_guess = opt.binest > 0 ? (size_t)opt.binest
                                            : 14 + total_bytes * (opt.intra_period > 1 ? 24 : 30) / 100 + ((size_t)1 << 20);
    const size_t bin_est = std::min(bin_cap, (bin_guess + page - 1) / page * page);
    uint8_t* in_map = nullptr; uint8_t* out_map = nullptr;           // non-null only when mapped AND pinned
    uint8_t* bin_map = nullptr;                                      // the .bin mapping (bin_cap bytes), pinned or not
    bool bin_pinned = false;                                         //   its first bin_est bytes are populated and pinned
    double t_map = 0, t_pin = 0;
    Gate hip_up, maps_settled, plan_ready;
    // Placement (csrc/icsp_topology.cpp): on a host with several NUMA nodes a device's threads are bound to the node the device
    // hangs off, and the pages of test_yuv.yuv that its chunks will be written into are allocated from a thread bound there
    // (first touch) instead of all at once by MAP_POPULATE.  One node (every box this has run on so far): nothing changes.
    const bool placement = icsp_numa_nodes() >= 2 && !getenv("ICSP_NO_PLACEMENT");
    std::vector<int> dev_node;                                       // per device, filled once the runtime is up (before plan_ready)
    std::vector<Chunk> chunks;
    int ndev_used = 1;
    size_t placed_bytes = 0;
    std::thread helper([&] {
        if (want_map) {
            const double t0 = now();
            void* out_raw = MAP_FAILED;
            if (fd_rec >= 0) out_raw = mmap(nullptr, total_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | (placement ? 0 : MAP_POPULATE), fd_rec, 0);
            void* in_raw = mmap(nullptr, total_bytes, PROT_READ, MAP_SHARED | MAP_POPULATE, fd_in, 0);
            void* bin_raw = MAP_FAILED;
            if (ftruncate(fd_bin, (off_t)bin_cap) == 0) bin_raw = mmap(nullptr, bin_cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd_bin, 0);
            if (bin_raw != MAP_FAILED && madvise(bin_raw, bin_est, MADV_POPULATE_WRITE) != 0)
                for (size_t o = 0; o < bin_est; o += page) ((volatile uint8_t*)bin_raw)[o] = 0;      // (kernels before 5.14)
            t_map = now() - t0;
            hip_up.wait();
            if (placement && out_raw != MAP_FAILED) {
                // the chunk table and the devices' nodes are known now: one thread per device, bound to the device's node,
                // touches the output pages of the device's chunks
                plan_ready.wait();
                const double tp = now();
                std::vector<std::thread> th;
                std::atomic<size_t> placed(0);
                for (int d = 0; d < ndev_used; d++)
                    th.emplace_back([&, d] {
                        (void)icsp_bind_thread_to_node(dev_node[d], nullptr);
                        for (size_t c = 0; c < chunks.size(); c++)
                            if (icsp_chunk_device((int)c, ndev_used) == d &&
                                icsp_populate_here((uint8_t*)out_raw + (size_t)chunks[c].first * fsz, (size_t)chunks[c].count * fsz) == ICSP_OK)
                                placed += (size_t)chunks[c].count * fsz;
                    });
                for (auto& t : th) t.join();
                placed_bytes = placed.load();
                t_map += now() - tp;
            }
            const double t1 = now();
            const bool pin = !opt.nopin;                          // --nopin: behave as if the runtime had refused every range
            if (pin && in_raw != MAP_FAILED && icsp_host_register(in_raw, total_bytes, 1) == ICSP_OK) in_map = (uint8_t*)in_raw;
            if (pin && out_raw != MAP_FAILED && icsp_host_register(out_raw, total_bytes, 0) == ICSP_OK) out_map = (uint8_t*)out_raw;
            if (bin_raw != MAP_FAILED) { bin_map = (uint8_t*)bin_raw; bin_pinned = pin && icsp_host_register(bin_raw, bin_est, 0) == ICSP_OK; }
            t_pin = now() - t1;
        }
        maps_settled.set();
    });

    // Closed-GOP chunks in a queue, taken by workers: one host thread + one context each (the analogue of encoding_thread,
    // ENC:186-213).  --EnMultiThread N: N workers; otherwise --gpus G (default 1) devices x --streams S workers per device.
    // With fewer devices than asked for, workers share devices round-robin, so every option still works on a one-GPU box.
    const int L = opt.intra_period > 0 ? opt.intra_period : 1;
    const int ngop = (n + L - 1) / L;
    // Short clips copy with the runtime's shader kernels instead of its DMA engines: an engine's queue costs about 6 ms when
    // a process first uses it, which engine a copy gets varies from run to run, and a clip that takes 2.4 ms after set-up
    // then takes 10 ms in two runs of three (measured; eight of eight at 2.4 ms this way).  Long clips keep the engines, which
    // move their data about 10 % faster beside the kernels.  (Decided before the runtime starts; never overrides the caller.)
    if ((uint64_t)n * nmb <= (uint64_t)1024 * 396) setenv("HSA_ENABLE_SDMA", "0", 0);
    const double t_hip0 = now();
    const int ndev_seen = icsp_device_count();                      // starts the HIP runtime
    const double t_hip = now() - t_hip0;
    hip_up.set();
    const int ndev = ndev_seen > 0 ? ndev_seen : 1;
    // a chunk: whole GOPs, by default as many frames as hold 512 CIF frames' macroblocks (at least one GOP)
    const int chunk_target = opt.chunk > 0 ? opt.chunk : std::max(1, (int)(512 * 396 / nmb));
    int chunk_gops = (chunk_target + L - 1) / L;
    int nworker;
    if (opt.multi_thread_mode > 0) nworker = opt.nthreads;
    else {
        // default: up to two workers per device, one per chunk on short clips (3000 CIF frames, transfers on the device's shared
        // up and down streams: 1 worker 115-119 k frames/s, 2 workers 170-185 k, 3, 4 and 6 the same with more set-up time)
        const int chunks_per_dev = (ngop + chunk_gops * std::max(1, opt.gpus) - 1) / (chunk_gops * std::max(1, opt.gpus));
        const int per_dev = opt.streams > 0 ? opt.streams : std::max(1, std::min(2, chunks_per_dev));
        nworker = (opt.gpus > 0 ? opt.gpus : 1) * per_dev;
    }
    nworker = std::max(1, std::min({ nworker, ngop, 64 }));
    // Several workers on a device already keep it busy from several streams; a second GOP-group stream in each context only
    // adds queues to share (3000 frames, 3 workers: 18.0 ms with one group each, 19-22 ms with two).
    // An all-intra batch in two parts only pays when several passes are in flight (a part then follows what its own stream
    // carries); every chunk here is encoded once and waited for.  Both go to every context through icsp_set_groups (an
    // environment variable set here, with the runtime's threads and the helper already running, would race with their getenv).
    const int p_groups = nworker > ndev && !getenv("ICSP_P_GROUPS") ? 1 : 0, i_groups = getenv("ICSP_I_GROUPS") ? 0 : 1;
    chunk_gops = std::max(1, std::min(chunk_gops, (ngop + nworker - 1) / nworker));     // every worker gets something to do
    const int chunk = chunk_gops * L;
    // The download side carries more than the upload side (reconstruction + bits) and sets the pace; it can start when the
    // first chunk has been uploaded, encoded and counted.  So with three chunks or more the first chunk of every device is a
    // quarter and the second a half of the rest (3000 CIF frames all-intra: 14.7 -> 14.3 ms).
    const bool ramp = ngop >= 3 * chunk_gops * ndev && chunk_gops >= 4;
    for (int f = 0, k = 0; f < n; k++) {
        int g = chunk_gops;
        if (ramp && k < ndev) g = chunk_gops / 4; else if (ramp && k < 2 * ndev) g = chunk_gops / 2;
        Chunk c; c.first = f; c.count = std::min(g * L, n - f); c.bits = c.at = 0; c.direct = false;
        f += c.count;
        chunks.push_back(std::move(c));
    }
    const int nchunks = (int)chunks.size();
    // Chunk c belongs to device c mod (devices in use): the devices advance through the clip together (a chunk's place in the
    // bitstream is known when every chunk before it has reported its length) and each device's ranges are known in advance.
    ndev_used = std::min(ndev, nworker);
    dev_node.assign(ndev_used, -1);
    if (placement) for (int d = 0; d < ndev_used; d++) dev_node[d] = icsp_device_numa_node(d);
    plan_ready.set();
    std::vector<Worker> workers(nworker);
    for (int d = 0; d < nworker; d++) {
        Worker& w = workers[d];
        w.device = d % ndev; w.rc = 0; w.chunks = 0; w.ctx = nullptr; w.body = nullptr; w.body_cap = 0; w.stage = nullptr; w.stage_in = nullptr; w.up_chunk = -1; w.up_rc = 0; w.up_done = true;
        w.t_setup = w.t_read = w.t_write = w.t_up = w.t_enc = w.t_count = w.t_turn = w.t_pack = w.t_down = 0;
        w.t_create = w.t_prepare = w.t_copystreams = w.t_mapwait = w.t_stage = w.t_warm = 0;
        w.node = -1; w.bound = 0;
    }

    // --hostpack: bring levels/flags/vectors back and run the sequential writer on the host (kept for cross-checking).
    std::vector<int16_t> levels;
    std::vector<uint8_t> acflag, mpm;
    std::vector<int8_t> mvd;
    if (opt.hostpack) { levels.resize(nmb * 384 * n); acflag.resize(nmb * 6 * n); mpm.resize(nmb * 4 * n); mvd.resize(nmb * 2 * n); }
    Barrier ready; ready.total = nworker;
    Cursor cursor;
    // per device: how many of ITS chunks have been taken; the k-th chunk of device d is chunk d + k * ndev_used
    std::vector<std::atomic<int>> next_on(ndev_used);
    for (auto& a : next_on) a.store(0);
    auto take_chunk = [&](int device) {
        const int d = device % ndev_used;
        const long long c = (long long)d + (long long)next_on[d].fetch_add(1) * ndev_used;
        return c < nchunks ? (int)c : nchunks;
    };
    // Several workers on a device: their uploads go through one stream of the device and their downloads through another
    // (icsp_copy_streams), so that the link carries both directions at once -- on their own streams all transfers of the device
    // end up on one DMA engine and run one at a time (3000 CIF frames: 18.6 ms = 1042 MB at the one-way rate; now 14-15 ms).
    // And they upload in turn (whole chunks, each at the full rate, the first worker encoding while the second uploads)
    // instead of all at once at a third of the rate each.
    const bool shared_copies = nworker > ndev;
    // One uploader thread per device makes all uploads of that device, one at a time (icsp_upload_sync): each runs at the
    // full rate, and none is submitted while the upload stream's DMA engine is busy -- a copy submitted then is given another
    // engine, whose first use costs milliseconds (3000 frames: 15 ms became 21-30 ms, varying from run to run).  A worker
    // hands over its NEXT chunk as soon as the current one is encoded (the frames are no longer needed; icsp_pack_count has
    // waited for the kernels), so the upload runs beside the packing and the downloads of the current chunk.
    struct Uploader { std::mutex m; std::condition_variable cv; std::deque<Worker*> q; bool stop = false; };
    std::vector<Uploader> uploaders(shared_copies ? ndev : 0);
    auto uploader_loop = [&](Uploader* u) {
        if (placement) (void)icsp_bind_thread_to_node(dev_node[(int)(u - uploaders.data()) % ndev_used], nullptr);
        for (;;) {
            Worker* w;
            {
                std::unique_lock<std::mutex> l(u->m);
                u->cv.wait(l, [&] { return u->stop || !u->q.empty(); });
                if (u->q.empty()) return;
                w = u->q.front(); u->q.pop_front();
            }
            const Chunk& ch = chunks[w->up_chunk];
            const size_t f = (size_t)ch.first, bytes = fsz * ch.count;
            int rc = 0;
            if (w->stage_in && !pread_all(fd_in, w->stage_in, bytes, (off_t)(f * fsz))) rc = ICSP_ERR_RANGE;
            if (!rc) rc = icsp_upload_sync(w->ctx, w->stage_in ? w->stage_in : in_map + f * fsz, 0, ch.count);
            { std::lock_guard<std::mutex> l(u->m); w->up_rc = rc; w->up_done = true; }
            u->cv.notify_all();
        }
    };
    // Contexts and the devices' transfer-stream pairs are made HERE, before any worker thread runs, one after the other PER DEVICE: a
    // stream (a hardware queue) takes 3 ms to create on an idle device and 12-55 ms while another thread's kernels or pinning calls
    // are in flight on it (every queue of the process is re-mapped; `tools/trace_copy_streams.sh`), which is what the workers'
    // concurrent set-up used to run into (3000 frames, two workers: create + copy streams 0.09-0.28 s -> see NEGATIVE_RESULTS.md).
    // Several devices: one creator thread per device, bound to the device's NUMA node first (the set-up would otherwise
    // grow linearly with the devices, and the contexts' host allocations would land on the main thread's node).
    {
        const int cmax = std::min(chunk, n);
        auto create_for = [&](int dev) {
            if (placement) { int bound = 0; (void)icsp_bind_thread_to_node(dev_node[dev % ndev_used], &bound); }
            bool pair_made = false;
            for (auto& w : workers) {
                if (w.device != dev) continue;
                double t0 = now();
                w.rc = icsp_create(&w.ctx, &params, w.device, cmax);
                if (!w.rc) w.rc = icsp_set_groups(w.ctx, p_groups, i_groups);
                // one chunk in all: the clip is encoded in a few milliseconds, a second stream takes longer than that to create
                if (!w.rc && nchunks == 1) w.rc = icsp_single_stream(w.ctx, 1);
                w.t_create = now() - t0; t0 = now();
                if (!w.rc && shared_copies && !pair_made) { w.rc = icsp_copy_streams(w.ctx, 1); pair_made = true; }
                w.t_copystreams = now() - t0;
            }
        };
        if (ndev_used <= 1) create_for(workers.empty() ? 0 : workers[0].device);
        else {
            std::vector<std::thread> creators;
            for (int d = 0; d < ndev_used; d++) creators.emplace_back(create_for, d);
            for (auto& t : creators) t.join();
        }
    }
    auto work = [&](Worker* w) {
        double t0 = now();
        if (placement) { w->node = dev_node[w->device % ndev_used]; (void)icsp_bind_thread_to_node(w->node, &w->bound); }
        const int cmax = std::min(chunk, n);
        double t1 = now();
        if (!w->rc) w->rc = icsp_prepare(w->ctx);
        w->t_prepare = now() - t1; t1 = now();
        if (!w->rc && shared_copies) w->rc = icsp_copy_streams(w->ctx, 1);          // (the device's pair exists: this only adopts it)
        w->t_copystreams += now() - t1; t1 = now();
        if (w->rc) w->err = std::string(icsp_strerror(w->rc)) + ": " + (w->ctx ? icsp_last_error(w->ctx) : "");
        maps_settled.wait();
        w->t_mapwait = now() - t1; t1 = now();
        // pinned staging (each allocation costs milliseconds) only for whichever side is not mapped
        const size_t cbytes = (fsz * cmax + 255) & ~(size_t)255;
        uint8_t* stage_in = nullptr; uint8_t* stage_out = nullptr;
        const size_t nstage = (in_map ? 0 : 1) + (out_map ? 0 : 1);
        if (!w->rc && nstage) {
            w->stage = (uint8_t*)icsp_host_alloc(cbytes * nstage);
            if (!w->stage) { w->rc = ICSP_ERR_MEM_ALLOC; w->err = "pinned host memory"; }
            stage_in = in_map ? nullptr : w->stage;
            w->stage_in = stage_in;
            stage_out = out_map ? nullptr : w->stage + (in_map ? 0 : cbytes);
        }
        // a chunk's bit string that cannot go straight into the .bin mapping comes back through a pinned buffer: almost
        // always far smaller than the frames; the worst case (icsp_bitstream_bound, 7x) only if the packer says so
        auto need_body = [&](size_t want) {
            if (w->body_cap >= want) return true;
            icsp_host_free(w->body);
            w->body = (uint8_t*)icsp_host_alloc(want);
            w->body_cap = w->body ? want : 0;
            return w->body != nullptr;
        };
        if (!w->rc && !opt.hostpack && !bin_pinned && !need_body(std::max<size_t>(cbytes / 2, (size_t)1 << 20))) { w->rc = ICSP_ERR_MEM_ALLOC; w->err = "pinned host memory"; }
        w->t_stage = now() - t1; t1 = now();
        // the first transfers from and into the newly pinned mappings can cost their hipMemcpyAsync calls milliseconds: spend
        // that now.  One frame read from the input mapping (into slot 0, which the first chunk overwrites); the slots' content
        // written into the output mapping; and, below, the black GOP's string into the .bin mapping, cleared again (the image
        // has to start out zeroed).  icsp_prepare has spent the stream's own first-transfer costs already.
        if (!w->rc) {
            // (whatever the slots hold goes to frames [0, cmax) of the output, which their chunks write again)
            if (in_map) (void)icsp_upload(w->ctx, in_map, 0, 1);
            if (out_map) (void)icsp_download(w->ctx, 0, cmax, nullptr, nullptr, nullptr, nullptr, out_map);
            (void)icsp_sync(w->ctx);
        }
        if (!w->rc && !opt.hostpack && bin_pinned && bin_est > 2 * page)
            (void)icsp_host_warm(w->ctx, bin_map + page, bin_est - 2 * page);     // zeros: the black GOP's string is too short for a DMA
        if (!w->rc && !opt.hostpack && bin_pinned) {
            uint64_t b = 0;
            if (icsp_pack_count(w->ctx, 0, std::min(L, cmax), &b) == ICSP_OK && 14 + (size_t)(b / 8) + 2 <= bin_est &&
                icsp_pack_into(w->ctx, 0, std::min(L, cmax), 0, bin_map + 14, bin_est - 14) == ICSP_OK)
                memset(bin_map + 14, 0, (size_t)(b / 8) + 1);
        }
        w->t_warm = now() - t1;
        w->t_setup = now() - t0 + w->t_create;
        {   // every worker arrives here, failed or not; the last arrival ends "init"
            std::unique_lock<std::mutex> l(ready.m);
            if (++ready.waiting == ready.total) { ready.t_last = now(); ready.cv.notify_all(); }
            else ready.cv.wait(l, [&] { return ready.waiting == ready.total; });
        }
        auto fail = [&](int rc, const char* what) {
            w->rc = rc; w->err = std::string(icsp_strerror(rc)) + ": " + (what ? what : icsp_last_error(w->ctx));
            { std::lock_guard<std::mutex> l(cursor.m); cursor.failed = true; }
            cursor.cv.notify_all();
        };
        if (w->rc) { fail(w->rc, w->err.c_str()); return; }
        // shared transfer streams: hand chunk c to the device's uploader / wait for it; else read (staged mode) + upload here
        auto post_upload = [&](int c) {
            Uploader& u = uploaders[w->device];
            { std::lock_guard<std::mutex> l(u.m); w->up_chunk = c; w->up_done = false; w->up_rc = 0; u.q.push_back(w); }
            u.cv.notify_all();
        };
        auto wait_upload = [&]() -> int {
            Uploader& u = uploaders[w->device];
            const double t = now();
            std::unique_lock<std::mutex> l(u.m);
            u.cv.wait(l, [&] { return w->up_done; });
            w->t_up += now() - t;
            return w->up_rc;
        };
        auto upload_here = [&](int c) -> int {
            const Chunk& ch = chunks[c];
            const size_t f = (size_t)ch.first, bytes = fsz * ch.count;
            double t = now();
            if (!in_map && !pread_all(fd_in, stage_in, bytes, (off_t)(f * fsz))) return ICSP_ERR_RANGE;
            w->t_read += now() - t; t = now();
            const int rc = icsp_upload(w->ctx, in_map ? in_map + f * fsz : stage_in, 0, ch.count);
            w->t_up += now() - t;
            return rc;
        };
        const bool early = shared_copies && !opt.hostpack;
        int c = take_chunk(w->device);
        if (c < nchunks && !opt.hostpack) {
            int rc;
            if (early) { post_upload(c); rc = wait_upload(); } else rc = upload_here(c);
            if (rc) { fail(rc, rc == ICSP_ERR_RANGE ? "short read" : nullptr); return; }
        }
        while (c < nchunks) {
            Chunk& ch = chunks[c];
            const int cn = ch.count;
            const size_t f = (size_t)ch.first, bytes = fsz * cn;
            uint8_t* rec = out_map ? out_map + f * fsz : stage_out;
            int rc, c2;
            if (opt.hostpack) {
                const uint8_t* src = in_map ? in_map + f * fsz : stage_in;
                t0 = now();
                if (!in_map && !pread_all(fd_in, stage_in, bytes, (off_t)(f * fsz))) { fail(ICSP_ERR_RANGE, "short read"); return; }
                w->t_read += now() - t0; t0 = now();
                rc = icsp_encode_gop(w->ctx, src, cn, levels.data() + f * nmb * 384, acflag.data() + f * nmb * 6,
                                     mpm.data() + f * nmb * 4, mvd.data() + f * nmb * 2, rec);
                w->t_enc += now() - t0;
                if (rc) { fail(rc, nullptr); return; }
                c2 = take_chunk(w->device);
            } else {
                t0 = now();
                rc = icsp_encode_resident(w->ctx, 0, cn);
                w->t_enc += now() - t0; t0 = now();
                uint64_t bits = 0;
                if (!rc) rc = icsp_pack_count(w->ctx, 0, cn, &bits);
                w->t_count += now() - t0;
                if (rc) { fail(rc, nullptr); return; }
                c2 = take_chunk(w->device);
                if (early && c2 < nchunks) post_upload(c2);
                t0 = now();
                {   // my turn: every earlier chunk has reported its length
                    std::unique_lock<std::mutex> l(cursor.m);
                    cursor.cv.wait(l, [&] { return cursor.turn == c || cursor.failed; });
                    if (cursor.failed) return;
                    ch.at = cursor.bits; ch.bits = bits;
                    cursor.bits += bits; cursor.turn++;
                }
                cursor.cv.notify_all();
                w->t_turn += now() - t0; t0 = now();
                const size_t end_byte = 14 + (size_t)((ch.at + bits + 7) / 8);
                ch.direct = bin_pinned && end_byte <= bin_est;
                if (ch.direct) rc = icsp_pack_into(w->ctx, 0, cn, ch.at, bin_map + 14, bin_est - 14);
                else {
                    const size_t nb = (size_t)((bits + 7) / 8);
                    if (!need_body(std::max<size_t>(nb, (size_t)1 << 20))) rc = ICSP_ERR_MEM_ALLOC;
                    else rc = icsp_pack_bits(w->ctx, 0, cn, w->body, w->body_cap, &bits);
                    if (!rc) ch.bytes.assign(w->body, w->body + nb);
                }
                w->t_pack += now() - t0; t0 = now();
                if (!rc) rc = icsp_download(w->ctx, 0, cn, nullptr, nullptr, nullptr, nullptr, rec);
                w->t_down += now() - t0;
                if (rc) { fail(rc, nullptr); return; }
            }
            t0 = now();
            if (!out_map && fd_rec >= 0) pwrite_all(fd_rec, stage_out, bytes, (off_t)(f * fsz));
            w->t_write += now() - t0;
            w->chunks++;
            if (!opt.hostpack && c2 < nchunks) {
                rc = early ? wait_upload() : upload_here(c2);
                if (rc) { fail(rc, rc == ICSP_ERR_RANGE ? "short read" : nullptr); return; }
            }
            c = c2;
        }
    };
    {
        std::vector<std::thread> th, up;
        for (auto& u : uploaders) up.emplace_back(uploader_loop, &u);
        for (int d = 1; d < nworker; d++) th.emplace_back(work, &workers[d]);
        work(&workers[0]);
        for (auto& t : th) t.join();
        for (auto& u : uploaders) { { std::lock_guard<std::mutex> l(u.m); u.stop = true; } u.cv.notify_all(); }
        for (auto& t : up) t.join();
    }
    helper.join();
    for (auto& w : workers)
        if (w.rc) { printf("[ERROR] GPU %d: %s\n", w.device, w.err.c_str()); exit(-1); }
    const double t_init_done = ready.t_last;
    const double t_encoded = now();

    for (int f = 0; f < n; f++)                                            // print_frame_end_message (ENC:44-48)
        printf("Encoding FRAME_%03d(%c) done!\n", f, (opt.intra_period == 0 || f % opt.intra_period == 0) ? 'I' : 'P');

    // makebitstream (ENC:4849-4900), the rest of it: header, the strings that did not go direct (cut into 1 MB runs and
    // placed by several threads), the reference's final byte; the image is the .bin mapping when there is one
    size_t nbytes = 0;
    int rc;
    const uint64_t total = cursor.bits;
    auto finish = [&](uint8_t* img, size_t cap, bool zeroed) -> int {
        if (opt.hostpack) return icsp_write_bitstream(&params, n, levels.data(), acflag.data(), mpm.data(), mvd.data(), img, cap, &nbytes);
        int r = zeroed ? icsp_bitstream_header(&params, total, img, cap, &nbytes) : icsp_bitstream_begin(&params, total, img, cap, &nbytes);
        if (r) return r;
        struct Run { uint64_t at; const uint8_t* p; uint64_t bits; };
        std::vector<Run> runs;
        const size_t kRun = (size_t)1 << 20;
        for (auto& ch : chunks) {
            if (ch.direct) continue;
            const size_t nb = (size_t)((ch.bits + 7) / 8);
            for (size_t o = 0; o < nb; o += kRun)
                runs.push_back({ ch.at + 8 * (uint64_t)o, ch.bytes.data() + o, std::min<uint64_t>(8 * (uint64_t)kRun, ch.bits - 8 * (uint64_t)o) });
        }
        if (!runs.empty()) {
            const int nt = (int)std::min<size_t>({ runs.size(), (size_t)16, (size_t)std::max(1u, std::thread::hardware_concurrency()) });
            std::atomic<size_t> next(0);
            std::atomic<int> bad(0);
            auto place = [&] {
                for (size_t k; (k = next.fetch_add(1)) < runs.size();)
                    if (int e = icsp_bitstream_place(img, cap, runs[k].at, runs[k].p, runs[k].bits)) bad = e;
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; t++) th.emplace_back(place);
            place();
            for (auto& t : th) t.join();
            if (bad.load()) return bad.load();
        }
        return icsp_bitstream_end(img, total);
    };
    double t_fin = 0, t_trunc = 0;
    if (bin_map) {
        double t0 = now();
        rc = finish(bin_map, bin_cap, true);                      // a fresh file's pages are zero
        t_fin = now() - t0; t0 = now();
        if (bin_pinned) icsp_host_unregister(bin_map);
        if (!rc && ftruncate(fd_bin, (off_t)nbytes) != 0) rc = ICSP_ERR_RANGE;      // (only after unpinning: truncating a file under
                                                                                   //  a live pinned mapping stalls the device's queues); unpinning takes 0.1 ms, the truncation
                                                                                   //  1 ms + 0.15 ms per MB of populated pages it frees
        t_trunc = now() - t0;
    } else {
        std::vector<uint8_t> bs(opt.hostpack ? bin_cap : 14 + (size_t)(total / 8) + 3);
        rc = finish(bs.data(), bs.size(), false);
        if (!rc && (ftruncate(fd_bin, 0) != 0 || !pwrite_all(fd_bin, bs.data(), nbytes, 0))) rc = ICSP_ERR_RANGE;
    }
    if (rc) { printf("[ERROR] %s in makebitstream\n", icsp_strerror(rc)); exit(-1); }
    close(fd_bin);
    if (fd_rec < 0) printf("fail to save yuv\n");
    else close(fd_rec);
    close(fd_in);
    const double t_files = now();
    if (opt.stats) {
        int ndirect = 0;
        for (auto& ch : chunks) ndirect += ch.direct ? 1 : 0;
        double su = 0;
        for (auto& w : workers) su = std::max(su, w.t_setup);
        const Worker& w0 = workers[0];
        int nbound = 0;
        for (auto& w : workers) nbound += w.bound;
        printf("[icsp_enc]{\"frames\": %d, \"workers\": %d, \"devices\": %d, \"chunk_frames\": %d, \"chunks\": %d, \"chunks_packed_into_bin_mapping\": %d, "
               "\"input_mapped\": %s, \"output_mapped\": %s, \"bin_mapped\": %s, "
               "\"init_s\": %.4f, \"encode_s\": %.4f, \"bitstream_and_files_s\": %.4f, \"map_files_s\": %.4f, \"pin_mappings_s\": %.4f, "
               "\"hip_start_s\": %.4f, \"setup_worker0\": {\"create_s\": %.4f, \"prepare_s\": %.4f, \"copy_streams_s\": %.4f, \"wait_for_mappings_s\": %.4f, "
               "\"staging_alloc_s\": %.4f, \"warm_transfers_s\": %.4f}, "
               "\"max_worker_setup_s\": %.4f, \"worker0\": {\"chunks\": %d, \"read_s\": %.4f, \"upload_s\": %.4f, \"encode_call_s\": %.4f, \"pack_count_s\": %.4f, "
               "\"turn_wait_s\": %.4f, \"pack_s\": %.4f, \"download_s\": %.4f, \"write_s\": %.4f}, "
               "\"numa\": {\"nodes\": %d, \"placement\": %s, \"device0_node\": %d, \"threads_bound\": %d, \"output_bytes_placed\": %zu}, "
               "\"bin_finish_s\": %.4f, \"bin_truncate_s\": %.4f, \"bin_bytes\": %zu, \"e2e_fps_excl_init\": %.1f, \"e2e_fps_incl_init\": %.1f}\n",
               n, nworker, std::min(ndev, nworker), chunk, nchunks, ndirect, in_map ? "true" : "false", out_map ? "true" : "false", bin_pinned ? "true" : "false",
               t_init_done - t_start, t_encoded - t_init_done, t_files - t_encoded, t_map, t_pin,
               t_hip, w0.t_create, w0.t_prepare, w0.t_copystreams, w0.t_mapwait, w0.t_stage, w0.t_warm, su,
               w0.chunks, w0.t_read, w0.t_up, w0.t_enc, w0.t_count, w0.t_turn, w0.t_pack, w0.t_down, w0.t_write,
               icsp_numa_nodes(), placement ? "true" : "false", dev_node.empty() ? -1 : dev_node[0], nbound, placed_bytes,
               t_fin, t_trunc, nbytes, n / (t_files - t_init_done), n / (t_files - t_start));
    }
    // The files are complete and closed.  Contexts, pinned buffers, mappings and the runtime itself are not torn down piece by
    // piece (tens of milliseconds of unmapping and freeing that produce nothing): the process ends here and the kernel reclaims
    // them.  ICSP_ENC_TEARDOWN=1 takes the long way (leak checkers, tests).
    fflush(stdout);
    if (!getenv("ICSP_ENC_TEARDOWN")) _exit(0);
    for (auto& w : workers) { icsp_host_free(w.stage); icsp_host_free(w.body); icsp_destroy(w.ctx); }
    return 0;
}
