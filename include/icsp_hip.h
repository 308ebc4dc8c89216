/*
 * icsp_hip.h — C ABI of the MI355X (gfx950) implementation of ICSPCodec's per-macroblock encode loop.
 *
 * The reference has no plugin/FFI interface; its de-facto boundary is the three frame functions the
 * scheduler calls:
 *     int  allintraPrediction(FrameData*, int nframes, int QstepDC, int QstepAC);  ICSP_Codec_Encoder.h:250, caller ENC:221
 *     void intraPrediction   (FrameData&,              int QstepDC, int QstepAC);  ICSP_Codec_Encoder.h:249, callers ENC:233, 202
 *     int  interPrediction   (FrameData& cur, FrameData& prev, int, int);          ICSP_Codec_Encoder.h:265, callers ENC:238, 207
 * plus, downstream, makebitstream (ICSP_Codec_Encoder.h:310, callers ENC:222, 242).
 * (ENC = /root/reference/source/encoder/ICSP_Codec_Encoder_source.cpp.)
 *
 * This library replaces them with plain-pointer entry points over flat planes.  Data that crosses:
 *   frames    uint8  [n][W*H*3/2]       planar I420 (Y, Cb, Cr) — the order YCbCrLoad reads (ENC:274-279)
 *   levels    int16  [n][nMB][6][64]    zig-zag order (ENC:3031-3094); blocks 0-3 Y (raster in the MB), 4 Cb, 5 Cr
 *                                        = intraReorderedblck8 / interReorderedblck8 / (intra|inter)Reorderedblck
 *   acflag    uint8  [n][nMB][6]        1 = all 63 AC levels are zero (intraACflag / interACflag, ENC:2784-2792)
 *   mpm_mode  uint8  [n][nMB][4]        bit0 MPMFlag, bit1 intraPredMode (I frames; 0 on P frames) (ENC:987-997)
 *   mvd       int8   [n][nMB][2]        differential motion vector x,y = bd.mv after mvPrediction (ENC:2353; 0 on I frames)
 *   recon     uint8  [n][W*H*3/2]       reconstructedY/Cb/Cr — what checkResultFrames dumps to test_yuv.yuv (ENC:6408-6410)
 * nMB = (W/16)*(H/16) <= 8704 (e.g. 2048x1088), macroblocks in raster order.  W, H multiples of 16, 32 <= W <= 4096, 16 <= H <= 2304.
 *
 * Errors: every function returns 0 (= reference SUCCESS, ICSP_Codec_Encoder.h:33-39) or a positive
 * icsp_status code; nothing calls exit().  There is NO CPU fallback: without a usable HIP device
 * icsp_create fails with ICSP_ERR_NO_DEVICE.
 * Environment: three variables a host may want (read once by icsp_create; results are identical in every setting; a value that is
 * not a whole number in the stated range makes icsp_create fail with ICSP_ERR_UNCORRECT_PARAM):
 *   ICSP_P_GROUPS  1..3  GOP groups whose P-step kernel chains run on separate streams (default 2; per context: icsp_set_groups)
 *   ICSP_I_GROUPS  1..2  parts (launches on separate streams) an all-intra batch of more frames than CUs is encoded in (default 2)
 *   ICSP_FAKE_DEVICES 2..64 (test hook, read once per process) the library presents that many devices, device d being physical device
 *                        d mod (real devices) with per-device records of its own: the multi-device paths of a host on a one-GPU box
 * The tuning and diagnostic overrides the kernels' authors use (ICSP_INTRA_FORM, ICSP_CHROMA_CAP, ICSP_TIMELINE_DUMP ...) are listed in
 * INTEGRATION.md ("Tuning and diagnostic overrides"); none of them changes a result either.
 * Launch path: a failed kernel launch, event record or cross-stream wait could silently drop an ordering edge and yield
 * wrong bits, so it POISONS the context: that call and every later call on the context return ICSP_ERR_HIP
 * (icsp_last_error names the first failure) until icsp_destroy.
 * Threading: one context per device; calls on one context must be serialised by the caller; distinct
 * contexts are independent (the host GOP dispatcher uses one thread per GPU, the analogue of
 * encoding_thread, ENC:186-213).
 */
#ifndef ICSP_HIP_H
#define ICSP_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library itself is built with -fvisibility=hidden: these entry points are all it exports */
#endif

typedef enum {
    ICSP_OK = 0,                 /* SUCCESS          (ICSP_Codec_Encoder.h:35) */
    ICSP_ERR_UNENOUGH_PARAM = 1, /* UNENOUGH_PARAM   (ICSP_Codec_Encoder.h:36) */
    ICSP_ERR_UNCORRECT_PARAM = 2,/* UNCORRECT_PARAM  (ICSP_Codec_Encoder.h:37) */
    ICSP_ERR_MEM_ALLOC = 3,      /* FAIL_MEM_ALLOC   (ICSP_Codec_Encoder.h:38) */
    ICSP_ERR_NO_DEVICE = 4,      /* no HIP device / runtime unusable */
    ICSP_ERR_HIP = 5,            /* a HIP call failed; see icsp_last_error() */
    ICSP_ERR_RANGE = 6           /* frame range outside the context's capacity / not GOP aligned */
} icsp_status;

typedef struct {
    int width, height;   /* luma size, multiples of 16 (the reference hard-codes 352x288, encoder_main.cpp:20) */
    int qp_dc, qp_ac;    /* quantiser steps, 1..255: one header byte each (README: 1, 8 or 16) */
    int intra_period;    /* 0 = every frame is an I frame (ALL_INTRA, ICSP_Codec_Encoder.h:18); k>0: frame n is I iff n%k==0 */
} icsp_params_t;

typedef struct icsp_ctx icsp_ctx_t; /* opaque: device buffers, stream and scratch for ONE device */

/* ---- lifetime ------------------------------------------------------------------------------ */
/* max_frames = capacity of the resident frame store (inputs + all outputs stay in HBM).
 * A context's HIP streams (up to four) are taken from, and on icsp_destroy given back to, a per-device pool that lives as long as
 * the process: which hardware queue a stream gets depends on the order the process created its streams in, so later contexts
 * inherit the early ones.  Create the first context before other HIP streams of the process where possible. */
int icsp_create(icsp_ctx_t** out, const icsp_params_t* params, int device_id, int max_frames);
int icsp_destroy(icsp_ctx_t* ctx);
const char* icsp_strerror(int status);
int icsp_device_count(void);          /* usable HIP devices (0 when there is none or the runtime cannot start) */
const char* icsp_last_error(const icsp_ctx_t* ctx);   /* text of the last HIP failure on this context */

/* ---- one-call host path: replaces the allintraPrediction / intraPrediction+interPrediction loop of
 *      single_thread_encoding (ENC:217-245) for frames [0, n).  Frame i is an I frame iff
 *      intra_period == 0 or i % intra_period == 0; closed GOPs are encoded concurrently.
 *      All pointers are caller-owned host memory; any output pointer may be NULL (skipped). ----- */
int icsp_encode_gop(icsp_ctx_t* ctx, const uint8_t* yuv420_in, int n,
                    int16_t* levels, uint8_t* acflag, uint8_t* mpm_mode, int8_t* mvd, uint8_t* recon);
/* Inside the call the frames are cut into chunks of whole GOPs (about 32 MB of results each) that go up, are encoded and come
 * down as a pipeline: uploads, kernels and downloads of neighbouring chunks run side by side on the device's two transfer
 * streams (icsp_copy_streams is switched on for the context by the first multi-chunk call; once per device and process that
 * costs some tens of milliseconds).  Arrays in pinned memory (icsp_host_alloc, or the caller's own memory after
 * icsp_host_register -- the frame buffers YCbCrLoad allocates once per clip, ENC:247-283, are the natural candidates) are read
 * and written by DMA directly: the call then runs at the link's rate; other memory goes through pinned staging buffers of the
 * context with a few helper threads.  Same bytes either way.
 *
 * icsp_encode_gop_packed: the same, but instead of the levels and the side arrays the caller gets the PACKED BODY of the frames
 * (what icsp_pack_bits returns: `*nbits` bits, to be framed by icsp_bitstream_assemble / _begin, _place, _end) -- makebitstream
 * (ENC.h:310, callers ENC:222, 242) needs nothing else, and it is 4 % of the levels' bytes.  recon may be NULL. */
int icsp_encode_gop_packed(icsp_ctx_t* ctx, const uint8_t* yuv420_in, int n, uint8_t* recon,
                           uint8_t* body, size_t body_cap, uint64_t* nbits);

/* ---- resident path (inputs already in HBM when the timed region starts) --------------------- */
int icsp_upload(icsp_ctx_t* ctx, const uint8_t* yuv420_in, int first_frame, int n);   /* H2D into slots [first, first+n) */
/* Encode slots [first, first+n); first must be GOP aligned (first % intra_period == 0).  Asynchronous
 * on the context's streams.  Consecutive calls are ordered against each other only as far as the data require: a range
 * disjoint from every range still in flight (the next chunk of a clip: closed GOPs are the reference's independent jobs,
 * ENC:186-213), or the very same range again, starts without waiting for the earlier calls; a range that partly overlaps
 * one in flight waits for everything.  Whatever reads results (icsp_sync, icsp_download, icsp_pack_*) or writes inputs
 * (icsp_upload) waits for every encode issued before it. */
int icsp_encode_resident(icsp_ctx_t* ctx, int first_frame, int n);
/* Several disjoint ranges as ONE batch: slots [first_frames[r], first_frames[r] + ns[r]) for r < k, each GOP aligned, none overlapping
 * another (ICSP_ERR_RANGE otherwise).  Every kernel of a step is launched once over all ranges (slot tables instead of arithmetic
 * progressions), so several short ranges -- chunks of different clips, the ends of GOP shards: the reference's independent GOP jobs,
 * ICSP_thread.cpp:47-56 -- cost what one long range costs (four ranges of 150 CIF frames, all-intra: 1.0 M frames/s one by one).
 * Same results as icsp_encode_resident on each range; asynchronous in the same way.  The same LIST encoded again follows its own previous
 * pass; overlap with what is still in flight is decided range by range: a list one of whose RANGES partly overlaps a range in flight
 * waits for everything, a list whose ranges merely interleave with those of another list or range does not. */
int icsp_encode_resident_many(icsp_ctx_t* ctx, int k, const int* first_frames, const int* ns);
int icsp_sync(icsp_ctx_t* ctx);
int icsp_download(icsp_ctx_t* ctx, int first_frame, int n,
                  int16_t* levels, uint8_t* acflag, uint8_t* mpm_mode, int8_t* mvd, uint8_t* recon);

/* Device-side view for zero-copy users (bench.py fills `frames` from a torch tensor and reads the
 * outputs in place).  Pointers are HBM addresses valid until icsp_destroy. */
typedef struct {
    void* frames; void* levels; void* acflag; void* mpm_mode; void* mvd; void* recon;
    void* stream;         /* hipStream_t the kernels are launched on */
    int max_frames, n_mb;
} icsp_device_view_t;
int icsp_device_view(icsp_ctx_t* ctx, icsp_device_view_t* out);
/* Scheduling knobs of ONE context (the environment variables ICSP_P_GROUPS / ICSP_I_GROUPS set the defaults of every
 * context of the process): GOP groups on separate streams (1..3) and parts of a large all-intra batch (1..2); 0 keeps the
 * current value.  Results never depend on them. */
int icsp_set_groups(icsp_ctx_t* ctx, int p_groups, int i_groups);
/* on != 0: every kernel of the context on its one stream (no chroma stream, no GOP-group streams).  A stream costs 10-25 ms of
 * set-up on this runtime, more than a short batch takes to encode; hosts that encode one chunk (icsp_enc on a short clip) or
 * run several contexts per device use it.  Results never depend on it. */
int icsp_single_stream(icsp_ctx_t* ctx, int on);

/* ---- debug taps used by the parity tests (not part of the reference boundary) ---------------- */
/* Raw motion vectors int8[n][nMB][2] (Reconstructedmv, ENC:2426) and chosen intra modes uint8[n][nMB][4]
 * (DPCMmodePred, ENC:889...) of slots [first, first+n); either pointer may be NULL. */
int icsp_download_debug(icsp_ctx_t* ctx, int first_frame, int n, int8_t* mv, uint8_t* intra_mode);
/* When enabled (before encoding), forward-DCT coefficients before DC prediction/quantisation are kept:
 * double[n][nMB][6][64], row-major [v][u] (DCT_block output, ENC:2685-2749).  Costs 8x the level store. */
int icsp_debug_keep_coef(icsp_ctx_t* ctx, int on);
int icsp_download_coef(icsp_ctx_t* ctx, int first_frame, int n, double* coef);
/* What the last icsp_encode_resident chose: intra luma kernel form (8 or 32 lanes per block), its waves per workgroup, whether the
 * 8-lane form wrote the reconstruction through its LDS ring (0/1), range placed whole on one chain stream (0/1), GOP groups, block
 * rows of the 8-lane form's wavefront chained in pairs (2) or not (0).  Any pointer may be NULL.  For reports. */
int icsp_debug_last_choice(icsp_ctx_t* ctx, int* intra_form, int* intra_waves, int* intra_recon_ring, int* whole_range, int* gop_groups, int* intra_row_group);
/* Test hook, needs no device: the placement the library would give k successive icsp_encode_resident calls on ranges (firsts[i], ns[i])
 * of a fresh context with default settings -- whole[i]: on ONE chain stream (the range is disjoint from the call before);
 * three[i]: three chain streams in turn (all-intra, three ranges in rotation); turn[i]: which chain stream (0, 1, 2).  Any
 * output may be NULL. */
int icsp_debug_plan_turns(int intra_period, int k, const int* firsts, const int* ns, int* whole, int* three, int* turn);
/* Streams idle in the pool of device `device_id` (of this process): what destroyed contexts left for the next one.  For tests. */
int icsp_debug_stream_pool(int device_id);
/* Test hook, needs no device: a context shell in the state a failed launch-path call leaves behind (poisoned).  Every entry
 * point answers ICSP_ERR_HIP on it without touching the runtime; release it with icsp_destroy. */
int icsp_debug_poisoned_context(icsp_ctx_t** out);

/* ---- per-kernel timing with HIP events on the launch stream ---------------------------------- */
enum { ICSP_K_INTRA_LUMA = 0, ICSP_K_CHROMA_DC, ICSP_K_RESIDUAL, ICSP_K_ME, ICSP_K_FRAME_SERIAL, ICSP_K_PACK, ICSP_K_DECODE, ICSP_K_COUNT };
/* on: 0 = off, 1 = time every kernel, otherwise a mask with bit (k+1) set for each kernel k to time (events cost a few
 * microseconds per launch, so the bench times only the dominant kernel inside its timed region). */
int icsp_profile_enable(icsp_ctx_t* ctx, int on);
int icsp_profile_reset(icsp_ctx_t* ctx);
/* Sums over launches since the last reset; syncs the stream. */
int icsp_profile_get(icsp_ctx_t* ctx, int kernel, double* total_ms, long long* launches);
const char* icsp_kernel_name(int kernel);

/* ---- host back end: bit-exact restatement of makebitstream (ENC:4849-6334) -------------------- */
/* Worst-case size in bytes of the .bin for n frames (header included). */
size_t icsp_bitstream_bound(const icsp_params_t* params, int n);
/* Packs header + body for frames [0, n) from the arrays above (host memory) into out (capacity cap).
 * Writes the byte count to *out_bytes: 14 header bytes + cntbits/8 + 1 (ENC:4895, 5029). */
int icsp_write_bitstream(const icsp_params_t* params, int n,
                         const int16_t* levels, const uint8_t* acflag, const uint8_t* mpm_mode, const int8_t* mvd,
                         uint8_t* out, size_t cap, size_t* out_bytes);


/* ---- device back end: the same body bits packed on the GPU ------------------------------------- */
/* The value code is a fixed table (ENC:5417-5602), so bit lengths, prefix sums and packing parallelise per block.
 * Packs the body (everything after the 14-byte header) of the already encoded slots [first, first+n), first GOP
 * aligned, as an MSB-first bit string, copies its ceil(nbits/8) bytes (zero padded) to host memory `body` and writes the
 * bit count to *nbits.  Only the bits cross PCIe instead of levels/flags/vectors (about 1/30 of the bytes at QP 16).
 * cap >= icsp_bitstream_bound() - 14 always suffices; ICSP_ERR_RANGE if the body does not fit cap. */
int icsp_pack_bits(icsp_ctx_t* ctx, int first_frame, int n, uint8_t* body, size_t cap, uint64_t* nbits);
/* The same in two steps, for hosts that build ONE image out of several batches without touching the bits on the CPU
 * (icsp_enc): icsp_pack_count runs the length and prefix-sum kernels and returns the bit count of the range; once the host
 * knows where the string starts in the body (at_bit = sum of the counts before it), icsp_pack_into packs it on the device
 * at that bit phase and copies its whole bytes to body_image + at_bit/8 (by DMA when the image is pinned memory, e.g. a
 * registered mapping of the .bin); the first and last byte, which neighbouring strings may share, are OR-ed in
 * atomically, so the image must start out zeroed and concurrent calls from several contexts are safe.  body_image is the
 * body, i.e. the .bin image + 14; cap its capacity in bytes.  Both synchronise.  icsp_pack_into must follow an
 * icsp_pack_count of the same range (ICSP_ERR_RANGE otherwise, or when the string does not fit cap). */
int icsp_pack_count(icsp_ctx_t* ctx, int first_frame, int n, uint64_t* nbits);
int icsp_pack_into(icsp_ctx_t* ctx, int first_frame, int n, uint64_t at_bit, uint8_t* body_image, size_t cap);
/* Creates now what an encode / pack would create on first use (GOP-group streams, packer buffers, first launches) by
 * encoding and packing up to two GOPs of black frames in slots [0, ...): for hosts that time or pipeline their batches. */
int icsp_prepare(icsp_ctx_t* ctx);
/* Host: header + the pieces (each an MSB-first bit string of piece_bits[i] bits, e.g. one per GPU shard, in frame
 * order) concatenated bit-wise + the reference's final byte (its bits right-aligned, ENC:4956) -> the .bin image,
 * identical to icsp_write_bitstream on the same frames.  *out_bytes = 14 + total_bits/8 + 1. */
int icsp_bitstream_assemble(const icsp_params_t* params, int npieces, const uint8_t* const* pieces,
                            const uint64_t* piece_bits, uint8_t* out, size_t cap, size_t* out_bytes);

/* The same in steps, for hosts that place the pieces of several shards from several threads (icsp_enc): begin zeroes the image
 * and writes the header, place puts one piece at its bit offset in the body (thread-safe for disjoint bit ranges), end
 * right-aligns the final partial byte. */
int icsp_bitstream_begin(const icsp_params_t* params, uint64_t total_bits, uint8_t* image, size_t cap, size_t* out_bytes);
/* begin without the zeroing: only the 14 header bytes (for an image that is zero already, e.g. a fresh file mapping that
 * icsp_pack_into has filled meanwhile); *out_bytes as above. */
int icsp_bitstream_header(const icsp_params_t* params, uint64_t total_bits, uint8_t* image, size_t cap, size_t* out_bytes);
int icsp_bitstream_place(uint8_t* image, size_t cap, uint64_t bit_offset, const uint8_t* piece, uint64_t piece_bits);
int icsp_bitstream_end(uint8_t* image, uint64_t total_bits);

/* ---- host memory and the transfers.  Every call that takes a host pointer works with any memory.  What the library KNOWS to be
 *      pinned -- ranges of icsp_host_alloc and icsp_host_register, and ranges for which the runtime names one allocation that
 *      covers them whole (someone else's hipHostMalloc) -- is the source / target of the DMA itself: PCIe speed, beside the
 *      kernels of other contexts.  Everything else goes through two pinned staging buffers per direction inside the library
 *      (helper threads copy a piece beside the transfer of the piece before): the HIP runtime is never handed a plain caller
 *      pointer.  (It would pin such a buffer on the fly and cache the pin; in a long-lived process whose heap ranges are trimmed
 *      and mapped again a later transfer then faults on the device -- round 4, tools/repro_fault.py.)  NULL when there is no
 *      device or no memory. ---- */
void* icsp_host_alloc(size_t bytes);
void icsp_host_free(void* p);
/* Pin a range the caller already owns (hipHostRegister) -- in particular a mapping of the input file (read_only != 0: mapped
 * without write permission) or of the output file (MAP_SHARED, pages populated): icsp_upload then reads the frames and
 * icsp_download writes the reconstruction by DMA from/into the page cache, with no staging copy on the host (the reference's
 * YCbCrLoad fread, ENC:247-283, and checkResultFrames fwrite, ENC:6376-6413, become the transfers themselves).  The range is
 * usable from every device.
 * `p` must lie on a page boundary (ICSP_ERR_UNCORRECT_PARAM otherwise); `bytes` is rounded up to whole pages, which must be the
 * caller's own to the end of the last one (true of every mapping, and of posix_memalign(4096, ...) blocks whose size was rounded
 * up).  ICSP_ERR_HIP when the runtime refuses the range (callers fall back to plain memory, i.e. the library's staging).
 * Unregister a range BEFORE its memory is freed or unmapped: a registration that outlives its pages makes the runtime treat
 * whatever is mapped there next as pinned.  Buffers that merely start or end inside a registered page are not treated as pinned
 * by this library (they are staged), whatever the runtime reports for their first byte. */
int icsp_host_register(void* p, size_t bytes, int read_only);
int icsp_host_unregister(void* p);
/* Spends a pinned range's first-use cost now: the context's stream writes `bytes` zero bytes (at most 16 MB) to it by DMA.  For
 * ranges that must start out zeroed anyway, i.e. the body image icsp_pack_into fills (a stream's first large transfer into a
 * newly pinned range can cost the call about 6 ms).  ICSP_ERR_UNCORRECT_PARAM for memory the library does not know to be pinned. */
int icsp_host_warm(icsp_ctx_t* ctx, void* pinned, size_t bytes);
/* Several contexts on one device: a stream's transfers go to the DMA engine the runtime gave its FIRST copy -- the lowest-
 * numbered engine idle at that moment -- so contexts set up one after the other on an idle device all share one engine, and
 * their transfers, uploads and downloads alike, then run one at a time (tools/probe_duplex.hip).
 * shared != 0: the context's uploads (icsp_upload) run on one stream shared by all contexts of its device and its downloads
 * (icsp_download, icsp_pack_into) on another, so that the link carries both directions at once.  A transfer then starts when
 * the context's own stream is idle and the call returns when it is complete (icsp_upload too), one at a time per direction
 * and device; 0: transfers on the context's own stream again (the default). */
int icsp_copy_streams(icsp_ctx_t* ctx, int shared);
/* icsp_upload on the device's shared upload stream (icsp_copy_streams must be on), returning when the frames are on the
 * device.  It changes nothing in the context, so a second host thread may call it while the context's own thread packs and
 * downloads an earlier batch -- the one exception to "calls on one context must be serialised".  The caller guarantees that
 * no kernel of the context that reads the frames is queued or running (the last encode has been waited for: icsp_pack_count,
 * icsp_sync or a download has returned) and starts the next encode only after this call has returned.  A host that makes
 * all uploads of a device from ONE thread this way, one at a time, also keeps them on one DMA engine: a copy submitted while
 * the stream's engine is busy is given another engine, whose first use costs milliseconds. */
int icsp_upload_sync(icsp_ctx_t* ctx, const uint8_t* yuv, int first, int n);

/* ---- placement on multi-socket hosts (csrc/icsp_topology.cpp; host only, sysfs, no libnuma).  The reference's GOP thread pool
 *      (ICSP_thread.cpp:39-77) places nothing; here frames move by DMA from and into pinned file mappings, so a device's host
 *      threads and the pages of its chunks belong on the socket the device hangs off.  Every call is a no-op where there is
 *      nothing to place (one node, unknown node, a container that forbids affinity changes). ---- */
/* PCI bus id "dddd:bb:dd.f" of a HIP device (hipDeviceGetPCIBusId); ICSP_ERR_NO_DEVICE / ICSP_ERR_RANGE (cap too small). */
int icsp_device_pci_bus_id(int device, char* out, int cap);
int icsp_numa_node_of_pci(const char* bus_id);            /* /sys/bus/pci/devices/<id>/numa_node; -1 unknown */
int icsp_device_numa_node(int device);                    /* the two above together; -1 unknown */
int icsp_numa_nodes(void);                                /* nodes that have CPUs (>= 1) */
int icsp_numa_cpus(int node, int* cpus, int cap);         /* CPUs of a node (count; cpus may be NULL); -1 no such node */
int icsp_parse_cpulist(const char* list, int* cpus, int cap);   /* "0-3,8,10-11" -> cpus; count, -1 malformed */
int icsp_bind_thread_to_node(int node, int* bound);      /* calling thread -> the node's CPUs; *bound: whether anything changed */
int icsp_populate_here(void* p, size_t bytes);            /* first-touch the pages of a writable mapping from this thread */
int icsp_chunk_device(int chunk, int ndev);               /* chunk -> device when a clip's chunks are dealt over ndev devices */

/* ---- decoder side: DEC = /root/reference/source/decoder/ICSP_Codec_Decoder_source.cpp ---- */
/* Host: readHeader (DEC:14-37).  intra_period is the header field as stored: 1 (or 0) = every frame intra (DEC.h:293). */
int icsp_parse_header(const uint8_t* bin, size_t nbytes, icsp_params_t* out);
/* Host: readBlockData (DEC:38-405): the .bin image -> the syntax arrays of frames [0, n) in the layouts above.
 * ICSP_ERR_RANGE when the stream ends early (the final macroblock may run into the reference's own garbled last byte). */
int icsp_parse_bitstream(const uint8_t* bin, size_t nbytes, int n,
                         int16_t* levels, uint8_t* acflag, uint8_t* mpm_mode, int8_t* mvd);
/* H2D of parsed syntax into slots [first, first+n) of a context created with the stream's parameters. */
int icsp_upload_syntax(icsp_ctx_t* ctx, int first_frame, int n, const int16_t* levels, const uint8_t* mpm_mode, const int8_t* mvd);
/* Device: allintraPredictionDecode / intraPredictionDecode / interPredictionDecode (DEC:2083-2272) on resident syntax
 * (uploaded, or left by icsp_encode_resident): decoded planes into `recon` of slots [first, first+n), first GOP aligned.
 * Uses the decoder's own cosine table (double literals, DEC.h:19-27), so the output is the reference DECODER's, which
 * differs from the encoder's reconstruction by a grey level on a few pixels.  Asynchronous; fetch with icsp_download. */
int icsp_decode_resident(icsp_ctx_t* ctx, int first_frame, int n);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
