// icsp_sched.cpp — the host half of libicsp_hip.so: contexts, the device's stream pool, flight records and placement rules
// (which range goes onto which stream, which form the I-frame luma launch takes), the frame schedule of a range (encode_range,
// encode_many, decode_range), transfers and caller memory, the one-call pipeline (icsp_encode_gop) and the C ABI of
// include/icsp_hip.h.  Host code only: every kernel is reached through the launch layer of icsp_device.hip (icsp_kernels.h).
//
// Replaces the reference's scheduler: single_thread_encoding ENC:217-245 (frame loop), encoding_thread / the GOP job queue
// ENC:186-213, ICSP_thread.cpp:39-77 (independent closed GOPs), YCbCrLoad ENC:247-283 and checkResultFrames ENC:6376-6413 (frames in,
// reconstruction out).  ENC = /root/reference/source/encoder/ICSP_Codec_Encoder_source.cpp.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <vector>
#include "icsp_hip.h"
#include "icsp_kernels.h"

using namespace icspk;

namespace {

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return ICSP_ERR_HIP; } } while (0)
// The launch path (kernel launches, event records, cross-stream waits): a failure there can silently remove an ordering edge
// and produce wrong bits, so it POISONS the context -- this call and every later one on the context return ICSP_ERR_HIP
// (icsp_last_error names the first failed call) until the context is destroyed.
#define HIPQ(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return poison(ctx, #call, e_); } while (0)
// every entry point that works on a context's device state
#define ENTER(ctx) do { if (!(ctx)) return ICSP_ERR_UNENOUGH_PARAM; if ((ctx)->sticky) return (ctx)->sticky; } while (0)

struct EvPair { hipEvent_t a, b; int kernel; int sid; };
constexpr int kMaxPGroups = 3;
constexpr int kMaxFlights = 4;
constexpr int kGopEvSets = 4;          // sets of per-stream events of icsp_encode_gop's chunks (a power of two)
// A range of frame slots whose encode may still be running on the context's streams and has not been joined onto `stream`.
// Two encodes need no ordering between them when their ranges are disjoint (closed GOPs / independent frames: the reference's
// GOP jobs, ENC:186-213) or identical (the same partition onto the same streams: stream order does it); anything else
// joins everything first.
// How a range is laid onto the streams is decided when its record is made and kept while it is in flight:
//   split (whole == false): the range's GOP groups / all-intra parts on `stream` and the group streams -- best when the same
//     range is encoded again and again, since only its own parts can run beside each other;
//   whole: everything on ONE of the chain streams (sidx: 0 = `stream`, 1 = pstream[1]; 2 = pstream[2] while three all-intra ranges
//     rotate), ranges taking turns -- best when the caller alternates between independent ranges: two (three) whole ranges are in
//     flight side by side instead of two halves.  Calls take the streams in turn; a range that comes back on another stream first
//     waits for its own previous pass (ev_done, recorded behind every whole pass), so ranges in rotation load the streams evenly.
struct Flight { int first, n; bool used, whole, done_valid; int sidx; hipEvent_t ev_done; hipEvent_t ev_p1[kMaxPGroups];    // ev_p1[k]: group k's first P step of the last pass over this range is done
                // a coalesced list of ranges (icsp_encode_resident_many): first / n are its hull; the list itself, and the slot tables of its
                // launches ([step][GOP], compacted per step) on the device and in pinned host memory
                int many_k; int* many_list; int* d_tab; int* h_tab; int tab_cap; hipEvent_t ev_tab; bool tab_up; };     // ev_tab: the tables' last upload has left h_tab

} // namespace

struct icsp_ctx {
    icsp_params_t p;
    Geo g;
    int device, slot, max_frames, intra_waves, n_cu;      // device: physical; slot: the caller's device number (per-device records)
    hipStream_t stream, stream2;      // stream2: I-frame chroma beside the luma wavefront kernel (all-intra); all I-frame kernels (IPPP)
    hipEvent_t ev_fork, ev_join;
    hipStream_t up_stream, down_stream;   // icsp_copy_streams: the device's shared transfer streams (null: transfers on `stream`)
    // all-intra batches: stream2's chroma work and the luma kernel touch disjoint data, so consecutive encodes need no
    // cross-stream events at all; the join is deferred until something reads results (s2_dirty), the fork happens only
    // after other work was queued on `stream` (st_ahead) or when an outside producer uses the stream (always_sync)
    bool s2_dirty, st_ahead, always_sync;
    // IPPP batches: the I frames run on stream2, every GOP group's P steps are a chain on the group's own stream, and
    // consecutive encodes overlap across calls as far as the data allow (encode_range): encodes of disjoint ranges, or of
    // the same range again, are not ordered against each other at all.  p_dirty: the group streams carry work `stream` has
    // not been ordered after yet; flight[]: the ranges that un-joined work belongs to.
    bool p_dirty;
    Flight flight[kMaxFlights];
    int last_first, last_n, rr;       // the range of the previous encode call (alternation between ranges -> whole placement); stream turn
    std::vector<std::pair<int, int>> last_list;   // ... its ranges when it was a list (last_first / last_n are then the list's hull): a plain range in a
                                                  // hole of the hull is as independent of the list as one outside it
    int prev2_first, prev2_n;         // ... and of the call before it (three ranges in rotation -> three chain streams, all-intra)
    bool chains3;                     // ICSP_CHAINS3=0: two chain streams whatever the rotation (comparison)
    int last_form, last_nw, last_ring, last_whole, last_groups, last_rowgroup;     // what the last encode chose (icsp_debug_last_choice)
    bool i_stream_b;                  // ICSP_I_STREAM_B=0: the I frames of every range on stream2 (as before round 5); default: those of a range
                                      // placed whole on chain stream 1 on a stream of their own (pstream[2])
    int chroma_cap;                   // ICSP_CHROMA_CAP: KB of LDS reserved (not used) by the all-intra chroma launch of a small range placed whole
                                      // (encode_range), on top of k_residual8's 16.9 KB.  Default 60: 77 KB per workgroup -- one per CU beside up to
                                      // three 21.7 KB workgroups of the 8-lane luma kernel, two on a CU without any.  0: nothing reserved
    bool single;                      // icsp_single_stream: every kernel on `stream`, no chroma stream, no group streams
    bool whole_ok;                    // ICSP_WHOLE=0: never place a range whole on one stream (comparison)
    int sticky;                       // ICSP_ERR_HIP once a call of the launch path has failed (HIPQ): the context is poisoned
    bool no_fuse;                     // ICSP_NO_FUSE=1: k_me<true> and k_frame_serial as separate launches (comparison / fallback)
    int force_intra_form;             // ICSP_INTRA_FORM: 8 or 32 lanes per block in the intra luma kernel (0 = chosen from the batch)
    int force_intra_group;            // ICSP_INTRA_GROUP: 8-lane form with the plain wavefront where one is built (1), rows in pairs (2), or chosen (0)
    int intra_waves_g2;               // waves of eight blocks that the widest step of the pairs wavefront needs
    int i_groups;                     // ICSP_I_GROUPS: parts an all-intra batch of more frames than CUs is launched in (1 or 2)
    int prio_lo;
    int p_groups, prio_hi;            // GOP groups whose P-step chains run on separate streams (created on first use: a stream costs
                                      // milliseconds to create, and an all-intra encode never needs them)
    hipStream_t pstream[kMaxPGroups]; // [0] unused (group 0 runs on `stream`)
    hipEvent_t ev_pjoin[kMaxPGroups];
    DevBufs b;
    PackBufs pk;                      // device bit packer scratch + body buffer, allocated on first icsp_pack_bits
    size_t pk_cap;                    // bytes of pk.out
    uint8_t* pk_host;                 // pinned: total bits of the last count (8 bytes) | head bytes at 64 | tail bytes at 192
    int pk_first, pk_n;               // the range icsp_pack_count last measured (-1: none)
    unsigned long long pk_total;      //   and its bits
    uint8_t* d_frames;
    // icsp_encode_gop's transfer pipeline: two pinned staging buffers each way for caller memory that is not pinned (allocated on
    // first use, sized to a chunk), the event a chunk's kernels are waited for by, helper threads for the staging copies
    uint8_t* gop_stage_in[2];
    uint8_t* gop_stage_out[2];
    size_t gop_stage_in_cap, gop_stage_out_cap;
    hipEvent_t gop_ev[kGopEvSets][2 + 3]; // per chunk in flight (two; four when every chunk is queued on arrival): one event per stream of the context (stream, stream2, group streams)
    struct CopyPool* gop_pool;
    // every transfer from or into caller memory that is not KNOWN to be pinned goes through those staging buffers (xfer_up /
    // xfer_down): the runtime never sees a plain caller pointer.  xfer_ev_*: the DMA that last used a staging buffer
    hipEvent_t xfer_ev_in[2], xfer_ev_out[2];
    bool xfer_in_busy[2];
    std::mutex stage_in_m;                // the upload direction's staging state (gop_stage_in, xfer_in_busy, their reallocation): icsp_upload_sync may run on
                                          // a helper thread beside the context's own thread
    struct CopyPool* up_pool;             // helper threads of the upload direction (icsp_upload_sync may run beside the context's own thread)
    bool keep_coef, profiling;
    unsigned prof_mask;               // which kernels get HIP events (icsp_profile_enable's argument, bit k = kernel k)
    std::vector<EvPair> ev_pending;
    FILE* tl_file; hipEvent_t tl_base;    // ICSP_TIMELINE_DUMP=<file> (diagnostics): every launch's start / end against one base event, written at every collect
    std::vector<EvPair> ev_pool;
    double prof_ms[ICSP_K_COUNT];
    long long prof_n[ICSP_K_COUNT];
    std::string err;
};

namespace {

// Tuning/diagnostic overrides read by icsp_create (documented in icsp_hip.h).  Unset: *out keeps its default and the result is
// true; set to a whole number inside [lo, hi]: taken; anything else: false (icsp_create then fails with UNCORRECT_PARAM
// instead of running in a mode nobody asked for).
bool env_int(const char* name, int lo, int hi, int* out)
{
    const char* v = getenv(name);
    if (!v) return true;
    char* end = nullptr;
    const long k = strtol(v, &end, 10);
    if (end == v || *end != 0 || k < lo || k > hi) return false;
    *out = (int)k;
    return true;
}

int collect_profile(icsp_ctx* ctx);
void gop_release(icsp_ctx* ctx);

int poison(icsp_ctx* ctx, const char* what, hipError_t e)
{
    if (!ctx->sticky) ctx->err = std::string(what) + ": " + hipGetErrorString(e) + " (launch path: the context is unusable from here on)";
    ctx->sticky = ICSP_ERR_HIP;
    return ICSP_ERR_HIP;
}

// f() launches kernels on `st`; the launch status is checked here (hipLaunchKernelGGL itself returns nothing)
template <typename F> int launch_timed(icsp_ctx* ctx, int kernel, hipStream_t st, F&& f)
{
    // (hipGetLastError returns -- and clears -- the last error of ANY earlier HIP call of this thread, the host's own included
    //  (torch, RCCL, a tolerated failure of ours): drained first, so that what is read after f() is f()'s.  Our own earlier
    //  launches were checked when they were made.)
    (void)hipGetLastError();
    if (!ctx->profiling || !((ctx->prof_mask >> kernel) & 1u)) { f(); HIPQ(hipGetLastError()); return 0; }
    if (ctx->ev_pool.empty() && ctx->ev_pending.size() >= 8192) collect_profile(ctx);     // keeps the list bounded (this one blocks)
    EvPair e;
    if (!ctx->ev_pool.empty()) { e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
    else {
        // no event to be had: the launch itself must still happen, it just goes untimed
        if (hipEventCreate(&e.a) != hipSuccess) { (void)hipGetLastError(); f(); HIPQ(hipGetLastError()); return 0; }
        if (hipEventCreate(&e.b) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(e.a); f(); HIPQ(hipGetLastError()); return 0; }
    }
    e.kernel = kernel;
    e.sid = st == ctx->stream ? 0 : st == ctx->stream2 ? 1 : st == ctx->pstream[1] ? 2 : st == ctx->pstream[2] ? 3 : 4;
    const hipError_t ea = hipEventRecord(e.a, st);
    f();
    const hipError_t el = hipGetLastError();
    const hipError_t eb = hipEventRecord(e.b, st);
    if (ea == hipSuccess && eb == hipSuccess) ctx->ev_pending.push_back(e);
    else { (void)hipGetLastError(); ctx->ev_pool.push_back(e); }       // timing events order nothing: a failed record only loses the sample
    HIPQ(el);
    return 0;
}
#define LT(...) do { if (int rc_ = launch_timed(__VA_ARGS__)) return rc_; } while (0)

int collect_profile(icsp_ctx* ctx)
{
    for (auto& e : ctx->ev_pending) {
        float ms = 0;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            ctx->prof_ms[e.kernel] += ms; ctx->prof_n[e.kernel] += 1;
            float t0 = 0;
            if (ctx->tl_file && hipEventElapsedTime(&t0, ctx->tl_base, e.a) == hipSuccess) fprintf(ctx->tl_file, "%d %d %.1f %.1f\n", e.kernel, e.sid, t0 * 1e3, (t0 + ms) * 1e3);
        } else (void)hipGetLastError();
        ctx->ev_pool.push_back(e);
    }
    ctx->ev_pending.clear();
    return 0;
}

int check_range(icsp_ctx* ctx, int first, int n)
{
    if (first < 0 || n < 0 || (long long)first + n > ctx->max_frames) return ICSP_ERR_RANGE;
    return 0;
}

// waves per I-frame workgroup: enough for the widest step of the block wavefront (2 blocks per wave)
int intra_waves_needed(const Geo& g)
{
    int widest = 0;
    const int nsteps = g.cols8 + 2 * (g.rows8 - 1);
    for (int t = 0; t < nsteps; t++) {
        int r_lo = t - (g.cols8 - 1); r_lo = (r_lo <= 0) ? 0 : (r_lo + 1) >> 1;
        int r_hi = (g.rows8 - 1 < (t >> 1)) ? g.rows8 - 1 : (t >> 1);
        if (r_hi - r_lo + 1 > widest) widest = r_hi - r_lo + 1;
    }
    return (widest + 1) / 2;
}

// the 8-lane kernel with block rows chained in groups of gc (k_intra_luma8<.., gc>): waves of eight slots that the widest step needs,
// slots starting at a multiple of gc
int intra_waves_chained(const Geo& g, int gc)
{
    int widest = 0;
    const int nsteps = g.cols8 + (g.rows8 - 1) + (g.rows8 - 1) / gc;
    for (int t = 0; t < nsteps; t++) {
        const int tp = t - (g.cols8 - 1);
        const int r_first = tp <= 0 ? 0 : gc * (tp / (gc + 1)) + std::min(tp % (gc + 1), gc);
        const int r_last = std::min(g.rows8 - 1, gc * (t / (gc + 1)) + std::min(t % (gc + 1), gc - 1));
        widest = std::max(widest, r_last - (r_first & ~(gc - 1)) + 1);
    }
    return (widest + 7) / 8;
}

void launch_intra_luma(icsp_ctx* ctx, const Geo& g, const FrameSel& fs, const DevBufs& b, int G, int G_all, hipStream_t st, bool beside_p_steps = false);

// Orders `stream` after everything queued on the context's other streams (chroma stream, GOP-group streams): called by whatever
// reads results, uploads, decodes, or encodes a range that partly overlaps one in flight.  No range is "in flight" afterwards.
int join_all(icsp_ctx* ctx)
{
    // whatever was un-joined is now queued in front of `stream` only: the next encode must make the other streams follow it
    // (fork_all), whoever the caller is (callers used to set st_ahead themselves, and one path did not)
    bool any = ctx->s2_dirty || ctx->p_dirty;
    for (auto& f : ctx->flight) any = any || f.used;
    if (any) ctx->st_ahead = true;
    if (ctx->s2_dirty) {
        HIPQ(hipEventRecord(ctx->ev_join, ctx->stream2));
        HIPQ(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        ctx->s2_dirty = false;
    }
    if (ctx->p_dirty) {
        for (int k = 1; k < kMaxPGroups; k++)
            if (ctx->pstream[k]) { HIPQ(hipEventRecord(ctx->ev_pjoin[k], ctx->pstream[k])); HIPQ(hipStreamWaitEvent(ctx->stream, ctx->ev_pjoin[k], 0)); }
        ctx->p_dirty = false;
    }
    for (auto& f : ctx->flight) f.used = false;
    return 0;
}

// the chroma stream and every GOP-group stream that exists follow what has been queued on `stream` so far (uploads, a join)
int fork_all(icsp_ctx* ctx)
{
    HIPQ(hipEventRecord(ctx->ev_fork, ctx->stream));
    if (ctx->stream2) HIPQ(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    for (int k = 1; k < kMaxPGroups; k++) if (ctx->pstream[k]) HIPQ(hipStreamWaitEvent(ctx->pstream[k], ctx->ev_fork, 0));
    ctx->st_ahead = false;
    return 0;
}

// The streams of a device's contexts are kept when a context goes and handed to the next one that asks (by priority).  Not to save
// their set-up time alone: WHICH hardware queue a stream gets depends on how many streams the process made before it, and the
// regime that keeps four streams busy (two IPPP ranges alternating: two chains, two I streams) runs 17 % slower when two other
// streams -- never used -- were made first, 35 % slower behind three, whatever GPU_MAX_HW_QUEUES says; streams made AFTER the four
// cost nothing (tools/exp_prelude.py, profiles/r05_stream_order.txt).  So a process's first context makes its streams early (the
// transfer streams come behind them, icsp_copy_streams) and later contexts inherit them instead of making new ones behind
// whatever else the process has created since.  A poisoned context's streams are destroyed, not kept.
struct StreamPool { std::mutex m; std::vector<std::pair<int, hipStream_t>> idle[64]; };
StreamPool& stream_pool() { static StreamPool* p = new StreamPool; return *p; }      // (never destroyed: contexts may go during static destruction)
hipError_t stream_get(int slot, int prio, hipStream_t* out)
{
    {
        StreamPool& sp = stream_pool();
        std::lock_guard<std::mutex> l(sp.m);
        auto& v = sp.idle[slot & 63];
        for (size_t i = 0; i < v.size(); i++)
            if (v[i].first == prio) { *out = v[i].second; v.erase(v.begin() + (long)i); return hipSuccess; }
    }
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
}
void stream_put(int slot, int prio, hipStream_t s, bool healthy)
{
    if (!s) return;
    if (!healthy || hipStreamQuery(s) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamDestroy(s); return; }
    StreamPool& sp = stream_pool();
    std::lock_guard<std::mutex> l(sp.m);
    sp.idle[slot & 63].emplace_back(prio, s);
}

// stream2 (I-frame chroma / all I-frame kernels) is created by the first encode or decode of a context that is not in
// single-stream mode: a stream costs 10-25 ms of set-up on this runtime (a hardware queue + its 4 MB and 16 MB buffers)
int second_stream(icsp_ctx* ctx)
{
    if (ctx->stream2 || ctx->single) return 0;
    HIPCHK(stream_get(ctx->slot, ctx->prio_lo, &ctx->stream2));
    ctx->st_ahead = true;                               // ordered after nothing yet: the next encode forks
    return 0;
}

// the streams of GOP groups 1.. are created by the first encode that needs them (a stream costs 2-10 ms), or by icsp_prepare
int group_streams(icsp_ctx* ctx, int ng)
{
    for (int k = 1; k < ng && k < kMaxPGroups; k++)
        if (!ctx->pstream[k]) {
            HIPCHK(stream_get(ctx->slot, ctx->prio_hi, &ctx->pstream[k]));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_pjoin[k], hipEventDisableTiming));
            ctx->st_ahead = true;                       // a new stream is ordered after nothing: the next encode forks
        }
    return 0;
}

// Admission of an encode of slots [first, first + n): finds or makes the range's flight record.
//   *same: the very same range is in flight (and allow_same): the new pass follows the old one stream by stream, no fork, no join;
//   *joined: the range partly overlapped one in flight (or the table was full): everything was joined onto `stream`, the
//            caller must fork.  A range disjoint from everything in flight needs neither.
//   whole: the placement this call wants (see Flight); a record of the same range with the other placement is a conflict
//          like a partial overlap (the range's frames would change streams), resolved the same way.
//   list (k >= 2 pairs of first, n; encode_many): first / n are then the list's hull, but overlap and identity are decided range by
//         range -- two lists whose ranges interleave are as independent as two disjoint ranges; a plain range and a list are never
//         "the same".
int flight_admit(icsp_ctx* ctx, int first, int n, bool allow_same, bool whole, Flight** out, bool* same, bool* joined, int k = 0, const int* list = nullptr)
{
    *same = false; *joined = false;
    Flight* hit = nullptr;
    bool overlap = false;
    auto ranges_of = [](const Flight& f, int i, int& a, int& m) { if (f.many_k) { a = f.many_list[2 * i]; m = f.many_list[2 * i + 1]; } else { a = f.first; m = f.n; } };
    for (auto& f : ctx->flight) {
        if (!f.used) continue;
        const int fk = f.many_k ? f.many_k : 1, nk = k ? k : 1;
        bool equal = (f.many_k > 0) == (k > 0) && fk == nk;
        for (int i = 0; equal && i < nk; i++) {
            int a, m; ranges_of(f, i, a, m);
            equal = k ? (a == list[2 * i] && m == list[2 * i + 1]) : (a == first && m == n);
        }
        if (equal) { hit = &f; continue; }
        for (int i = 0; i < nk && !overlap; i++) {
            const int a0 = k ? list[2 * i] : first, m0 = k ? list[2 * i + 1] : n;
            for (int j = 0; j < fk && !overlap; j++) { int a, m; ranges_of(f, j, a, m); overlap = a0 < a + m && a < a0 + m0; }
        }
    }
    if (hit && allow_same && !overlap && hit->whole == whole) { *same = true; *out = hit; return 0; }
    Flight* slot = nullptr;
    if (!hit && !overlap) for (auto& f : ctx->flight) if (!f.used) { slot = &f; break; }
    if (!slot) {
        if (int rc = join_all(ctx)) return rc;
        *joined = true;
        slot = &ctx->flight[0];
    }
    slot->used = true; slot->first = first; slot->n = n; slot->whole = whole; slot->sidx = 0; slot->done_valid = false; slot->many_k = 0;
    *out = slot;
    return 0;
}

int flight_events(icsp_ctx* ctx, Flight* f, int ng)
{
    for (int k = 0; k < ng && k < kMaxPGroups; k++)
        if (!f->ev_p1[k]) HIPCHK(hipEventCreateWithFlags(&f->ev_p1[k], hipEventDisableTiming));
    if (f->whole && !f->ev_done) HIPCHK(hipEventCreateWithFlags(&f->ev_done, hipEventDisableTiming));
    return 0;
}

// The three launches of one P step over the frames `fs` selects (the i-th frame of every GOP of a chain) on stream sk.
int launch_p_step(icsp_ctx* ctx, const DevBufs& b, const FrameSel& fs, hipStream_t sk)
{
    const Geo& g = ctx->g;
    // small frames: the four-state search rides in the serial kernel's launch (one kernel boundary less per step;
    // nobody waits inside that launch, see k_serial_fused); else two launches
    const bool fused = p_step_fusable(g) && !ctx->no_fuse;
    LT(ctx, ICSP_K_ME, sk, [&] { me_search(g, fs, b, !fused, sk); });
    LT(ctx, ICSP_K_FRAME_SERIAL, sk, [&] { frame_serial(g, fs, b, fused, sk); });
    LT(ctx, ICSP_K_RESIDUAL, sk, [&] { residual(g, fs, b, true, sk); });
    return 0;
}

// Where a call's range goes, from the calls before it alone (no device state: icsp_debug_plan_turns runs it on the CPU,
// tests/test_turns.py).  whole: the range is disjoint from the previous call's -- the caller alternates between independent ranges;
// three: so are this one, the previous one and the one before that, all-intra: three chain streams in turn; turn: the chain stream
// (0 = `stream`, 1, 2 = pstream[1], pstream[2]) a whole range takes.  The state moves on as a side effect.
struct TurnState { int last_first, last_n, prev2_first, prev2_n, rr; };
struct Turn { bool whole, three; int turn; };
Turn plan_turn(TurnState& t, int first, int n, int L, bool may_whole, bool chains3, const std::vector<std::pair<int, int>>* last_list = nullptr)
{
    auto apart = [](int a, int an, int b_, int bn) { return a >= b_ + bn || b_ >= a + an; };
    Turn r{ false, false, 0 };
    bool apart_last = apart(first, n, t.last_first, t.last_n);
    if (!apart_last && last_list && !last_list->empty()) {           // the previous call was a list: range by range, not its hull
        apart_last = true;
        for (auto& q : *last_list) apart_last = apart_last && apart(first, n, q.first, q.second);
    }
    r.whole = may_whole && t.last_n > 0 && apart_last;
    r.three = chains3 && r.whole && L == 1 && t.prev2_n > 0 && apart(first, n, t.prev2_first, t.prev2_n) &&
              apart(t.last_first, t.last_n, t.prev2_first, t.prev2_n);
    t.prev2_first = t.last_first; t.prev2_n = t.last_n;
    t.last_first = first; t.last_n = n;
    if (r.whole) {
        if (!r.three) t.rr &= 1;
        r.turn = t.rr;
        t.rr = r.three ? (t.rr + 1) % 3 : (t.rr ^ 1);
    }
    return r;
}

int encode_range(icsp_ctx* ctx, int first, int n)
{
    const Geo& g = ctx->g;
    const int L = ctx->p.intra_period > 0 ? ctx->p.intra_period : 1;
    if (first % L != 0) return ICSP_ERR_RANGE;
    if (n == 0) return 0;
    DevBufs b = ctx->b;
    if (!ctx->keep_coef) b.coef = nullptr;
    if (int rc = second_stream(ctx)) return rc;
    const bool single = ctx->single;                   // everything on `stream`: no cross-stream ordering at all
    hipStream_t st = ctx->stream, s2 = single ? ctx->stream : ctx->stream2;
    const int G = (n + L - 1) / L;
    const int cwgs = chroma_wgs_per_frame(g);                         // k_residual8 workgroups per frame, chroma waves only
    // GOP groups (L > 1): a P step is a chain of dependent kernels of which the serial one is latency-bound (one workgroup
    // per frame) and leaves most of the chip idle, so the GOPs are split into groups, each running its own chain on its own
    // stream: one group's serial kernel overlaps the others' search and residual kernels.  Groups touch disjoint frames,
    // hence disjoint slots of every buffer.
    const bool lazy = !ctx->always_sync;
    // A caller that alternates between independent ranges (this call's range is not the previous call's and does not touch
    // it) gets every range WHOLE on one of the two chain streams, taking turns: two whole batches side by side keep twice the
    // frames in flight that the two halves of one batch do.
    // Three all-intra ranges in rotation (this one, the one before and the one before that pairwise disjoint): THREE chain streams in
    // turn, so that three whole batches are in flight instead of two -- with the chroma stream that makes the four streams the runtime
    // has hardware queues for.  CIF, frames/s with two chain streams -> three: three ranges of 100 / 150 / 200 / 220 frames 0.91 -> 1.03,
    // 1.10 -> 1.48, 1.43 -> 1.89, 1.58 -> 2.03 M; of 250 / 300 / 350 / 400 frames 1.76 -> 2.16, 1.72 -> 2.16, 1.92 -> 2.34, 1.98 -> 2.34 M; four
    // of 75 / 150 / 300: 0.70 -> 0.95, 1.12 -> 1.46, 1.70 -> 2.10 M; 352x576 3 x 100: 0.53 -> 0.64 M, 704x576 3 x 50 / 100: 0.17 -> 0.25, 0.31 ->
    // 0.39 M, 720p 3 x 30: 58 -> 86 k, 1088p 3 x 15: 16 -> 24 k (profiles/r05_exp_chains3.txt).  Two ranges alternating stay as they were
    // (a range follows its own previous pass: two in flight is all there can be).
    TurnState ts{ ctx->last_first, ctx->last_n, ctx->prev2_first, ctx->prev2_n, ctx->rr };
    const Turn turn = plan_turn(ts, first, n, L, !single && lazy && ctx->whole_ok, ctx->chains3, &ctx->last_list);
    ctx->last_list.clear();
    const bool whole = turn.whole, three = turn.three;
    ctx->last_first = ts.last_first; ctx->last_n = ts.last_n; ctx->prev2_first = ts.prev2_first; ctx->prev2_n = ts.prev2_n; ctx->rr = ts.rr;
    int NG = ctx->p_groups;
    // keep every group's launches wide enough to be worth splitting: a dozen GOPs per group (tools/sweep_regimes.py, one CIF range
    // encoded again and again: 10 GOPs 0.358 M frames/s in one group against 0.341 M in two, 20 GOPs 0.679 / 0.656, 25 GOPs 0.762 / 0.812)
    // Frames of 2048 macroblocks and more run the serial kernel as a launch of its own (60 us at 1088p, the chip idle beside it): there
    // three GOPs per group are enough -- 1088p, one range again and again, 6 GOPs 45.8 -> 49.1 k frames/s, 13 GOPs (a rank's share of
    // configs[4] split eight ways) 62.8 -> 68.9 k, 25 GOPs and more two groups either way.  (Splitting 4CIF / 720p / 352x576 ranges
    // earlier than a dozen GOPs per group loses 1-4 %: profiles/r06_ab_group_threshold.txt.)
    const int min_gops = g.nmb >= 2048 ? 3 : 12;
    if (NG > G / min_gops) NG = G / min_gops;
    if (NG < 1 || L == 1 || whole || single) NG = 1;
    auto group_lo = [&](int k) { return (int)((long long)G * k / NG); };
    // Which ranges are in flight decides the ordering against earlier calls (flight_admit): the same range again, or a range
    // disjoint from all of them -- the next chunk of a clip, the reference's independent GOP jobs (ENC:186-213) -- is not
    // ordered against them at all; the chroma / group streams follow `stream` (fork) only when it carries something they must
    // wait for (an upload, a join).
    ctx->last_whole = whole; ctx->last_groups = NG;
    Flight* F = nullptr;
    bool same = false, joined = false;
    if (int rc = flight_admit(ctx, first, n, lazy, whole, &F, &same, &joined)) return rc;
    if (int rc = flight_events(ctx, F, NG)) return rc;
    bool moved = false;                                // a whole range on the other stream than its previous pass
    if (whole) {
        if (int rc = group_streams(ctx, three ? 3 : 2)) return rc;
        moved = same && F->sidx != turn.turn;
        F->sidx = turn.turn;                             // calls take the chain streams in turn (plan_turn)
    }
    // stream of chain / part k of this range
    auto chain_stream = [&](int k) { return whole ? (F->sidx ? ctx->pstream[F->sidx] : st) : (k == 0 ? st : ctx->pstream[k]); };
    if (moved && F->done_valid) HIPQ(hipStreamWaitEvent(chain_stream(0), F->ev_done, 0));
    if (L == 1) {
        // ---- all-intra: the I frame of every GOP.  Chroma of an I frame does not depend on its luma (no pixel prediction,
        //      ENC:4347-4349), so its kernels run on a second stream beside the latency-bound luma wavefront kernel.
        FrameSel fs{ first, L, G, nullptr };
        // More frames than CUs: some CUs carry two frames and finish half again as late as the others, and a launch lasts as
        // long as its slowest workgroup.  Two launches on two streams (unequal parts, so that they do not fall into step), each
        // following only what its own stream carries, keep the early finishers busy: a part starts as soon as the part before
        // it on its stream is through (300 CIF frames: 0.94 M -> 1.04 M frames/s; 600: +1 %).
        // (... unless one launch fills the CUs evenly, two workgroups each: from 1.66 to 2 frames per CU -- CIF, one range again and again,
        //  450 / 500 / 512 frames: 1.32 / 1.43 / 1.47 M frames/s in two parts against 1.40 / 1.54 / 1.57 M in one; 400 frames 1.33 / 1.28,
        //  560 frames 1.61 / 1.50; 352x576 the same)
        const bool even2 = 20 * G >= 33 * ctx->n_cu && G <= 2 * ctx->n_cu;
        const int NGI = (G > ctx->n_cu && !whole && !single && !even2) ? ctx->i_groups : 1;
        // the chroma launch may take the one-workgroup-per-CU form (below): frames whose luma workgroups have at most three waves
        // (the room left on a CU was measured for those), and as long as a CU's share of the chroma units, at 4.5 us each, stays
        // within 0.85 of the luma step (1.67 us per wavefront step with two batches in flight) -- CIF: up to 367 frames
        const int luma_steps = g.cols8 + 2 * (g.rows8 - 1);
        // (three batches in flight: only while they leave the CUs room, up to 2.75 luma workgroups per CU -- three CIF ranges of 220 frames
        //  2.03 M frames/s with the one-per-CU launch, 1.96 M without; of 250 frames 1.80 M with, 2.16 M without -- and up to 2 per CU for
        //  the longer-lived workgroups of tall frames: 352x576, three ranges of 150 / 175 / 200 frames +1 % / level / -17 % with it)
        const bool tall = g.rows8 * 2 >= g.cols8 * 3;
        const bool cap_ok = whole && ctx->chroma_cap && (ctx->intra_waves * 2 + 7) / 8 <= 3 &&
                            270LL * G * cwgs <= 85LL * luma_steps * ctx->n_cu &&
                            (!three || (tall ? 3LL * G <= 2LL * ctx->n_cu : 12LL * G <= 11LL * ctx->n_cu));
        if (NGI > 1) { if (int rc = group_streams(ctx, NGI)) return rc; }
        if (!single && !same && (joined || !lazy || ctx->st_ahead)) { if (int rc = fork_all(ctx)) return rc; }
        for (int k = 0; k < NGI; k++) {
            const int g0 = k == 0 ? 0 : 2 * G / 5, g1 = k + 1 == NGI ? G : 2 * G / 5;
            hipStream_t sk = chain_stream(k);
            FrameSel fk{ first + g0, L, g1 - g0, nullptr };
            // frames in flight at once (which decides the kernel form): whole placement -> another batch like this one beside it
            LT(ctx, ICSP_K_INTRA_LUMA, sk, [&] { launch_intra_luma(ctx, g, fk, b, g1 - g0, whole ? (three ? 3 : 2) * G : G, sk); });
        }
        // A range placed whole runs beside another range's luma launch, and with up to about 1.4 frames per CU its chroma launches
        // have slack on the second stream (0.14 ms of kernels per 0.2 ms step at 300 frames).  Left to itself k_residual8 fills
        // every CU with eight workgroups of four waves, and the luma workgroups of the next launch, which have to start together to
        // end together, find the CUs full and land unevenly: luma launches of 415-525 us beside chroma against 385 us alone.
        // So there the chroma blocks go to ONE workgroup per CU (k_residual8_strided: n_cu workgroups, each every n_cu-th
        // unit), which also RESERVES LDS it does not use -- 77 KB with the reservation: the dispatcher then cannot put two of
        // them on a CU that holds luma workgroups, and room for three luma workgroups is always left.  Measured, two alternating
        // 300-frame batches: plain 1.44 M frames/s; k_residual8 itself with the reservation (one workgroup per CU at a time, but
        // a dispatch between any two) 1.51 M, its chroma launch 78 -> 165 us; strided with the reservation 1.57 M (0.13 ms, and
        // the step is the luma chains' again); strided without it 1.44 M, two workgroups per CU 1.49 M.  Two batches of 280 / 320 /
        // 350 frames: 1.38 -> 1.51, 1.50 -> 1.64, 1.59 -> 1.70 M; three batches of 300 in rotation 1.39 -> 1.52 M; 352x576, two
        // batches of 300: 0.79 -> 0.85 M.  With more frames per batch the one-per-CU launch no longer fits the step (two batches of
        // 400: 1.72 -> 1.52 M, of 600: 1.98 -> 1.61 M; a batch on its own, 3390 frames: 2.26 -> 2.01 M), and larger frames have larger
        // luma workgroups (4CIF, two batches of 300: 0.47 -> 0.39 M; 720p -5 %, 1088p -10 %): there, and beside the 32-lane luma
        // form (CIF, 250 frames: 1.44 -> 1.31 M), the chroma launch is the plain one (cap_ok above).
        // (A chroma stream of its own for the second chain's range does not help: the two launches then share the one
        //  workgroup slot per CU -- 0.25 ms each, 1.43 M frames/s.)
        const bool cap = cap_ok && ctx->last_form == 8;
        const size_t cap_lds = cap ? (size_t)ctx->chroma_cap * 1024 : 0;
        LT(ctx, ICSP_K_CHROMA_DC, s2, [&] { chroma_dc(g, fs, b, s2); });
        if (cap)    // one workgroup per CU, each taking every n_cu-th unit: no dispatch between a CU's units (0.165 -> 0.13 ms)
            LT(ctx, ICSP_K_RESIDUAL, s2, [&] { residual_one_per_cu(g, fs, b, ctx->n_cu, cap_lds, s2); });
        else
            LT(ctx, ICSP_K_RESIDUAL, s2, [&] { residual(g, fs, b, false, s2); });
        if (!single) ctx->s2_dirty = true;
        if (NGI > 1 || (whole && F->sidx)) ctx->p_dirty = true;
        if (whole) { HIPQ(hipEventRecord(F->ev_done, chain_stream(0))); F->done_valid = true; }
        if (!lazy) { if (int rc = join_all(ctx)) return rc; }
        return 0;
    }
    // ---- IPPP.  The I frames of all groups run on stream2 (chroma kernels, then the luma wavefront kernel); every group's P
    // steps are one chain on the group's own stream, which waits for the I frames.  When the same range is encoded again while
    // in flight, the passes overlap as far as the data allow: a group's chain follows its own previous pass in stream order,
    // and the I frames only wait for the FIRST P step of every group's previous pass over that range -- the last reader of what
    // the I kernels overwrite (the I frames' reconstruction; later P steps read and write their own slots and the slot before
    // them only).  So the I frames of the next pass, a latency-bound launch on a few CUs, run beside P steps 2.. of this one
    // -- or beside another range's P steps -- instead of in front of an idle chip.
    if (int rc = group_streams(ctx, NG)) return rc;
    // The I frames of two alternating ranges are latency-bound launches on a few CUs each (0.22 ms for 30 CIF frames); following each
    // other on stream2 they set the pace of that regime (0.43 ms per round of two ranges).  So those of a range placed whole on
    // chain stream 1 get a stream of their own (pstream[2]): two ranges of 30 GOPs alternating 1.235 -> 1.37 M frames/s, three in
    // rotation 1.19 -> 1.32 M, two of 15 GOPs 0.65 -> 0.97 M, two of 60 GOPs +1 % (profiles/r05_exp_istream.txt; ICSP_I_STREAM_B=0:
    // as before.  Round 3 measured the same idea as a loss, 1.05 -> 0.86 M, when the chroma kernels still rode on stream2.)
    if (whole && ctx->i_stream_b && F->sidx) { if (int rc = group_streams(ctx, 3)) return rc; s2 = ctx->pstream[2]; }
    if (single) { /* one stream: stream order is the order */ }
    else if (same) { for (int k = 0; k < NG; k++) HIPQ(hipStreamWaitEvent(s2, F->ev_p1[k], 0)); }
    else if (joined || !lazy || ctx->st_ahead) { if (int rc = fork_all(ctx)) return rc; }   // after what was queued on `stream` (uploads ...)
    {
        FrameSel fs{ first, L, G, nullptr };
        // (Rounds 3-4 took the chroma kernels of a range placed whole to the front of its chain stream, when the I frames of both
        //  alternating ranges shared stream2 and set the pace.  With an I stream per chain it is the chains that set it, and the
        //  chroma kernels are better off in front of the luma kernel again: two ranges of 30 GOPs 1.38 -> 1.43 M frames/s, three in
        //  rotation 1.33 -> 1.37 M, two of 15 GOPs 0.96 -> 0.99 M, of 60 GOPs level -- profiles/r05_exp_istream.txt.)
        LT(ctx, ICSP_K_CHROMA_DC, s2, [&] { chroma_dc(g, fs, b, s2); });
        LT(ctx, ICSP_K_RESIDUAL, s2, [&] { residual(g, fs, b, false, s2); });
        // (frames in flight: a range placed whole has another one's I frames, on their own stream, beside its own)
        LT(ctx, ICSP_K_INTRA_LUMA, s2, [&] { launch_intra_luma(ctx, g, fs, b, G, whole ? 2 * G : G, s2, true); });
        if (!single) {
            HIPQ(hipEventRecord(ctx->ev_join, s2));
            for (int k = 0; k < NG; k++) HIPQ(hipStreamWaitEvent(chain_stream(k), ctx->ev_join, 0));
        }
    }
    // every chain is ordered after stream2's work; `stream` itself is one of them unless the range went whole onto the group stream
    if (whole && F->sidx) ctx->s2_dirty = true; else ctx->s2_dirty = false;
    if (NG > 1 || (whole && F->sidx)) ctx->p_dirty = true;
    // P step i of GOP group k on stream sk; -1: no GOP of the group has a frame i
    auto p_step = [&](int k, int i, hipStream_t sk) -> int {
        const int g0 = group_lo(k), g1 = group_lo(k + 1);
        int Gi = 0;
        for (int gop = g0; gop < g1; gop++) if (gop * L + i < n) Gi++;
        if (Gi == 0) return -1;
        FrameSel fs{ first + g0 * L + i, L, Gi, nullptr };
        return launch_p_step(ctx, b, fs, sk);
    };
    for (int i = 1; i < L; i++) {
        bool any = false;
        for (int k = 0; k < NG; k++) {
            hipStream_t sk = chain_stream(k);
            const int rc = p_step(k, i, sk);
            if (rc > 0) return rc;
            any = any || rc == 0;
            if (i == 1) HIPQ(hipEventRecord(F->ev_p1[k], sk));             // the I frames of the next pass over this range may start
        }
        if (!any) break;
    }
    if (whole) { HIPQ(hipEventRecord(F->ev_done, chain_stream(0))); F->done_valid = true; }
    // a zero-copy consumer on `stream` (icsp_device_view) must find every group's results ordered before it
    if (!lazy) { if (int rc = join_all(ctx)) return rc; }
    return 0;
}

// The slot tables of every flight record, made once per context by its first list, OUTSIDE the flight logic (growing a
// record's tables after its admission joined everything -- which also cleared the record just admitted -- and stalled the host).
// Ranges start at multiples of L and are disjoint, so a list holds at most ceil(max_frames / L) GOPs: L rows of that many slots.
int many_tables(icsp_ctx* ctx, int L)
{
    const int cap = L * ((ctx->max_frames + L - 1) / L);
    for (auto& f : ctx->flight) {
        if (f.tab_cap >= cap) continue;
        if (hipMalloc((void**)&f.d_tab, sizeof(int) * cap) != hipSuccess || hipHostMalloc((void**)&f.h_tab, sizeof(int) * cap, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError(); ctx->err = "hipMalloc slot tables"; return ICSP_ERR_MEM_ALLOC;
        }
        f.tab_cap = cap;
    }
    return 0;
}

// Several disjoint resident ranges as ONE batch (icsp_encode_resident_many): every kernel of a step is launched once over the frames
// of all ranges -- slot tables instead of arithmetic progressions (FrameSel::table) -- so that a host holding several short ranges
// (chunks of different clips, the ends of GOP shards) gets the launches of one long range: four ranges of 150 CIF frames cost four
// launches of 150 workgroups per pass when given one by one, one launch of 600 here.  Placement: everything on ONE of the two chain
// streams, calls taking them in turn (the `whole` placement of encode_range), one GOP group.  Ordering against earlier calls range by
// range (flight_admit): lists whose ranges interleave are independent of each other.
int encode_many(icsp_ctx* ctx, int k, const int* firsts, const int* ns)
{
    const Geo& g = ctx->g;
    const int L = ctx->p.intra_period > 0 ? ctx->p.intra_period : 1;
    std::vector<std::pair<int, int>> rs;
    for (int r = 0; r < k; r++) {
        if (int rc = check_range(ctx, firsts[r], ns[r])) return rc;
        if (firsts[r] % L != 0) return ICSP_ERR_RANGE;
        if (ns[r] > 0) rs.emplace_back(firsts[r], ns[r]);
    }
    if (rs.empty()) return 0;
    if (rs.size() == 1 || ctx->single || ctx->always_sync) {             // (one stream / an outside consumer on it: the plain calls, in list order)
        for (auto& r : rs) if (int rc = encode_range(ctx, r.first, r.second)) return rc;
        return 0;
    }
    std::vector<std::pair<int, int>> sorted = rs;
    std::sort(sorted.begin(), sorted.end());
    for (size_t r = 1; r < sorted.size(); r++) if (sorted[r - 1].first + sorted[r - 1].second > sorted[r].first) return ICSP_ERR_RANGE;     // ranges must not overlap
    const int hull_first = sorted.front().first, hull_n = sorted.back().first + sorted.back().second - hull_first;
    // GOPs of the list, in list order; step i covers the GOPs that have a frame i
    std::vector<int> gop_first, gop_len;
    for (auto& r : rs) for (int f = 0; f < r.second; f += L) { gop_first.push_back(r.first + f); gop_len.push_back(std::min(L, r.second - f)); }
    const int G = (int)gop_first.size();
    DevBufs b = ctx->b;
    if (!ctx->keep_coef) b.coef = nullptr;
    if (int rc = second_stream(ctx)) return rc;
    if (int rc = group_streams(ctx, 2)) return rc;
    hipStream_t st = ctx->stream, s2 = ctx->stream2;
    ctx->prev2_first = ctx->last_first; ctx->prev2_n = ctx->last_n;
    ctx->last_first = hull_first; ctx->last_n = hull_n; ctx->last_whole = 1; ctx->last_groups = 1;
    ctx->last_list = rs;
    if (int rc = many_tables(ctx, L)) return rc;
    Flight* F = nullptr;
    bool same = false, joined = false;
    std::vector<int> flat;
    for (auto& r : rs) { flat.push_back(r.first); flat.push_back(r.second); }
    if (int rc = flight_admit(ctx, hull_first, hull_n, true, true, &F, &same, &joined, (int)rs.size(), flat.data())) return rc;
    if (int rc = flight_events(ctx, F, 1)) return rc;
    if (!same) {
        // the list and its tables: rows of G slots, row i compacted to the GOPs with a frame i (row 0: every GOP)
        int* nl = (int*)realloc(F->many_list, sizeof(int) * 2 * rs.size());
        if (!nl) return ICSP_ERR_MEM_ALLOC;
        F->many_list = nl; F->many_k = (int)rs.size();
        for (size_t r = 0; r < rs.size(); r++) { nl[2 * r] = rs[r].first; nl[2 * r + 1] = rs[r].second; }
        const int need = L * G;
        if (need > F->tab_cap) { ctx->err = "slot tables: a list needs more rows than max_frames allows"; return ICSP_ERR_RANGE; }     // (cannot happen: many_tables)
        // h_tab is about to be overwritten: its last upload must have left it (the device side is ordered by the join that every
        // reuse of a record for another list goes through)
        if (F->tab_up) { HIPCHK(hipEventSynchronize(F->ev_tab)); F->tab_up = false; }
        if (!F->ev_tab) HIPCHK(hipEventCreateWithFlags(&F->ev_tab, hipEventDisableTiming));
        for (int i = 0; i < L; i++) {
            int c = 0;
            for (int q = 0; q < G; q++) if (gop_len[q] > i) F->h_tab[i * G + c++] = gop_first[q] + i;
        }
        // up on `stream`, and every stream that launches from the tables follows (fork_all)
        HIPCHK(hipMemcpyAsync(F->d_tab, F->h_tab, sizeof(int) * need, hipMemcpyHostToDevice, st));
        HIPQ(hipEventRecord(F->ev_tab, st)); F->tab_up = true;
        ctx->st_ahead = true;
    }
    ctx->rr &= 1;
    bool moved = same && F->sidx != ctx->rr;
    F->sidx = ctx->rr; ctx->rr ^= 1;
    hipStream_t cs = F->sidx ? ctx->pstream[1] : st;
    if (moved && F->done_valid) HIPQ(hipStreamWaitEvent(cs, F->ev_done, 0));
    auto count_of = [&](int i) { int c = 0; for (int q = 0; q < G; q++) c += gop_len[q] > i; return c; };
    if (L == 1) {
        FrameSel fs{ 0, 0, G, F->d_tab };
        if (!same && (joined || ctx->st_ahead)) { if (int rc = fork_all(ctx)) return rc; }
        LT(ctx, ICSP_K_INTRA_LUMA, cs, [&] { launch_intra_luma(ctx, g, fs, b, G, 2 * G, cs); });
        LT(ctx, ICSP_K_CHROMA_DC, s2, [&] { chroma_dc(g, fs, b, s2); });
        LT(ctx, ICSP_K_RESIDUAL, s2, [&] { residual(g, fs, b, false, s2); });
        ctx->s2_dirty = true;
        if (F->sidx) ctx->p_dirty = true;
        HIPQ(hipEventRecord(F->ev_done, cs)); F->done_valid = true;
        return 0;
    }
    // IPPP: the I frames of every GOP on stream2 (or the second I stream), then one chain of P steps on the chain stream (encode_range's order of events)
    if (ctx->i_stream_b && F->sidx) { if (int rc = group_streams(ctx, 3)) return rc; s2 = ctx->pstream[2]; }
    if (same) HIPQ(hipStreamWaitEvent(s2, F->ev_p1[0], 0));
    else if (joined || ctx->st_ahead) { if (int rc = fork_all(ctx)) return rc; }
    {
        FrameSel fs{ 0, 0, G, F->d_tab };
        LT(ctx, ICSP_K_CHROMA_DC, s2, [&] { chroma_dc(g, fs, b, s2); });
        LT(ctx, ICSP_K_RESIDUAL, s2, [&] { residual(g, fs, b, false, s2); });
        LT(ctx, ICSP_K_INTRA_LUMA, s2, [&] { launch_intra_luma(ctx, g, fs, b, G, 2 * G, s2, true); });
        HIPQ(hipEventRecord(ctx->ev_join, s2));
        HIPQ(hipStreamWaitEvent(cs, ctx->ev_join, 0));
    }
    if (F->sidx) { ctx->s2_dirty = true; ctx->p_dirty = true; } else ctx->s2_dirty = false;
    auto p_step = [&](int i, hipStream_t sk) -> int {               // -1: no GOP of the list has a frame i
        const int Gi = count_of(i);
        if (Gi == 0) return -1;
        FrameSel fs{ 0, 0, Gi, F->d_tab + i * G };
        return launch_p_step(ctx, b, fs, sk);
    };
    for (int i = 1; i < L; i++) {
        const int rc = p_step(i, cs);
        if (rc > 0) return rc;
        if (i == 1) HIPQ(hipEventRecord(F->ev_p1[0], cs));              // the I frames of the next pass over this list may start
        if (rc < 0) break;
    }
    HIPQ(hipEventRecord(F->ev_done, cs)); F->done_valid = true;
    return 0;
}

// Decode slots [first, first+n) from the syntax arrays in place (levels, mpm, mvd) into recon.  One launch resolves every
// frame's serial chains, then the I frames' luma wavefront runs beside their chroma, then one parallel launch per P step.
int decode_range(icsp_ctx* ctx, int first, int n)
{
    const Geo& g = ctx->g;
    const int L = ctx->p.intra_period > 1 ? ctx->p.intra_period : 1;     // header period 0 or 1: every frame intra (DEC.h:293)
    if (first % L != 0) return ICSP_ERR_RANGE;
    if (n == 0) return 0;
    DevBufs b = ctx->b;
    b.coef = nullptr;
    if (int rc = second_stream(ctx)) return rc;
    const bool single = ctx->single;
    hipStream_t st = ctx->stream, s2 = single ? ctx->stream : ctx->stream2;
    const int G = (n + L - 1) / L;
    if (int rc = join_all(ctx)) return rc;
    ctx->st_ahead = true;
    LT(ctx, ICSP_K_DECODE, st, [&] {
        dec_serial(g, first, n, L, b, st);
    });
    {
        FrameSel fs{ first, L, G, nullptr };
        if (!single) { HIPQ(hipEventRecord(ctx->ev_fork, st)); HIPQ(hipStreamWaitEvent(s2, ctx->ev_fork, 0)); }
        LT(ctx, ICSP_K_DECODE, st, [&] {
            const int diag = g.rows8 < g.cols8 ? g.rows8 : g.cols8;              // widest anti-diagonal, 2 blocks per wave
            const int need = (diag + 1) / 2;
            const int nw = (G > ctx->n_cu) ? (need < 8 ? need : 8) : need;
            dec_intra_luma(g, fs, b, nw, st);
        });
        LT(ctx, ICSP_K_DECODE, s2, [&] { dec_blocks(g, fs, b, 4, 2, 0, s2); });
        if (!single) { HIPQ(hipEventRecord(ctx->ev_join, s2)); HIPQ(hipStreamWaitEvent(st, ctx->ev_join, 0)); }
    }
    for (int i = 1; i < L; i++) {
        int Gi = 0;
        for (int gop = 0; gop < G; gop++) if (gop * L + i < n) Gi++;
        if (Gi == 0) break;
        FrameSel fs{ first + i, L, Gi, nullptr };
        LT(ctx, ICSP_K_DECODE, st, [&] { dec_blocks(g, fs, b, 0, 6, 1, st); });
    }
    return 0;
}

// Which form of the I-frame luma kernel a launch takes.  Three forms:
//   32-lane (k_intra_luma32): two blocks per wave, as many waves as the widest wavefront step needs -- the latency form, best while
//     every frame has a CU of its own (with more frames than CUs, capped at 8 waves x 128 VGPRs so that two workgroups share a CU);
//   8-lane, block rows chained in PAIRS (k_intra_luma8<.., 2>): 96 steps per CIF frame instead of 114, four waves -- one per SIMD --,
//     26 KB of LDS; from where frames share CUs, at every load;
//   8-lane plain (k_intra_luma8<.., 0>): frames whose widest pairs step does not fit eight waves (720p, 1088p).
// The only decision with thresholds is where the latency form ends, in frames in flight per CU (G_all / CUs) by geometry class and by
// what runs beside the launch.  The thresholds are data: each a crossover measured by tools/sweep_regimes.py (profiles/r05_sweep.json:
// 300 regimes x forced knobs, this round's kernels and stream layout), named beside its entry.
enum GeoClass { GEO_CIF, GEO_TALL, GEO_4CIF, GEO_WIDE };            // by the waves of two blocks the widest plain step needs (<= 16 / 17-24 / more) and the aspect
enum Beside { BESIDE_ALONE, BESIDE_RANGE, BESIDE_P_STEPS, BESIDE_P_STEPS_MANY };   // nothing / another range's launches / P-step kernels (up to, more than 12 I frames)
struct FormRule { int geo, pairs, beside, lat_end20; const char* measured; };     // pairs, beside: -1 = any; lat_end20: the 32-lane form up to lat_end20 / 20 frames per CU
const FormRule kFormRules[] = {
    { GEO_WIDE, -1, -1,                  0,  "1280x720, 1920x1088: the 32-lane form 15-28 % behind at any load (two rounds and more per step)" },
    { GEO_4CIF,  1, BESIDE_P_STEPS,      0,  "704x576, up to 12 I frames beside P steps: pairs +3 %" },
    { GEO_4CIF,  1, BESIDE_P_STEPS_MANY, 0,  "704x576, more I frames beside P steps: pairs +2-5 % (15 I frames, one range or several; 25 with two ranges or three) except 25 I frames of ONE range (-3 %)" },
    { GEO_4CIF,  1, BESIDE_ALONE,        0,  "704x576, a launch on its own: pairs at any load (100 frames 0.167 / 0.173 M frames/s)" },
    { GEO_4CIF,  1, BESIDE_RANGE,        16, "704x576, two ranges of 100 frames alternating: 0.313 M 32-lane / 0.291 M pairs; of 250: 0.48 / 0.53 M" },
    { GEO_4CIF,  0, BESIDE_P_STEPS,      4,  "4CIF-class frames too wide for pairs, beside P steps (round 3's rule for the plain form)" },
    { GEO_4CIF,  0, BESIDE_P_STEPS_MANY, 4,  "as above" },
    { GEO_4CIF,  0, -1,                  16, "4CIF-class frames too wide for pairs (round 3's rule for the plain form)" },
    { GEO_TALL, -1, BESIDE_P_STEPS,      0,  "352x576, up to 12 I frames beside P steps: pairs +2-4 % (5 / 10 / 12 I frames, one to three ranges: profiles/r05_sweep.json)" },
    { GEO_TALL, -1, BESIDE_P_STEPS_MANY, 2,  "352x576, 13-25 I frames beside P steps: 32-lane +1-5 % (17 / 20 I frames); 30 / 50 I frames: pairs +2-5 %" },
    { GEO_TALL, -1, BESIDE_ALONE,        0,  "352x576, a launch on its own: pairs (50 frames +5.6 %, 100-150 +1 %, 175-200 -1.3 %, 300 and more +20 %)" },
    { GEO_TALL, -1, BESIDE_RANGE,        20, "352x576: two ranges of 125 frames 0.66 M 32-lane / 0.59 M pairs, of 175 frames 0.62 / 0.79 M; one frame per CU level" },
    { GEO_CIF,  -1, BESIDE_P_STEPS,      4,  "CIF I step beside P steps, up to 12 I frames in flight: 32-lane (10 GOPs of one range +5 %)" },
    { GEO_CIF,  -1, BESIDE_P_STEPS_MANY, 4,  "CIF, more: pairs from 52 I frames in flight -- one range of 40 GOPs 32-lane +17 %, of 60 GOPs pairs +1 %; two ranges (both count) of 20 GOPs 32-lane +5 %, of 27 / 30 / 40 GOPs pairs +0-4 / +3 / +5 %" },
    { GEO_CIF,  -1, -1,                  20, "CIF: pairs from one frame per CU on (two ranges of 150 / 300 / 3390 frames: 1.00 / 1.41 / 1.47 M 32-lane, 1.05 / 1.84 / 2.36 M pairs)" },
};

// G: frames of this launch; G_all: frames in flight at once (other GOP groups / the other range launch theirs beside this one)
// beside_p_steps: the I step of an IPPP range -- P-step kernels of other GOP groups / ranges share the chip with this launch
void launch_intra_luma(icsp_ctx* ctx, const Geo& g, const FrameSel& fs, const DevBufs& b, int G, int G_all, hipStream_t st, bool beside_p_steps)
{
    const int need = ctx->intra_waves;                              // waves of two blocks for the widest plain step
    const int need8 = (need * 2 + 7) / 8;                           // ... of eight blocks
    const int nwp = ctx->intra_waves_g2;                            // ... of eight blocks for the widest pairs step
    const bool pairs_ok = nwp >= 1 && nwp <= 8;
    const int geo = need > 24 ? GEO_WIDE : need > 16 ? GEO_4CIF : (g.rows8 * 2 >= g.cols8 * 3) ? GEO_TALL : GEO_CIF;
    const int beside = beside_p_steps ? (G_all > 12 ? BESIDE_P_STEPS_MANY : BESIDE_P_STEPS) : (G_all == G ? BESIDE_ALONE : BESIDE_RANGE);
    int form = ctx->force_intra_form;
    if (!form) {
        int lat_end20 = 20;
        for (const FormRule& r : kFormRules)
            if (r.geo == geo && (r.pairs < 0 || r.pairs == (pairs_ok ? 1 : 0)) && (r.beside < 0 || r.beside == beside)) { lat_end20 = r.lat_end20; break; }
        form = 20 * G_all > lat_end20 * ctx->n_cu ? 8 : 32;
        if (ctx->force_intra_group) form = 8;                      // (asking for a wavefront of the 8-lane kernel asks for that kernel)
    }
    ctx->last_rowgroup = 0;
    if (form == 8 && pairs_ok && ctx->force_intra_group != 1) {
        ctx->last_form = 8; ctx->last_ring = true; ctx->last_rowgroup = 2;
        ctx->last_nw = intra_luma8_pairs(g, fs, b, nwp, st);
        return;
    }
    if (form == 8) {
        // the plain wavefront (frames too wide for pairs; ICSP_INTRA_GROUP=1: any frame -- smaller ones leave waves of the six idle)
        bool ring = false;
        ctx->last_form = 8; ctx->last_nw = intra_luma8_plain(g, fs, b, need8, &ring, st); ctx->last_ring = ring;
        return;
    }
    const int nw = G_all > ctx->n_cu ? (need < 8 ? need : 8) : need;
    ctx->last_form = 32; ctx->last_ring = false;
    ctx->last_nw = intra_luma32(g, fs, b, nw, st);
}

} // namespace

// Helper threads for the staging copies of icsp_encode_gop (caller memory that is not pinned goes through pinned buffers; one
// thread copies 10-12 GB/s, the link moves 57 each way).  run(n, f) executes f(0..n-1) on the caller and the helpers and
// returns when all are done.  One job at a time (the context's calls are serialised by contract; the uploader thread of a
// call uses a pool of its own).
struct CopyPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::function<void(int)> job;
    int n_items = 0, next = 0, running = 0, gen = 0;
    bool stop = false;
    explicit CopyPool(int helpers)
    {
        for (int k = 0; k < helpers; k++) th.emplace_back([this] { loop(); });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv_job.notify_all();
        for (auto& t : th) t.join();
    }
    void loop()
    {
        int seen = 0;
        std::unique_lock<std::mutex> l(m);
        for (;;) {
            cv_job.wait(l, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            work(l);
        }
    }
    void work(std::unique_lock<std::mutex>& l)      // called with the lock held
    {
        while (next < n_items) {
            const int k = next++;
            running++;
            l.unlock();
            job(k);
            l.lock();
            running--;
        }
        if (running == 0) cv_done.notify_all();
    }
    void run(int n, std::function<void(int)> f)
    {
        std::unique_lock<std::mutex> l(m);
        job = std::move(f); n_items = n; next = 0; gen++;
        cv_job.notify_all();
        work(l);
        cv_done.wait(l, [&] { return next >= n_items && running == 0; });
    }
    // dst[0, bytes) = src[0, bytes) in slices of at least 1 MB over all threads
    void copy(void* dst, const void* src, size_t bytes)
    {
        const size_t slices = std::max<size_t>(1, std::min<size_t>(th.size() + 1, bytes >> 20));
        if (slices == 1) { memcpy(dst, src, bytes); return; }
        const size_t per = ((bytes + slices - 1) / slices + 4095) & ~(size_t)4095;
        run((int)slices, [=](int k) {
            const size_t o = (size_t)k * per;
            if (o < bytes) memcpy((char*)dst + o, (const char*)src + o, std::min(per, bytes - o));
        });
    }
};

namespace {
void gop_release(icsp_ctx* ctx)
{
    delete ctx->gop_pool; ctx->gop_pool = nullptr;
    delete ctx->up_pool; ctx->up_pool = nullptr;
    for (int k = 0; k < 2; k++) {
        if (ctx->xfer_ev_in[k]) (void)hipEventDestroy(ctx->xfer_ev_in[k]);
        if (ctx->xfer_ev_out[k]) (void)hipEventDestroy(ctx->xfer_ev_out[k]);
        ctx->xfer_ev_in[k] = ctx->xfer_ev_out[k] = nullptr; ctx->xfer_in_busy[k] = false;
        if (ctx->gop_stage_in[k]) (void)hipHostFree(ctx->gop_stage_in[k]);
        if (ctx->gop_stage_out[k]) (void)hipHostFree(ctx->gop_stage_out[k]);
        ctx->gop_stage_in[k] = ctx->gop_stage_out[k] = nullptr;
    }
    for (auto& set : ctx->gop_ev) for (auto& e : set) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    ctx->gop_stage_in_cap = ctx->gop_stage_out_cap = 0;
}

// ------------------------------------------------------------------------------------------------ transfers and caller memory
// The runtime pins the host buffer of a large transfer from PLAIN memory on the fly and keeps such pins in a cache of its own;
// a long-lived process that frees and re-allocates hundreds of megabytes there (glibc trims and re-maps the ranges) sooner or
// later has a transfer routed through a pin that no longer matches the pages behind the address: "Memory access fault by GPU"
// (round 4: about one long test session in two; tools/repro_fault.py).  So this library never hands the runtime a plain caller
// pointer: memory it KNOWS to be pinned is the DMA source / target itself, everything else goes through two pinned staging
// buffers per direction, filled / emptied by a few helper threads beside the transfer of the piece before.
// Known to be pinned: ranges of icsp_host_alloc and icsp_host_register (a table of our own), and ranges for which the runtime
// names ONE allocation that covers them whole (someone else's hipHostMalloc).  A buffer that only starts or ends inside a
// registered page -- which the runtime's per-pointer attributes report as pinned -- is not.
struct PinnedRanges {
    std::mutex m;
    std::vector<std::pair<uintptr_t, size_t>> r;
    void add(const void* p, size_t n) { std::lock_guard<std::mutex> l(m); r.emplace_back((uintptr_t)p, n); }
    void remove(const void* p)
    {
        std::lock_guard<std::mutex> l(m);
        for (size_t k = 0; k < r.size(); k++) if (r[k].first == (uintptr_t)p) { r[k] = r.back(); r.pop_back(); return; }
    }
    bool covers(const void* p, size_t n)
    {
        std::lock_guard<std::mutex> l(m);
        const uintptr_t a = (uintptr_t)p;
        for (auto& e : r) if (a >= e.first && a - e.first <= e.second && n <= e.second - (a - e.first)) return true;
        return false;
    }
};
PinnedRanges& pinned_ranges() { static PinnedRanges t; return t; }

bool host_pinned(const void* p, size_t bytes)
{
    if (!p || !bytes) return true;
    if (pinned_ranges().covers(p, bytes)) return true;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }      // unknown to the runtime: pageable
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return false;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)a.devicePointer) != hipSuccess) { (void)hipGetLastError(); return false; }
    const uintptr_t d = (uintptr_t)a.devicePointer, b = (uintptr_t)base;
    return d >= b && d - b <= size && bytes <= size - (d - b);
}

constexpr size_t kXferPiece = (size_t)16 << 20;
// both staging buffers of a direction hold at least `need` bytes (never shrinks)
int stage_reserve(icsp_ctx* ctx, bool in, size_t need)
{
    uint8_t** buf = in ? ctx->gop_stage_in : ctx->gop_stage_out;
    size_t& cap = in ? ctx->gop_stage_in_cap : ctx->gop_stage_out_cap;
    hipEvent_t* ev = in ? ctx->xfer_ev_in : ctx->xfer_ev_out;
    for (int k = 0; k < 2; k++) if (!ev[k]) HIPCHK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    if (need <= cap && buf[0] && buf[1]) return 0;
    for (int k = 0; k < 2; k++) {
        if (in && ctx->xfer_in_busy[k]) { (void)hipEventSynchronize(ev[k]); ctx->xfer_in_busy[k] = false; }     // a DMA may still read the old buffer
        if (buf[k]) (void)hipHostFree(buf[k]);
        buf[k] = nullptr;
        if (hipHostMalloc((void**)&buf[k], need, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError(); buf[k] = nullptr; cap = 0;
            ctx->err = in ? "hipHostMalloc staging (in)" : "hipHostMalloc staging (out)";
            return ICSP_ERR_MEM_ALLOC;
        }
    }
    cap = need;
    return 0;
}
CopyPool* copy_pool(CopyPool*& slot)
{
    if (!slot) slot = new (std::nothrow) CopyPool((int)std::min(5u, std::max(2u, std::thread::hardware_concurrency()) / 2));
    return slot;
}
// before the staging buffers of the upload direction are written by anything else (icsp_encode_gop's uploader)
void xfer_in_drain(icsp_ctx* ctx)
{
    for (int k = 0; k < 2; k++) if (ctx->xfer_in_busy[k]) { (void)hipEventSynchronize(ctx->xfer_ev_in[k]); ctx->xfer_in_busy[k] = false; }
}

// Host -> device on `st`.  Pinned source: one asynchronous copy.  Anything else: piece by piece through the staging pair; the call
// returns when the last piece has been COPIED into its staging buffer -- the caller's memory is no longer read, the last DMAs may
// still run (like the direct copy; a later use of the buffers waits for them).
int xfer_up(icsp_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st)
{
    if (!bytes) return 0;
    if (host_pinned(src, bytes)) { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); return 0; }
    std::lock_guard<std::mutex> stage_lock(ctx->stage_in_m);
    if (int rc = stage_reserve(ctx, true, std::min(bytes, kXferPiece))) return rc;
    CopyPool* pool = copy_pool(ctx->up_pool);
    if (!pool) return ICSP_ERR_MEM_ALLOC;
    const size_t piece = std::min(ctx->gop_stage_in_cap, kXferPiece);
    for (size_t o = 0, k = 0; o < bytes; o += piece, k++) {
        const size_t nb = std::min(piece, bytes - o);
        const int b = (int)(k & 1);
        if (ctx->xfer_in_busy[b]) { HIPCHK(hipEventSynchronize(ctx->xfer_ev_in[b])); ctx->xfer_in_busy[b] = false; }
        pool->copy(ctx->gop_stage_in[b], (const char*)src + o, nb);
        HIPCHK(hipMemcpyAsync((char*)dst + o, ctx->gop_stage_in[b], nb, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(ctx->xfer_ev_in[b], st));
        ctx->xfer_in_busy[b] = true;
    }
    return 0;
}
// Device -> host on `st`.  Pinned target: one asynchronous copy (the caller waits for the stream).  Anything else: through the
// staging pair, the copy-out of a piece beside the DMA of the next; returns when `dst` holds every byte.
int xfer_down(icsp_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st)
{
    if (!bytes || !dst) return 0;
    if (host_pinned(dst, bytes)) { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st)); return 0; }
    if (int rc = stage_reserve(ctx, false, std::min(bytes, kXferPiece))) return rc;
    CopyPool* pool = copy_pool(ctx->gop_pool);
    if (!pool) return ICSP_ERR_MEM_ALLOC;
    const size_t piece = std::min(ctx->gop_stage_out_cap, kXferPiece);
    size_t prev_o = 0, prev_nb = 0;
    for (size_t o = 0, k = 0; o < bytes; o += piece, k++) {
        const size_t nb = std::min(piece, bytes - o);
        const int b = (int)(k & 1);
        HIPCHK(hipMemcpyAsync(ctx->gop_stage_out[b], (const char*)src + o, nb, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(ctx->xfer_ev_out[b], st));
        if (k) pool->copy((char*)dst + prev_o, ctx->gop_stage_out[b ^ 1], prev_nb);       // (its DMA was waited for in the round before)
        HIPCHK(hipEventSynchronize(ctx->xfer_ev_out[b]));
        prev_o = o; prev_nb = nb;
        if (o + nb >= bytes) pool->copy((char*)dst + o, ctx->gop_stage_out[b], nb);
    }
    return 0;
}
} // namespace

// ================================================================================================ C ABI
extern "C" {

const char* icsp_strerror(int s)
{
    switch (s) {
    case ICSP_OK: return "success";
    case ICSP_ERR_UNENOUGH_PARAM: return "unenough parameters";
    case ICSP_ERR_UNCORRECT_PARAM: return "uncorrect parameters";
    case ICSP_ERR_MEM_ALLOC: return "fail memory allocation";
    case ICSP_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case ICSP_ERR_HIP: return "HIP runtime error";
    case ICSP_ERR_RANGE: return "frame range outside capacity or not GOP aligned";
    default: return "unknown reason";
    }
}

const char* icsp_last_error(const icsp_ctx_t* ctx) { return ctx ? ctx->err.c_str() : ""; }

// Test hook: ICSP_FAKE_DEVICES=N (2..64) makes the library present N devices, device d being physical device d mod (real
// devices), each with its OWN per-device records (search tables uploaded, shared transfer streams, transfer turns) -- so that
// the multi-device paths of a host (one uploader thread, one stream pair, one table upload per device) run with several
// device records on a one-GPU box.  Results never depend on it.
static int fake_devices()
{
    static const int n = [] { const char* v = getenv("ICSP_FAKE_DEVICES"); const int k = v ? atoi(v) : 0; return (k >= 2 && k <= 64) ? k : 0; }();
    return n;
}

int icsp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    return fake_devices() ? fake_devices() : n;
}

void* icsp_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    pinned_ranges().add(p, bytes ? bytes : 1);
    return p;
}

void icsp_host_free(void* p) { if (p) { pinned_ranges().remove(p); (void)hipHostFree(p); } }

// Uploads of all contexts of a device on one stream, downloads on another.  A stream's transfers go to the DMA engine its
// first copy was given -- the lowest-numbered one idle at that moment, chosen among the engines of that copy's direction -- so
// the streams of contexts set up one after the other all sit on ONE engine and every transfer of the device, up or down, runs
// alone (icsp_enc, 3 workers, 3000 CIF frames: 18.6 ms = 1042 MB at the one-way rate).  A stream that only ever uploads and one
// that only ever downloads sit on two engines whatever the order of events, and the link runs both ways at once
// (tools/probe_duplex.hip: 17.0 -> 11.1 ms for 456 MB each way).
int icsp_copy_streams(icsp_ctx_t* ctx, int shared)
{
    ENTER(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = join_all(ctx)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (!shared) { ctx->up_stream = ctx->down_stream = nullptr; return ICSP_OK; }
    static std::mutex m;
    static hipStream_t up[64], down[64];
    std::lock_guard<std::mutex> l(m);
    const int d = ctx->slot & 63;
    if (!up[d] && !ctx->single) {
        // the device's compute streams before its transfer streams (StreamPool: the order the process makes its streams in decides
        // which hardware queues the busy ones get)
        if (int rc = second_stream(ctx)) return rc;
        if (int rc = group_streams(ctx, kMaxPGroups)) return rc;
    }
    if (!up[d]) {
        // The two streams must sit on two DMA engines.  A stream keeps the engine its first copy was given, the lowest idle one
        // at that moment.  So the download stream's first copy is made while the upload stream is kept busy, and the pair is
        // then timed: an upload and a download together must take clearly less than the two one after the other (measured
        // here: 0.18 ms against 0.31 ms for 8 MB each way); if not, the download stream is made anew.  (Even so about one
        // icsp_enc process in thirty still ends up with all its transfers taking turns, 18 ms instead of 13.6 for 3000 frames;
        // one in thirteen before uploads and downloads were made one at a time per stream.)
        const bool trace = getenv("ICSP_TRACE_CREATE") != nullptr;
        auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double tph = tnow();
        auto phase = [&](const char* what) { if (trace) { const double t = tnow(); fprintf(stderr, "[icsp_copy_streams] %-26s %8.3f ms\n", what, (t - tph) * 1e3); tph = t; } };
        hipStream_t a = nullptr, b = nullptr;
        uint8_t* h = nullptr;
        const size_t nb = (size_t)8 << 20;
        uint8_t* dv = nullptr;                                         // device scratch: nothing of the context is touched
        // whatever leaves this block early (a failed stream creation) must not leak the probe's stream and its two 16 MB buffers
        struct Probe { hipStream_t* a; hipStream_t* b; uint8_t** h; uint8_t** dv; bool keep;
                       ~Probe() { if (*h) (void)hipHostFree(*h); if (*dv) (void)hipFree(*dv);
                                  if (!keep) { if (*a) (void)hipStreamDestroy(*a); if (*b) (void)hipStreamDestroy(*b); }
                                  (void)hipGetLastError(); } } probe{ &a, &b, &h, &dv, false };
        HIPCHK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        phase("upload stream");
        if (hipHostMalloc((void**)&h, 2 * nb, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h = nullptr; }
        phase("hipHostMalloc 16 MB");
        if (h && hipMalloc((void**)&dv, 2 * nb) != hipSuccess) { (void)hipGetLastError(); dv = nullptr; (void)hipHostFree(h); h = nullptr; }
        // (on the new stream: a plain hipMemset would make the runtime create its null stream's queue first)
        if (dv) { (void)hipMemsetAsync(dv, 0, 2 * nb, a); (void)hipStreamSynchronize(a); }
        phase("hipMalloc + memset");
        auto seconds = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        if (h) {
            memset(h, 0, 2 * nb);
            (void)hipMemcpyAsync(dv, h, nb, hipMemcpyHostToDevice, a);       // first use of the upload stream
            (void)hipStreamSynchronize(a);
            phase("first upload");
        }
        for (int attempt = 0; attempt < 4; attempt++) {
            HIPCHK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
            if (!h) break;
            const int busy = (int)std::min<size_t>(64, ((size_t)96 << 20) / nb);        // about 2 ms of uploads queued
            for (int k = 0; k < busy; k++) (void)hipMemcpyAsync(dv, h, nb, hipMemcpyHostToDevice, a);
            (void)hipMemcpyAsync(h + nb, dv + nb, nb, hipMemcpyDeviceToHost, b);   // first use of the download stream, under them
            (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
            double t_seq = 1e9, t_both = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                double t = seconds();
                (void)hipMemcpyAsync(dv, h, nb, hipMemcpyHostToDevice, a); (void)hipStreamSynchronize(a);
                (void)hipMemcpyAsync(h + nb, dv + nb, nb, hipMemcpyDeviceToHost, b); (void)hipStreamSynchronize(b);
                t_seq = std::min(t_seq, seconds() - t);
                t = seconds();
                (void)hipMemcpyAsync(dv, h, nb, hipMemcpyHostToDevice, a);
                (void)hipMemcpyAsync(h + nb, dv + nb, nb, hipMemcpyDeviceToHost, b);
                (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
                t_both = std::min(t_both, seconds() - t);
            }
            phase("download stream + timing");
            if (t_both < 0.8 * t_seq) break;
            if (attempt < 3) { (void)hipStreamDestroy(b); b = nullptr; }
        }
        probe.keep = true;                                             // (the buffers go with the guard)
        up[d] = a; down[d] = b;
    }
    ctx->up_stream = up[d]; ctx->down_stream = down[d];
    return ICSP_OK;
}

int icsp_host_register(void* p, size_t bytes, int read_only)
{
    if (!p || !bytes) return ICSP_ERR_UNENOUGH_PARAM;
    // whole pages only: a registration covers whole pages anyway, and a range that shares its first page with something else
    // cannot be told from that something by the runtime.  The length is rounded up to the page (the caller owns the rest of its
    // last page: true of every mapping and of aligned allocations whose size was rounded up, icsp_hip.h).
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    if ((uintptr_t)p % page) return ICSP_ERR_UNCORRECT_PARAM;
    const size_t whole = (bytes + page - 1) / page * page;
    const unsigned flags = hipHostRegisterPortable | (read_only ? hipHostRegisterReadOnly : 0u);
    if (hipHostRegister(p, whole, flags) != hipSuccess) { (void)hipGetLastError(); return ICSP_ERR_HIP; }
    pinned_ranges().add(p, whole);
    return ICSP_OK;
}

int icsp_host_unregister(void* p)
{
    if (!p) return ICSP_ERR_UNENOUGH_PARAM;
    pinned_ranges().remove(p);
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return ICSP_ERR_HIP; }
    return ICSP_OK;
}

const char* icsp_kernel_name(int k)
{
    static const char* names[ICSP_K_COUNT] = { "k_intra_luma", "k_chroma_dc", "k_residual", "k_me", "k_frame_serial", "k_pack", "k_decode" };
    return (k >= 0 && k < ICSP_K_COUNT) ? names[k] : "?";
}

int icsp_create(icsp_ctx_t** out, const icsp_params_t* p, int device_id, int max_frames)
{
    if (!out || !p) return ICSP_ERR_UNENOUGH_PARAM;
    *out = nullptr;
    if (p->width % 16 || p->height % 16 || p->width < 32 || p->width > 4096 || p->height < 16 || p->height > 2304 ||
        (p->width / 16) * (p->height / 16) > 8704 ||      /* k_frame_serial keeps 15 bytes of LDS per macroblock */
        p->qp_dc <= 0 || p->qp_ac <= 0 || p->qp_dc > 255 || p->qp_ac > 255 ||      /* one header byte each (ENC.h:207-208) */
        p->intra_period < 0 || max_frames <= 0)
        return ICSP_ERR_UNCORRECT_PARAM;
    // ICSP_TRACE_CREATE=1: where the set-up time goes, phase by phase, on stderr (tools/cold_first.sh)
    const bool trace = getenv("ICSP_TRACE_CREATE") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tph = tnow();
    auto phase = [&](const char* what) { if (trace) { const double t = tnow(); fprintf(stderr, "[icsp_create] %-28s %8.3f ms\n", what, (t - tph) * 1e3); tph = t; } };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= (fake_devices() ? fake_devices() : ndev)) return ICSP_ERR_NO_DEVICE;
    const int slot_id = device_id;                     // the caller's device number: index of the per-device records
    device_id %= ndev;                                 // the physical device (the same number unless ICSP_FAKE_DEVICES is set)
    if (hipSetDevice(device_id) != hipSuccess) return ICSP_ERR_NO_DEVICE;
    phase("device count + set device");
    icsp_ctx* ctx = new (std::nothrow) icsp_ctx();
    if (!ctx) return ICSP_ERR_MEM_ALLOC;
    ctx->p = *p; ctx->device = device_id; ctx->slot = slot_id; ctx->max_frames = max_frames;
    ctx->keep_coef = false; ctx->profiling = false; ctx->prof_mask = 0;
    memset(ctx->prof_ms, 0, sizeof(ctx->prof_ms)); memset(ctx->prof_n, 0, sizeof(ctx->prof_n));
    ctx->tl_file = nullptr; ctx->tl_base = nullptr;
    Geo& g = ctx->g;
    g.W = p->width; g.H = p->height; g.sw = g.W / 16; g.sh = g.H / 16; g.nmb = g.sw * g.sh;
    g.cols8 = 2 * g.sw; g.rows8 = 2 * g.sh; g.cw = g.W / 2; g.ch = g.H / 2;
    g.qdc = p->qp_dc; g.qac = p->qp_ac;
    g.prio = 1; g.bands = 1;
    // magic = floor(2^32/q) + 1 (== ceil(2^32/q) unless q is a power of two): |t|/q == umulhi(|t|, magic) for |t| < 2^16, and
    // strictly above 2^32/q, which the signed form in the DC chains needs (a negative multiple of q must not divide exactly)
    g.mdc = (uint32_t)(0x100000000ull / (unsigned)g.qdc + 1);               // unused when q == 1 (would not fit 32 bits)
    g.mac = (uint32_t)(0x100000000ull / (unsigned)g.qac + 1);
    g.qpow2 = ((g.qdc & (g.qdc - 1)) == 0 && (g.qac & (g.qac - 1)) == 0) ? 1 : 0;
    { int v_ = g.qpow2; if (!env_int("ICSP_QUANT_POW2", 0, 1, &v_)) { delete ctx; return ICSP_ERR_UNCORRECT_PARAM; } g.qpow2 = g.qpow2 && v_; }
    g.idc = 1.0 / (double)g.qdc; g.iac = 1.0 / (double)g.qac;
    g.msw = (uint32_t)(0x100000000ull / (unsigned)g.sw + 1);
    g.mtpr = (uint32_t)(0x100000000ull / (unsigned)((g.sw + 1) / 2) + 1);
    g.fsz = (long long)g.W * g.H * 3 / 2;
    ctx->intra_waves = intra_waves_needed(g);
    ctx->intra_waves_g2 = intra_waves_chained(g, 2);
    ctx->n_cu = 256;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && v > 0) ctx->n_cu = v; }
    memset(&ctx->b, 0, sizeof(ctx->b));
    memset(&ctx->pk, 0, sizeof(ctx->pk)); ctx->pk_cap = 0; ctx->pk_host = nullptr; ctx->pk_first = -1; ctx->pk_n = 0; ctx->pk_total = 0;
    ctx->stream = nullptr; ctx->stream2 = nullptr; ctx->ev_fork = nullptr; ctx->ev_join = nullptr;
    ctx->up_stream = nullptr; ctx->down_stream = nullptr;
    ctx->gop_stage_in[0] = ctx->gop_stage_in[1] = ctx->gop_stage_out[0] = ctx->gop_stage_out[1] = nullptr;
    ctx->gop_stage_in_cap = ctx->gop_stage_out_cap = 0; memset(ctx->gop_ev, 0, sizeof(ctx->gop_ev)); ctx->gop_pool = nullptr;
    ctx->xfer_ev_in[0] = ctx->xfer_ev_in[1] = ctx->xfer_ev_out[0] = ctx->xfer_ev_out[1] = nullptr;
    ctx->xfer_in_busy[0] = ctx->xfer_in_busy[1] = false; ctx->up_pool = nullptr;
    ctx->s2_dirty = false; ctx->st_ahead = true; ctx->always_sync = false;
    ctx->p_dirty = false; ctx->sticky = 0;
    memset(ctx->flight, 0, sizeof(ctx->flight));
    ctx->last_first = 0; ctx->last_n = 0; ctx->rr = 0; ctx->single = false;
    { int v_ = 1; if (!env_int("ICSP_I_STREAM_B", 0, 1, &v_)) { delete ctx; return ICSP_ERR_UNCORRECT_PARAM; } ctx->i_stream_b = v_ != 0; }
    { int v_ = 1; if (!env_int("ICSP_CHAINS3", 0, 1, &v_)) { delete ctx; return ICSP_ERR_UNCORRECT_PARAM; } ctx->chains3 = v_ != 0; }
    ctx->prev2_first = ctx->prev2_n = 0;
    { int v_ = 60; if (!env_int("ICSP_CHROMA_CAP", 0, 120, &v_)) { delete ctx; return ICSP_ERR_UNCORRECT_PARAM; } ctx->chroma_cap = v_; }
    ctx->last_form = ctx->last_nw = ctx->last_ring = ctx->last_whole = ctx->last_groups = 0;
    { int w_ = 1; if (!env_int("ICSP_WHOLE", 0, 1, &w_)) { delete ctx; return ICSP_ERR_UNCORRECT_PARAM; } ctx->whole_ok = w_ != 0; }
    int no_fuse = 0, force_slices = 0;
    ctx->force_intra_form = 0; ctx->force_intra_group = 0; ctx->last_rowgroup = 0;
    for (int k = 0; k < kMaxPGroups; k++) { ctx->pstream[k] = nullptr; ctx->ev_pjoin[k] = nullptr; }
    ctx->p_groups = 2;                 // measured, one range of 30 / 60 / 339 CIF GOPs again and again: 1 group 0.91 / 1.27 / 1.90 M frames/s,
                                       // 2 groups 0.97 / 1.48 / 1.94 M, 3 groups 0.97 / 1.44 / 1.91 M (round 3 saw 0.27 M with three: see StreamPool)
    ctx->i_groups = 2;
    if (!env_int("ICSP_NO_FUSE", 0, 1, &no_fuse) || !env_int("ICSP_P_GROUPS", 1, kMaxPGroups, &ctx->p_groups) || !env_int("ICSP_I_GROUPS", 1, 2, &ctx->i_groups) ||
        !env_int("ICSP_INTRA_FORM", 8, 32, &ctx->force_intra_form) ||
        (ctx->force_intra_form != 0 && ctx->force_intra_form != 8 && ctx->force_intra_form != 32) ||
        !env_int("ICSP_INTRA_GROUP", 0, 2, &ctx->force_intra_group) ||
        !env_int("ICSP_XCD_SLICES", 0, 64, &force_slices) ||
        !env_int("ICSP_SERIAL_PRIO", 0, 1, &g.prio) || !env_int("ICSP_SERIAL_BANDS", 0, 1, &g.bands)) { delete ctx; return ICSP_ERR_UNCORRECT_PARAM; }
    ctx->no_fuse = no_fuse != 0;
    set_xcd_slices(force_slices);
    const size_t nf = (size_t)max_frames, nmb = (size_t)g.nmb;
    auto fail = [&](int code, const char* what, hipError_t e) { ctx->err = std::string(what) + ": " + hipGetErrorString(e); icsp_destroy(ctx); return code; };
    hipError_t e;
    // the main stream carries the latency-bound kernels and gets the higher priority; stream2 (I-frame chroma) fills in
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) { (void)hipGetLastError(); prio_lo = prio_hi = 0; }
    ctx->prio_lo = prio_lo; ctx->prio_hi = prio_hi;        // (before the first stream: icsp_destroy hands streams back by priority)
    if ((e = stream_get(ctx->slot, prio_hi, &ctx->stream)) != hipSuccess) return fail(ICSP_ERR_HIP, "hipStreamCreate", e);
    phase("priority range + stream");
    ctx->prio_lo = prio_lo;            // stream2 is created by the first encode / decode that uses it (second_stream)
    if ((e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)) != hipSuccess) return fail(ICSP_ERR_HIP, "hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming)) != hipSuccess) return fail(ICSP_ERR_HIP, "hipEventCreate", e);
    if ((e = kernel_attributes()) != hipSuccess) return fail(ICSP_ERR_HIP, "hipFuncSetAttribute", e);
    ctx->prio_hi = prio_hi;            // the streams of the additional GOP groups are created by the first P step that uses them
    phase("events + function attributes");
#define ALLOC(ptr, bytes) if ((e = hipMalloc((void**)&(ptr), (bytes))) != hipSuccess) return fail(ICSP_ERR_MEM_ALLOC, "hipMalloc " #ptr, e)
    ALLOC(ctx->d_frames, nf * g.fsz);
    ctx->b.frames = ctx->d_frames;
    ALLOC(ctx->b.recon, nf * g.fsz);
    ALLOC(ctx->b.levels, nf * nmb * 384 * sizeof(int16_t));
    ALLOC(ctx->b.acflag, nf * nmb * 6);
    ALLOC(ctx->b.mpm, nf * nmb * 4);
    ALLOC(ctx->b.mvd, nf * nmb * 2);
    ALLOC(ctx->b.mv, nf * nmb * 2);
    ALLOC(ctx->b.imode, nf * nmb * 4);
    ALLOC(ctx->b.me_ent, nf * nmb * 4 * sizeof(uint32_t));
    ALLOC(ctx->b.me_sums, nf * nmb * 24 * sizeof(int16_t));
    ALLOC(ctx->b.me_flag, nf * sizeof(int));
    ALLOC(ctx->b.me_done, nf * sizeof(int));
    ALLOC(ctx->b.dcpred, nf * nmb * 6 * sizeof(int16_t));
#undef ALLOC
    phase("13 hipMalloc");
#define ZERO(ptr, bytes) if ((e = hipMemsetAsync((ptr), 0, (bytes), ctx->stream)) != hipSuccess) return fail(ICSP_ERR_HIP, "hipMemsetAsync " #ptr, e)
    ZERO(ctx->b.me_flag, nf * sizeof(int));            // k_me raises it, the serial kernel of the same step clears it
    ZERO(ctx->b.me_done, nf * sizeof(int));            // arrival tickets: 0 between launches
    ZERO(ctx->b.mpm, nf * nmb * 4);
    ZERO(ctx->b.mvd, nf * nmb * 2);
    ZERO(ctx->b.mv, nf * nmb * 2);
    ZERO(ctx->b.imode, nf * nmb * 4);
#undef ZERO
    phase("6 hipMemsetAsync (enqueue)");
    {   // the search tables are the same for every context: once per device and process (the call costs 7-12 ms)
        static std::mutex m;
        static bool loaded[64] = {};
        std::lock_guard<std::mutex> lock(m);
        if (slot_id >= 64 || !loaded[slot_id]) {
            // (on the context's stream: the plain call would make the runtime create its null stream's queue, 10 ms of set-up)
            if ((e = upload_search_tables(ctx->stream)) != hipSuccess ||
                (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(ICSP_ERR_HIP, "hipMemcpyToSymbolAsync", e);
            if (slot_id < 64) loaded[slot_id] = true;
        }
    }
    phase("search tables + their sync");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(ICSP_ERR_HIP, "hipStreamSynchronize", e);
    phase("final sync");
    if (const char* tl = getenv("ICSP_TIMELINE_DUMP")) {
        // diagnostics: every launch between HIP events, "kernel stream start_us end_us" against one base event (tools/timeline_events.py);
        // the host pays two event records per launch
        ctx->tl_file = fopen(tl, "a");
        if (ctx->tl_file && hipEventCreate(&ctx->tl_base) == hipSuccess && hipEventRecord(ctx->tl_base, ctx->stream) == hipSuccess) {
            ctx->profiling = true; ctx->prof_mask = 0xffffffffu;
            for (int i = 0; i < 6000; i++) {
                EvPair ep;
                if (hipEventCreate(&ep.a) != hipSuccess) break;
                if (hipEventCreate(&ep.b) != hipSuccess) { (void)hipEventDestroy(ep.a); break; }
                ep.kernel = 0; ep.sid = 0;
                ctx->ev_pool.push_back(ep);
            }
        }
        (void)hipGetLastError();
    }
    *out = ctx;
    return ICSP_OK;
}

int icsp_destroy(icsp_ctx_t* ctx)
{
    if (!ctx) return ICSP_OK;
    // best effort from here on: nothing can be done about a failing call, and none of them orders anything
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    for (int k = 1; k < kMaxPGroups; k++) if (ctx->pstream[k]) (void)hipStreamSynchronize(ctx->pstream[k]);
    if (ctx->tl_file) { collect_profile(ctx); fclose(ctx->tl_file); ctx->tl_file = nullptr; }
    if (ctx->tl_base) (void)hipEventDestroy(ctx->tl_base);
    for (auto& e : ctx->ev_pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : ctx->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    void* bufs[] = { ctx->d_frames, ctx->b.recon, ctx->b.levels, ctx->b.acflag, ctx->b.mpm, ctx->b.mvd, ctx->b.mv, ctx->b.imode, ctx->b.me_ent,
                     ctx->b.me_sums, ctx->b.me_flag, ctx->b.me_done, ctx->b.dcpred, ctx->b.coef, ctx->pk.grp_bits, ctx->pk.grp_off,
                     ctx->pk.chunk_bits, ctx->pk.chunk_base, ctx->pk.out };
    for (void* q : bufs) if (q) (void)hipFree(q);
    if (ctx->pk_host) (void)hipHostFree(ctx->pk_host);
    gop_release(ctx);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    for (auto& f : ctx->flight) {
        if (f.ev_done) (void)hipEventDestroy(f.ev_done);
        for (int k = 0; k < kMaxPGroups; k++) if (f.ev_p1[k]) (void)hipEventDestroy(f.ev_p1[k]);
        if (f.ev_tab) (void)hipEventDestroy(f.ev_tab);
        if (f.d_tab) (void)hipFree(f.d_tab);
        if (f.h_tab) (void)hipHostFree(f.h_tab);
        free(f.many_list);
    }
    // the streams stay with the device for its next context (stream_get); they were waited for above
    const bool healthy = ctx->sticky == 0;
    stream_put(ctx->slot, ctx->prio_hi, ctx->stream, healthy);
    stream_put(ctx->slot, ctx->prio_lo, ctx->stream2, healthy);
    for (int k = 1; k < kMaxPGroups; k++) { if (ctx->ev_pjoin[k]) (void)hipEventDestroy(ctx->ev_pjoin[k]); stream_put(ctx->slot, ctx->prio_hi, ctx->pstream[k], healthy); }
    (void)hipGetLastError();
    delete ctx;
    return ICSP_OK;
}

namespace {
// Transfers of a context that uses the device's shared transfer streams (icsp_copy_streams): uploads run on one stream,
// downloads on another; a transfer starts when the context's own stream is idle and the host waits for it, one at a time per
// stream and device.
static std::mutex g_up_turn[64];       // one upload at a time on a device's shared upload stream (see g_down_turn)
static int copy_up(icsp_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    hipStream_t st = ctx->stream, up = ctx->up_stream;
    if (!up) return xfer_up(ctx, dst, src, bytes, st);
    HIPCHK(hipStreamSynchronize(st));                              // whatever still reads the destination
    std::lock_guard<std::mutex> l(g_up_turn[ctx->slot & 63]);
    if (int rc = xfer_up(ctx, dst, src, bytes, up)) return rc;
    HIPCHK(hipStreamSynchronize(up));
    return 0;
}
// One download at a time on a device's shared download stream: a copy submitted while the stream's engine is busy is given
// another engine -- possibly the upload stream's, if that one happens to be idle -- and from then on the two directions take
// turns on it.  The turn is taken when the context's kernels are through, so that it covers the copies alone.
static std::mutex g_down_turn[64];
struct DownTurn { std::mutex* m = nullptr; ~DownTurn() { if (m) m->unlock(); } };
static hipStream_t down_of(icsp_ctx* ctx) { return ctx->down_stream ? ctx->down_stream : ctx->stream; }
static int copy_down_begin(icsp_ctx* ctx, DownTurn& turn)
{
    if (!ctx->down_stream) return 0;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    turn.m = &g_down_turn[ctx->slot & 63];
    turn.m->lock();
    return 0;
}
static int copy_down_end(icsp_ctx* ctx)
{
    HIPCHK(hipStreamSynchronize(down_of(ctx)));                   // (the shared stream carries this context's copies alone: the turn)
    return 0;
}
} // namespace

int icsp_upload(icsp_ctx_t* ctx, const uint8_t* yuv, int first, int n)
{
    ENTER(ctx);
    if (!yuv) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = join_all(ctx)) return rc;                                      // chroma kernels of an earlier encode may still read the frames
    ctx->st_ahead = true;
    if (int rc = copy_up(ctx, ctx->d_frames + (size_t)first * ctx->g.fsz, yuv, (size_t)n * ctx->g.fsz)) return rc;
    return ICSP_OK;
}

// Upload on the device's shared upload stream, returning when the frames are on the device.  It reads the context (device,
// frame buffer, geometry, the shared stream) and changes nothing in it, so another host thread may run it while the
// context's own thread packs and downloads an earlier batch -- see icsp_hip.h for what the caller has to guarantee.
int icsp_upload_sync(icsp_ctx_t* ctx, const uint8_t* yuv, int first, int n)
{
    ENTER(ctx);
    if (!yuv) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = check_range(ctx, first, n)) return rc;
    if (!ctx->up_stream) return ICSP_ERR_UNCORRECT_PARAM;
    if (hipSetDevice(ctx->device) != hipSuccess) return ICSP_ERR_HIP;
    std::lock_guard<std::mutex> l(g_up_turn[ctx->slot & 63]);
    // (frames that are not in pinned memory go through the context's upload staging buffers, which nothing else uses meanwhile)
    if (int rc = xfer_up(ctx, ctx->d_frames + (size_t)first * ctx->g.fsz, yuv, (size_t)n * ctx->g.fsz, ctx->up_stream)) return rc;
    if (hipStreamSynchronize(ctx->up_stream) != hipSuccess) { (void)hipGetLastError(); return ICSP_ERR_HIP; }
    return ICSP_OK;
}

int icsp_encode_resident(icsp_ctx_t* ctx, int first, int n)
{
    ENTER(ctx);
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    return encode_range(ctx, first, n);
}

int icsp_encode_resident_many(icsp_ctx_t* ctx, int k, const int* first_frames, const int* ns)
{
    ENTER(ctx);
    if (k < 0 || (k > 0 && (!first_frames || !ns))) return ICSP_ERR_UNENOUGH_PARAM;
    HIPCHK(hipSetDevice(ctx->device));
    return encode_many(ctx, k, first_frames, ns);
}

int icsp_sync(icsp_ctx_t* ctx)
{
    ENTER(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = join_all(ctx)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // event pairs are read out in icsp_profile_get / icsp_profile_reset, not here: a caller timing "launch ... icsp_sync"
    // must not pay for the bookkeeping of the profiler
    return ICSP_OK;
}

int icsp_download(icsp_ctx_t* ctx, int first, int n, int16_t* levels, uint8_t* acflag, uint8_t* mpm, int8_t* mvd, uint8_t* recon)
{
    ENTER(ctx);
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nmb = ctx->g.nmb, f = first, c = n;
    if (int rc = join_all(ctx)) return rc;
    DownTurn turn;
    if (int rc = copy_down_begin(ctx, turn)) return rc;
    hipStream_t st = down_of(ctx);
    if (int rc = xfer_down(ctx, levels, ctx->b.levels + f * nmb * 384, c * nmb * 384 * sizeof(int16_t), st)) return rc;
    if (int rc = xfer_down(ctx, acflag, ctx->b.acflag + f * nmb * 6, c * nmb * 6, st)) return rc;
    if (int rc = xfer_down(ctx, mpm, ctx->b.mpm + f * nmb * 4, c * nmb * 4, st)) return rc;
    if (int rc = xfer_down(ctx, mvd, ctx->b.mvd + f * nmb * 2, c * nmb * 2, st)) return rc;
    if (int rc = xfer_down(ctx, recon, ctx->b.recon + f * ctx->g.fsz, c * ctx->g.fsz, st)) return rc;
    if (int rc = copy_down_end(ctx)) return rc;
    if (ctx->profiling) collect_profile(ctx);
    return ICSP_OK;
}

int icsp_upload_syntax(icsp_ctx_t* ctx, int first, int n, const int16_t* levels, const uint8_t* mpm, const int8_t* mvd)
{
    ENTER(ctx);
    if (!levels || !mpm || !mvd) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nmb = ctx->g.nmb, f = first, c = n;
    hipStream_t st = ctx->stream;
    if (int rc = join_all(ctx)) return rc;
    ctx->st_ahead = true;
    if (int rc = xfer_up(ctx, ctx->b.levels + f * nmb * 384, levels, c * nmb * 384 * sizeof(int16_t), st)) return rc;
    if (int rc = xfer_up(ctx, ctx->b.mpm + f * nmb * 4, mpm, c * nmb * 4, st)) return rc;
    if (int rc = xfer_up(ctx, ctx->b.mvd + f * nmb * 2, mvd, c * nmb * 2, st)) return rc;
    HIPCHK(hipStreamSynchronize(st));
    return ICSP_OK;
}

int icsp_decode_resident(icsp_ctx_t* ctx, int first, int n)
{
    ENTER(ctx);
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    return decode_range(ctx, first, n);
}

// ---- device bit packer (icsp_pack.hip.inc).  Two steps: lengths + scans (the host learns the number of bits), then the
// packing itself at a bit phase the host chooses.
// bytes of body buffer for a string of `bits` bits placed at at0 = 8 * A + sh (A < 64 bytes of phase, sh < 8 bits: k_pack's at0
// is at most 511): the string's bytes + 1 (bit phase) + 64 (byte phase) + the 16 bytes past the end that k_pack_zero may
// clear, rounded up to a dword.  Invariant: at0 / 8 + ceil((sh + bits) / 8) + 16 <= pack_bytes(bits).
static inline size_t pack_bytes(unsigned long long bits) { return ((size_t)(bits / 8) + 1 + 64 + 16 + 3) & ~(size_t)3; }

static int pack_alloc(icsp_ctx* ctx)
{
    if (ctx->pk.out) return ICSP_OK;
    if (ctx->pk.grp_bits) {                                            // (a failed pack_reserve left only the body buffer missing)
        if (hipMalloc((void**)&ctx->pk.out, ctx->pk_cap = pack_bytes(8ull << 20)) == hipSuccess) return ICSP_OK;
        (void)hipGetLastError(); ctx->pk.out = nullptr; ctx->pk_cap = 0; ctx->err = "hipMalloc bit packer body";
        return ICSP_ERR_MEM_ALLOC;
    }
    const Geo& g = ctx->g;
    const long long cap_grps = pack_groups(g, ctx->max_frames);
    const size_t chunks = (size_t)pack_chunks(cap_grps);
    // the body buffer starts at a third of the frames' size (what QP >= 8 needs) and grows to what a count asks for
    // (pack_reserve): the worst case, seven times the frames, would be gigabytes that almost no stream ever touches
    ctx->pk_cap = std::min(pack_bytes(icsp_bitstream_bound(&ctx->p, ctx->max_frames) * 8ull),
                           pack_bytes((unsigned long long)ctx->max_frames * (unsigned long long)ctx->g.fsz * 8ull / 3 + (8ull << 20)));
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->pk.grp_bits, (size_t)cap_grps * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->pk.grp_off, (size_t)cap_grps * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->pk.chunk_bits, chunks * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->pk.chunk_base, (chunks + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->pk.out, ctx->pk_cap);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->pk_host, 256, hipHostMallocDefault);     // total bits | head | tail
    if (e != hipSuccess) {
        (void)hipGetLastError();
        void* bufs[] = { ctx->pk.grp_bits, ctx->pk.grp_off, ctx->pk.chunk_bits, ctx->pk.chunk_base, ctx->pk.out };
        for (void* q : bufs) if (q) (void)hipFree(q);
        if (ctx->pk_host) (void)hipHostFree(ctx->pk_host);
        memset(&ctx->pk, 0, sizeof(ctx->pk)); ctx->pk_host = nullptr;
        ctx->err = std::string("hipMalloc bit packer: ") + hipGetErrorString(e);
        return ICSP_ERR_MEM_ALLOC;
    }
    return ICSP_OK;
}

// makes room in the body buffer for a string of `bits` bits at any phase icsp_pack_into may ask for; a realloc drops whatever
// the buffer held (nothing does between a count and its packing)
static int pack_reserve(icsp_ctx* ctx, unsigned long long bits)
{
    const size_t need = pack_bytes(bits);
    if (need <= ctx->pk_cap) return ICSP_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->pk.out); ctx->pk.out = nullptr; ctx->pk_cap = 0;
    const size_t want = need + need / 4;                               // headroom: the next batches are about as long
    if (hipMalloc((void**)&ctx->pk.out, want) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc((void**)&ctx->pk.out, need) != hipSuccess) { ctx->pk.out = nullptr; ctx->err = "hipMalloc bit packer body"; (void)hipGetLastError(); return ICSP_ERR_MEM_ALLOC; }
        ctx->pk_cap = need;
    } else ctx->pk_cap = want;
    return ICSP_OK;
}

static int pack_check(icsp_ctx* ctx, int first, int n)
{
    if (int rc = check_range(ctx, first, n)) return rc;
    const int L = ctx->p.intra_period > 0 ? ctx->p.intra_period : 1;
    return (first % L != 0) ? ICSP_ERR_RANGE : ICSP_OK;
}

// lengths + scans of slots [first, first + n) on stream kst, the total read back: returns when the host has it.  No ordering against
// the context's other streams here: the caller has made sure the range's encode is complete or queued in front of kst.
static int pack_count_on(icsp_ctx* ctx, int first, int n, hipStream_t kst, unsigned long long* total_out)
{
    if (int rc = pack_alloc(ctx)) return rc;
    const Geo& g = ctx->g;
    const int L = ctx->p.intra_period > 0 ? ctx->p.intra_period : 1;
    const int nchunk = pack_chunks(pack_groups(g, n));
    const DevBufs& b = ctx->b;
    const PackBufs& pk = ctx->pk;
    LT(ctx, ICSP_K_PACK, kst, [&] { bits_count_scan(g, first, n, L, b, pk, kst); });
    unsigned long long* total = (unsigned long long*)ctx->pk_host;
    HIPCHK(hipMemcpyAsync(total, pk.chunk_base + nchunk, 8, hipMemcpyDeviceToHost, kst));
    HIPCHK(hipStreamSynchronize(kst));
    if (int rc = pack_reserve(ctx, *total)) return rc;
    *total_out = *total;
    return ICSP_OK;
}

int icsp_pack_count(icsp_ctx_t* ctx, int first, int n, uint64_t* nbits)
{
    ENTER(ctx);
    if (!nbits) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = pack_check(ctx, first, n)) return rc;
    *nbits = 0;
    ctx->pk_first = -1;
    if (n == 0) { ctx->pk_first = first; ctx->pk_n = 0; ctx->pk_total = 0; return ICSP_OK; }
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = join_all(ctx)) return rc;
    unsigned long long total = 0;
    if (int rc = pack_count_on(ctx, first, n, ctx->stream, &total)) return rc;
    ctx->pk_first = first; ctx->pk_n = n; ctx->pk_total = total;
    *nbits = total;
    return ICSP_OK;
}

// the packing kernels for the range icsp_pack_count last measured, the string starting at bit `sh` (0..7) of pk.out
static int pack_write(icsp_ctx* ctx, int first, int n, unsigned sh, hipStream_t st = nullptr)
{
    const Geo& g = ctx->g;
    const int L = ctx->p.intra_period > 0 ? ctx->p.intra_period : 1;
    if (!st) st = ctx->stream;
    const DevBufs& b = ctx->b;
    const PackBufs& pk = ctx->pk;
    LT(ctx, ICSP_K_PACK, st, [&] { bits_pack(g, first, n, L, b, pk, sh, st); });
    return ICSP_OK;
}

// A string of `total` bits (what pack_count_on last measured) goes to body_image + at_bit / 8 in two steps.
// It is packed at the byte phase (mod 64) and bit phase it has in the image, so that device byte j and its place in the image are
// congruent mod 64 and the bulk goes as one aligned copy (DMA engines crawl on odd addresses: 6 GB/s instead of 57 measured).  The
// ragged head and tail (< 64 bytes each) come back through a pinned scratch; their outermost bytes may be shared with the
// neighbouring strings and are OR-ed in, the rest is stored.
static int pack_place_kernels(icsp_ctx* ctx, int first, int n, uint64_t at_bit, const uint8_t* body_image, hipStream_t kst)
{
    const size_t A = (size_t)((uintptr_t)(body_image + (size_t)(at_bit >> 3)) & 63);
    return pack_write(ctx, first, n, (unsigned)(A * 8 + (unsigned)(at_bit & 7)), kst);
}
// the copies, on stream ds (ordered after the packing kernels by the caller); returns when the bytes are in the image
static int pack_place_copy(icsp_ctx* ctx, unsigned long long total, uint64_t at_bit, uint8_t* body_image, hipStream_t ds)
{
    const unsigned sh = (unsigned)(at_bit & 7);
    const size_t nb = (size_t)((sh + total + 7) / 8), b0 = (size_t)(at_bit >> 3);
    uint8_t* dst = body_image + b0;
    const size_t A = (size_t)((uintptr_t)dst & 63);
    const uint8_t* out = (const uint8_t*)ctx->pk.out;              // device byte A + j  <->  dst[j]
    const size_t lo = A, hi = A + nb;
    size_t ilo = (lo + 1 + 63) & ~(size_t)63, ihi = (hi - 1) & ~(size_t)63;      // aligned interior, first and last byte excluded
    if (ihi <= ilo) ilo = ihi = hi;                                // short string: everything through the scratch
    const size_t nhead = std::min(ilo, hi) - lo, ntail = hi - std::max(ihi, lo + nhead);
    uint8_t* head = ctx->pk_host + 64;
    uint8_t* tail = ctx->pk_host + 192;                           // (head: up to 127 bytes when there is no interior)
    if (ihi > ilo) { if (int rc = xfer_down(ctx, dst + (ilo - lo), out + ilo, ihi - ilo, ds)) return rc; }    // (an image in plain memory: staged)
    if (nhead) HIPCHK(hipMemcpyAsync(head, out + lo, nhead, hipMemcpyDeviceToHost, ds));
    if (ntail) HIPCHK(hipMemcpyAsync(tail, out + hi - ntail, ntail, hipMemcpyDeviceToHost, ds));
    HIPCHK(hipStreamSynchronize(ds));
    for (size_t j = 0; j < nhead; j++) {
        if (j == 0 || j == nb - 1) __atomic_fetch_or(&dst[j], head[j], __ATOMIC_RELAXED);
        else dst[j] = head[j];
    }
    for (size_t j = 0; j < ntail; j++) {
        const size_t d = nb - ntail + j;
        if (d == nb - 1) __atomic_fetch_or(&dst[d], tail[j], __ATOMIC_RELAXED);
        else dst[d] = tail[j];
    }
    return ICSP_OK;
}

int icsp_pack_into(icsp_ctx_t* ctx, int first, int n, uint64_t at_bit, uint8_t* body_image, size_t cap)
{
    ENTER(ctx);
    if (!body_image) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = pack_check(ctx, first, n)) return rc;
    if (ctx->pk_first != first || ctx->pk_n != n) { ctx->err = "icsp_pack_into without icsp_pack_count of the same range"; return ICSP_ERR_RANGE; }
    if (ctx->pk_total == 0) return ICSP_OK;
    const unsigned sh = (unsigned)(at_bit & 7);
    const size_t nb = (size_t)((sh + ctx->pk_total + 7) / 8), b0 = (size_t)(at_bit >> 3);
    if (b0 + nb > cap || b0 + nb < b0) return ICSP_ERR_RANGE;
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = pack_place_kernels(ctx, first, n, at_bit, body_image, ctx->stream)) return rc;
    DownTurn turn;
    if (int rc = copy_down_begin(ctx, turn)) return rc;             // (shared download stream: waits for the kernels, takes the device's turn)
    if (int rc = pack_place_copy(ctx, ctx->pk_total, at_bit, body_image, down_of(ctx))) return rc;
    if (ctx->profiling) collect_profile(ctx);
    return ICSP_OK;
}

int icsp_pack_bits(icsp_ctx_t* ctx, int first, int n, uint8_t* body, size_t cap, uint64_t* nbits)
{
    ENTER(ctx);
    if (!body || !nbits) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = icsp_pack_count(ctx, first, n, nbits)) return rc;
    const uint64_t total = *nbits;
    if (total == 0) return ICSP_OK;
    const size_t nbytes = (size_t)((total + 7) / 8);
    if (nbytes > cap) { *nbits = 0; return ICSP_ERR_RANGE; }
    if (int rc = pack_write(ctx, first, n, 0)) return rc;
    hipStream_t st = ctx->stream;
    if (int rc = xfer_down(ctx, body, ctx->pk.out, nbytes, st)) return rc;
    HIPCHK(hipStreamSynchronize(st));
    if (ctx->profiling) collect_profile(ctx);
    return ICSP_OK;
}

// A stream's first large transfer into a newly pinned range can cost the hipMemcpyAsync call about 6 ms on this runtime
// (transfers of less than a megabyte take another path and do not count).  This writes `bytes` ZERO bytes (at most 16 MB) from
// the packer's scratch buffer to `pinned`, so a host can spend that during set-up on a range that must start out zeroed anyway.
int icsp_host_warm(icsp_ctx_t* ctx, void* pinned, size_t bytes)
{
    ENTER(ctx);
    if (!pinned) return ICSP_ERR_UNENOUGH_PARAM;
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = join_all(ctx)) return rc;
    if (int rc = pack_alloc(ctx)) return rc;
    const size_t nb = std::min(bytes, (size_t)16 << 20), piece = std::min(nb, ctx->pk_cap);
    if (nb == 0) return ICSP_OK;
    if (!host_pinned(pinned, nb)) return ICSP_ERR_UNCORRECT_PARAM;      // this call is about a pinned range's first use; plain memory has none
    ctx->pk_first = -1;                                 // the scratch no longer holds a counted string
    HIPCHK(hipMemsetAsync(ctx->pk.out, 0, piece, ctx->stream));
    for (size_t o = 0; o < nb; o += piece)
        HIPCHK(hipMemcpyAsync((uint8_t*)pinned + o, ctx->pk.out, std::min(piece, nb - o), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ICSP_OK;
}

// Everything an encode or pack creates on first use (the GOP-group streams, the packer's buffers, the kernels' first
// launches) is created now, by encoding one GOP of black frames and packing it -- a host that times or pipelines its
// batches calls this while it sets up (icsp_enc).  The frame store's first GOP is overwritten.
int icsp_prepare(icsp_ctx_t* ctx)
{
    ENTER(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    const int L = ctx->p.intra_period > 0 ? ctx->p.intra_period : 1;
    const int n = std::min(ctx->max_frames, std::max(L, 2 * L <= ctx->max_frames ? 2 * L : L));    // two GOPs when they fit: both group streams
    if (int rc = join_all(ctx)) return rc;
    // batches of 8+ GOPs run as p_groups chains, all-intra batches of more frames than CUs in i_groups parts
    // (a context in single-stream mode never uses them: icsp_enc sets that mode for one-chunk clips to save exactly this set-up)
    const int ngs = ctx->single ? 1 : L > 1 ? (ctx->max_frames >= 8 * L ? ctx->p_groups : 1) : (ctx->max_frames > ctx->n_cu ? ctx->i_groups : 1);
    if (ngs > 1) {
        if (int rc = group_streams(ctx, ngs)) return rc;
        for (int k = 1; k < ngs; k++) HIPCHK(hipMemsetAsync(ctx->b.me_done, 0, sizeof(int), ctx->pstream[k]));    // first use of the queue
        for (int k = 1; k < ngs; k++) HIPCHK(hipStreamSynchronize(ctx->pstream[k]));
    }
    // A stream's first host-to-device DMA costs about 6 ms inside the hipMemcpyAsync call, whatever its size (a transfer queue
    // is set up), and so does its first LARGE device-to-host one that follows a kernel of ours (transfers below a megabyte, and
    // ones behind the runtime's own fill kernels, take another path).  Both are spent here on a scratch buffer.  (Pinned by
    // hipHostMalloc rather than registered and unregistered on the spot: contexts are prepared concurrently, and pinning
    // calls racing with other threads' transfers are best avoided.)
    const size_t nb = std::min<size_t>((size_t)4 << 20, (size_t)ctx->max_frames * ctx->g.fsz);
    void* h = nullptr;
    if (hipHostMalloc(&h, nb, hipHostMallocDefault) == hipSuccess) {
        memset(h, 0, std::min<size_t>(nb, (size_t)1 << 20));
        (void)hipMemcpyAsync(ctx->d_frames, h, std::min<size_t>(nb, (size_t)1 << 20), hipMemcpyHostToDevice, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
    } else { h = nullptr; (void)hipGetLastError(); }
    HIPCHK(hipMemsetAsync(ctx->d_frames, 0, (size_t)n * ctx->g.fsz, ctx->stream));
    ctx->st_ahead = true;
    int rc = encode_range(ctx, 0, n);
    uint64_t bits = 0;
    const int npk = (n / L) * L;
    if (!rc && npk > 0) {
        rc = icsp_pack_count(ctx, 0, npk, &bits);
        if (!rc) rc = pack_write(ctx, 0, npk, 0);
        if (!rc && h) (void)hipMemcpyAsync(h, ctx->pk.out, std::min(nb, ctx->pk_cap), hipMemcpyDeviceToHost, ctx->stream);
    }
    ctx->pk_first = -1;
    if (!rc) rc = icsp_sync(ctx);
    if (h) (void)hipHostFree(h);
    return rc;
}

namespace {
// The one-call host path as a pipeline: the frames go up chunk by chunk (whole GOPs) on the device's upload stream from a helper
// thread, every chunk is encoded as soon as it is there, and its levels and reconstruction come down on the device's download
// stream while the next chunk is being encoded and the one after it uploaded -- transfers in both directions and kernels side by
// side, which a plain upload / encode / download sequence (18.8 k CIF frames/s, round 3) never has.  Caller memory that is pinned
// (icsp_host_alloc, icsp_host_register) is the DMA source / target itself; memory that is not goes through two pinned staging
// buffers each way, filled and emptied by a few helper threads (one thread copies 10-12 GB/s, the link moves 57).  The small arrays
// (ACflags, mode bits, vector differences: 12 bytes per macroblock) come down once at the end.
// pack: instead of (or beside) the levels the caller gets the packed body (icsp_pack_bits) -- 4 % of the bytes.
// Replaces the loop of single_thread_encoding (ENC:217-245) + YCbCrLoad's frames (ENC:247-283) as input; same bytes as
// icsp_upload + icsp_encode_resident + icsp_download.
static int gop_pipeline(icsp_ctx* ctx, const uint8_t* yuv, int n, int16_t* levels, uint8_t* acflag, uint8_t* mpm, int8_t* mvd, uint8_t* recon,
                 uint8_t* body, size_t body_cap, uint64_t* nbits)
{
    ENTER(ctx);
    if (!yuv) return ICSP_ERR_UNENOUGH_PARAM;
    if (body && !nbits) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = check_range(ctx, 0, n)) return rc;
    if (nbits) *nbits = 0;
    if (n == 0) return ICSP_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const Geo& g = ctx->g;
    const size_t fsz = (size_t)g.fsz, nmb = (size_t)g.nmb, lvf = nmb * 384 * sizeof(int16_t);       // bytes per frame: input / recon, levels
    const int L = ctx->p.intra_period > 0 ? ctx->p.intra_period : 1;
    // Chunks of whole GOPs.  The downloads carry three times the bytes of the uploads and set the pace, so the first chunk is
    // small -- the download stream starts early -- and every next one twice as large (its upload and its kernels, a quarter of a
    // millisecond for CIF I frames whatever their number, then fit under the download of the one before), up to about 32 MB of
    // results, from where a transfer runs at the link's rate anyway (300 CIF frames: 19 + 38 + 76 + 67 + ... against five chunks of
    // 67: 89 k -> 100 k frames/s from pinned memory).
    const size_t out_per_frame = (levels ? lvf : 0) + (recon ? fsz : 0) + fsz / 4;                 // (never 0)
    long long cf = (long long)(((size_t)32 << 20) / out_per_frame);                                // frames of the largest chunk
    cf = std::max<long long>(L, (cf + L - 1) / L * L);
    if (cf > n) cf = n;
    std::vector<int> c_first, c_n;
    // Only the bits come back (body, no levels, no reconstruction): nothing big goes down, and the call is as long as its uploads plus whatever the LAST
    // chunk still needs once it has arrived (its kernels -- an I-frame launch takes 0.2 ms whatever its size --, its packing, its bits).
    // So: few chunks, the last one a quarter of the frames (at most a quarter of the largest chunk), the others equal; every chunk's bits
    // are packed and fetched behind its own encode while the next chunk's kernels run (below).  300 CIF frames: 225 + 75.
    const bool tail_bound = body && !levels && !recon;              // (with the reconstruction to bring down the downloads set the pace: doubling chunks, below)
    if (tail_bound) {
        // halves: 1/2, 1/4, ... of what is left, down to about a sixth of the largest chunk (300 CIF frames: 150 + 75 + 75 -> 150, 100, 50)
        const long long gops = (n + L - 1) / L;
        std::vector<long long> g;                                   // chunk sizes in GOPs
        long long left = gops;
        const long long cap = std::max<long long>(1, cf / L);
        while (left > 0) {
            long long k = std::min<long long>(cap, std::max<long long>(1, (left + 1) / 2));
            if (g.size() >= 2 || left <= std::max<long long>(1, gops / 6)) k = std::min(left, cap);      // the third chunk takes the rest
            g.push_back(k); left -= k;
        }
        long long f = 0;
        for (long long k : g) { const long long fr = std::min<long long>(k * L, n - f); c_first.push_back((int)f); c_n.push_back((int)fr); f += fr; }
    } else {
        long long sz = std::max<long long>(L, ((n + 15) / 16 + L - 1) / L * L);
        for (long long f = 0; f < n; ) {
            const long long k = std::min<long long>(std::min(sz, cf), n - f);
            c_first.push_back((int)f); c_n.push_back((int)k);
            f += k;
            sz *= 2;
        }
        // (a last chunk of a GOP or two is not worth a turn of its own)
        if (c_n.size() >= 2 && c_n.back() < c_n[c_n.size() - 2] / 4 && c_n[c_n.size() - 2] + c_n.back() <= cf + cf / 4) {
            c_n[c_n.size() - 2] += c_n.back(); c_n.pop_back(); c_first.pop_back();
        }
    }
    const int nc = (int)c_n.size();
    for (int k : c_n) cf = std::max<long long>(cf, k);                                             // staging buffers hold the largest
    if (nc == 1 && !ctx->up_stream) {
        // one chunk: nothing to overlap, and the shared transfer streams cost tens of milliseconds to set up
        if (int rc = icsp_upload(ctx, yuv, 0, n)) return rc;
        if (int rc = icsp_encode_resident(ctx, 0, n)) return rc;
        if (int rc = icsp_download(ctx, 0, n, levels, acflag, mpm, mvd, recon)) return rc;
        return body ? icsp_pack_bits(ctx, 0, n, body, body_cap, nbits) : ICSP_OK;
    }
    if (!ctx->up_stream) { if (int rc = icsp_copy_streams(ctx, 1)) return rc; }          // (once per device and process: two DMA engines)
    if (int rc = join_all(ctx)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));                    // whatever still reads the frame store or writes the results
    for (int k = 0; k < kGopEvSets; k++) for (auto& e : ctx->gop_ev[k]) if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const bool in_direct = host_pinned(yuv, (size_t)n * fsz);
    const bool lv_direct = host_pinned(levels, (size_t)n * lvf), rc_direct = host_pinned(recon, (size_t)n * fsz);
    const size_t need_in = in_direct ? 0 : (size_t)cf * fsz;
    const size_t need_out = ((levels && !lv_direct) ? (size_t)cf * lvf : 0) + ((recon && !rc_direct) ? (size_t)cf * fsz : 0);
    std::lock_guard<std::mutex> stage_lock(ctx->stage_in_m);        // (to the end of the call: the uploader thread below fills the staging pair)
    xfer_in_drain(ctx);                                            // (an earlier staged upload may still be reading the buffers)
    if (need_in) { if (int rc = stage_reserve(ctx, true, need_in)) return rc; }
    if (need_out) { if (int rc = stage_reserve(ctx, false, need_out)) return rc; }
    if ((need_in || need_out) && !copy_pool(ctx->gop_pool)) return ICSP_ERR_MEM_ALLOC;

    // ICSP_TRACE_GOP=1 (diagnostics): when every phase of every chunk ended, in microseconds from here, on stderr
    static const bool trace_gop = getenv("ICSP_TRACE_GOP") != nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    auto tr = [&](const char* what, int c) {
        if (trace_gop) fprintf(stderr, "[gop] %8.1f us  %-14s chunk %d (%d frames)\n",
                               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count(), what, c, c < nc ? c_n[c] : 0);
    };
    tr("set-up done", 0);
    // ---- uploader thread: chunk after chunk onto the device's upload stream (one transfer at a time per stream: copy_up's rule)
    std::mutex um; std::condition_variable ucv;
    int uploaded = 0, up_rc = 0;
    std::atomic<bool> cancel{ false };
    std::thread uploader([&] {
        if (hipSetDevice(ctx->device) != hipSuccess) { (void)hipGetLastError(); std::lock_guard<std::mutex> l(um); up_rc = ICSP_ERR_HIP; ucv.notify_all(); return; }
        CopyPool* pool = need_in ? new (std::nothrow) CopyPool(2) : nullptr;       // its own helpers: the caller's pool empties the other direction meanwhile
        for (int c = 0; c < nc && !cancel.load(); c++) {
            const size_t f0 = (size_t)c_first[c], cn = (size_t)c_n[c];
            const uint8_t* src = yuv + f0 * fsz;
            if (!in_direct) {
                if (pool) pool->copy(ctx->gop_stage_in[c & 1], src, cn * fsz); else memcpy(ctx->gop_stage_in[c & 1], src, cn * fsz);
                src = ctx->gop_stage_in[c & 1];
            }
            bool ok;
            {
                std::lock_guard<std::mutex> t(g_up_turn[ctx->slot & 63]);
                ok = hipMemcpyAsync(ctx->d_frames + f0 * fsz, src, cn * fsz, hipMemcpyHostToDevice, ctx->up_stream) == hipSuccess &&
                     hipStreamSynchronize(ctx->up_stream) == hipSuccess;
            }
            std::lock_guard<std::mutex> l(um);
            if (!ok) { (void)hipGetLastError(); up_rc = ICSP_ERR_HIP; ucv.notify_all(); break; }
            uploaded = c + 1;
            ucv.notify_all();
            tr("uploaded", c);
        }
        delete pool;
    });
    struct Joiner { std::thread& t; std::atomic<bool>& c; ~Joiner() { c.store(true); if (t.joinable()) t.join(); } } joiner{ uploader, cancel };

    // ---- this thread: encode chunk c, then bring chunk c-1 down while c runs; the staged results of c-2 leave meanwhile
    int rc = 0;
    uint64_t at_bit = 0;                                           // body: bits placed so far
    if (body) { if ((rc = pack_alloc(ctx))) return rc; ctx->pk_first = -1; }
    auto stage_off_recon = [&](size_t cn) { return (levels && !lv_direct) ? cn * lvf : (size_t)0; };
    auto unstage = [&](int c) {                                                    // staging buffer -> the caller's arrays
        const size_t f0 = (size_t)c_first[c], cn = (size_t)c_n[c];
        const uint8_t* st = ctx->gop_stage_out[c & 1];
        if (levels && !lv_direct) ctx->gop_pool->copy((char*)levels + f0 * lvf, st, cn * lvf);
        if (recon && !rc_direct) ctx->gop_pool->copy(recon + f0 * fsz, st + stage_off_recon(cn), cn * fsz);
    };
    // Bits only: a third thread queues every chunk's kernels the moment the chunk has arrived, so that this one can sit in the packing
    // and fetching of the chunk before (its host waits were what delayed the next encode: gpurun trace, profiles/r06_gop_trace.txt).  The
    // two threads touch disjoint parts of the context (flight records and streams there; packer buffers and the download stream here);
    // with per-kernel timing on they would share its event lists, so then the chunks are queued from this thread as below.
    const bool use_launcher = tail_bound && nc > 1 && !ctx->profiling;
    if (use_launcher && !ctx->single) {
        // every stream an encode may want exists before the launcher starts: this thread reads the stream handles (the per-chunk
        // events below) while that one runs encode_range, which would otherwise create them on first use
        if ((rc = second_stream(ctx))) return rc;
        if ((rc = group_streams(ctx, kMaxPGroups))) return rc;
    }
    int queued = 0, q_rc = 0, packed_upto = -1;
    auto queue_chunk = [&](int c) -> int {
        const size_t f0 = (size_t)c_first[c], cn = (size_t)c_n[c];
        if (int r = encode_range(ctx, (int)f0, (int)cn)) return r;
        tr("encode queued", c);
        hipStream_t sts[5] = { ctx->stream, ctx->stream2, ctx->pstream[0], ctx->pstream[1], ctx->pstream[2] };
        for (int k = 0; k < 5; k++)
            if (sts[k] && (k < 2 || sts[k] != ctx->stream) && hipEventRecord(ctx->gop_ev[c & (kGopEvSets - 1)][k], sts[k]) != hipSuccess) return poison(ctx, "hipEventRecord", hipGetLastError());
        return 0;
    };
    std::thread launcher;
    if (use_launcher) launcher = std::thread([&] {
        if (hipSetDevice(ctx->device) != hipSuccess) { (void)hipGetLastError(); std::lock_guard<std::mutex> l(um); q_rc = ICSP_ERR_HIP; ucv.notify_all(); return; }
        for (int c = 0; c < nc && !cancel.load(); c++) {
            {
                std::unique_lock<std::mutex> l(um);
                ucv.wait(l, [&] { return uploaded > c || up_rc || cancel.load(); });
                if (up_rc || cancel.load()) return;
            }
            // (chunk c's set of events is free once chunk c - kGopEvSets has been packed)
            {
                std::unique_lock<std::mutex> l(um);
                ucv.wait(l, [&] { return packed_upto >= c - kGopEvSets || cancel.load(); });
                if (cancel.load()) return;
            }
            const int r = queue_chunk(c);
            std::lock_guard<std::mutex> l(um);
            if (r) { q_rc = r; ucv.notify_all(); return; }
            queued = c + 1;
            ucv.notify_all();
        }
    });
    struct Joiner2 { std::thread& t; std::atomic<bool>& c; std::condition_variable& cv; ~Joiner2() { c.store(true); cv.notify_all(); if (t.joinable()) t.join(); } } joiner2{ launcher, cancel, ucv };
    for (int c = 0; c <= nc && !rc; c++) {
        if (c < nc && use_launcher) {
            // (queued by the launcher thread; this loop only follows it: chunk c - 1 is handled below once chunk c - 1 is queued)
        } else if (c < nc) {
            {
                std::unique_lock<std::mutex> l(um);
                ucv.wait(l, [&] { return uploaded > c || up_rc; });
                if (up_rc) { rc = up_rc; ctx->err = "icsp_encode_gop: upload failed"; break; }
            }
            // "chunk c is through": an event behind what each stream of the context carries now -- everything of chunk c, nothing
            // of chunk c + 1 -- instead of a join, so that the chunks' kernels overlap the way encode_range lets disjoint ranges
            if ((rc = queue_chunk(c))) break;
        }
        if (c >= 1) {
            const int d = c - 1;
            const size_t f0 = (size_t)c_first[d], cn = (size_t)c_n[d];
            if (use_launcher) {
                std::unique_lock<std::mutex> l(um);
                ucv.wait(l, [&] { return queued > d || q_rc || up_rc; });
                if (q_rc || up_rc) { rc = q_rc ? q_rc : up_rc; if (up_rc) ctx->err = "icsp_encode_gop: upload failed"; break; }
            }
            {
                hipStream_t sts[5] = { ctx->stream, ctx->stream2, ctx->pstream[0], ctx->pstream[1], ctx->pstream[2] };
                for (int k = 0; k < 5 && !rc; k++)     // (an event never recorded counts as complete; a stream created since then carries later chunks only)
                    if (sts[k] && hipEventSynchronize(ctx->gop_ev[d & (kGopEvSets - 1)][k]) != hipSuccess) { rc = ICSP_ERR_HIP; ctx->err = "hipEventSynchronize"; (void)hipGetLastError(); }
                if (rc) break;
            }
            tr("kernels done", d);
            {
                std::lock_guard<std::mutex> t(g_down_turn[ctx->slot & 63]);
                hipStream_t ds = ctx->down_stream;
                uint8_t* st = ctx->gop_stage_out[d & 1];
                hipError_t e = hipSuccess;
                if (levels) e = hipMemcpyAsync(lv_direct ? (void*)((char*)levels + f0 * lvf) : (void*)st, (const char*)ctx->b.levels + f0 * lvf, cn * lvf, hipMemcpyDeviceToHost, ds);
                if (e == hipSuccess && recon) e = hipMemcpyAsync(rc_direct ? recon + f0 * fsz : st + stage_off_recon(cn), ctx->b.recon + f0 * fsz, cn * fsz, hipMemcpyDeviceToHost, ds);
                if (e == hipSuccess && d >= 1 && need_out) unstage(d - 1);        // (the other staging buffer, beside the transfer)
                if (e == hipSuccess) e = hipStreamSynchronize(ds);
                tr("results down", d);
                if (e != hipSuccess) { (void)hipGetLastError(); rc = ICSP_ERR_HIP; ctx->err = std::string("icsp_encode_gop download: ") + hipGetErrorString(e); break; }
                if (tail_bound) {
                    // chunk d's bits, on the download stream (its encode is complete: the events above; chunk d + 1 is being encoded on the
                    // context's own streams meanwhile): count, pack at the bit the body has reached, fetch
                    unsigned long long bits_d = 0;
                    if ((rc = pack_count_on(ctx, (int)f0, (int)cn, ds, &bits_d))) break;
                    tr("bits counted", d);
                    const size_t b0 = (size_t)(at_bit >> 3), nbd = (size_t)(((at_bit & 7) + bits_d + 7) / 8);
                    if (b0 + nbd > body_cap) { rc = ICSP_ERR_RANGE; break; }
                    if (bits_d) {
                        if ((at_bit & 7) == 0) body[b0] = 0;                      // the outermost bytes of a string are OR-ed into the image
                        if (nbd > 1 || (at_bit & 7) == 0) body[b0 + nbd - 1] = 0;
                        if ((rc = pack_place_kernels(ctx, (int)f0, (int)cn, at_bit, body, ds))) break;
                        if ((rc = pack_place_copy(ctx, bits_d, at_bit, body, ds))) break;
                        at_bit += bits_d;
                    }
                    tr("bits down", d);
                    { std::lock_guard<std::mutex> l(um); packed_upto = d; }
                    ucv.notify_all();
                }
            }
        }
    }
    cancel.store(true);
    ucv.notify_all();
    if (launcher.joinable()) launcher.join();
    if (uploader.joinable()) uploader.join();
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if (need_out) unstage(nc - 1);
    // The small arrays of the whole range, and the packed body.  Plain caller memory gets these through the staging buffer too, in
    // pieces: the call never hands the runtime a large plain pointer (which it would pin on the fly: DESIGN.md section 4e).
    if (int r2 = join_all(ctx)) return r2;
    auto fetch = [&](void* dst, const void* dev, size_t bytes) -> int {
        if (!dst || !bytes) return 0;
        DownTurn turn;
        if (int r2 = copy_down_begin(ctx, turn)) return r2;
        if (int r2 = xfer_down(ctx, dst, dev, bytes, down_of(ctx))) return r2;
        return copy_down_end(ctx);
    };
    const size_t nmb6 = (size_t)n * nmb;
    if (int r2 = fetch(acflag, ctx->b.acflag, nmb6 * 6)) return r2;
    if (int r2 = fetch(mpm, ctx->b.mpm, nmb6 * 4)) return r2;
    if (int r2 = fetch(mvd, ctx->b.mvd, nmb6 * 2)) return r2;
    if (tail_bound) *nbits = at_bit;                                // (every chunk's bits were packed and fetched behind its encode)
    else if (body) {
        if (int r2 = icsp_pack_count(ctx, 0, n, nbits)) return r2;
        const size_t nbytes = (size_t)((*nbits + 7) / 8);
        if (nbytes > body_cap) { *nbits = 0; return ICSP_ERR_RANGE; }
        if (nbytes) {
            if (int r2 = pack_write(ctx, 0, n, 0)) return r2;
            if (int r2 = fetch(body, ctx->pk.out, nbytes)) return r2;
        }
    }
    if (ctx->profiling) collect_profile(ctx);
    return ICSP_OK;
}
} // namespace

int icsp_encode_gop(icsp_ctx_t* ctx, const uint8_t* yuv, int n, int16_t* levels, uint8_t* acflag, uint8_t* mpm, int8_t* mvd, uint8_t* recon)
{
    return gop_pipeline(ctx, yuv, n, levels, acflag, mpm, mvd, recon, nullptr, 0, nullptr);
}

int icsp_encode_gop_packed(icsp_ctx_t* ctx, const uint8_t* yuv, int n, uint8_t* recon, uint8_t* body, size_t cap, uint64_t* nbits)
{
    if (!body || !nbits) return ICSP_ERR_UNENOUGH_PARAM;
    return gop_pipeline(ctx, yuv, n, nullptr, nullptr, nullptr, nullptr, recon, body, cap, nbits);
}

// Scheduling knobs of one context (what ICSP_P_GROUPS / ICSP_I_GROUPS set for every context of the process): 0 keeps a value.
// Results never depend on them.  Joins first, so the next encode starts from a clean slate whatever was in flight.
int icsp_set_groups(icsp_ctx_t* ctx, int p_groups, int i_groups)
{
    ENTER(ctx);
    if (p_groups < 0 || p_groups > kMaxPGroups || i_groups < 0 || i_groups > 2) return ICSP_ERR_UNCORRECT_PARAM;
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = join_all(ctx)) return rc;
    ctx->st_ahead = true;
    if (p_groups) ctx->p_groups = p_groups;
    if (i_groups) ctx->i_groups = i_groups;
    return ICSP_OK;
}

// Everything on the context's one stream: no chroma stream, no GOP-group streams, no cross-stream events.  For hosts that
// encode one short batch (a stream costs 10-25 ms of set-up: icsp_enc on a clip of one chunk) or keep a device busy from
// several contexts anyway.  Results never depend on it.
int icsp_single_stream(icsp_ctx_t* ctx, int on)
{
    ENTER(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = join_all(ctx)) return rc;
    ctx->st_ahead = true;
    ctx->single = on != 0;
    return ICSP_OK;
}

// What the last icsp_encode_resident chose (bench.py puts it beside its figures, so that a line explains its own regime):
// form of the intra luma kernel (8 / 32 lanes per block), its waves per workgroup, reconstruction through the LDS ring or not, whether the range went
// whole onto one chain stream, GOP groups.  Any pointer may be null.
int icsp_debug_plan_turns(int intra_period, int k, const int* firsts, const int* ns, int* whole, int* three, int* turn)
{
    if (k < 0 || (k > 0 && (!firsts || !ns))) return ICSP_ERR_UNENOUGH_PARAM;
    TurnState t{ 0, 0, 0, 0, 0 };
    const int L = intra_period > 0 ? intra_period : 1;
    for (int i = 0; i < k; i++) {
        const Turn r = plan_turn(t, firsts[i], ns[i], L, true, true);
        if (whole) whole[i] = r.whole;
        if (three) three[i] = r.three;
        if (turn) turn[i] = r.turn;
    }
    return ICSP_OK;
}

int icsp_debug_stream_pool(int device_id)
{
    if (device_id < 0) return 0;
    StreamPool& sp = stream_pool();
    std::lock_guard<std::mutex> l(sp.m);
    return (int)sp.idle[device_id & 63].size();
}

int icsp_debug_last_choice(icsp_ctx_t* ctx, int* intra_form, int* intra_waves, int* intra_recon_ring, int* whole_range, int* gop_groups, int* intra_row_group)
{
    ENTER(ctx);
    if (intra_row_group) *intra_row_group = ctx->last_rowgroup;
    if (intra_form) *intra_form = ctx->last_form;
    if (intra_waves) *intra_waves = ctx->last_nw;
    if (intra_recon_ring) *intra_recon_ring = ctx->last_ring;
    if (whole_range) *whole_range = ctx->last_whole;
    if (gop_groups) *gop_groups = ctx->last_groups;
    return ICSP_OK;
}

int icsp_device_pci_bus_id(int device, char* out, int cap)
{
    if (!out || cap < 13) return ICSP_ERR_RANGE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= (fake_devices() ? fake_devices() : ndev)) { (void)hipGetLastError(); return ICSP_ERR_NO_DEVICE; }
    device %= ndev;
    if (hipDeviceGetPCIBusId(out, cap, device) != hipSuccess) { (void)hipGetLastError(); return ICSP_ERR_HIP; }
    return ICSP_OK;
}

int icsp_device_numa_node(int device)
{
    char id[32];
    if (icsp_device_pci_bus_id(device, id, (int)sizeof(id)) != ICSP_OK) return -1;
    return icsp_numa_node_of_pci(id);
}

int icsp_device_view(icsp_ctx_t* ctx, icsp_device_view_t* v)
{
    ENTER(ctx);
    if (!v) return ICSP_ERR_UNENOUGH_PARAM;
    v->frames = ctx->d_frames; v->levels = ctx->b.levels; v->acflag = ctx->b.acflag; v->mpm_mode = ctx->b.mpm;
    v->mvd = ctx->b.mvd; v->recon = ctx->b.recon; v->stream = (void*)ctx->stream;
    v->max_frames = ctx->max_frames; v->n_mb = ctx->g.nmb;
    if (int rc = join_all(ctx)) return rc;
    ctx->always_sync = true;           // an outside producer/consumer now shares `stream`: order stream2 against it every time
    return ICSP_OK;
}

int icsp_download_debug(icsp_ctx_t* ctx, int first, int n, int8_t* mv, uint8_t* imode)
{
    ENTER(ctx);
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nmb = ctx->g.nmb, f = first, c = n;
    if (int rc = join_all(ctx)) return rc;
    if (int rc = xfer_down(ctx, mv, ctx->b.mv + f * nmb * 2, c * nmb * 2, ctx->stream)) return rc;
    if (int rc = xfer_down(ctx, imode, ctx->b.imode + f * nmb * 4, c * nmb * 4, ctx->stream)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ICSP_OK;
}

int icsp_debug_keep_coef(icsp_ctx_t* ctx, int on)
{
    ENTER(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    if (on && !ctx->b.coef) {
        hipError_t e = hipMalloc((void**)&ctx->b.coef, (size_t)ctx->max_frames * ctx->g.nmb * 384 * sizeof(double));
        if (e != hipSuccess) { ctx->err = std::string("hipMalloc coef: ") + hipGetErrorString(e); return ICSP_ERR_MEM_ALLOC; }
    }
    ctx->keep_coef = on != 0;
    return ICSP_OK;
}

int icsp_download_coef(icsp_ctx_t* ctx, int first, int n, double* coef)
{
    ENTER(ctx);
    if (!coef) return ICSP_ERR_UNENOUGH_PARAM;
    if (!ctx->b.coef) return ICSP_ERR_UNCORRECT_PARAM;
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t per = (size_t)ctx->g.nmb * 384;
    if (int rc = join_all(ctx)) return rc;
    if (int rc = xfer_down(ctx, coef, ctx->b.coef + (size_t)first * per, (size_t)n * per * sizeof(double), ctx->stream)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ICSP_OK;
}

int icsp_profile_enable(icsp_ctx_t* ctx, int on)
{
    ENTER(ctx);
    ctx->profiling = on != 0;
    ctx->prof_mask = (on == 1) ? 0xffffffffu : ((unsigned)on >> 1);     // 1 = every kernel; otherwise bit (k+1) selects kernel k
    if (on) {
        // events are created here, not at the first timed launch: hipEventCreate inside a measured region would be billed to it
        HIPCHK(hipSetDevice(ctx->device));
        while (ctx->ev_pool.size() + ctx->ev_pending.size() < 256) {
            EvPair e;
            if (hipEventCreate(&e.a) != hipSuccess) break;
            if (hipEventCreate(&e.b) != hipSuccess) { (void)hipEventDestroy(e.a); break; }
            e.kernel = 0;
            ctx->ev_pool.push_back(e);
        }
    }
    return ICSP_OK;
}

int icsp_profile_reset(icsp_ctx_t* ctx)
{
    ENTER(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    collect_profile(ctx);
    memset(ctx->prof_ms, 0, sizeof(ctx->prof_ms)); memset(ctx->prof_n, 0, sizeof(ctx->prof_n));
    return ICSP_OK;
}

int icsp_profile_get(icsp_ctx_t* ctx, int kernel, double* total_ms, long long* launches)
{
    if (!ctx || kernel < 0 || kernel >= ICSP_K_COUNT) return ICSP_ERR_UNCORRECT_PARAM;
    if (ctx->sticky) return ctx->sticky;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    collect_profile(ctx);
    if (total_ms) *total_ms = ctx->prof_ms[kernel];
    if (launches) *launches = ctx->prof_n[kernel];
    return ICSP_OK;
}

// Test hook (tests/test_host_cpu.py, no device needed): a context shell in the state a failed launch-path call leaves behind.
// Every entry point must keep answering ICSP_ERR_HIP without touching the runtime; icsp_destroy releases it.
int icsp_debug_poisoned_context(icsp_ctx_t** out)
{
    if (!out) return ICSP_ERR_UNENOUGH_PARAM;
    icsp_ctx* ctx = new (std::nothrow) icsp_ctx();
    if (!ctx) return ICSP_ERR_MEM_ALLOC;
    ctx->p = icsp_params_t{ 352, 288, 16, 16, 0 };
    ctx->device = 0; ctx->slot = 0; ctx->max_frames = 1;
    memset(&ctx->g, 0, sizeof(ctx->g)); memset(&ctx->b, 0, sizeof(ctx->b)); memset(&ctx->pk, 0, sizeof(ctx->pk));
    memset(ctx->flight, 0, sizeof(ctx->flight));
    ctx->last_first = ctx->last_n = ctx->rr = 0; ctx->whole_ok = true; ctx->single = false; ctx->prio_lo = 0; ctx->i_stream_b = false; ctx->chains3 = false; ctx->prev2_first = ctx->prev2_n = 0; ctx->chroma_cap = 60;
    ctx->stream = ctx->stream2 = nullptr; ctx->ev_fork = ctx->ev_join = nullptr; ctx->up_stream = ctx->down_stream = nullptr;
    for (int k = 0; k < kMaxPGroups; k++) { ctx->pstream[k] = nullptr; ctx->ev_pjoin[k] = nullptr; }
    ctx->d_frames = nullptr; ctx->pk_host = nullptr; ctx->pk_cap = 0; ctx->pk_first = -1; ctx->pk_n = 0; ctx->pk_total = 0;
    ctx->gop_stage_in[0] = ctx->gop_stage_in[1] = ctx->gop_stage_out[0] = ctx->gop_stage_out[1] = nullptr;
    ctx->gop_stage_in_cap = ctx->gop_stage_out_cap = 0; memset(ctx->gop_ev, 0, sizeof(ctx->gop_ev)); ctx->gop_pool = nullptr;
    ctx->xfer_ev_in[0] = ctx->xfer_ev_in[1] = ctx->xfer_ev_out[0] = ctx->xfer_ev_out[1] = nullptr;
    ctx->xfer_in_busy[0] = ctx->xfer_in_busy[1] = false; ctx->up_pool = nullptr;
    ctx->force_intra_group = 0; ctx->intra_waves_g2 = 0; ctx->last_rowgroup = 0;
    ctx->s2_dirty = ctx->st_ahead = ctx->always_sync = ctx->p_dirty = false;
    ctx->keep_coef = ctx->profiling = false; ctx->prof_mask = 0;
    poison(ctx, "icsp_debug_poisoned_context", hipErrorUnknown);
    *out = ctx;
    return ICSP_OK;
}

} // extern "C"
