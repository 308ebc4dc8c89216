// icsp_kernels.h — internal seam between the two translation units of libicsp_hip.so (not installed, not part of the C ABI):
//   icsp_device.hip   the gfx950 kernels and, at its end, this launch layer: one plain function per kernel (or per kernel family)
//                     that picks the instantiation, sizes the grid / LDS and enqueues on the stream it is given;
//   icsp_sched.cpp    host only: contexts, streams and their pool, flight records, placement rules, transfers, the C ABI
//                     (include/icsp_hip.h).  It never names a kernel.
// A launcher only ENQUEUES; the caller reads hipGetLastError() behind it (launch_timed in icsp_sched.cpp).
#ifndef ICSP_KERNELS_H
#define ICSP_KERNELS_H
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace icspk __attribute__((visibility("hidden"))) {     // internal to libicsp_hip.so: not exported

// ------------------------------------------------------------------------------------------------ kernel arguments
struct Geo {
    int W, H, sw, sh, nmb, cols8, rows8, cw, ch;
    int qdc, qac;
    int prio;                         // the serial kernel's chain waves raise their issue priority (ICSP_SERIAL_PRIO)
    int bands;                        // tall frames: the DC chain's bands as waves of one continued wavefront (ICSP_SERIAL_BANDS)
    uint32_t mdc, mac;                // floor(2^32/q) + 1: |t|/q == umulhi(|t|, m) for |t| < 2^16, q > 1
    uint32_t msw, mtpr;               // the same for sw and for the tiles per row (sw + 1) / 2: n / sw == umulhi(n, msw) for n < 2^16
    int qpow2;                        // both quantiser steps are powers of two: the quantiser is one fused multiply-add + one conversion (quant_pow2)
    double idc, iac;                  //   1 / qdc, 1 / qac (exact then)
    long long fsz;                    // bytes per frame = W*H*3/2
};
struct FrameSel { int first, stride, count; const int* table; };     // item i -> frame slot first + i*stride, or table[i] (a coalesced list of ranges)
struct DevBufs {
    const uint8_t* frames; uint8_t* recon;
    int16_t* levels; uint8_t* acflag; uint8_t* mpm; int8_t* mvd;
    int8_t* mv; uint8_t* imode;       // debug taps / inter-kernel data
    uint32_t* me_ent;                 // [slot][nmb][4] packed (mvx, mvy, next state)
    int* me_flag;                     // [slot] set by k_me<false> when a macroblock of the frame broke out of its walk early
    int* me_done;                     // [slot] k_serial_fused: arrival tickets of a flagged frame's workgroups (0 between launches)
    int16_t* me_sums;                 // [slot][nmb][4][6] residual block sums of a P-frame MB for each search state
    int16_t* dcpred;                  // [slot][nmb][6] DC predictors
    double* coef;                     // optional [slot][nmb][6][64]
};
struct PackBufs {                     // device bit packer (icsp_pack.hip.inc)
    uint32_t* grp_bits;               // [group]
    uint32_t* grp_off;                // [group] bit offset inside its chunk
    unsigned long long* chunk_bits;   // [chunk]
    unsigned long long* chunk_base;   // [chunk + 1] exclusive scan of chunk_bits; [nchunk] = total body bits
    uint32_t* out;                    // body, dwords
};
constexpr int kGrpUnits = 8;          // bit packer: 8x8 blocks per group (one wave), groups per scan chunk
constexpr int kChunkGrps = 2048;

// ------------------------------------------------------------------------------------------------ set-up
hipError_t kernel_attributes();                        // dynamic-LDS limits of the kernels that need more than 64 KB (per icsp_create)
hipError_t upload_search_tables(hipStream_t st);       // the motion search's walk tables -> constant memory (once per device and process)
void set_xcd_slices(int bands);                        // ICSP_XCD_SLICES (experiments); 0 = automatic

// ------------------------------------------------------------------------------------------------ I frames
// waves per workgroup of the builds the forms come in; each returns the NW of the build it launched
int intra_luma32(const Geo&, const FrameSel&, const DevBufs&, int nw_needed, hipStream_t);             // latency form, two blocks per wave
int intra_luma8_pairs(const Geo&, const FrameSel&, const DevBufs&, int nw_needed, hipStream_t);        // 8-lane form, block rows chained in pairs (ring)
int intra_luma8_plain(const Geo&, const FrameSel&, const DevBufs&, int nw8_needed, bool* ring, hipStream_t);   // 8-lane form, plain wavefront
void chroma_dc(const Geo&, const FrameSel&, const DevBufs&, hipStream_t);
int  chroma_wgs_per_frame(const Geo&);                 // k_residual8 workgroups per frame, chroma waves only
// k_residual8: inter == false: the chroma blocks of I frames; true: all six blocks of P frames
void residual(const Geo&, const FrameSel&, const DevBufs&, bool inter, hipStream_t);
// the I-frame chroma blocks through ONE workgroup per CU, each taking every n_cu-th unit and reserving reserve_lds bytes it does not use
void residual_one_per_cu(const Geo&, const FrameSel&, const DevBufs&, int n_cu, size_t reserve_lds, hipStream_t);

// ------------------------------------------------------------------------------------------------ P step
bool p_step_fusable(const Geo&);                       // small frames: the four-state search rides in the serial kernel's launch
void me_search(const Geo&, const FrameSel&, const DevBufs&, bool with_full_search, hipStream_t);       // k_me<false> (+ k_me<true> when not fused)
void frame_serial(const Geo&, const FrameSel&, const DevBufs&, bool fused, hipStream_t);               // k_serial_fused / k_frame_serial

// ------------------------------------------------------------------------------------------------ decoder, bit packer
void dec_serial(const Geo&, int first, int n, int L, const DevBufs&, hipStream_t);
void dec_intra_luma(const Geo&, const FrameSel&, const DevBufs&, int nw_needed, hipStream_t);
void dec_blocks(const Geo&, const FrameSel&, const DevBufs&, int kbase, int kcount, int inter, hipStream_t);
inline long long pack_groups(const Geo& g, long long n) { return (n * g.nmb * 6 + kGrpUnits - 1) / kGrpUnits; }
inline int pack_chunks(long long ngrp) { return (int)((ngrp + kChunkGrps - 1) / kChunkGrps); }
void bits_count_scan(const Geo&, int first, int n, int L, const DevBufs&, const PackBufs&, hipStream_t);     // lengths + the two scans
void bits_pack(const Geo&, int first, int n, int L, const DevBufs&, const PackBufs&, unsigned at_bit, hipStream_t);

} // namespace icspk
#endif
