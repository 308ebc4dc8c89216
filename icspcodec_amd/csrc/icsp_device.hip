// icsp_device.hip — HIP kernels (gfx950 / CDNA4) + the device half of the C ABI (include/icsp_hip.h).
//
// Replaces, for whole batches of frames resident in HBM, the reference's frame encoders
//   intraPrediction   ENC:556-643     (+ DPCM_pix_block 851, DCT_block 2685, DPCM_DC_block 3643, Quantization_block 2750,
//                                       reordering 2894, IQuantization_block 2797, IDPCM_DC_block 3991, IDCT_block 2825,
//                                       IDPCM_pix_block 1500, intraCbCr 1876, intraImgReconstruct 1904)
//   interPrediction   ENC:1986-2072   (+ motionEstimation 2073, motionCompensation 2156, mvPrediction 2353,
//                                       interYReconstruct 2298, interCbCr 2625)
// ENC = /root/reference/source/encoder/ICSP_Codec_Encoder_source.cpp.
//
// Arithmetic contract: DCT/IDCT accumulate in IEEE double in the reference's index order
// with separate multiply and add.  This file MUST be compiled with -ffp-contract=off (build() greps the ISA of a
// -DICSP_NO_FMA compile for v_fma_f64 / v_fmac_f64: the compiler must make none).  Only bit-safe shortcuts are taken: x*1.0
// is skipped, terms whose integer factor is zero for the whole block are skipped (they add +-0 to a sum that started at
// +0), and the first pass of each transform, whose products are exact, uses explicit fused multiply-adds (icsp_blk8.hip.inc,
// proof against the reference's object code: oracle/fma_proof.c).
//
// Mapping (icsp_blk8.hip.inc): 8 lanes per 8x8 block for the throughput kernels (a lane owns one row or column, 8 blocks
// per wave; also the intra kernel's throughput form), 32 lanes per block for the latency-bound intra kernel; the two 1-D
// passes of each transform exchange data through a 528-byte LDS tile per block.
// This translation unit holds the device code and, at its end, the launch layer (icsp_kernels.h): the host scheduler, the
// transfers and the C ABI are icsp_sched.cpp, which never names a kernel.
// Pieces: icsp_me.hip.inc (motion search over 2x2-macroblock tiles, per-frame serial kernel with the DC-DPCM chains,
// last-arriver hand-off of the fused launch), icsp_blk8.hip.inc (transform chain, k_residual8, k_intra_luma32,
// k_intra_luma8), icsp_pack.hip.inc (bit packer), icsp_dec.hip.inc (decoder, §8 f4);
// the host-only parts of the ABI are icsp_sched.cpp (contexts, streams, scheduling, transfers) and icsp_bitstream.cpp (bit
// writer / assembler / parser).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "icsp_hip.h"
#include "icsp_kernels.h"










namespace {
using namespace icspk;          // Geo, FrameSel, DevBufs, PackBufs: the kernels' argument types (icsp_kernels.h)

// ------------------------------------------------------------------------------------------------ constants
constexpr double kIrt2 = 0x1.6a09e667f3bccp-1;   // 1.0/sqrt(2.0) in double (ENC.h:199); pinned by tests/test_oracle_golden.py

struct CosTab { double v[64]; };
// ENC.h:190-198 holds cos((2x+1)u*pi/16) as 6-digit float literals; only cos(k*pi/16), k=0..7 occur.
constexpr CosTab make_costab()
{
    CosTab t{};
    const float mag[8] = { 1.0f, 0.980785f, 0.92388f, 0.83147f, 0.707107f, 0.55557f, 0.382683f, 0.19509f };
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++) {
            int m = ((2 * x + 1) * u) % 32;
            if (m > 16) m = 32 - m;
            float f = (m > 8) ? -mag[16 - m] : mag[m];
            t.v[u * 8 + x] = (double)f;           // float literal promoted to double, as at ENC:2715
        }
    return t;
}
__constant__ CosTab c_cos = make_costab();

struct ZigZag { uint8_t pos[64]; };               // raster index (v*8+u) -> scan position
constexpr ZigZag make_zigzag()
{
    ZigZag z{};
    int k = 0;
    for (int s = 0; s < 15; s++) {                 // JPEG zig-zag == ENC:3031-3094
        if (s & 1) { for (int r = (s < 8 ? 0 : s - 7); r <= (s < 8 ? s : 7); r++) z.pos[r * 8 + (s - r)] = (uint8_t)k++; }
        else       { for (int r = (s < 8 ? s : 7); r >= (s < 8 ? 0 : s - 7); r--) z.pos[r * 8 + (s - r)] = (uint8_t)k++; }
    }
    return z;
}
__constant__ ZigZag c_zz = make_zigzag();
struct ZigZagCol { unsigned long long col[8]; };  // column u: the scan positions of (v = 0..7, u), one byte each, v = 0 lowest
constexpr ZigZagCol make_zigzag_col()
{
    const ZigZag z = make_zigzag();
    ZigZagCol c{};
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) c.col[u] |= (unsigned long long)z.pos[v * 8 + u] << (8 * v);
    return c;
}
__constant__ ZigZagCol c_zzcol = make_zigzag_col();

// motion-search walk tables (filled by the host at icsp_create from its own simulation of ENC:2111-2125)
struct MeTables {
    int8_t  un_dx[132], un_dy[132];   // the 129 distinct offsets of the four walks
    uint8_t walk_u[4][64];            // (state, step) -> index into un_*
    uint32_t pack[144];               // per offset u (the last one repeated to 144): what k_me keeps in LDS, precomputed --
                                      //   low half: dword offset of the candidate's first window row | byte shift << 14,
                                      //   high half: (dx & 0xff) | (dy & 0xff) << 8
    int     n_union;
};
__constant__ MeTables c_me;

#ifdef ICSP_DIAG
#ifndef ICSP_DIAG_BLOCK
#define ICSP_DIAG_BLOCK 0        // which workgroup of the launch is stamped
#endif
// Diagnostic build only (tools/diag_intra.hip): per-phase shader-cycle shares of one wave, never in the product build.
__device__ unsigned long long g_diag[16];
#define DIAG_DECL unsigned long long dg_t0 = 0, dg_acc[8] = {0,0,0,0,0,0,0,0}; const bool dg_on = (blockIdx.x == ICSP_DIAG_BLOCK && threadIdx.x < 64);
#define DIAG_START if (dg_on) { __builtin_amdgcn_sched_barrier(0); dg_t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); }
#define DIAG_STAMP(n) if (dg_on) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); dg_acc[n] += t_ - dg_t0; dg_t0 = t_; __builtin_amdgcn_sched_barrier(0); }
#define DIAG_END if (dg_on && (threadIdx.x == 0)) { for (int z_ = 0; z_ < 8; z_++) g_diag[z_] = dg_acc[z_]; g_diag[8] = __builtin_amdgcn_s_memrealtime() - dg_rt0; g_diag[9] = __builtin_amdgcn_s_memtime() - dg_c0; }
#define DIAG_BEGIN unsigned long long dg_rt0 = __builtin_amdgcn_s_memrealtime(), dg_c0 = __builtin_amdgcn_s_memtime();
#else
#define DIAG_DECL
#define DIAG_START
#define DIAG_STAMP(n)
#define DIAG_END
#define DIAG_BEGIN
#endif

#ifdef ICSP_DIAG8
// Diagnostic build only (tools/diag_intra8.hip): per-phase shader cycles of wave 0 of one workgroup of k_intra_luma8
#ifndef ICSP_DIAG_BLOCK
#define ICSP_DIAG_BLOCK 0
#endif
__device__ unsigned long long g_diag8[20];
struct Dg8 { unsigned long long t0, acc[14]; bool on; };
#define DG8_DECL Dg8 dg8; dg8.t0 = 0; for (int z_ = 0; z_ < 14; z_++) dg8.acc[z_] = 0; dg8.on = (blockIdx.x == ICSP_DIAG_BLOCK && threadIdx.x < 64); \
                 const unsigned long long dg8_rt0 = __builtin_amdgcn_s_memrealtime(), dg8_c0 = __builtin_amdgcn_s_memtime();
#define DG8_START(d) if ((d).on) { __builtin_amdgcn_sched_barrier(0); (d).t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); }
// (waits for everything outstanding first: a phase is charged with the memory and LDS latency it started)
#define DG8_STAMP(d, n) if ((d).on) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); (d).acc[n] += t_ - (d).t0; (d).t0 = t_; __builtin_amdgcn_sched_barrier(0); }
#define DG8_END if (dg8.on && threadIdx.x == 0) { for (int z_ = 0; z_ < 14; z_++) g_diag8[z_] = dg8.acc[z_]; g_diag8[14] = __builtin_amdgcn_s_memrealtime() - dg8_rt0; g_diag8[15] = __builtin_amdgcn_s_memtime() - dg8_c0; }
#define DG8_ARG , Dg8* dgp
#define DG8_PASS , &dg8
#define DG8_OFF , (Dg8*)nullptr
#define DG8_IN(n) if (dgp) { DG8_STAMP(*dgp, n) }
#else
#define DG8_DECL
#define DG8_START(d)
#define DG8_STAMP(d, n)
#define DG8_END
#define DG8_ARG
#define DG8_PASS
#define DG8_OFF
#define DG8_IN(n)
#endif

#ifdef ICSP_TIMELINE
// Diagnostic build only (tools/timeline.hip): every workgroup of the P-step kernels logs (start, end) in 100 MHz
// s_memrealtime ticks into a slot of its own (kernel, GOP group, block): no atomics, so the logging does not serialise the
// workgroups; a later launch of the same kernel overwrites an earlier one, what is read back is the LAST P step of a pass.
struct TlEntry { unsigned long long t0, t1; };
__device__ TlEntry g_tl[8 * 2 * 8192];
#define TL_BEGIN const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();
#define TL_END(id, grp) if (threadIdx.x == 0 && blockIdx.x < 8192) { TlEntry* e_ = &g_tl[((id) * 2 + (grp)) * 8192 + blockIdx.x]; e_->t0 = tl_t0; e_->t1 = __builtin_amdgcn_s_memrealtime(); }
#else
#define TL_BEGIN
#define TL_END(id, grp)
#endif


// ------------------------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// frame slot of item `item` of a launch (wave-uniform: a scalar load where there is a table)
__device__ __forceinline__ int fs_slot(const FrameSel& fs, int item) { return fs.table ? fs.table[item] : fs.first + item * fs.stride; }

template <int CTRL> __device__ __forceinline__ int dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// sum over the 64 lanes, result uniform
__device__ __forceinline__ int wave_sum(int v)
{
    v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp<0x141>(v);   // row_half_mirror
    v += dpp<0x140>(v);   // row_mirror  -> every lane holds its 16-lane row sum
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_min(int v)
{
    v = min(v, dpp<0xB1>(v));
    v = min(v, dpp<0x4E>(v));
    v = min(v, dpp<0x141>(v));
    v = min(v, dpp<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int median3(int a, int b, int c)   // the reference's if-chain (ENC:3677-3679)
{
    if ((a > b) && (a > c)) return (b > c) ? b : c;
    else if ((b > a) && (b > c)) return (a > c) ? a : c;
    else return (a > b) ? a : b;
}
__device__ __forceinline__ unsigned long long ballot64(bool c) { return __builtin_amdgcn_ballot_w64(c); }
// Lane mask of x != 0.  hipcc lowers a ballot whose condition has a second use (or sits next to one) to v_cmp, v_cndmask 0/1,
// v_cmp again -- three vector instructions; this is the one v_cmp it should be.  gfx940-class parts need two wait states
// between a vector instruction writing an SGPR and a vector instruction reading it; the compiler inserts them for its own
// code but cannot see into this statement, so they are part of it.
__device__ __forceinline__ unsigned long long ballot_ne0(int x)
{
    unsigned long long m;
    asm volatile("v_cmp_ne_u32_e64 %0, 0, %1\n\ts_nop 1" : "=s"(m) : "v"(x));
    return m;
}
__device__ __forceinline__ int med3i(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }   // same value as median3, branch-free (v_med3_i32)
__device__ __forceinline__ int clip255(int t) { return min(max(t, 0), 255); }

// ------------------------------------------------------------------------------------------------ prediction fetch
// Sample of the reference's padded image (getPaddingImage, ENC:2227-2269) at padded coordinates (py,px): edge
// replication, except that the last padded row and column are never written and stay 0.
__device__ __forceinline__ int pad_fetch(const uint8_t* plane, int w, int h, int pad, int py, int px)
{
    // branch-free: the clamped address is always valid, so the load is unconditional and several fetches can be in flight
    const int y = min(max(py - pad, 0), h - 1), x = min(max(px - pad, 0), w - 1);
    const int v = plane[y * w + x];
    const bool zero = (py == h + 2 * pad - 1) | (px == w + 2 * pad - 1);
    return zero ? 0 : v;
}

// XCD-aware placement of a launch's workgroups.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share
// one: observed on gfx950, used for speed only, never for correctness), and every XCD has an L2 of its own.  A frame's
// workgroups read overlapping lines (search windows, prediction rows) and the next kernel of the P step reads what this
// one wrote, so all workgroups of frame f of a launch -- in every kernel of the step -- get block indices congruent to
// f mod 8: the frame's lines are fetched into ONE L2 instead of up to eight (k_me moved 5.7x its algorithmic bytes with a
// linear mapping).  Block b -> (frame, unit within the frame); grids are padded to 8 * ceil(frames / 8) frames, blocks of
// the padding frames return at once.
// When a launch has few frames (or they are large) a frame is cut into `slices` runs of consecutive units (bands of the
// frame) and the (frame, slice) pairs are dealt over the XCDs instead, so that all 8 XCDs have work.
struct XcdUnit { int frame, unit; };          // unit >= per_frame or frame >= frames: padding, the workgroup returns at once
// 1-D form (k_serial_fused, whose grid has the serial workgroups in front): b = block index among the units' workgroups
__device__ __forceinline__ XcdUnit xcd_unit(int b, int per_frame, int slices)
{
    const int per_vf = (per_frame + slices - 1) / slices;           // everything here is wave-uniform: scalar arithmetic
    const int x = b & 7, j = b >> 3, q = j / per_vf;
    const int vf = x + 8 * q, frame = vf / slices, slice = vf - frame * slices;
    const int unit = slice * per_vf + (j - q * per_vf);
    return XcdUnit{ unit < per_frame ? frame : 0x7fffffff, unit };
}
// 2-D form: grid (8 * per_vf, ceil(frames * slices / 8)), x fastest, so the linear workgroup number -- and with it the XCD --
// is what the 1-D form has, but the two quotients come as blockIdx.x >> 3 and blockIdx.y instead of by division (an integer
// division of wave-uniform values still costs five vector instructions, one of them a reciprocal, and a dozen scalar
// ones: with three of them a kernel spent 4 % of its instructions finding out which workgroup it is).  slices: a power of two.
__device__ __forceinline__ XcdUnit xcd_unit2(int per_frame, int slices)
{
    const int sl = __builtin_ctz((unsigned)slices);
    const int per_vf = (per_frame + slices - 1) >> sl;
    const int vf = ((int)blockIdx.x & 7) + 8 * (int)blockIdx.y, frame = vf >> sl, slice = vf & (slices - 1);
    const int unit = slice * per_vf + ((int)blockIdx.x >> 3);
    return XcdUnit{ unit < per_frame ? frame : 0x7fffffff, unit };
}
int g_force_slices = 0;                                             // ICSP_XCD_SLICES (experiments); 0 = automatic
inline int xcd_slices(int frames, int per_frame)
{
    int s = 1;                                                      // a power of two up to 8: frames * s pairs deal out evenly ...
    if (g_force_slices > 0) { while (2 * s <= g_force_slices && 2 * s <= per_frame) s *= 2; return s; }
    while (s < 8 && frames * s < 32 && per_frame / (2 * s) >= 32) s *= 2;       // ... in slices of at least 32 units
    return s;
}
inline unsigned xcd_grid(int frames, int per_frame, int slices)
{
    const int per_vf = (per_frame + slices - 1) / slices;
    return 8u * (unsigned)((frames * slices + 7) / 8) * (unsigned)per_vf;
}
inline dim3 xcd_grid2(int frames, int per_frame, int slices)
{
    const int per_vf = (per_frame + slices - 1) / slices;
    return dim3(8u * (unsigned)per_vf, (unsigned)((frames * slices + 7) / 8));
}

#include "icsp_me.hip.inc"

// ------------------------------------------------------------------------------------------------ I-frame chroma DC chain
// Chroma of an I frame is transformed without pixel prediction (ENC:4347-4349), so only its DC-DPCM is serial
// (CDPCM_DC_block ENC:4420-4514, CIDPCM_DC_block ENC:4515-4609).  One workgroup per (frame, plane): block sums S of the
// raw pixels (8 lanes per block, v_sad_u8 against 0) go to LDS -- the DC coefficient is ((S*irt2)*irt2)*0.25 exactly,
// both passes of the u=0/v=0 cosine row being exact integer sums -- then one wave walks the chain and the predictors
// are written back for k_residual8.
__global__ __launch_bounds__(256) void k_chroma_dc(Geo g, FrameSel fs, DevBufs b)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ int s_rec[2][512];
    __shared__ int16_t s_dummy;
    __shared__ int s_dummy32;
    const int slot = fs_slot(fs, (int)blockIdx.x);
    const int pl = blockIdx.y;                                 // 0 = Cb, 1 = Cr
    const int cols = g.sw, rows = g.sh, nblk = g.nmb;
    int16_t* s_sp = (int16_t*)s_dyn;                           // [nmb] block sum in, predictor out
    const long long fb = (long long)slot * g.nmb;
    const uint8_t* plane = b.frames + slot * g.fsz + (long long)g.W * g.H + (long long)pl * g.cw * g.ch;
    const int i = threadIdx.x & 7;
    for (int n0 = 0; n0 < nblk; n0 += 32) {
        const int n = min(n0 + (int)(threadIdx.x >> 3), nblk - 1);
        const uint2 row = *(const uint2*)(plane + ((n / cols) * 8 + i) * g.cw + (n % cols) * 8);
        int v = (int)__builtin_amdgcn_sad_u8(row.y, 0, __builtin_amdgcn_sad_u8(row.x, 0, 0));
        v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v);      // sum over the 8 rows of the block
        if (i == 0) s_sp[n] = (int16_t)v;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        const int q = g.qdc;
        const uint32_t mg = g.mdc;
        if (rows <= 64) dc_chain_rows64<false>(s_sp, &s_dummy, cols, rows, q, mg, l);
        else            dc_chain_banded<false>(s_sp, &s_dummy, &s_dummy32, &s_rec[0][0], cols, rows, q, mg, l);
    }
    __syncthreads();
    for (int n = threadIdx.x; n < nblk; n += 256) b.dcpred[(fb + n) * 6 + 4 + pl] = s_sp[n];
}

#include "icsp_blk8.hip.inc"
#include "icsp_pack.hip.inc"
#include "icsp_dec.hip.inc"

void build_me_tables(MeTables& t)
{
    // host simulation of the walk of motionEstimation (ENC:2111-2125) from each of the four reachable states;
    // state index = iterations mod 4 starting from (flag, xflag, yflag) = (0, +1, -1) (ENC:2095)
    memset(&t, 0, sizeof(t));
    int n = 0;
    for (int s = 0; s < 4; s++) {
        int flag = 0, xflag = 1, yflag = -1;
        for (int i = 0; i < s; i++) { if (!flag) { flag = 1; xflag = -xflag; } else { flag = 0; yflag = -yflag; } }
        int x0 = 0, y0 = 0, xcnt = 0, ycnt = 0;
        for (int k = 0; k < 64; k++) {
            if (!flag) { if (xflag <= 0) x0 += xcnt; else x0 -= xcnt; flag = 1; xcnt++; xflag = -xflag; }
            else       { if (yflag < 0)  y0 += ycnt; else y0 -= ycnt; flag = 0; ycnt++; yflag = -yflag; }
            int u = -1;
            for (int j = 0; j < n; j++) if (t.un_dx[j] == x0 && t.un_dy[j] == y0) { u = j; break; }
            if (u < 0) { u = n++; t.un_dx[u] = (int8_t)x0; t.un_dy[u] = (int8_t)y0; }
            t.walk_u[s][k] = (uint8_t)u;
        }
    }
    t.n_union = n;
    for (int u = 0; u < 144; u++) {
        const int v = u < n ? u : n - 1;
        const int dx = t.un_dx[v], dy = t.un_dy[v], ox = 16 + dx, oy = 16 + dy;
        t.pack[u] = (uint32_t)((oy * kWinDw + (ox >> 2)) | ((ox & 3) << 14)) | (uint32_t)((dx & 0xff) | ((dy & 0xff) << 8)) << 16;
    }
}

} // namespace

// ================================================================================================ launch layer (icsp_kernels.h)
// What icsp_sched.cpp sees of this file: one function per kernel or kernel family.  Each picks the instantiation, sizes the grid
// and the LDS from the geometry and enqueues on the stream it is given; none waits, none touches a context.
namespace icspk {

hipError_t kernel_attributes()
{
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)k_dec_serial, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)) != hipSuccess) return e;
    // (k_residual8_strided's dynamic LDS is the reservation of the one-per-CU chroma launch: ICSP_CHROMA_CAP, up to 120 KB)
    if ((e = hipFuncSetAttribute((const void*)k_residual8_strided, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)) != hipSuccess) return e;
    // k_frame_serial stages a frame's block sums, vectors and states in dynamic LDS: 15 bytes per macroblock
    return hipFuncSetAttribute((const void*)k_frame_serial, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
}

hipError_t upload_search_tables(hipStream_t st)
{
    MeTables t; build_me_tables(t);
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(c_me), &t, sizeof(t), 0, hipMemcpyHostToDevice, st);
}

void set_xcd_slices(int bands) { g_force_slices = bands; }

// ---- I frames
inline size_t intra8_lds_bytes(const Geo& g, int record_rows = 2) { return (size_t)g.W + g.H + 4 * (size_t)record_rows * (g.cols8 + 2); }     // neighbour state of one frame
// the plain wavefront (t = c8 + 2 r8): built for the frames whose widest PAIRS step does not fit eight waves -- six and eight waves
// with the reconstruction ring (one round per step), twelve and sixteen without it (1088p: 15 waves' worth per step)
template <int NW> static void launch_intra8(const Geo& g, const FrameSel& fs, const DevBufs& b, hipStream_t st)
{
    const int G = fs.count;
    // (two builds of every variant: power-of-two quantiser steps and any other steps -- blk8_chain, QM)
    if constexpr (NW <= 8) {
        const size_t lds = intra8_lds_bytes(g) + (size_t)ring_slots(NW) * 64 * kRingBlocks;
        if (g.qpow2) hipLaunchKernelGGL((k_intra_luma8<NW, true, 0, true>), dim3(G), dim3(NW * 64), lds, st, g, fs, b);
        else         hipLaunchKernelGGL((k_intra_luma8<NW, true, 0, false>), dim3(G), dim3(NW * 64), lds, st, g, fs, b);
    } else {
        if (g.qpow2) hipLaunchKernelGGL((k_intra_luma8<NW, false, 0, true>), dim3(G), dim3(NW * 64), intra8_lds_bytes(g), st, g, fs, b);
        else         hipLaunchKernelGGL((k_intra_luma8<NW, false, 0, false>), dim3(G), dim3(NW * 64), intra8_lds_bytes(g), st, g, fs, b);
    }
}
// rows chained in pairs (always with the ring; NW covers the widest step: one round)
template <int NW> static void launch_intra8_pairs(const Geo& g, const FrameSel& fs, const DevBufs& b, hipStream_t st)
{
    const int G = fs.count;
    const size_t lds = intra8_lds_bytes(g, 4) + (size_t)ring_slots_exact(NW) * 64 * kRingBlocks;
    if (g.qpow2) hipLaunchKernelGGL((k_intra_luma8<NW, true, 2, true>), dim3(G), dim3(NW * 64), lds, st, g, fs, b);
    else         hipLaunchKernelGGL((k_intra_luma8<NW, true, 2, false>), dim3(G), dim3(NW * 64), lds, st, g, fs, b);
}

int intra_luma8_pairs(const Geo& g, const FrameSel& fs, const DevBufs& b, int nwp, hipStream_t st)
{
    if (nwp <= 1)      { launch_intra8_pairs<1>(g, fs, b, st); return 1; }
    else if (nwp <= 2) { launch_intra8_pairs<2>(g, fs, b, st); return 2; }
    else if (nwp <= 3) { launch_intra8_pairs<3>(g, fs, b, st); return 3; }
    else if (nwp <= 4) { launch_intra8_pairs<4>(g, fs, b, st); return 4; }
    else if (nwp <= 5) { launch_intra8_pairs<5>(g, fs, b, st); return 5; }
    else if (nwp <= 6) { launch_intra8_pairs<6>(g, fs, b, st); return 6; }
    launch_intra8_pairs<8>(g, fs, b, st);
    return 8;
}

int intra_luma8_plain(const Geo& g, const FrameSel& fs, const DevBufs& b, int need8, bool* ring, hipStream_t st)
{
    if (ring) *ring = need8 <= 8;
    if (need8 <= 6)       { launch_intra8<6>(g, fs, b, st); return 6; }
    else if (need8 <= 8)  { launch_intra8<8>(g, fs, b, st); return 8; }
    else if (need8 <= 12) { launch_intra8<12>(g, fs, b, st); return 12; }
    launch_intra8<16>(g, fs, b, st);
    return 16;
}

int intra_luma32(const Geo& g, const FrameSel& fs, const DevBufs& b, int nw, hipStream_t st)
{
    const int G = fs.count;
    if (nw <= 2)       { hipLaunchKernelGGL((k_intra_luma32<2, 1>), dim3(G), dim3(128), 0, st, g, fs, b); return 2; }
    else if (nw <= 4)  { hipLaunchKernelGGL((k_intra_luma32<4, 1>), dim3(G), dim3(256), 0, st, g, fs, b); return 4; }
    else if (nw <= 6)  { hipLaunchKernelGGL((k_intra_luma32<6, 3>), dim3(G), dim3(384), 0, st, g, fs, b); return 6; }
    else if (nw <= 8)  { hipLaunchKernelGGL((k_intra_luma32<8, 4>), dim3(G), dim3(512), 0, st, g, fs, b); return 8; }
    else if (nw <= 11) { hipLaunchKernelGGL((k_intra_luma32<11, 1>), dim3(G), dim3(704), 0, st, g, fs, b); return 11; }
    hipLaunchKernelGGL((k_intra_luma32<16, 1>), dim3(G), dim3(1024), 0, st, g, fs, b);
    return 16;
}

void chroma_dc(const Geo& g, const FrameSel& fs, const DevBufs& b, hipStream_t st)
{
    hipLaunchKernelGGL(k_chroma_dc, dim3(fs.count, 2), dim3(256), (size_t)g.nmb * 2, st, g, fs, b);
}

int chroma_wgs_per_frame(const Geo& g) { return ((g.nmb + 3) / 4 + 3) / 4; }

void residual(const Geo& g, const FrameSel& fs, const DevBufs& b, bool inter, hipStream_t st)
{
    // workgroups per frame: luma + chroma waves of eight blocks (P frames), chroma waves only (I frames)
    const int wgs = inter ? ((g.nmb + 1) / 2 + (g.nmb + 3) / 4 + 3) / 4 : chroma_wgs_per_frame(g);
    const int sl = xcd_slices(fs.count, wgs);
    hipLaunchKernelGGL(k_residual8, xcd_grid2(fs.count, wgs, sl), dim3(256), 0, st, g, fs, b, inter ? 1 : 0, wgs, sl);
}

void residual_one_per_cu(const Geo& g, const FrameSel& fs, const DevBufs& b, int n_cu, size_t reserve_lds, hipStream_t st)
{
    hipLaunchKernelGGL(k_residual8_strided, dim3(n_cu), dim3(256), reserve_lds, st, g, fs, b, 0, chroma_wgs_per_frame(g));
}

// ---- P step.  Search workgroups: one per 2x2 macroblock tile; the four-state search takes a run of tiles per workgroup once
// one-per-tile would mean more than about 4096 workgroups, which in the usual case (no flag up) do nothing but get dispatched
bool p_step_fusable(const Geo& g) { return g.nmb < 2048; }
namespace {
struct PStepShape { int tiles, run, runs, s_tiles, s_runs; };
PStepShape p_step_shape(const Geo& g, int Gi)
{
    PStepShape s;
    s.tiles = ((g.sw + 1) / 2) * ((g.sh + 1) / 2);
    int run = (int)(((long long)Gi * s.tiles + 4095) / 4096);
    s.run = run < 1 ? 1 : (run > 32 ? 32 : run);
    s.runs = (s.tiles + s.run - 1) / s.run;
    s.s_tiles = xcd_slices(Gi, s.tiles); s.s_runs = xcd_slices(Gi, s.runs);
    return s;
}
}
void me_search(const Geo& g, const FrameSel& fs, const DevBufs& b, bool with_full_search, hipStream_t st)
{
    const int Gi = fs.count;
    const PStepShape s = p_step_shape(g, Gi);
    hipLaunchKernelGGL((k_me<false>), xcd_grid2(Gi, s.tiles, s.s_tiles), dim3(256), 0, st, g, fs, b, s.tiles, s.s_tiles, 1);
    if (with_full_search) hipLaunchKernelGGL((k_me<true>), xcd_grid2(Gi, s.runs, s.s_runs), dim3(256), 0, st, g, fs, b, s.tiles, s.s_runs, s.run);
}

void frame_serial(const Geo& g, const FrameSel& fs, const DevBufs& b, bool fused, hipStream_t st)
{
    const int Gi = fs.count;
    const size_t serial_lds = serial_lds_bytes(g.nmb, g.sw, g.sh);
    if (fused) {
        const PStepShape s = p_step_shape(g, Gi);
        const unsigned n_serial8 = 8u * (unsigned)((Gi + 7) / 8);
        hipLaunchKernelGGL(k_serial_fused, dim3(n_serial8 + xcd_grid(Gi, s.runs, s.s_runs)), dim3(256), serial_lds, st, g, fs, b, (int)n_serial8, s.runs, s.s_runs, s.run, s.tiles);
    } else {
        // (the serial kernel on its own: 1024 threads for the staging loops of large frames)
        hipLaunchKernelGGL(k_frame_serial, dim3(Gi), dim3(g.nmb >= 2048 ? 1024 : 256), serial_lds, st, g, fs, b);
    }
}

// ---- decoder
template <int NW> static void launch_dec_luma(const Geo& g, const FrameSel& fs, const DevBufs& b, hipStream_t st)
{
    hipLaunchKernelGGL((k_dec_intra_luma32<NW>), dim3(fs.count), dim3(NW * 64), 0, st, g, fs, b);
}
void dec_serial(const Geo& g, int first, int n, int L, const DevBufs& b, hipStream_t st)
{
    hipLaunchKernelGGL(k_dec_serial, dim3(n), dim3(256), (size_t)g.nmb * 16, st, g, first, n, L, b);
}
void dec_intra_luma(const Geo& g, const FrameSel& fs, const DevBufs& b, int nw, hipStream_t st)
{
    if (nw <= 2)       launch_dec_luma<2>(g, fs, b, st);
    else if (nw <= 4)  launch_dec_luma<4>(g, fs, b, st);
    else if (nw <= 6)  launch_dec_luma<6>(g, fs, b, st);
    else if (nw <= 8)  launch_dec_luma<8>(g, fs, b, st);
    else if (nw <= 11) launch_dec_luma<11>(g, fs, b, st);
    else               launch_dec_luma<16>(g, fs, b, st);
}
void dec_blocks(const Geo& g, const FrameSel& fs, const DevBufs& b, int kbase, int kcount, int inter, hipStream_t st)
{
    const long long nblk = (long long)fs.count * g.nmb * kcount;
    hipLaunchKernelGGL(k_dec_blocks, dim3((unsigned)((nblk + 31) / 32)), dim3(256), 0, st, g, fs, b, kbase, kcount, inter);
}

// ---- bit packer
void bits_count_scan(const Geo& g, int first, int n, int L, const DevBufs& b, const PackBufs& pk, hipStream_t st)
{
    const long long ngrp = pack_groups(g, n);
    const int nchunk = pack_chunks(ngrp);
    hipLaunchKernelGGL(k_bits_count, dim3((unsigned)((ngrp + 3) / 4)), dim3(256), 0, st, g, first, n, L, ngrp, b, pk);
    hipLaunchKernelGGL(k_bits_scan, dim3(nchunk), dim3(256), 0, st, ngrp, pk);
    hipLaunchKernelGGL(k_chunk_base, dim3(1), dim3(256), 0, st, nchunk, pk);
}
void bits_pack(const Geo& g, int first, int n, int L, const DevBufs& b, const PackBufs& pk, unsigned at_bit, hipStream_t st)
{
    const long long ngrp = pack_groups(g, n);
    hipLaunchKernelGGL(k_pack_zero, dim3((unsigned)((ngrp + 255) / 256)), dim3(256), 0, st, ngrp, pk, at_bit);
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((ngrp + 3) / 4)), dim3(256), 0, st, g, first, n, L, ngrp, b, pk, at_bit);
}

} // namespace icspk
