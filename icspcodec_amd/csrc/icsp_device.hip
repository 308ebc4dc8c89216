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
// with separate multiply and add.  This file MUST be compiled with -ffp-contract=off (build() greps the ISA for
// v_fma_f64 / v_fmac_f64).  Only bit-safe shortcuts are taken: x*1.0 is skipped, terms whose integer factor is zero
// for the whole block are skipped (they add +-0 to a sum that started at +0).
//
// Mapping: one 64-lane wavefront per 8x8 block, lane = (row r = lane>>3, column c = lane&7).  The two 1-D passes of
// each transform exchange data through a 512-byte LDS tile private to the wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <string>
#include <vector>
#include "icsp_hip.h"

namespace {

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

// motion-search walk tables (filled by the host at icsp_create from its own simulation of ENC:2111-2125)
struct MeTables {
    int8_t  un_dx[132], un_dy[132];   // the 129 distinct offsets of the four walks
    uint8_t walk_u[4][64];            // (state, step) -> index into un_*
    int     n_union;
};
__constant__ MeTables c_me;

// ------------------------------------------------------------------------------------------------ geometry
struct Geo {
    int W, H, sw, sh, nmb, cols8, rows8, cw, ch;
    int qdc, qac;
    uint32_t mdc, mac;                // ceil(2^32/q): |t|/q == umulhi(|t|, m) for |t| < 2^16, q > 1
    long long fsz;                    // bytes per frame = W*H*3/2
};
struct FrameSel { int first, stride, count; };     // item i -> frame slot first + i*stride
struct DevBufs {
    const uint8_t* frames; uint8_t* recon;
    int16_t* levels; uint8_t* acflag; uint8_t* mpm; int8_t* mvd;
    int8_t* mv; uint8_t* imode;       // debug taps / inter-kernel data
    uint32_t* me_ent;                 // [slot][nmb][4] packed (mvx, mvy, next state)
    int16_t* sums;                    // [slot][nmb][6] residual block sums
    int16_t* dcpred;                  // [slot][nmb][6] DC predictors
    double* coef;                     // optional [slot][nmb][6][64]
};

// ------------------------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

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
__device__ __forceinline__ int clip255(int t) { return min(max(t, 0), 255); }

// ------------------------------------------------------------------------------------------------ 8x8 transforms
// Per-lane slices of the cosine table, loaded once per wave.
struct LaneTab {
    double row_c[8];   // cos[c][k]  forward pass 1 (u = c)
    double row_r[8];   // cos[r][k]  forward pass 2 (v = r)
    double col_c[8];   // cos[k][c]  inverse pass 1 (x = c)
    double col_r[8];   // cos[k][r]  inverse pass 2 (y = r)
    int r, c, zzpos;
};
__device__ __forceinline__ void lane_tab_init(LaneTab& T)
{
    int l = lane_id();
    T.r = l >> 3; T.c = l & 7;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        T.row_c[k] = c_cos.v[T.c * 8 + k];
        T.row_r[k] = c_cos.v[T.r * 8 + k];
        T.col_c[k] = c_cos.v[k * 8 + T.c];
        T.col_r[k] = c_cos.v[k * 8 + T.r];
    }
    T.zzpos = c_zz.pos[l];
}

// DCT_block (ENC:2685-2749) for the lane's output coefficient (v = r, u = c).  sx: the wave's 64-double LDS tile.
__device__ __forceinline__ double fdct8x8(const LaneTab& T, double* sx, int err)
{
    sx[T.r * 8 + T.c] = (double)err;
    __builtin_amdgcn_wave_barrier();
    double s = 0.0;
#pragma unroll
    for (int x = 0; x < 8; x++) s += sx[T.r * 8 + x] * T.row_c[x];       // tmp[v][u] += err[v][x]*cos[u][x]
    __builtin_amdgcn_wave_barrier();
    sx[T.r * 8 + T.c] = s;
    __builtin_amdgcn_wave_barrier();
    double o = 0.0;
#pragma unroll
    for (int y = 0; y < 8; y++) o += sx[y * 8 + T.c] * T.row_r[y];       // out[v][u] += tmp[y][u]*cos[v][y]
    __builtin_amdgcn_wave_barrier();
    // row 0 then column 0 times irt2 (DC gets both, ENC:2732-2736); x*1.0 == x so the select is bit-safe
    o = o * ((T.r == 0) ? kIrt2 : 1.0);
    o = o * ((T.c == 0) ? kIrt2 : 1.0);
    return o * (1. / 4.);
}

// IDCT_block (ENC:2825-2893) for the lane's output sample (y = r, x = c).  iq: the lane's dequantised coefficient
// (v = r, u = c).  nz: ballot of iq != 0 over the block, used to skip all-zero columns / rows (bit-safe, §9 Q3).
__device__ __forceinline__ double idct8x8(const LaneTab& T, double* sx, int iq, unsigned long long nz)
{
    double b = (double)iq;
    sx[T.r * 8 + T.c] = (T.c == 0) ? kIrt2 * b : b;                      // Cu[u]*iq[y][u], Cu[0] = irt2, else 1.0
    __builtin_amdgcn_wave_barrier();
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < 8; u++)
        if (nz & (0x0101010101010101ull << u)) s += sx[T.r * 8 + u] * T.col_c[u];   // tmp[y][x] += (Cu*iq[y][u])*cos[u][x]
    __builtin_amdgcn_wave_barrier();
    sx[T.r * 8 + T.c] = (T.r == 0) ? kIrt2 * s : s;                      // Cv[v]*tmp[v][x]
    __builtin_amdgcn_wave_barrier();
    double o = 0.0;
#pragma unroll
    for (int v = 0; v < 8; v++)
        if (nz & (0xffull << (8 * v))) o += sx[v * 8 + T.c] * T.col_r[v];            // out[y][x] += (Cv*tmp[v][x])*cos[v][y]
    __builtin_amdgcn_wave_barrier();
    return o * (1. / 4.);
}

// quantise -> ACflag -> zig-zag store -> dequantise (ENC:2750-2824 luma, 4610-4686 chroma).  coef already has the DC
// predictor subtracted on lane 0.  Returns the dequantised value with the predictor added back on lane 0.
__device__ __forceinline__ int quant_store(const LaneTab& T, double coef, int dcpred, int qdc, int qac, bool chroma,
                                           int16_t* lv, uint8_t* acflag)
{
    int l = lane_id();
    int q = (l == 0) ? qdc : qac;
    int t = chroma ? (int)floor(coef + 0.5) : (int)(coef + 0.5);
    int lvl = t / q;
    unsigned long long nzac = __ballot(lvl != 0 && l != 0);
    lv[T.zzpos] = (int16_t)lvl;
    if (l == 0) *acflag = (nzac == 0) ? 1 : 0;
    int iq = lvl * q;
    if (l == 0) iq += dcpred;
    return iq;
}

// ------------------------------------------------------------------------------------------------ intra luma
// One workgroup per I frame.  8x8 blocks are processed along the 2:1 wavefront t = c8 + 2*r8: block (r8,c8) needs
// the reconstructed pixels of L and U, the modes of L, UL, U and the reconstructed DC of L, U, UR (or UL).
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_intra_luma(Geo g, FrameSel fs, DevBufs b)
{
    __shared__ double s_x[NW][64];
    __shared__ uint8_t s_bot[4096];       // bottom row of the newest reconstructed block in each pixel column
    __shared__ uint8_t s_right[2304];     // right column of the newest reconstructed block in each pixel row
    __shared__ uint8_t s_mode[2][512];    // intra modes, two rolling block rows
    __shared__ int s_rec[2][512];         // reconstructed DC, two rolling block rows

    const int slot = fs.first + blockIdx.x * fs.stride;
    const uint8_t* Y = b.frames + slot * g.fsz;
    uint8_t* rY = b.recon + slot * g.fsz;
    const int wave = threadIdx.x >> 6, l = lane_id();
    LaneTab T; lane_tab_init(T);
    double* sx = s_x[wave];

    const int nsteps = g.cols8 + 2 * (g.rows8 - 1);
    for (int t = 0; t < nsteps; t++) {
        int r_lo = t - (g.cols8 - 1); r_lo = (r_lo <= 0) ? 0 : (r_lo + 1) >> 1;
        int r_hi = min(g.rows8 - 1, t >> 1);
        for (int r8 = r_lo + wave; r8 <= r_hi; r8 += NW) {
            const int c8 = t - 2 * r8;
            const bool upav = r8 > 0, leav = c8 > 0;
            const int mb = (r8 >> 1) * g.sw + (c8 >> 1), k = (r8 & 1) * 2 + (c8 & 1);
            const long long blk = ((long long)slot * g.nmb + mb) * 6 + k;
            const int cur = Y[(r8 * 8 + T.r) * g.W + c8 * 8 + T.c];
            // neighbours
            const uint32_t u0 = *(const uint32_t*)&s_bot[c8 * 8], u1 = *(const uint32_t*)&s_bot[c8 * 8 + 4];
            const uint32_t l0 = *(const uint32_t*)&s_right[r8 * 8], l1 = *(const uint32_t*)&s_right[r8 * 8 + 4];
            const int upx = upav ? (int)((((T.c & 4) ? u1 : u0) >> ((T.c & 3) * 8)) & 0xff) : 128;
            const int ley = leav ? (int)((((T.r & 4) ? l1 : l0) >> ((T.r & 3) * 8)) & 0xff) : 128;
            const int sumU = upav ? (int)__builtin_amdgcn_sad_u8(u1, 0, __builtin_amdgcn_sad_u8(u0, 0, 0)) : 1024;
            const int sumL = leav ? (int)__builtin_amdgcn_sad_u8(l1, 0, __builtin_amdgcn_sad_u8(l0, 0, 0)) : 1024;
            // three candidate residuals (DPCM_pix_0/1/2, ENC:644-743).  Mode 2: (int)(cur - k/16.0) truncates toward
            // zero, and cur - k/16 is exact in double, so it equals the C quotient (16*cur - k)/16.
            const int ksum = sumL + sumU;
            const int e0 = cur - upx, e1 = cur - ley, e2 = (16 * cur - ksum) / 16;
            const int s01 = wave_sum(abs(e0) | (abs(e1) << 16));
            const int sae2 = wave_sum(abs(e2));
            const int sae0 = s01 & 0xffff, sae1 = s01 >> 16;
            int m;
            if (!upav && !leav) m = 2;
            else if (!upav)     m = (sae2 > sae1) ? 1 : 2;                         // ENC:1009
            else if (!leav)     m = (sae2 > sae0) ? 0 : 2;                         // ENC:1161
            else { int mn = min(min(sae0, sae1), sae2); m = (mn == sae0) ? 0 : (mn == sae1) ? 1 : 2; }   // ENC:1314-1332
            const int err = (m == 0) ? e0 : (m == 1) ? e1 : e2;
            // most-probable-mode signalling (ENC:1334-1350)
            int mpm = 0, ipm = 0;
            if (upav || leav) {
                int p;
                if (!upav)      p = s_mode[r8 & 1][c8 - 1];
                else if (!leav) p = s_mode[(r8 - 1) & 1][c8];
                else p = median3(s_mode[r8 & 1][c8 - 1], s_mode[(r8 - 1) & 1][c8 - 1], s_mode[(r8 - 1) & 1][c8]);
                mpm = (m == p);
                if (!mpm) ipm = (p == 0) ? ((m == 1) ? 0 : 1) : ((m == 0) ? 0 : 1);
            }
            // DC predictor on the 8x8 grid (ENC:3652-3818)
            int dcp;
            if (r8 == 0 && c8 == 0) dcp = 1024;
            else if (r8 == 0) dcp = s_rec[0][c8 - 1];
            else if (c8 == 0) dcp = s_rec[(r8 - 1) & 1][0];
            else {
                int L = s_rec[r8 & 1][c8 - 1], U = s_rec[(r8 - 1) & 1][c8];
                if (((r8 & 1) && (c8 & 1)) || c8 == g.cols8 - 1) dcp = median3(L, s_rec[(r8 - 1) & 1][c8 - 1], U);
                else dcp = median3(L, U, s_rec[(r8 - 1) & 1][c8 + 1]);
            }
            double coef = fdct8x8(T, sx, err);
            if (b.coef) b.coef[blk * 64 + l] = coef;
            if (l == 0) coef = coef - dcp;
            const int iq = quant_store(T, coef, dcp, g.qdc, g.qac, false, b.levels + blk * 64, b.acflag + blk);
            const unsigned long long nz = __ballot(iq != 0);
            const double v = idct8x8(T, sx, iq, nz);
            // reconstruction: the SUM idct + prediction is truncated (ENC:754, 767, 800, 843)
            int px;
            if (m == 0)      px = (int)(v + (double)upx);
            else if (m == 1) px = (int)(v + (double)ley);
            else             px = (int)(v + (double)ksum / 16.0);
            px = clip255(px);
            rY[(r8 * 8 + T.r) * g.W + c8 * 8 + T.c] = (uint8_t)px;
            if (T.r == 7) s_bot[c8 * 8 + T.c] = (uint8_t)px;
            if (T.c == 7) s_right[r8 * 8 + T.r] = (uint8_t)px;
            if (l == 0) {
                s_mode[r8 & 1][c8] = (uint8_t)m;
                s_rec[r8 & 1][c8] = iq;
                b.mpm[((long long)slot * g.nmb + mb) * 4 + k] = (uint8_t)(mpm | (ipm << 1));
                b.imode[((long long)slot * g.nmb + mb) * 4 + k] = (uint8_t)m;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ prediction fetch
// Sample of the reference's padded image (getPaddingImage, ENC:2227-2269) at padded coordinates (py,px): edge
// replication, except that the last padded row and column are never written and stay 0.
__device__ __forceinline__ int pad_fetch(const uint8_t* plane, int w, int h, int pad, int py, int px)
{
    if (py == h + 2 * pad - 1 || px == w + 2 * pad - 1) return 0;
    int y = min(max(py - pad, 0), h - 1), x = min(max(px - pad, 0), w - 1);
    return plane[y * w + x];
}

// Residual sample + prediction for the lane's pixel of block (slot, mb, k).  inter=false: chroma of an I frame
// (no prediction, ENC:4347-4349).  inter=true: motion compensated (ENC:2156-2226 luma, 2500-2557 chroma, mv/2).
__device__ __forceinline__ void block_sample(const Geo& g, const DevBufs& b, int slot, int prev_slot, int mb, int k,
                                             bool inter, int r, int c, int& cur, int& pred)
{
    const int R = mb / g.sw, C = mb % g.sw;
    const uint8_t* F = b.frames + slot * g.fsz;
    if (k < 4) {
        const int y = R * 16 + (k >> 1) * 8 + r, x = C * 16 + (k & 1) * 8 + c;
        cur = F[y * g.W + x];
        pred = 0;
        if (inter) {
            const int8_t* mv = b.mv + ((long long)slot * g.nmb + mb) * 2;
            pred = pad_fetch(b.recon + prev_slot * g.fsz, g.W, g.H, 16, y - mv[1] + 16, x - mv[0] + 16);
        }
    } else {
        const long long off = (long long)g.W * g.H + (k == 5 ? g.cw * g.ch : 0);
        const int y = R * 8 + r, x = C * 8 + c;
        cur = F[off + y * g.cw + x];
        pred = 0;
        if (inter) {
            const int8_t* mv = b.mv + ((long long)slot * g.nmb + mb) * 2;
            pred = pad_fetch(b.recon + prev_slot * g.fsz + off, g.cw, g.ch, 8, y - mv[1] / 2 + 8, x - mv[0] / 2 + 8);
        }
    }
}

// ------------------------------------------------------------------------------------------------ block sums
// S = sum of the 64 residual samples of a block.  The DC coefficient is ((S*irt2)*irt2)*0.25 exactly (the u=0 /
// v=0 cosine row is 1.0, so both passes are exact integer sums), which lets the serial DC-DPCM chain run on S
// before the parallel transform kernel.
__global__ __launch_bounds__(256) void k_block_sums(Geo g, FrameSel fs, DevBufs b, int kbase, int kcount, int inter)
{
    const long long id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long total = (long long)fs.count * g.nmb * kcount;
    if (id >= total) return;
    const int k = kbase + (int)(id % kcount);
    const int mb = (int)((id / kcount) % g.nmb);
    const int slot = fs.first + (int)(id / ((long long)kcount * g.nmb)) * fs.stride;
    const int l = lane_id();
    int cur, pred;
    block_sample(g, b, slot, slot - 1, mb, k, inter != 0, l >> 3, l & 7, cur, pred);
    const int s = wave_sum(cur - pred);
    if (l == 0) b.sums[((long long)slot * g.nmb + mb) * 6 + k] = (int16_t)s;
}

// ------------------------------------------------------------------------------------------------ DC chain
// Serial DC-DPCM (DPCM_DC_block ENC:3643, IDPCM_DC_block 3991, CDPCM_DC_block 4420, CIDPCM_DC_block 4515) on block
// sums, one wave per (frame, plane), along the 2:1 wavefront.  Writes the predictor of every block.
__global__ __launch_bounds__(64) void k_dc_chain(Geo g, FrameSel fs, DevBufs b, int chain_base)
{
    __shared__ int s_rec[2][512];
    const int slot = fs.first + blockIdx.x * fs.stride;
    const int chain = chain_base + blockIdx.y;                 // 0 = luma 8x8 grid, 1 = Cb, 2 = Cr
    const bool luma = chain == 0;
    const int cols = luma ? g.cols8 : g.sw, rows = luma ? g.rows8 : g.sh;
    const int l = lane_id();
    const long long fb = (long long)slot * g.nmb;
    const int nsteps = cols + 2 * (rows - 1);
    for (int t = 0; t < nsteps; t++) {
        int r_lo = t - (cols - 1); r_lo = (r_lo <= 0) ? 0 : (r_lo + 1) >> 1;
        const int r_hi = min(rows - 1, t >> 1);
        for (int r = r_lo + l; r <= r_hi; r += 64) {
            const int c = t - 2 * r;
            int p;
            if (r == 0 && c == 0) p = 1024;
            else if (r == 0) p = s_rec[0][c - 1];
            else if (c == 0) p = s_rec[(r - 1) & 1][0];
            else {
                const int L = s_rec[r & 1][c - 1], U = s_rec[(r - 1) & 1][c];
                const bool lul = luma ? (((r & 1) && (c & 1)) || c == cols - 1) : (c == cols - 1);
                p = lul ? median3(L, s_rec[(r - 1) & 1][c - 1], U) : median3(L, U, s_rec[(r - 1) & 1][c + 1]);
            }
            const int mb = luma ? (r >> 1) * g.sw + (c >> 1) : r * g.sw + c;
            const int k = luma ? (r & 1) * 2 + (c & 1) : 3 + chain;
            const double S = (double)b.sums[(fb + mb) * 6 + k];
            double dc = ((S * kIrt2) * kIrt2) * (1. / 4.);
            dc = dc - p;
            const int t0 = luma ? (int)(dc + 0.5) : (int)floor(dc + 0.5);
            const int rec = (t0 / g.qdc) * g.qdc + p;
            b.dcpred[(fb + mb) * 6 + k] = (int16_t)p;
            s_rec[r & 1][c] = rec;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ residual blocks
// Fully parallel transform chain for blocks whose residual does not depend on this frame's reconstruction: chroma of
// I frames and all six blocks of P-frame macroblocks.  One wave per block; DC predictors come from k_dc_chain.
__global__ __launch_bounds__(256) void k_residual(Geo g, FrameSel fs, DevBufs b, int kbase, int kcount, int inter)
{
    __shared__ double s_x[4][64];
    const long long id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long total = (long long)fs.count * g.nmb * kcount;
    if (id >= total) return;
    const int k = kbase + (int)(id % kcount);
    const int mb = (int)((id / kcount) % g.nmb);
    const int slot = fs.first + (int)(id / ((long long)kcount * g.nmb)) * fs.stride;
    const int l = lane_id();
    LaneTab T; lane_tab_init(T);
    double* sx = s_x[threadIdx.x >> 6];
    int cur, pred;
    block_sample(g, b, slot, slot - 1, mb, k, inter != 0, T.r, T.c, cur, pred);
    const long long blk = ((long long)slot * g.nmb + mb) * 6 + k;
    const int dcp = b.dcpred[blk];
    double coef = fdct8x8(T, sx, cur - pred);
    if (b.coef) b.coef[blk * 64 + l] = coef;
    if (l == 0) coef = coef - dcp;
    const bool chroma = k >= 4;
    const int iq = quant_store(T, coef, dcp, g.qdc, g.qac, chroma, b.levels + blk * 64, b.acflag + blk);
    const unsigned long long nz = __ballot(iq != 0);
    const double v = idct8x8(T, sx, iq, nz);
    int px;
    if (!inter)      px = (int)v;                      // I-frame chroma: clip(trunc(idct))          ENC:1964-1971
    else if (!chroma) px = pred + (int)v;              // P luma: residual truncated, then added     ENC:4812, 2343
    else             px = (int)((double)pred + v);     // P chroma: the SUM is truncated             ENC:2605-2612
    px = clip255(px);
    const int R = mb / g.sw, C = mb % g.sw;
    uint8_t* O = b.recon + slot * g.fsz;
    if (k < 4) O[(R * 16 + (k >> 1) * 8 + T.r) * g.W + C * 16 + (k & 1) * 8 + T.c] = (uint8_t)px;
    else O[(long long)g.W * g.H + (k == 5 ? g.cw * g.ch : 0) + (R * 8 + T.r) * g.cw + C * 8 + T.c] = (uint8_t)px;
}

// ------------------------------------------------------------------------------------------------ motion search
// One wave per macroblock.  The current block and its 48x48 search window (prediction of the reference's padded
// image) are staged in LDS; the 129 distinct candidate offsets of the four search-direction states are evaluated
// with v_sad_u8; then each state's 64-step walk is resolved with the reference's rules (first strict minimum wins;
// a second zero-SAD candidate breaks the walk and wins, ENC:2130-2141).  Output: for each start state the motion
// vector and the state the next macroblock starts in.
constexpr int kWinStride = 52;   // bytes per LDS window row (48 + pad, 13 dwords: odd => rows spread over banks)
__global__ __launch_bounds__(256) void k_me_sad(Geo g, FrameSel fs, DevBufs b)
{
    __shared__ uint32_t s_win[4][48 * kWinStride / 4];
    __shared__ uint32_t s_cur[4][64];
    __shared__ int s_sad[4][132];
    const int wave = threadIdx.x >> 6, l = lane_id();
    const long long id = (long long)blockIdx.x * 4 + wave;
    if (id >= (long long)fs.count * g.nmb) return;
    const int mb = (int)(id % g.nmb);
    const int slot = fs.first + (int)(id / g.nmb) * fs.stride;
    const int R = mb / g.sw, C = mb % g.sw;
    const uint8_t* Y = b.frames + slot * g.fsz;
    const uint8_t* P = b.recon + (slot - 1) * g.fsz;
    uint32_t* win = s_win[wave];
    uint32_t* curl = s_cur[wave];
    // current block: 64 dwords
    curl[l] = *(const uint32_t*)(Y + (R * 16 + (l >> 2)) * g.W + C * 16 + (l & 3) * 4);
    // window: padded coordinates origin (R*16, C*16), 48 rows x 12 dwords
    const bool interior = (R > 0) && (C > 0) && (R < g.sh - 1) && (C < g.sw - 1);
    for (int j = l; j < 48 * 12; j += 64) {
        const int wr = j / 12, wc = j % 12;
        uint32_t v;
        if (interior) v = *(const uint32_t*)(P + (R * 16 - 16 + wr) * g.W + C * 16 - 16 + wc * 4);
        else {
            v = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) v |= (uint32_t)pad_fetch(P, g.W, g.H, 16, R * 16 + wr, C * 16 + wc * 4 + q) << (8 * q);
        }
        win[wr * (kWinStride / 4) + wc] = v;
    }
    __builtin_amdgcn_wave_barrier();
    // SADs of the union candidates: lane handles p = l, l+64, and everyone redundantly p = 128
    for (int p = l; p < c_me.n_union; p += 64) {
        const int ox = 16 + c_me.un_dx[p], oy = 16 + c_me.un_dy[p];
        const int a = ox >> 2, sh = ox & 3;
        uint32_t sad = 0;
#pragma unroll 4
        for (int i = 0; i < 16; i++) {
            const uint32_t* wrow = win + (oy + i) * (kWinStride / 4) + a;
            const uint32_t w0 = wrow[0], w1 = wrow[1], w2 = wrow[2], w3 = wrow[3], w4 = wrow[4];
            sad = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w1, w0, sh), curl[i * 4 + 0], sad);
            sad = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w2, w1, sh), curl[i * 4 + 1], sad);
            sad = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w3, w2, sh), curl[i * 4 + 2], sad);
            sad = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w4, w3, sh), curl[i * 4 + 3], sad);
        }
        s_sad[wave][p] = (int)sad;
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t ent = 0;
    for (int s = 0; s < 4; s++) {
        const int u = c_me.walk_u[s][l];
        const int v = s_sad[wave][u];
        const unsigned long long z = __ballot(v == 0);
        int kbest, iters;
        const unsigned long long z2 = z & (z - 1);
        if (z2) { kbest = __builtin_ctzll(z2); iters = kbest + 1; }           // break on the second zero (ENC:2136-2141)
        else    { kbest = wave_min((v << 6) | l) & 63; iters = 64; }          // first strict minimum (ENC:2130)
        const int ub = c_me.walk_u[s][kbest];
        const int mvx = -c_me.un_dx[ub], mvy = -c_me.un_dy[ub];               // mv = cur - best (ENC:2145-2146)
        const uint32_t e = (uint32_t)(mvx & 0xff) | ((uint32_t)(mvy & 0xff) << 8) | ((uint32_t)((s + iters) & 3) << 16);
        if (l == s) ent = e;
    }
    if (l < 4) b.me_ent[((long long)slot * g.nmb + mb) * 4 + l] = ent;
}

// Resolve the search-direction state along the macroblock raster (it is carried from MB to MB and only changes on
// an early break, ENC:2095 vs 2106-2109), emit raw motion vectors and their differential coding (mvPrediction,
// ENC:2353-2425, including the `(y1>x3)` typo).  One wave per P frame.
__global__ __launch_bounds__(64) void k_me_resolve(Geo g, FrameSel fs, DevBufs b)
{
    extern __shared__ int8_t s_mv[];                      // [nmb][2] raw motion vectors of this frame
    const int slot = fs.first + blockIdx.x * fs.stride;
    const int l = lane_id();
    const uint32_t* ent = b.me_ent + (long long)slot * g.nmb * 4;
    int8_t* mv = b.mv + (long long)slot * g.nmb * 2;
    int8_t* mvd = b.mvd + (long long)slot * g.nmb * 2;
    // fast path: no macroblock changes the state
    bool same = true;
    for (int n = l; n < g.nmb; n += 64)
#pragma unroll
        for (int s = 0; s < 4; s++) same = same && (((ent[n * 4 + s] >> 16) & 3) == (uint32_t)s);
    if (__ballot(!same) == 0) {
        for (int n = l; n < g.nmb; n += 64) { uint32_t e = ent[n * 4]; s_mv[n * 2] = (int8_t)(e & 0xff); s_mv[n * 2 + 1] = (int8_t)((e >> 8) & 0xff); }
    } else if (l == 0) {
        int s = 0;
        for (int n = 0; n < g.nmb; n++) {
            uint32_t e = ent[n * 4 + s];
            s_mv[n * 2] = (int8_t)(e & 0xff); s_mv[n * 2 + 1] = (int8_t)((e >> 8) & 0xff);
            s = (e >> 16) & 3;
        }
    }
    __syncthreads();
    for (int n = l; n < g.nmb; n += 64) {
        int px, py;
        const int sw = g.sw;
        if (n == 0) { px = 8; py = 8; }
        else if (n / sw == 0) { px = s_mv[(n - 1) * 2]; py = s_mv[(n - 1) * 2 + 1]; }
        else if (n % sw == 0) { px = s_mv[(n - sw) * 2]; py = s_mv[(n - sw) * 2 + 1]; }
        else {
            const int i1 = n - 1;
            const int i2 = (n % sw == sw - 1) ? n - sw - 1 : n - sw;
            const int i3 = (n % sw == sw - 1) ? n - sw : n - sw + 1;
            const int x1 = s_mv[i1 * 2], x2 = s_mv[i2 * 2], x3 = s_mv[i3 * 2];
            const int y1 = s_mv[i1 * 2 + 1], y2 = s_mv[i2 * 2 + 1], y3 = s_mv[i3 * 2 + 1];
            px = median3(x1, x2, x3);
            if ((y1 > y2) && (y1 > y3))      py = (y2 > y3) ? y2 : y3;
            else if ((y2 > y1) && (y2 > y3)) py = (y1 > x3) ? y1 : y3;        // the reference's typo, kept (ENC:2399)
            else                              py = (y1 > y2) ? y1 : y2;
        }
        mv[n * 2] = s_mv[n * 2]; mv[n * 2 + 1] = s_mv[n * 2 + 1];
        mvd[n * 2] = (int8_t)(s_mv[n * 2] - px);
        mvd[n * 2 + 1] = (int8_t)(s_mv[n * 2 + 1] - py);
    }
}

#include "icsp_blk8.hip.inc"

// ------------------------------------------------------------------------------------------------ host side
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return ICSP_ERR_HIP; } } while (0)

struct EvPair { hipEvent_t a, b; int kernel; };

} // namespace

struct icsp_ctx {
    icsp_params_t p;
    Geo g;
    int device, max_frames, intra_waves;
    hipStream_t stream;
    DevBufs b;
    uint8_t* d_frames;
    bool keep_coef, profiling;
    std::vector<EvPair> ev_pending;
    std::vector<EvPair> ev_pool;
    double prof_ms[ICSP_K_COUNT];
    long long prof_n[ICSP_K_COUNT];
    std::string err;
};

namespace {

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
}

template <typename F> int launch_timed(icsp_ctx* ctx, int kernel, F&& f)
{
    if (!ctx->profiling) { f(); return 0; }
    EvPair e;
    if (!ctx->ev_pool.empty()) { e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
    else { if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return ICSP_ERR_HIP; }
    e.kernel = kernel;
    hipEventRecord(e.a, ctx->stream);
    f();
    hipEventRecord(e.b, ctx->stream);
    ctx->ev_pending.push_back(e);
    return 0;
}

int collect_profile(icsp_ctx* ctx)
{
    for (auto& e : ctx->ev_pending) {
        float ms = 0;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            ctx->prof_ms[e.kernel] += ms; ctx->prof_n[e.kernel] += 1;
        }
        ctx->ev_pool.push_back(e);
    }
    ctx->ev_pending.clear();
    return 0;
}

int check_range(icsp_ctx* ctx, int first, int n)
{
    if (first < 0 || n < 0 || first + n > ctx->max_frames) return ICSP_ERR_RANGE;
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

void launch_intra_luma(icsp_ctx* ctx, const Geo& g, const FrameSel& fs, const DevBufs& b, int G, hipStream_t st);

int encode_range(icsp_ctx* ctx, int first, int n)
{
    const Geo& g = ctx->g;
    const int L = ctx->p.intra_period > 0 ? ctx->p.intra_period : 1;
    if (first % L != 0) return ICSP_ERR_RANGE;
    if (n == 0) return 0;
    DevBufs b = ctx->b;
    if (!ctx->keep_coef) b.coef = nullptr;
    hipStream_t st = ctx->stream;
    const int G = (n + L - 1) / L;
    // ---- step 0: the I frame of every GOP
    {
        FrameSel fs{ first, L, G };
        launch_timed(ctx, ICSP_K_INTRA_LUMA, [&] { launch_intra_luma(ctx, g, fs, b, G, st); });
        const long long nblk = (long long)G * g.nmb * 2;
        launch_timed(ctx, ICSP_K_BLOCK_SUMS, [&] { hipLaunchKernelGGL(k_block_sums, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, st, g, fs, b, 4, 2, 0); });
        launch_timed(ctx, ICSP_K_DC_CHAIN, [&] { hipLaunchKernelGGL(k_dc_chain, dim3(G, 2), dim3(64), 0, st, g, fs, b, 1); });
        launch_timed(ctx, ICSP_K_RESIDUAL, [&] { hipLaunchKernelGGL(k_residual8, dim3((unsigned)((nblk + 31) / 32)), dim3(256), 0, st, g, fs, b, 4, 2, 0); });
    }
    // ---- steps 1..L-1: the i-th P frame of every GOP that has one
    for (int i = 1; i < L; i++) {
        int Gi = 0;
        for (int gop = 0; gop < G; gop++) if (gop * L + i < n) Gi++;
        if (Gi == 0) break;
        FrameSel fs{ first + i, L, Gi };
        const long long nmbs = (long long)Gi * g.nmb, nblk = nmbs * 6;
        launch_timed(ctx, ICSP_K_ME_SAD, [&] { hipLaunchKernelGGL(k_me_sad, dim3((unsigned)((nmbs + 3) / 4)), dim3(256), 0, st, g, fs, b); });
        launch_timed(ctx, ICSP_K_ME_RESOLVE, [&] { hipLaunchKernelGGL(k_me_resolve, dim3(Gi), dim3(64), (size_t)g.nmb * 2, st, g, fs, b); });
        launch_timed(ctx, ICSP_K_BLOCK_SUMS, [&] { hipLaunchKernelGGL(k_block_sums, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, st, g, fs, b, 0, 6, 1); });
        launch_timed(ctx, ICSP_K_DC_CHAIN, [&] { hipLaunchKernelGGL(k_dc_chain, dim3(Gi, 3), dim3(64), 0, st, g, fs, b, 0); });
        launch_timed(ctx, ICSP_K_RESIDUAL, [&] { hipLaunchKernelGGL(k_residual8, dim3((unsigned)((nblk + 31) / 32)), dim3(256), 0, st, g, fs, b, 0, 6, 1); });
    }
    HIPCHK(hipGetLastError());
    return 0;
}

void launch_intra_luma(icsp_ctx* ctx, const Geo& g, const FrameSel& fs, const DevBufs& b, int G, hipStream_t st)
{
    // 32-lane form: two blocks per wave; enough waves for the widest step, capped at 16 (wider steps take rounds)
    const int need = ctx->intra_waves;
    if (need <= 2)       hipLaunchKernelGGL(k_intra_luma32<2>, dim3(G), dim3(128), 0, st, g, fs, b);
    else if (need <= 4)  hipLaunchKernelGGL(k_intra_luma32<4>, dim3(G), dim3(256), 0, st, g, fs, b);
    else if (need <= 6)  hipLaunchKernelGGL(k_intra_luma32<6>, dim3(G), dim3(384), 0, st, g, fs, b);
    else if (need <= 8)  hipLaunchKernelGGL(k_intra_luma32<8>, dim3(G), dim3(512), 0, st, g, fs, b);
    else if (need <= 11) hipLaunchKernelGGL(k_intra_luma32<11>, dim3(G), dim3(704), 0, st, g, fs, b);
    else                 hipLaunchKernelGGL(k_intra_luma32<16>, dim3(G), dim3(1024), 0, st, g, fs, b);
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

const char* icsp_kernel_name(int k)
{
    static const char* names[ICSP_K_COUNT] = { "k_intra_luma", "k_block_sums", "k_dc_chain", "k_residual", "k_me_sad", "k_me_resolve" };
    return (k >= 0 && k < ICSP_K_COUNT) ? names[k] : "?";
}

int icsp_create(icsp_ctx_t** out, const icsp_params_t* p, int device_id, int max_frames)
{
    if (!out || !p) return ICSP_ERR_UNENOUGH_PARAM;
    *out = nullptr;
    if (p->width % 16 || p->height % 16 || p->width < 32 || p->width > 4096 || p->height < 16 || p->height > 2304 ||
        p->qp_dc <= 0 || p->qp_ac <= 0 || p->intra_period < 0 || max_frames <= 0)
        return ICSP_ERR_UNCORRECT_PARAM;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return ICSP_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return ICSP_ERR_NO_DEVICE;
    icsp_ctx* ctx = new (std::nothrow) icsp_ctx();
    if (!ctx) return ICSP_ERR_MEM_ALLOC;
    ctx->p = *p; ctx->device = device_id; ctx->max_frames = max_frames;
    ctx->keep_coef = false; ctx->profiling = false;
    memset(ctx->prof_ms, 0, sizeof(ctx->prof_ms)); memset(ctx->prof_n, 0, sizeof(ctx->prof_n));
    Geo& g = ctx->g;
    g.W = p->width; g.H = p->height; g.sw = g.W / 16; g.sh = g.H / 16; g.nmb = g.sw * g.sh;
    g.cols8 = 2 * g.sw; g.rows8 = 2 * g.sh; g.cw = g.W / 2; g.ch = g.H / 2;
    g.qdc = p->qp_dc; g.qac = p->qp_ac;
    g.mdc = (uint32_t)((0x100000000ull + g.qdc - 1) / (unsigned)g.qdc);     // unused when q == 1 (would not fit 32 bits)
    g.mac = (uint32_t)((0x100000000ull + g.qac - 1) / (unsigned)g.qac);
    g.fsz = (long long)g.W * g.H * 3 / 2;
    ctx->intra_waves = intra_waves_needed(g);
    memset(&ctx->b, 0, sizeof(ctx->b));
    ctx->stream = nullptr;
    const size_t nf = (size_t)max_frames, nmb = (size_t)g.nmb;
    auto fail = [&](int code, const char* what, hipError_t e) { ctx->err = std::string(what) + ": " + hipGetErrorString(e); icsp_destroy(ctx); return code; };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return fail(ICSP_ERR_HIP, "hipStreamCreate", e);
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
    ALLOC(ctx->b.sums, nf * nmb * 6 * sizeof(int16_t));
    ALLOC(ctx->b.dcpred, nf * nmb * 6 * sizeof(int16_t));
#undef ALLOC
    hipMemsetAsync(ctx->b.mpm, 0, nf * nmb * 4, ctx->stream);
    hipMemsetAsync(ctx->b.mvd, 0, nf * nmb * 2, ctx->stream);
    hipMemsetAsync(ctx->b.mv, 0, nf * nmb * 2, ctx->stream);
    hipMemsetAsync(ctx->b.imode, 0, nf * nmb * 4, ctx->stream);
    MeTables t; build_me_tables(t);
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_me), &t, sizeof(t))) != hipSuccess) return fail(ICSP_ERR_HIP, "hipMemcpyToSymbol", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(ICSP_ERR_HIP, "hipStreamSynchronize", e);
    *out = ctx;
    return ICSP_OK;
}

int icsp_destroy(icsp_ctx_t* ctx)
{
    if (!ctx) return ICSP_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (auto& e : ctx->ev_pending) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto& e : ctx->ev_pool) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    hipFree(ctx->d_frames); hipFree(ctx->b.recon); hipFree(ctx->b.levels); hipFree(ctx->b.acflag); hipFree(ctx->b.mpm);
    hipFree(ctx->b.mvd); hipFree(ctx->b.mv); hipFree(ctx->b.imode); hipFree(ctx->b.me_ent); hipFree(ctx->b.sums);
    hipFree(ctx->b.dcpred); hipFree(ctx->b.coef);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return ICSP_OK;
}

int icsp_upload(icsp_ctx_t* ctx, const uint8_t* yuv, int first, int n)
{
    if (!ctx || !yuv) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(ctx->d_frames + (size_t)first * ctx->g.fsz, yuv, (size_t)n * ctx->g.fsz, hipMemcpyHostToDevice, ctx->stream));
    return ICSP_OK;
}

int icsp_encode_resident(icsp_ctx_t* ctx, int first, int n)
{
    if (!ctx) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    return encode_range(ctx, first, n);
}

int icsp_sync(icsp_ctx_t* ctx)
{
    if (!ctx) return ICSP_ERR_UNENOUGH_PARAM;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->profiling) collect_profile(ctx);
    return ICSP_OK;
}

int icsp_download(icsp_ctx_t* ctx, int first, int n, int16_t* levels, uint8_t* acflag, uint8_t* mpm, int8_t* mvd, uint8_t* recon)
{
    if (!ctx) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nmb = ctx->g.nmb, f = first, c = n;
    hipStream_t st = ctx->stream;
    if (levels) HIPCHK(hipMemcpyAsync(levels, ctx->b.levels + f * nmb * 384, c * nmb * 384 * sizeof(int16_t), hipMemcpyDeviceToHost, st));
    if (acflag) HIPCHK(hipMemcpyAsync(acflag, ctx->b.acflag + f * nmb * 6, c * nmb * 6, hipMemcpyDeviceToHost, st));
    if (mpm) HIPCHK(hipMemcpyAsync(mpm, ctx->b.mpm + f * nmb * 4, c * nmb * 4, hipMemcpyDeviceToHost, st));
    if (mvd) HIPCHK(hipMemcpyAsync(mvd, ctx->b.mvd + f * nmb * 2, c * nmb * 2, hipMemcpyDeviceToHost, st));
    if (recon) HIPCHK(hipMemcpyAsync(recon, ctx->b.recon + f * ctx->g.fsz, c * ctx->g.fsz, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (ctx->profiling) collect_profile(ctx);
    return ICSP_OK;
}

int icsp_encode_gop(icsp_ctx_t* ctx, const uint8_t* yuv, int n, int16_t* levels, uint8_t* acflag, uint8_t* mpm, int8_t* mvd, uint8_t* recon)
{
    if (int rc = icsp_upload(ctx, yuv, 0, n)) return rc;
    if (int rc = icsp_encode_resident(ctx, 0, n)) return rc;
    return icsp_download(ctx, 0, n, levels, acflag, mpm, mvd, recon);
}

int icsp_device_view(icsp_ctx_t* ctx, icsp_device_view_t* v)
{
    if (!ctx || !v) return ICSP_ERR_UNENOUGH_PARAM;
    v->frames = ctx->d_frames; v->levels = ctx->b.levels; v->acflag = ctx->b.acflag; v->mpm_mode = ctx->b.mpm;
    v->mvd = ctx->b.mvd; v->recon = ctx->b.recon; v->stream = (void*)ctx->stream;
    v->max_frames = ctx->max_frames; v->n_mb = ctx->g.nmb;
    return ICSP_OK;
}

int icsp_download_debug(icsp_ctx_t* ctx, int first, int n, int8_t* mv, uint8_t* imode)
{
    if (!ctx) return ICSP_ERR_UNENOUGH_PARAM;
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nmb = ctx->g.nmb, f = first, c = n;
    if (mv) HIPCHK(hipMemcpyAsync(mv, ctx->b.mv + f * nmb * 2, c * nmb * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (imode) HIPCHK(hipMemcpyAsync(imode, ctx->b.imode + f * nmb * 4, c * nmb * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ICSP_OK;
}

int icsp_debug_keep_coef(icsp_ctx_t* ctx, int on)
{
    if (!ctx) return ICSP_ERR_UNENOUGH_PARAM;
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
    if (!ctx || !coef) return ICSP_ERR_UNENOUGH_PARAM;
    if (!ctx->b.coef) return ICSP_ERR_UNCORRECT_PARAM;
    if (int rc = check_range(ctx, first, n)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t per = (size_t)ctx->g.nmb * 384;
    HIPCHK(hipMemcpyAsync(coef, ctx->b.coef + (size_t)first * per, (size_t)n * per * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ICSP_OK;
}

int icsp_profile_enable(icsp_ctx_t* ctx, int on) { if (!ctx) return ICSP_ERR_UNENOUGH_PARAM; ctx->profiling = on != 0; return ICSP_OK; }

int icsp_profile_reset(icsp_ctx_t* ctx)
{
    if (!ctx) return ICSP_ERR_UNENOUGH_PARAM;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    collect_profile(ctx);
    memset(ctx->prof_ms, 0, sizeof(ctx->prof_ms)); memset(ctx->prof_n, 0, sizeof(ctx->prof_n));
    return ICSP_OK;
}

int icsp_profile_get(icsp_ctx_t* ctx, int kernel, double* total_ms, long long* launches)
{
    if (!ctx || kernel < 0 || kernel >= ICSP_K_COUNT) return ICSP_ERR_UNCORRECT_PARAM;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    collect_profile(ctx);
    if (total_ms) *total_ms = ctx->prof_ms[kernel];
    if (launches) *launches = ctx->prof_n[kernel];
    return ICSP_OK;
}

} // extern "C"
