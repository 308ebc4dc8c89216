/*
 * icsp_oracle.h — CPU restatement of ICSPCodec's per-macroblock encode loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  Nothing under
 * icspcodec_amd/ links, imports or calls it, and the product fails loudly without its
 * HIP library rather than falling back to this code.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py)
 * against fixtures dumped from the compiled reference itself (oracle/_ref, built from
 * /root/reference by oracle/Makefile; generator tools/make_golden.py): function-level
 * DCT/IDCT/quant/SAD/padding/ME vectors and stream-level .bin + recon SHA-256.
 *
 * Citations: ENC = /root/reference/source/encoder/ICSP_Codec_Encoder_source.cpp,
 *            ENC.h = /root/reference/source/encoder/ICSP_Codec_Encoder.h.
 *
 * Data layout (flat planes, no per-block heap objects):
 *   frame      uint8  [W*H*3/2]        planar I420: Y, Cb, Cr            (ENC:274-279)
 *   levels     int16  [nMB][6][64]     zig-zag order; blocks 0-3 Y, 4 Cb, 5 Cr
 *   acflag     uint8  [nMB][6]         1 = all 63 AC levels are zero      (ENC:2784-2792)
 *   mpm_mode   uint8  [nMB][4]         bit0 MPMFlag, bit1 intraPredMode   (I frames)
 *   mvd        int8   [nMB][2]         differential mv x,y                (P frames, ENC:2353)
 *   recon      uint8  [W*H*3/2]        what the reference dumps to test_yuv.yuv (ENC:6408-6410)
 * nMB = (W/16)*(H/16), MB raster order; W,H multiples of 16, W >= 32.
 */
#ifndef ICSP_ORACLE_H
#define ICSP_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- block-level restatements (known-answer tested against the reference object) ---- */
void icsp_oracle_costable(double out[64]);                       /* ENC.h:190-198, [u][x], float promoted */
double icsp_oracle_irt2(void);                                   /* ENC.h:199 */
void icsp_oracle_dct8x8(const int in[64], double out[64]);       /* ENC:2685-2749 (== CDCT ENC:4338-4419) */
void icsp_oracle_idct8x8(const int in[64], double out[64]);      /* ENC:2825-2893 (== CIDCT ENC:4687-4768) */
int  icsp_oracle_quant_luma(double coef, int qstep);             /* ENC:2780 */
int  icsp_oracle_quant_chroma(double coef, int qstep);           /* ENC:4642 */
void icsp_oracle_zigzag(const int in[64], int out[64]);          /* ENC:3014-3096 */
void icsp_oracle_pad(const uint8_t* src, uint8_t* dst, int pad, int w, int h);  /* ENC:2227-2269, dst calloc'ed semantics */
int  icsp_oracle_sad16(const uint8_t* cur, int cur_stride, const uint8_t* ref, int ref_stride); /* ENC:2283-2297 */
/* the 64-step walk of motionEstimation for start state s (0..3): offsets relative to the MB */
void icsp_oracle_me_walk(int state, int dx[64], int dy[64]);     /* ENC:2111-2125 */
/* whole-frame ME with the carried direction state; mv = cur - best (ENC:2073-2155).
 * nsad (optional) receives the number of SAD evaluations per MB. */
void icsp_oracle_me_frame(const uint8_t* curY, const uint8_t* prev_reconY, int w, int h,
                          int* mvx, int* mvy, int* nsad);

/* ---- frame-level ---- */
/* dbg_coef (optional): double[nMB][6][64] DCT output before DC-DPCM/quant (row-major v,u).
 * dbg_mode (optional): uint8[nMB][4] chosen intra mode 0=V 1=H 2=DC. */
void icsp_oracle_intra_frame(const uint8_t* frame, int w, int h, int qp_dc, int qp_ac,
                             int16_t* levels, uint8_t* acflag, uint8_t* mpm_mode, uint8_t* recon,
                             double* dbg_coef, uint8_t* dbg_mode);           /* ENC:556-643, 1876-1983 */
/* dbg_mv (optional): int8[nMB][2] raw motion vectors. */
void icsp_oracle_inter_frame(const uint8_t* frame, const uint8_t* prev_recon, int w, int h,
                             int qp_dc, int qp_ac,
                             int16_t* levels, uint8_t* acflag, int8_t* mvd, uint8_t* recon,
                             double* dbg_coef, int8_t* dbg_mv);              /* ENC:1986-2072 */

/* ---- sequence-level: closed GOPs, optional pthread job queue (the --EnMultiThread analogue,
 *      ICSP_thread.cpp:39-77 / ENC:186-213).  intra_period 0 => every frame I (ENC:219-224).
 *      Outputs are [nframes] x the per-frame layouts above; mpm_mode rows of P frames and mvd rows
 *      of I frames are zero-filled.  Returns 0. */
int icsp_oracle_encode_sequence(const uint8_t* yuv, int nframes, int w, int h,
                                int qp_dc, int qp_ac, int intra_period, int nthreads,
                                int16_t* levels, uint8_t* acflag, uint8_t* mpm_mode, int8_t* mvd,
                                uint8_t* recon);

/* ---- decoder restatement (icsp_oracle_dec.c; DEC = /root/reference/source/decoder/ICSP_Codec_Decoder_source.cpp) ---- */
void icsp_oracle_dec_costable(double out[64]);                   /* DEC.h:19-27: double literals, unlike the encoder's floats */
void icsp_oracle_dec_idct8x8(const int in[64], double out[64]);  /* DEC:3331-3445 (== CIDCT DEC:4220-4300) */
int icsp_oracle_parse_header(const uint8_t* bin, size_t nbytes, int* w, int* h, int* qdc, int* qac, int* period);   /* DEC:14-37 */
/* readBlockData (DEC:38-405): the .bin image -> the encoder-side arrays for nframes frames.  0, or -1 on a short stream. */
int icsp_oracle_parse(const uint8_t* bin, size_t nbytes, int nframes,
                      int16_t* levels, uint8_t* acflag, uint8_t* mpm_mode, int8_t* mvd);
void icsp_oracle_decode_intra(const int16_t* levels, const uint8_t* mpm_mode, int w, int h, int qp_dc, int qp_ac,
                              uint8_t* out);                                               /* DEC:2154-2219 */
void icsp_oracle_decode_inter(const int16_t* levels, const int8_t* mvd, const uint8_t* prev_decoded, int w, int h,
                              int qp_dc, int qp_ac, uint8_t* out, int8_t* dbg_mv);         /* DEC:2220-2272 */
/* IcspCodec::decoding (DEC.h:286-312); intra_period <= 1 => every frame I.  out = [nframes][W*H*3/2] = check_test_*_yuv.yuv */
int icsp_oracle_decode_sequence(const int16_t* levels, const uint8_t* mpm_mode, const int8_t* mvd, int nframes,
                                int w, int h, int qp_dc, int qp_ac, int intra_period, uint8_t* out);
double icsp_oracle_psnr_y(const uint8_t* orig, const uint8_t* dec, int nframes, int w, int h);   /* DEC.h:331-352 */

#ifdef __cplusplus
}
#endif
#endif
