/*
 * ref_harness.cpp — C-ABI wrappers that call the REFERENCE's own functions (compiled from
 * /root/reference by oracle/Makefile into oracle/_ref/libicsp_ref.so) on plain arrays.
 *
 * TEST INFRASTRUCTURE ONLY.  Used here, where /root/reference exists, by tools/make_golden.py to
 * dump the golden vectors committed under tests/golden/, and by tests/test_oracle_vs_ref.py to
 * compare the restatement (icsp_oracle.c) with the reference on fresh random inputs.  It contains
 * no reference source: it includes the reference header at build time (-I) and links its objects.
 *
 * The reference's stage functions free() their inputs (ENC:2747, 2795, 2823, 1498, 1874), so every
 * stage buffer handed to them is malloc'ed exactly the way intraPrediction does (ENC:573-600).
 */
#include "ICSP_Codec_Encoder.h"
#include <stdint.h>
#include <string.h>

extern "C" {

static Block8i** alloc8i(int n) { Block8i** p = (Block8i**)malloc(sizeof(Block8i*) * n); for (int i = 0; i < n; i++) p[i] = (Block8i*)malloc(sizeof(Block8i)); return p; }
static Block8d** alloc8d(int n) { Block8d** p = (Block8d**)malloc(sizeof(Block8d*) * n); for (int i = 0; i < n; i++) p[i] = (Block8d*)malloc(sizeof(Block8d)); return p; }

void ref_costable(double out[64]) { for (int u = 0; u < 8; u++) for (int x = 0; x < 8; x++) out[u * 8 + x] = (double)costable[u][x]; }
double ref_irt2(void) { return irt2; }

void ref_dct_block(const int in[64], double out[64])
{
    BlockData bd; memset(&bd, 0, sizeof(bd));
    bd.intraErrblck = alloc8i(1); bd.intraDCTblck = alloc8d(1);
    memcpy(bd.intraErrblck[0]->block, in, sizeof(int) * 64);
    DCT_block(bd, 0, 8, INTRA);                 /* frees intraErrblck[0] */
    memcpy(out, bd.intraDCTblck[0]->block, sizeof(double) * 64);
    free(bd.intraDCTblck[0]); free(bd.intraDCTblck); free(bd.intraErrblck);
}

void ref_cdct_block(const int in[64], double out[64])
{
    CBlockData cb; memset(&cb, 0, sizeof(cb));
    cb.interErrblck = (Block8i*)malloc(sizeof(Block8i)); cb.interDCTblck = (Block8d*)malloc(sizeof(Block8d));
    memcpy(cb.interErrblck->block, in, sizeof(int) * 64);
    CDCT_block(cb, 8, INTER);                   /* frees interErrblck */
    memcpy(out, cb.interDCTblck->block, sizeof(double) * 64);
    free(cb.interDCTblck);
}

void ref_idct_block(const int in[64], double out[64])
{
    BlockData bd; memset(&bd, 0, sizeof(bd));
    bd.intraInverseQuanblck = alloc8i(1); bd.intraInverseDCTblck = alloc8d(1);
    memcpy(bd.intraInverseQuanblck[0]->block, in, sizeof(int) * 64);
    IDCT_block(bd, 0, 8, INTRA);
    memcpy(out, bd.intraInverseDCTblck[0]->block, sizeof(double) * 64);
    free(bd.intraInverseQuanblck[0]); free(bd.intraInverseQuanblck);
    free(bd.intraInverseDCTblck[0]); free(bd.intraInverseDCTblck);
}

void ref_cidct_block(const int in[64], double out[64])
{
    CBlockData cb; memset(&cb, 0, sizeof(cb));
    cb.intraInverseQuanblck = (Block8i*)malloc(sizeof(Block8i)); cb.intraInverseDCTblck = (Block8d*)malloc(sizeof(Block8d));
    memcpy(cb.intraInverseQuanblck->block, in, sizeof(int) * 64);
    CIDCT_block(cb, 8, INTRA);
    memcpy(out, cb.intraInverseDCTblck->block, sizeof(double) * 64);
    free(cb.intraInverseQuanblck); free(cb.intraInverseDCTblck);
}

/* luma quantiser + ACflag + zig-zag + dequantiser (ENC:2750-2824, 2894-2912) */
void ref_quant_luma(const double coef[64], int qdc, int qac, int q[64], int zz[64], int iq[64], int* acflag)
{
    BlockData bd; memset(&bd, 0, sizeof(bd));
    bd.blocksize2 = 8;
    bd.intraDCTblck = alloc8d(1); bd.intraQuanblck = alloc8i(1); bd.intraInverseQuanblck = alloc8i(1);
    memcpy(bd.intraDCTblck[0]->block, coef, sizeof(double) * 64);
    Quantization_block(bd, 0, 8, qdc, qac, INTRA);   /* frees intraDCTblck[0] */
    memcpy(q, bd.intraQuanblck[0]->block, sizeof(int) * 64);
    *acflag = bd.intraACflag[0];
    reordering(bd, 0, INTRA);
    memcpy(zz, bd.intraReorderedblck8[0], sizeof(int) * 64);
    IQuantization_block(bd, 0, 8, qdc, qac, INTRA);  /* frees intraQuanblck[0] */
    memcpy(iq, bd.intraInverseQuanblck[0]->block, sizeof(int) * 64);
    free(bd.intraReorderedblck8[0]); free(bd.intraInverseQuanblck[0]);
    free(bd.intraDCTblck); free(bd.intraQuanblck); free(bd.intraInverseQuanblck);
}

/* chroma quantiser (floor rule, ENC:4610-4686) */
void ref_quant_chroma(const double coef[64], int qdc, int qac, int q[64], int zz[64], int iq[64], int* acflag)
{
    CBlockData cb; memset(&cb, 0, sizeof(cb));
    cb.blocksize = 8;
    cb.intraDCTblck = (Block8d*)malloc(sizeof(Block8d)); cb.intraQuanblck = (Block8i*)malloc(sizeof(Block8i));
    cb.intraInverseQuanblck = (Block8i*)malloc(sizeof(Block8i));
    memcpy(cb.intraDCTblck->block, coef, sizeof(double) * 64);
    CQuantization_block(cb, 8, qdc, qac, INTRA);     /* frees intraDCTblck */
    memcpy(q, cb.intraQuanblck->block, sizeof(int) * 64);
    *acflag = cb.intraACflag;
    Creordering(cb, INTRA);
    memcpy(zz, cb.intraReorderedblck, sizeof(int) * 64);
    CIQuantization_block(cb, 8, qdc, qac, INTRA);    /* frees intraQuanblck */
    memcpy(iq, cb.intraInverseQuanblck->block, sizeof(int) * 64);
    free(cb.intraReorderedblck); free(cb.intraInverseQuanblck);
}

void ref_pad(const uint8_t* src, uint8_t* dst, int pad, int w, int h)
{
    int pw = w + 2 * pad, ph = h + 2 * pad;
    memset(dst, 0, (size_t)pw * ph);                 /* the reference callocs (ENC:2085) */
    getPaddingImage((unsigned char*)src, dst, pw, pad, w, h);
}

int ref_sad16(const uint8_t cur[256], const uint8_t ref[256])
{
    unsigned char a[16][16], b[16][16];
    memcpy(a, cur, 256); memcpy(b, ref, 256);
    return getSAD(a, b, 16);
}

/* Entropy code of one value: returns the number of bits, bits[] one per byte (ENC:5417-5602) */
int ref_code_value(int v, uint8_t bits[32])
{
    int n = 0;
    unsigned char* r = DCentropy(v, n);
    memcpy(bits, r, n); free(r);
    return n;
}

/* Whole frames through the reference's own loader structures + frame encoders, mirroring
 * single_thread_encoding (ENC:217-245) minus the bitstream writer.  Extracts everything the
 * bitstream writer would read (ENC:5057-5128, 5152-5234) plus the internal decisions.
 *   levels int16[n][nMB][6][64], acflag u8[n][nMB][6], mpm u8[n][nMB][4] (bit0 MPMFlag, bit1
 *   intraPredMode), mvd i8[n][nMB][2], recon u8[n][W*H*3/2], modes u8[n][nMB][4], mv i8[n][nMB][2] */
int ref_encode_frames(const uint8_t* yuv, int nframes, int w, int h, int qdc, int qac, int period,
                      int16_t* levels, uint8_t* acflag, uint8_t* mpm, int8_t* mvd, uint8_t* recon,
                      uint8_t* modes, int8_t* mv)
{
    IcspCodec* ic = new IcspCodec;                    /* leaked on purpose: the reference's dtor frees half-owned data */
    size_t ysz = (size_t)w * h, csz = ysz / 4, fsz = ysz + 2 * csz;
    ic->YCbCr.nframe = nframes; ic->YCbCr.width = w; ic->YCbCr.height = h;
    ic->YCbCr.Ys = (unsigned char*)malloc(ysz * nframes);
    ic->YCbCr.Cbs = (unsigned char*)malloc(csz * nframes);
    ic->YCbCr.Crs = (unsigned char*)malloc(csz * nframes);
    for (int i = 0; i < nframes; i++) {                /* what YCbCrLoad's fread loop does (ENC:274-279) */
        memcpy(ic->YCbCr.Ys + i * ysz, yuv + i * fsz, ysz);
        memcpy(ic->YCbCr.Cbs + i * csz, yuv + i * fsz + ysz, csz);
        memcpy(ic->YCbCr.Crs + i * csz, yuv + i * fsz + ysz + csz, csz);
    }
    if (splitFrames(*ic) != 0) return -1;
    ic->frames->nblocks8 = 0;
    if (splitBlocks(*ic, 16, 8) != 0) return -2;
    FrameData* fr = ic->frames;
    int nmb = (w / 16) * (h / 16);
    for (int n = 0; n < nframes; n++) {
        int isI = (period == 0) || (n % period == 0);
        if (isI) intraPrediction(fr[n], qdc, qac);
        else     interPrediction(fr[n], fr[n - 1], qdc, qac);
        for (int b = 0; b < nmb; b++) {
            BlockData& bd = fr[n].blocks[b];
            size_t o = ((size_t)n * nmb + b);
            for (int k = 0; k < 4; k++) {
                int* z = isI ? bd.intraReorderedblck8[k] : bd.interReorderedblck8[k];
                for (int i = 0; i < 64; i++) levels[(o * 6 + k) * 64 + i] = (int16_t)z[i];
                acflag[o * 6 + k] = (uint8_t)(isI ? bd.intraACflag[k] : bd.interACflag[k]);
                mpm[o * 4 + k] = isI ? (uint8_t)(bd.MPMFlag[k] | (bd.intraPredMode[k] << 1)) : 0;
                modes[o * 4 + k] = isI ? (uint8_t)bd.DPCMmodePred[k] : 0;
            }
            CBlockData* cb[2] = { &fr[n].Cbblocks[b], &fr[n].Crblocks[b] };
            for (int p = 0; p < 2; p++) {
                int* z = isI ? cb[p]->intraReorderedblck : cb[p]->interReorderedblck;
                for (int i = 0; i < 64; i++) levels[(o * 6 + 4 + p) * 64 + i] = (int16_t)z[i];
                acflag[o * 6 + 4 + p] = (uint8_t)(isI ? cb[p]->intraACflag : cb[p]->interACflag);
            }
            mvd[o * 2] = isI ? 0 : (int8_t)bd.mv.x;   mvd[o * 2 + 1] = isI ? 0 : (int8_t)bd.mv.y;
            mv[o * 2] = isI ? 0 : (int8_t)bd.Reconstructedmv.x; mv[o * 2 + 1] = isI ? 0 : (int8_t)bd.Reconstructedmv.y;
        }
        memcpy(recon + n * fsz, fr[n].reconstructedY, ysz);
        memcpy(recon + n * fsz + ysz, fr[n].reconstructedCb, csz);
        memcpy(recon + n * fsz + ysz + csz, fr[n].reconstructedCr, csz);
    }
    return 0;
}

/* motionEstimation alone on one frame pair (ENC:2073-2155): raw mv per MB */
int ref_me_frame(const uint8_t* curY, const uint8_t* prevY, int w, int h, int* mvx, int* mvy)
{
    int sw = w / 16, sh = h / 16, nmb = sw * sh;
    FrameData cur, prev; memset(&cur, 0, sizeof(cur)); memset(&prev, 0, sizeof(prev));
    cur.blocks = (BlockData*)calloc(nmb, sizeof(BlockData));
    cur.nblocks16 = nmb; cur.nblocks8 = 4; cur.splitWidth = sw; cur.splitHeight = sh;
    for (int b = 0; b < nmb; b++) {
        cur.blocks[b].blocksize1 = 16; cur.blocks[b].blocksize2 = 8;
        cur.blocks[b].originalblck16 = (Block16u*)malloc(sizeof(Block16u));
        for (int y = 0; y < 16; y++)
            memcpy(cur.blocks[b].originalblck16->block[y], curY + ((b / sw) * 16 + y) * w + (b % sw) * 16, 16);
    }
    prev.reconstructedY = (unsigned char*)prevY;
    motionEstimation(cur, prev);
    for (int b = 0; b < nmb; b++) { mvx[b] = cur.blocks[b].mv.x; mvy[b] = cur.blocks[b].mv.y; free(cur.blocks[b].originalblck16); }
    free(cur.blocks);
    return 0;
}

} /* extern "C" */
