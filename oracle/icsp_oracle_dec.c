/*
 * icsp_oracle_dec.c — CPU restatement of ICSPCodec's DECODER: stream parser + reconstruction.
 *
 * TEST INFRASTRUCTURE ONLY (see icsp_oracle.h).  Parity status: PINNED against the compiled reference decoder
 * (oracle/_ref/icsp_ref_dec, built from /root/reference by oracle/Makefile): tests/test_decoder.py compares the decoded
 * planes byte for byte on I and P streams; tests/golden/decoded.json holds SHA-256 of the reference decoder's output for
 * the GPU box, where that binary's sources do not exist.
 *
 * DEC   = /root/reference/source/decoder/ICSP_Codec_Decoder_source.cpp
 * DEC.h = /root/reference/source/decoder/ICSP_Codec_Decoder.h
 *
 * The decoder mirrors the encoder's reconstruction path with ONE numeric difference: its cosine table is written as
 * double literals (DEC.h:19-27) where the encoder's is float literals promoted to double (ENC.h:190-198), so the two
 * reconstructions differ by one grey level on a few pixels (chroma of I frames, everything in P frames).
 */
#include "icsp_oracle.h"
#include "icsp_oracle_internal.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static const double k_cosmag_d[8] = { 1.0, 0.980785, 0.92388, 0.83147, 0.707107, 0.55557, 0.382683, 0.19509 };   /* DEC.h:19-27 */
static double d_cos[8][8];
static double d_irt2;
static int d_zz[64];
static pthread_once_t d_once = PTHREAD_ONCE_INIT;

static void dec_init(void)
{
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++) {
            int m = ((2 * x + 1) * u) % 32;
            if (m > 16) m = 32 - m;
            d_cos[u][x] = (m > 8) ? -k_cosmag_d[16 - m] : k_cosmag_d[m];
        }
    d_irt2 = 1.0 / sqrt(2.0);                                   /* DEC.h:28 */
    int k = 0;                                                  /* izigzagScanning, DEC:2829-2912 */
    for (int s = 0; s < 15; s++) {
        if (s & 1) { for (int r = (s < 8 ? 0 : s - 7); r <= (s < 8 ? s : 7); r++) d_zz[k++] = r * 8 + (s - r); }
        else       { for (int r = (s < 8 ? s : 7); r >= (s < 8 ? 0 : s - 7); r--) d_zz[k++] = r * 8 + (s - r); }
    }
}

void icsp_oracle_dec_costable(double out[64]) { pthread_once(&d_once, dec_init); memcpy(out, d_cos, sizeof(d_cos)); }

/* IDCT_block / CIDCT_block (DEC:3331-3445, 4220-4300): same loops as the encoder's, double table. */
void icsp_oracle_dec_idct8x8(const int in[64], double out[64])
{
    pthread_once(&d_once, dec_init);
    double tmp[8][8], C[8];
    C[0] = d_irt2;
    for (int i = 1; i < 8; i++) C[i] = 1.;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            double s = 0;
            for (int u = 0; u < 8; u++) s += C[u] * (double)in[y * 8 + u] * d_cos[u][x];
            tmp[y][x] = s;
        }
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++) {
            double s = 0;
            for (int v = 0; v < 8; v++) s += C[v] * tmp[v][x] * d_cos[v][y];
            out[y * 8 + x] = s;
        }
    for (int i = 0; i < 64; i++) out[i] *= (1. / 4.);
}

/* ---------------------------------------------------------------- stream parser */

typedef struct { const uint8_t* p; size_t nbits, at; int bad; } bitreader;

static int rd(bitreader* b)
{
    if (b->at >= b->nbits) { b->bad = 1; b->at++; return 0; }     /* past the end: zeros (the reference reads its calloc'ed tail) */
    int v = (b->p[b->at >> 3] >> (7 - (b->at & 7))) & 1;        /* MSB first (DEC:64-70) */
    b->at++;
    return v;
}
static int rdn(bitreader* b, int n) { int v = 0; for (int i = 0; i < n; i++) v = (v << 1) | rd(b); return v; }

/* one value of the DC / AC / MV code (DCientropy DEC:407-608, ACientropy DEC:810-1021, MVientropy DEC:2274-2653):
 * 00 -> 0; 010 s -> 1; 011/100/101/110 s + e bits (e = 1..4) -> 2^e + bits; (e-2) ones, 0, s, e bits for e = 5..11 */
static int rd_value(bitreader* b)
{
    int c = rdn(b, 2);
    if (c == 0) return 0;
    c = (c << 1) | rd(b);
    int e, a;
    if (c == 2) { e = 0; }
    else if (c < 7) { e = c - 2; }
    else { e = 5; while (e < 11 && rd(b) == 1) e++; if (e == 11) { /* nine ones read; the terminating 0 follows */ if (rd(b) != 0) b->bad = 1; } }
    int s = rd(b);
    a = (1 << e) + rdn(b, e);
    return s ? a : -a;
}

int icsp_oracle_parse_header(const uint8_t* bin, size_t nbytes, int* w, int* h, int* qdc, int* qac, int* period)
{
    if (nbytes < 14) return -1;
    if (!(bin[0] == 0 && bin[1] == 73 && bin[2] == 67 && bin[3] == 83 && bin[4] == 80)) return -1;     /* "\0ICSP" (DEC:18) */
    *h = bin[5] | (bin[6] << 8); *w = bin[7] | (bin[8] << 8);
    *qdc = bin[9]; *qac = bin[10];
    int outro = bin[12] | (bin[13] << 8);
    *period = (outro & 0x1F80) >> 7;                            /* DEC:29 */
    return 0;
}

static void rd_block(bitreader* b, int16_t* lv, uint8_t* acflag)
{
    lv[0] = (int16_t)rd_value(b);
    int ac = rd(b);
    *acflag = (uint8_t)ac;
    if (ac) { b->at += 63; for (int i = 1; i < 64; i++) lv[i] = 0; }                  /* DEC:127-132 */
    else for (int i = 1; i < 64; i++) lv[i] = (int16_t)rd_value(b);
}

/* readBlockData (DEC:38-405).  Frame n is intra iff period <= 1 or n % period == 0 (the reference divides by a header
 * period of 0; its all-intra streams carry 1, DEC.h:293).  Returns 0, or -1 if the stream ends early. */
int icsp_oracle_parse(const uint8_t* bin, size_t nbytes, int nframes,
                      int16_t* levels, uint8_t* acflag, uint8_t* mpm_mode, int8_t* mvd)
{
    int w, h, qdc, qac, period;
    if (icsp_oracle_parse_header(bin, nbytes, &w, &h, &qdc, &qac, &period)) return -1;
    const int nmb = (w / 16) * (h / 16);
    bitreader b = { bin + 14, (nbytes - 14) * 8, 0, 0 };
    for (int f = 0; f < nframes; f++) {
        const int intra = period <= 1 || f % period == 0;
        for (int n = 0; n < nmb; n++) {
            const size_t o = (size_t)f * nmb + n;
            if (intra) {
                mvd[o * 2] = mvd[o * 2 + 1] = 0;
                for (int k = 0; k < 4; k++) {
                    int flag = rd(&b), mode = rd(&b);
                    mpm_mode[o * 4 + k] = (uint8_t)(flag | (mode << 1));
                    rd_block(&b, levels + (o * 6 + k) * 64, acflag + o * 6 + k);
                }
            } else {
                (void)rd(&b);                                                          /* MVmodeflag (DEC:286) */
                mvd[o * 2] = (int8_t)rd_value(&b);
                mvd[o * 2 + 1] = (int8_t)rd_value(&b);
                for (int k = 0; k < 4; k++) { mpm_mode[o * 4 + k] = 0; rd_block(&b, levels + (o * 6 + k) * 64, acflag + o * 6 + k); }
            }
            rd_block(&b, levels + (o * 6 + 4) * 64, acflag + o * 6 + 4);
            rd_block(&b, levels + (o * 6 + 5) * 64, acflag + o * 6 + 5);
            /* running past the end is an error — except inside the very last macroblock, whose tail the reference pair
             * itself garbles: the encoder leaves the final partial byte right-aligned (ENC:4956), the decoder reads it
             * MSB-first (DEC:64-70), so the last few values can decode to longer codes than were written */
            if ((b.bad || b.at > b.nbits) && !(f == nframes - 1 && n == nmb - 1)) return -1;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- reconstruction */

static uint8_t clip255(int t) { t = (t > 255) ? 255 : t; t = (t < 0) ? 0 : t; return (uint8_t)t; }

/* reorderingblck + IQuantization_block + IDPCM_DC_block + IDCT_block for one block (DEC:2768-3445). Returns the DC. */
static int inverse_chain(const int16_t* lv, int dcpred, int qdc, int qac, double idct[64])
{
    int iq[64];
    for (int i = 0; i < 64; i++) iq[d_zz[i]] = lv[i] * ((d_zz[i] == 0) ? qdc : qac);
    iq[0] += dcpred;
    icsp_oracle_dec_idct8x8(iq, idct);
    return iq[0];
}

/* intraPredictionDecode (DEC:2154-2219): mode from MPMFlag / intraPredMode (IDPCM_pix_block, DEC:3446-3821),
 * pixels = clip(trunc(idct + predictor)) (IDPCM_pix_0/1/2, DEC:3822-3915); chroma = clip(trunc(idct)) (DEC:4056-4068). */
void icsp_oracle_decode_intra(const int16_t* levels, const uint8_t* mpm_mode, int w, int h, int qdc, int qac, uint8_t* out)
{
    pthread_once(&d_once, dec_init);
    const int sw = w / 16, sh = h / 16, nmb = sw * sh, cols8 = 2 * sw, rows8 = 2 * sh, cw = w / 2, ch = h / 2;
    uint8_t *rY = out, *rCb = out + w * h, *rCr = rCb + cw * ch;
    int* mode = (int*)calloc((size_t)rows8 * cols8, sizeof(int));
    int* recdc = (int*)calloc((size_t)rows8 * cols8, sizeof(int));
    int* crec[2] = { (int*)calloc(nmb, sizeof(int)), (int*)calloc(nmb, sizeof(int)) };
    for (int n = 0; n < nmb; n++) {
        const int R = n / sw, C = n % sw;
        for (int k = 0; k < 4; k++) {
            const int r8 = 2 * R + (k >> 1), c8 = 2 * C + (k & 1), upav = r8 > 0, leav = c8 > 0;
            const int flag = mpm_mode[n * 4 + k] & 1, bit = (mpm_mode[n * 4 + k] >> 1) & 1;
            int m;
            if (!upav && !leav) m = 2;                                              /* DEC:3461 */
            else {
                int p;
                if (!upav)      p = mode[r8 * cols8 + c8 - 1];
                else if (!leav) p = mode[(r8 - 1) * cols8 + c8];
                else p = icspo_median3(mode[r8 * cols8 + c8 - 1], mode[(r8 - 1) * cols8 + c8 - 1], mode[(r8 - 1) * cols8 + c8]);
                if (flag) m = p;
                else if (p == 0) m = bit ? 2 : 1;
                else if (p == 2) m = bit ? 1 : 0;
                else             m = bit ? 2 : 0;
            }
            mode[r8 * cols8 + c8] = m;
            double idct[64];
            recdc[r8 * cols8 + c8] = inverse_chain(levels + (n * 6 + k) * 64, icspo_luma_dcpred(recdc, r8, c8, cols8), qdc, qac, idct);
            uint8_t* o = rY + (r8 * 8) * w + c8 * 8;
            const uint8_t* up = o - w;
            const uint8_t* le = o - 1;
            double pl = 0, pu = 0;
            if (!leav) pl = 128 * 8; else for (int i = 0; i < 8; i++) pl += le[i * w];
            if (!upav) pu = 128 * 8; else for (int i = 0; i < 8; i++) pu += up[i];
            const double predVal = (pl + pu) / (double)(8 + 8);
            /* the left column must be read before this block's own pixels overwrite nothing of it: le is outside the block */
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    int t;
                    if (m == 0)      t = (int)(idct[y * 8 + x] + (upav ? (int)up[x] : 128));
                    else if (m == 1) t = (int)(idct[y * 8 + x] + (leav ? (int)le[y * w] : 128));
                    else             t = (int)(idct[y * 8 + x] + predVal);
                    o[y * w + x] = clip255(t);
                }
        }
        for (int pl = 0; pl < 2; pl++) {
            uint8_t* dst = (pl ? rCr : rCb) + (R * 8) * cw + C * 8;
            double idct[64];
            crec[pl][n] = inverse_chain(levels + (n * 6 + 4 + pl) * 64, icspo_chroma_dcpred(crec[pl], n, sw), qdc, qac, idct);
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    int t = (int)((idct[y * 8 + x] > 255) ? 255 : idct[y * 8 + x]);
                    dst[y * cw + x] = (uint8_t)((t < 0) ? 0 : t);
                }
        }
    }
    free(mode); free(recdc); free(crec[0]); free(crec[1]);
}

/* interPredictionDecode (DEC:2220-2272): ImvPrediction (DEC:4301-4370, with its `(y1>x3)` slip), luma = clip(pred +
 * trunc(idct)) (mergeBlock DEC:3953-3985, interYReconstruct DEC:4371-4419), chroma = clip(trunc(pred + idct)) with
 * mv/2 (DEC:2699-2767); getPaddingImage leaves the last padded row and column zero (DEC:4420-4460). */
void icsp_oracle_decode_inter(const int16_t* levels, const int8_t* mvd, const uint8_t* prev, int w, int h, int qdc, int qac,
                              uint8_t* out, int8_t* dbg_mv)
{
    pthread_once(&d_once, dec_init);
    const int sw = w / 16, sh = h / 16, nmb = sw * sh, cols8 = 2 * sw, rows8 = 2 * sh, cw = w / 2, ch = h / 2;
    const uint8_t *pY = prev, *pCb = prev + w * h, *pCr = pCb + cw * ch;
    uint8_t *rY = out, *rCb = out + w * h, *rCr = rCb + cw * ch;
    int* mx = (int*)malloc(sizeof(int) * nmb);
    int* my = (int*)malloc(sizeof(int) * nmb);
    int* recdc = (int*)calloc((size_t)rows8 * cols8, sizeof(int));
    const int pw = w + 32, ph = h + 32;
    uint8_t* pad = (uint8_t*)malloc((size_t)pw * ph);
    icsp_oracle_pad(pY, pad, 16, w, h);
    for (int n = 0; n < nmb; n++) {
        const int R = n / sw, C = n % sw;
        int px, py;
        icspo_mv_pred(mx, my, n, sw, &px, &py);
        mx[n] = mvd[n * 2] + px; my[n] = mvd[n * 2 + 1] + py;
        if (dbg_mv) { dbg_mv[n * 2] = (int8_t)mx[n]; dbg_mv[n * 2 + 1] = (int8_t)my[n]; }
        const int refx = C * 16 - mx[n] + 16, refy = R * 16 - my[n] + 16;
        for (int k = 0; k < 4; k++) {
            const int r8 = 2 * R + (k >> 1), c8 = 2 * C + (k & 1), oy = (k >> 1) * 8, ox = (k & 1) * 8;
            double idct[64];
            recdc[r8 * cols8 + c8] = inverse_chain(levels + (n * 6 + k) * 64, icspo_luma_dcpred(recdc, r8, c8, cols8), qdc, qac, idct);
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    long idx = (long)(refy + oy + y) * pw + refx + ox + x;
                    int p = (idx >= 0 && idx < (long)pw * ph) ? pad[idx] : 0;        /* corrupt vectors must not fault the checker */
                    rY[(R * 16 + oy + y) * w + C * 16 + ox + x] = clip255(p + (int)idct[y * 8 + x]);
                }
        }
    }
    free(pad);
    const int cpw = cw + 16, cph = ch + 16;
    uint8_t* cpad = (uint8_t*)malloc((size_t)cpw * cph);
    int* crec = (int*)calloc(nmb, sizeof(int));
    for (int pl = 0; pl < 2; pl++) {
        uint8_t* dst = pl ? rCr : rCb;
        icsp_oracle_pad(pl ? pCr : pCb, cpad, 8, cw, ch);
        memset(crec, 0, sizeof(int) * nmb);
        for (int n = 0; n < nmb; n++) {
            const int R = n / sw, C = n % sw;
            const int refx = C * 8 - (mx[n] / 2) + 8, refy = R * 8 - (my[n] / 2) + 8;
            double idct[64];
            crec[n] = inverse_chain(levels + (n * 6 + 4 + pl) * 64, icspo_chroma_dcpred(crec, n, sw), qdc, qac, idct);
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    long idx = (long)(refy + y) * cpw + refx + x;
                    int p = (idx >= 0 && idx < (long)cpw * cph) ? cpad[idx] : 0;
                    dst[(R * 8 + y) * cw + C * 8 + x] = clip255((int)(p + idct[y * 8 + x]));
                }
        }
    }
    free(cpad); free(crec); free(recdc); free(mx); free(my);
}

/* IcspCodec::decoding (DEC.h:286-312): frame n is intra iff period <= 1 or n % period == 0. */
int icsp_oracle_decode_sequence(const int16_t* levels, const uint8_t* mpm_mode, const int8_t* mvd, int nframes,
                                int w, int h, int qdc, int qac, int period, uint8_t* out)
{
    const size_t nmb = (size_t)(w / 16) * (h / 16), fsz = (size_t)w * h * 3 / 2;
    for (int f = 0; f < nframes; f++) {
        if (period <= 1 || f % period == 0)
            icsp_oracle_decode_intra(levels + f * nmb * 384, mpm_mode + f * nmb * 4, w, h, qdc, qac, out + f * fsz);
        else
            icsp_oracle_decode_inter(levels + f * nmb * 384, mvd + f * nmb * 2, out + (f - 1) * fsz, w, h, qdc, qac, out + f * fsz, 0);
    }
    return 0;
}

/* the decoder's quality line (DEC.h:331-352): mean over frames of 20*log10(255/sqrt(MSE)) on luma */
double icsp_oracle_psnr_y(const uint8_t* orig, const uint8_t* dec, int nframes, int w, int h)
{
    const size_t fsz = (size_t)w * h * 3 / 2;
    double psnr = 0;
    for (int f = 0; f < nframes; f++) {
        double mse = 0;
        for (int i = 0; i < w * h; i++) { double d = (double)orig[f * fsz + i] - (double)dec[f * fsz + i]; mse += d * d; }
        mse /= (double)(w * h);
        psnr += 20. * log10(255. / sqrt(mse));
    }
    return psnr / (double)nframes;
}
