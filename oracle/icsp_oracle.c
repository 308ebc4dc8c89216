/*
 * icsp_oracle.c — CPU restatement of ICSPCodec's per-macroblock encode loop (see icsp_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; never by the product path.  Parity status: PINNED against the compiled
 * reference (tests/test_oracle_golden.py, fixtures from tools/make_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off (oracle/Makefile).  The reference's results depend on IEEE-754
 * double arithmetic in a fixed order with separate multiply and add, so FMA
 * contraction must stay off and no loop below may be re-associated.
 *
 * ENC = /root/reference/source/encoder/ICSP_Codec_Encoder_source.cpp
 */
#include "icsp_oracle.h"
#include "icsp_oracle_internal.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

/* ---------------------------------------------------------------- constants */

/* ENC.h:190-198: const float costable[8][8] holds cos((2x+1)u*pi/16) as 6-significant-digit
 * literals.  Only eight magnitudes occur, cos(k*pi/16) for k = 0..7; the table is their signed
 * fold.  Values are float literals promoted to double at each use (ENC:2715, 2726, 2864, 2875). */
static const float k_cosmag[8] = { 1.0f, 0.980785f, 0.92388f, 0.83147f, 0.707107f, 0.55557f, 0.382683f, 0.19509f };

static double g_cos[8][8]; /* [u][x] */
static double g_irt2;
static int g_zz[64];       /* zig-zag: scan position -> raster index (row*8+col) */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void init_tables(void)
{
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++) {
            int m = ((2 * x + 1) * u) % 32;        /* angle in units of pi/16, period 32 */
            if (m > 16) m = 32 - m;                 /* cos(-a) = cos(a) */
            float v = (m > 8) ? -k_cosmag[16 - m] : k_cosmag[m]; /* cos(pi - a) = -cos(a); m == 8 never occurs */
            g_cos[u][x] = (double)v;
        }
    g_irt2 = 1.0 / sqrt(2.0);                       /* ENC.h:199 */
    /* ENC:3031-3094 is the JPEG zig-zag: walk anti-diagonals, alternating direction */
    int k = 0;
    for (int s = 0; s < 15; s++) {
        if (s & 1) { for (int r = (s < 8 ? 0 : s - 7); r <= (s < 8 ? s : 7); r++) g_zz[k++] = r * 8 + (s - r); }
        else       { for (int r = (s < 8 ? s : 7); r >= (s < 8 ? 0 : s - 7); r--) g_zz[k++] = r * 8 + (s - r); }
    }
}
static void ensure_init(void) { pthread_once(&g_once, init_tables); }

void icsp_oracle_costable(double out[64]) { ensure_init(); memcpy(out, g_cos, sizeof(g_cos)); }
double icsp_oracle_irt2(void) { ensure_init(); return g_irt2; }

/* ---------------------------------------------------------------- block kernels */

/* ENC:2685-2749.  tmp[v][u] = sum_x in[v][x]*cos[u][x]; out[v][u] = sum_y tmp[y][u]*cos[v][y];
 * row 0 and column 0 times irt2 (DC twice), everything times 1/4. */
void icsp_oracle_dct8x8(const int in[64], double out[64])
{
    ensure_init();
    double tmp[8][8];
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++) {
            double s = 0;
            for (int x = 0; x < 8; x++) s += (double)in[v * 8 + x] * g_cos[u][x];
            tmp[v][u] = s;
        }
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) {
            double s = 0;
            for (int y = 0; y < 8; y++) s += tmp[y][u] * g_cos[v][y];
            out[v * 8 + u] = s;
        }
    for (int i = 0; i < 8; i++) { out[0 * 8 + i] *= g_irt2; out[i * 8 + 0] *= g_irt2; }
    for (int i = 0; i < 64; i++) out[i] *= (1. / 4.);
}

/* ENC:2825-2893.  tmp[y][x] = sum_u (Cu*in[y][u])*cos[u][x]; out[y][x] = sum_v (Cv*tmp[v][x])*cos[v][y]; times 1/4. */
void icsp_oracle_idct8x8(const int in[64], double out[64])
{
    ensure_init();
    double tmp[8][8];
    double C[8];
    C[0] = g_irt2;
    for (int i = 1; i < 8; i++) C[i] = 1.;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            double s = 0;
            for (int u = 0; u < 8; u++) s += C[u] * (double)in[y * 8 + u] * g_cos[u][x];
            tmp[y][x] = s;
        }
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++) {
            double s = 0;
            for (int v = 0; v < 8; v++) s += C[v] * tmp[v][x] * g_cos[v][y];
            out[y * 8 + x] = s;
        }
    for (int i = 0; i < 64; i++) out[i] *= (1. / 4.);
}

int icsp_oracle_quant_luma(double c, int q)   { return (int)(c + 0.5) / q; }          /* ENC:2780 */
int icsp_oracle_quant_chroma(double c, int q) { return (int)floor(c + 0.5) / q; }     /* ENC:4642 */

void icsp_oracle_zigzag(const int in[64], int out[64])
{
    ensure_init();
    for (int k = 0; k < 64; k++) out[k] = in[g_zz[k]];
}

int icspo_median3(int a, int b, int c)   /* ENC:3677-3679 and every other median site */
{
    if ((a > b) && (a > c)) return (b > c) ? b : c;
    else if ((b > a) && (b > c)) return (a > c) ? a : c;
    else return (a > b) ? a : b;
}

/* ---------------------------------------------------------------- padding, SAD, ME */

/* ENC:2227-2269.  dst is (w+2pad) x (h+2pad), zero-initialised by the caller in the reference
 * (calloc, ENC:2085); the last padded row and column are never written and stay 0. */
void icsp_oracle_pad(const uint8_t* src, uint8_t* dst, int pad, int w, int h)
{
    int pw = w + 2 * pad, ph = h + 2 * pad;
    memset(dst, 0, (size_t)pw * ph);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[(y + pad) * pw + (x + pad)] = src[y * w + x];
    for (int y = 0; y < pad; y++)
        for (int x = 0; x < w; x++) {
            dst[y * pw + (pad + x)] = src[x];
            dst[(y + pad + h - 1) * pw + (x + pad)] = src[(h - 1) * w + x];
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < pad; x++) {
            dst[(y + pad) * pw + x] = src[y * w];
            dst[(y + pad) * pw + x + (w + pad - 1)] = src[y * w + (w - 1)];
        }
    for (int y = 0; y < pad; y++)
        for (int x = 0; x < pad; x++) {
            dst[y * pw + x] = src[0];
            dst[y * pw + x + (w + pad - 1)] = src[w - 1];
            dst[(y + pad + h - 1) * pw + x] = src[(h - 1) * w];
            dst[(y + pad + h - 1) * pw + x + (pad + w - 1)] = src[h * w - 1];
        }
}

int icsp_oracle_sad16(const uint8_t* cur, int cs, const uint8_t* ref, int rs)
{
    int sad = 0;
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) sad += abs((int)cur[y * cs + x] - (int)ref[y * rs + x]);
    return sad;
}

/* Direction state of motionEstimation (ENC:2095): (flag, xflag, yflag) starts (0,+1,-1) and is NOT
 * reset per macroblock.  Each iteration toggles flag and flips xflag or yflag, so the state after k
 * iterations depends only on k mod 4; state index s = iterations so far mod 4. */
static void me_state(int s, int* flag, int* xflag, int* yflag)
{
    int f = 0, xf = 1, yf = -1;
    for (int i = 0; i < (s & 3); i++) { if (!f) { f = 1; xf *= -1; } else { f = 0; yf *= -1; } }
    *flag = f; *xflag = xf; *yflag = yf;
}

void icsp_oracle_me_walk(int state, int dx[64], int dy[64])
{
    int flag, xflag, yflag, x0 = 0, y0 = 0, xcnt = 0, ycnt = 0;
    me_state(state, &flag, &xflag, &yflag);
    for (int cnt = 0; cnt < 64; cnt++) {
        if (!flag) { if (xflag <= 0) x0 += xcnt; else x0 -= xcnt; flag = 1; xcnt++; xflag *= -1; }
        else       { if (yflag < 0)  y0 += ycnt; else y0 -= ycnt; flag = 0; ycnt++; yflag *= -1; }
        dx[cnt] = x0; dy[cnt] = y0;
    }
}

void icsp_oracle_me_frame(const uint8_t* curY, const uint8_t* prevY, int w, int h, int* mvx, int* mvy, int* nsad)
{
    const int pad = 16, pw = w + 32, ph = h + 32;
    uint8_t* p = (uint8_t*)malloc((size_t)pw * ph);
    icsp_oracle_pad(prevY, p, pad, w, h);
    int sw = w / 16, total = sw * (h / 16);
    int flag = 0, xflag = 1, yflag = -1;                 /* ENC:2095, frame scope */
    int tempX = 0, tempY = 0;
    for (int n = 0; n < total; n++) {
        int min = INT_MAX, cnt = 0, evals = 0;
        int cntX0, cntY0, x0, y0, xcnt = 0, ycnt = 0;
        cntX0 = x0 = (n % sw) * 16;
        cntY0 = y0 = (n / sw) * 16;
        while (cnt < 64) {
            if (!flag) { if (xflag <= 0) x0 += xcnt; else x0 -= xcnt; flag = 1; xcnt++; xflag *= -1; }
            else       { if (yflag < 0)  y0 += ycnt; else y0 -= ycnt; flag = 0; ycnt++; yflag *= -1; }
            int sad = icsp_oracle_sad16(curY + cntY0 * w + cntX0, w, p + (pad + y0) * pw + (pad + x0), pw);
            evals++;
            if (min > sad) { min = sad; tempX = x0; tempY = y0; }
            else if (sad == 0) { tempX = x0; tempY = y0; break; }    /* ENC:2136-2141 */
            cnt++;
        }
        mvx[n] = cntX0 - tempX;                             /* ENC:2145-2146 */
        mvy[n] = cntY0 - tempY;
        if (nsad) nsad[n] = evals;
    }
    free(p);
}

/* ---------------------------------------------------------------- shared transform chain */

/* DCT -> DC-DPCM -> quant -> ACflag -> zig-zag -> dequant -> inverse DC-DPCM -> IDCT for one 8x8
 * block (ENC:507-513, 2043-2049, 1882-1888, 2651-2657).  Returns the reconstructed DC. */
static int transform_chain(const int err[64], int dcpred, int qdc, int qac, int chroma,
                           int16_t* lv, uint8_t* acflag, double idct[64], double* dbg_coef)
{
    double coef[64];
    int q[64], zz[64], iq[64];
    icsp_oracle_dct8x8(err, coef);
    if (dbg_coef) memcpy(dbg_coef, coef, sizeof(coef));
    coef[0] = coef[0] - dcpred;                                    /* ENC:3659 etc. */
    for (int i = 0; i < 64; i++) {
        int qs = (i == 0) ? qdc : qac;
        q[i] = chroma ? icsp_oracle_quant_chroma(coef[i], qs) : icsp_oracle_quant_luma(coef[i], qs);
    }
    int ac = 1;
    for (int i = 1; i < 64; i++) if (q[i] != 0) { ac = 0; break; }
    *acflag = (uint8_t)ac;
    icsp_oracle_zigzag(q, zz);
    for (int i = 0; i < 64; i++) lv[i] = (int16_t)zz[i];
    for (int i = 0; i < 64; i++) iq[i] = q[i] * ((i == 0) ? qdc : qac);   /* ENC:2818 */
    iq[0] = iq[0] + dcpred;                                        /* ENC:3991-4337 */
    icsp_oracle_idct8x8(iq, idct);
    return iq[0];
}

/* DC predictor on the 8x8 luma grid (ENC:3652-3818; same map for the INTER buffers 3822-3988).
 * Restated from the four macroblock-position cases: frame's first block 1024; first block row L;
 * first block column U; blk3 of any MB and blk1 of the last MB column median(L,UL,U); everything
 * else median(L,U,UR).  rec is the reconstructed-DC grid [rows8][cols8]. */
int icspo_luma_dcpred(const int* rec, int r8, int c8, int cols8)
{
    if (r8 == 0 && c8 == 0) return 1024;
    if (r8 == 0) return rec[c8 - 1];
    if (c8 == 0) return rec[(r8 - 1) * cols8];
    int L = rec[r8 * cols8 + c8 - 1], U = rec[(r8 - 1) * cols8 + c8];
    if (((r8 & 1) && (c8 & 1)) || c8 == cols8 - 1) return icspo_median3(L, rec[(r8 - 1) * cols8 + c8 - 1], U);
    return icspo_median3(L, U, rec[(r8 - 1) * cols8 + c8 + 1]);
}

/* DC predictor on the chroma block grid (ENC:4482-4513). */
int icspo_chroma_dcpred(const int* rec, int n, int sw)
{
    if (n == 0) return 1024;
    if (n / sw == 0) return rec[n - 1];
    if (n % sw == 0) return rec[n - sw];
    if (n % sw == sw - 1) return icspo_median3(rec[n - 1], rec[n - sw - 1], rec[n - sw]);
    return icspo_median3(rec[n - 1], rec[n - sw], rec[n - sw + 1]);
}

static uint8_t clip255(int t) { t = (t > 255) ? 255 : t; t = (t < 0) ? 0 : t; return (uint8_t)t; }

/* ---------------------------------------------------------------- intra frame */

void icsp_oracle_intra_frame(const uint8_t* frame, int w, int h, int qdc, int qac,
                             int16_t* levels, uint8_t* acflag, uint8_t* mpm_mode, uint8_t* recon,
                             double* dbg_coef, uint8_t* dbg_mode)
{
    ensure_init();
    const int sw = w / 16, sh = h / 16, nmb = sw * sh;
    const int cols8 = 2 * sw, rows8 = 2 * sh, cw = w / 2, ch = h / 2;
    const uint8_t *Y = frame, *Cb = frame + w * h, *Cr = Cb + cw * ch;
    uint8_t *rY = recon, *rCb = recon + w * h, *rCr = rCb + cw * ch;
    int* mode = (int*)calloc((size_t)rows8 * cols8, sizeof(int));
    int* recdc = (int*)calloc((size_t)rows8 * cols8, sizeof(int));
    int* crec[2] = { (int*)calloc(nmb, sizeof(int)), (int*)calloc(nmb, sizeof(int)) };

    for (int n = 0; n < nmb; n++) {
        int R = n / sw, C = n % sw;
        for (int k = 0; k < 4; k++) {
            int r8 = 2 * R + (k >> 1), c8 = 2 * C + (k & 1);
            int upav = r8 > 0, leav = c8 > 0;
            const uint8_t* cur = Y + (r8 * 8) * w + c8 * 8;
            const uint8_t* up = rY + (r8 * 8 - 1) * w + c8 * 8;      /* bottom row of the block above (ENC:665) */
            const uint8_t* le = rY + (r8 * 8) * w + c8 * 8 - 1;      /* right column of the left block (ENC:696) */
            int e0[64], e1[64], e2[64], sae0 = 0, sae1 = 0, sae2 = 0;
            /* mode 0 (ENC:644-672), mode 1 (ENC:673-703): missing neighbour => 128 */
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    int c = cur[y * w + x];
                    e0[y * 8 + x] = c - (upav ? (int)up[x] : 128);
                    e1[y * 8 + x] = c - (leav ? (int)le[y * w] : 128);
                    sae0 += abs(e0[y * 8 + x]);
                    sae1 += abs(e1[y * 8 + x]);
                }
            /* mode 2 (ENC:704-743): mean of 8 left + 8 up in double, residual truncated toward zero */
            double pl = 0, pu = 0;
            if (!leav) pl = 128 * 8; else for (int i = 0; i < 8; i++) pl += le[i * w];
            if (!upav) pu = 128 * 8; else for (int i = 0; i < 8; i++) pu += up[i];
            double predVal = (pl + pu) / (double)(8 + 8);
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    e2[y * 8 + x] = (int)((int)cur[y * w + x] - predVal);
                    sae2 += abs(e2[y * 8 + x]);
                }
            /* decision (ENC:886-998 ... 1305-1482) */
            int m;
            if (!upav && !leav) m = 2;
            else if (!upav)     m = (sae2 > sae1) ? 1 : 2;
            else if (!leav)     m = (sae2 > sae0) ? 0 : 2;
            else { int mn = sae0 < sae1 ? sae0 : sae1; mn = mn < sae2 ? mn : sae2;
                   m = (mn == sae0) ? 0 : (mn == sae1) ? 1 : 2; }
            mode[r8 * cols8 + c8] = m;
            if (dbg_mode) dbg_mode[n * 4 + k] = (uint8_t)m;
            /* most-probable-mode signalling (ENC:910-921, 978-997) */
            int mpm = 0, ipm = 0;
            if (upav || leav) {
                int p;
                if (!upav)      p = mode[r8 * cols8 + c8 - 1];
                else if (!leav) p = mode[(r8 - 1) * cols8 + c8];
                else p = icspo_median3(mode[r8 * cols8 + c8 - 1], mode[(r8 - 1) * cols8 + c8 - 1], mode[(r8 - 1) * cols8 + c8]);
                mpm = (m == p);
                if (!mpm) {
                    if (p == 0)      ipm = (m == 1) ? 0 : 1;
                    else if (p == 2) ipm = (m == 0) ? 0 : 1;
                    else             ipm = (m == 0) ? 0 : 1;
                }
            }
            mpm_mode[n * 4 + k] = (uint8_t)(mpm | (ipm << 1));
            const int* err = (m == 0) ? e0 : (m == 1) ? e1 : e2;
            double idct[64];
            int pred = icspo_luma_dcpred(recdc, r8, c8, cols8);
            recdc[r8 * cols8 + c8] = transform_chain(err, pred, qdc, qac, 0, levels + (n * 6 + k) * 64,
                                                     acflag + n * 6 + k, idct,
                                                     dbg_coef ? dbg_coef + (n * 6 + k) * 64 : 0);
            /* reconstruction: the SUM idct+pred is truncated (ENC:754,767,787,800,843) */
            uint8_t* out = rY + (r8 * 8) * w + c8 * 8;
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    int t;
                    if (m == 0)      t = (int)(idct[y * 8 + x] + (upav ? (int)up[x] : 128));
                    else if (m == 1) t = (int)(idct[y * 8 + x] + (leav ? (int)le[y * w] : 128));
                    else             t = (int)(idct[y * 8 + x] + predVal);
                    out[y * w + x] = clip255(t);
                }
        }
        /* chroma: DCT of the raw pixels, no pixel prediction (ENC:4347-4349, 1876-1903);
         * recon = clip(trunc(idct)) (ENC:1964-1971) */
        for (int pl = 0; pl < 2; pl++) {
            const uint8_t* src = (pl ? Cr : Cb) + (R * 8) * cw + C * 8;
            uint8_t* dst = (pl ? rCr : rCb) + (R * 8) * cw + C * 8;
            int err[64];
            for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) err[y * 8 + x] = src[y * cw + x];
            double idct[64];
            int pred = icspo_chroma_dcpred(crec[pl], n, sw);
            crec[pl][n] = transform_chain(err, pred, qdc, qac, 1, levels + (n * 6 + 4 + pl) * 64,
                                          acflag + n * 6 + 4 + pl, idct,
                                          dbg_coef ? dbg_coef + (n * 6 + 4 + pl) * 64 : 0);
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    int t = (int)((idct[y * 8 + x] > 255) ? 255 : idct[y * 8 + x]);
                    dst[y * cw + x] = (uint8_t)((t < 0) ? 0 : t);
                }
        }
    }
    free(mode); free(recdc); free(crec[0]); free(crec[1]);
}

/* ---------------------------------------------------------------- inter frame */

/* MV predictor (ENC:2353-2425): 8 for MB 0, left on row 0, up on column 0, else a median of
 * (L,UL,U) on the last column or (L,U,UR) elsewhere.  The y component carries the reference's
 * `(y1>x3)` typo (ENC:2399, 2418) and is therefore not always a median. */
void icspo_mv_pred(const int* mx, const int* my, int n, int sw, int* px, int* py)
{
    if (n == 0) { *px = 8; *py = 8; return; }
    if (n / sw == 0) { *px = mx[n - 1]; *py = my[n - 1]; return; }
    if (n % sw == 0) { *px = mx[n - sw]; *py = my[n - sw]; return; }
    int i1 = n - 1, i2, i3;
    if (n % sw == sw - 1) { i2 = n - sw - 1; i3 = n - sw; } else { i2 = n - sw; i3 = n - sw + 1; }
    int x1 = mx[i1], x2 = mx[i2], x3 = mx[i3], y1 = my[i1], y2 = my[i2], y3 = my[i3];
    *px = icspo_median3(x1, x2, x3);
    if ((y1 > y2) && (y1 > y3))      *py = (y2 > y3) ? y2 : y3;
    else if ((y2 > y1) && (y2 > y3)) *py = (y1 > x3) ? y1 : y3;
    else                              *py = (y1 > y2) ? y1 : y2;
}

void icsp_oracle_inter_frame(const uint8_t* frame, const uint8_t* prev, int w, int h, int qdc, int qac,
                             int16_t* levels, uint8_t* acflag, int8_t* mvd, uint8_t* recon,
                             double* dbg_coef, int8_t* dbg_mv)
{
    ensure_init();
    const int sw = w / 16, sh = h / 16, nmb = sw * sh;
    const int cols8 = 2 * sw, rows8 = 2 * sh, cw = w / 2, ch = h / 2;
    const uint8_t *Y = frame, *Cb = frame + w * h, *Cr = Cb + cw * ch;
    const uint8_t *pY = prev, *pCb = prev + w * h, *pCr = pCb + cw * ch;
    uint8_t *rY = recon, *rCb = recon + w * h, *rCr = rCb + cw * ch;
    int* mx = (int*)malloc(sizeof(int) * nmb);
    int* my = (int*)malloc(sizeof(int) * nmb);
    icsp_oracle_me_frame(Y, pY, w, h, mx, my, 0);                       /* ENC:1998 */

    const int pw = w + 32, ph = h + 32;
    uint8_t* pad = (uint8_t*)malloc((size_t)pw * ph);
    icsp_oracle_pad(pY, pad, 16, w, h);                                 /* ENC:2176, 2316 */
    int* recdc = (int*)calloc((size_t)rows8 * cols8, sizeof(int));

    for (int n = 0; n < nmb; n++) {
        int R = n / sw, C = n % sw;
        /* motionCompensation uses the raw mv (ENC:2185-2186); mvPrediction then replaces bd.mv by the
         * difference (ENC:2353) and ImvPrediction restores Reconstructedmv == raw mv (ENC:2426). */
        int px, py;
        icspo_mv_pred(mx, my, n, sw, &px, &py);
        mvd[n * 2 + 0] = (int8_t)(mx[n] - px);
        mvd[n * 2 + 1] = (int8_t)(my[n] - py);
        if (dbg_mv) { dbg_mv[n * 2] = (int8_t)mx[n]; dbg_mv[n * 2 + 1] = (int8_t)my[n]; }
        int refx = C * 16 - mx[n] + 16, refy = R * 16 - my[n] + 16;
        for (int k = 0; k < 4; k++) {
            int r8 = 2 * R + (k >> 1), c8 = 2 * C + (k & 1);
            int oy = (k >> 1) * 8, ox = (k & 1) * 8;
            int err[64];
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++)
                    err[y * 8 + x] = (int)Y[(R * 16 + oy + y) * w + C * 16 + ox + x]
                                   - (int)pad[(refy + oy + y) * pw + refx + ox + x];    /* ENC:2194 */
            double idct[64];
            int pred = icspo_luma_dcpred(recdc, r8, c8, cols8);
            recdc[r8 * cols8 + c8] = transform_chain(err, pred, qdc, qac, 0, levels + (n * 6 + k) * 64,
                                                     acflag + n * 6 + k, idct,
                                                     dbg_coef ? dbg_coef + (n * 6 + k) * 64 : 0);
            /* mergeBlock(INTER) truncates each IDCT value (ENC:4812-4836); interYReconstruct adds the
             * integer residual to the prediction and clips (ENC:2343-2346) */
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    int res = (int)idct[y * 8 + x];
                    int t = pad[(refy + oy + y) * pw + refx + ox + x] + res;
                    rY[(R * 16 + oy + y) * w + C * 16 + ox + x] = clip255(t);
                }
        }
    }
    free(pad);

    /* chroma (ENC:2625-2682): mv/2 with C division (ENC:2538-2539), pad 8, residual chain with the
     * chroma quantiser, recon = clip(trunc(pred + idct)) with the SUM truncated (ENC:2605-2612) */
    const int cpw = cw + 16, cph = ch + 16;
    uint8_t* cpad = (uint8_t*)malloc((size_t)cpw * cph);
    int* crec = (int*)calloc(nmb, sizeof(int));
    for (int pl = 0; pl < 2; pl++) {
        const uint8_t* src = pl ? Cr : Cb;
        uint8_t* dst = pl ? rCr : rCb;
        icsp_oracle_pad(pl ? pCr : pCb, cpad, 8, cw, ch);
        memset(crec, 0, sizeof(int) * nmb);
        for (int n = 0; n < nmb; n++) {
            int R = n / sw, C = n % sw;
            int refx = C * 8 - (mx[n] / 2) + 8, refy = R * 8 - (my[n] / 2) + 8;
            int err[64];
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++)
                    err[y * 8 + x] = (int)src[(R * 8 + y) * cw + C * 8 + x] - (int)cpad[(refy + y) * cpw + refx + x];
            double idct[64];
            int pred = icspo_chroma_dcpred(crec, n, sw);
            crec[n] = transform_chain(err, pred, qdc, qac, 1, levels + (n * 6 + 4 + pl) * 64,
                                      acflag + n * 6 + 4 + pl, idct,
                                      dbg_coef ? dbg_coef + (n * 6 + 4 + pl) * 64 : 0);
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    int t = (int)(cpad[(refy + y) * cpw + refx + x] + idct[y * 8 + x]);
                    dst[(R * 8 + y) * cw + C * 8 + x] = clip255(t);
                }
        }
    }
    free(cpad); free(crec); free(recdc); free(mx); free(my);
}

/* ---------------------------------------------------------------- sequence + GOP job queue */

typedef struct {
    const uint8_t* yuv; int nframes, w, h, qdc, qac, period;
    int16_t* levels; uint8_t* acflag; uint8_t* mpm; int8_t* mvd; uint8_t* recon;
    int next_job, njobs; pthread_mutex_t mu;
} seq_t;

static void encode_gop(seq_t* s, int start, int end)   /* frames [start, end], first is I (ENC:202-208) */
{
    const int nmb = (s->w / 16) * (s->h / 16);
    const size_t fsz = (size_t)s->w * s->h * 3 / 2;
    for (int f = start; f <= end; f++) {
        int16_t* lv = s->levels + (size_t)f * nmb * 384;
        uint8_t* ac = s->acflag + (size_t)f * nmb * 6;
        uint8_t* mp = s->mpm + (size_t)f * nmb * 4;
        int8_t* mv = s->mvd + (size_t)f * nmb * 2;
        if (f == start) {
            memset(mv, 0, (size_t)nmb * 2);
            icsp_oracle_intra_frame(s->yuv + f * fsz, s->w, s->h, s->qdc, s->qac, lv, ac, mp, s->recon + f * fsz, 0, 0);
        } else {
            memset(mp, 0, (size_t)nmb * 4);
            icsp_oracle_inter_frame(s->yuv + f * fsz, s->recon + (f - 1) * fsz, s->w, s->h, s->qdc, s->qac,
                                    lv, ac, mv, s->recon + f * fsz, 0, 0);
        }
    }
}

static void* seq_worker(void* arg)
{
    seq_t* s = (seq_t*)arg;
    for (;;) {
        pthread_mutex_lock(&s->mu);
        int j = s->next_job++;
        pthread_mutex_unlock(&s->mu);
        if (j >= s->njobs) break;
        int period = s->period > 0 ? s->period : 1;
        int start = j * period, end = start + period - 1;
        if (end >= s->nframes) end = s->nframes - 1;
        encode_gop(s, start, end);
    }
    return 0;
}

int icsp_oracle_encode_sequence(const uint8_t* yuv, int nframes, int w, int h, int qdc, int qac,
                                int intra_period, int nthreads,
                                int16_t* levels, uint8_t* acflag, uint8_t* mpm_mode, int8_t* mvd, uint8_t* recon)
{
    seq_t s;
    s.yuv = yuv; s.nframes = nframes; s.w = w; s.h = h; s.qdc = qdc; s.qac = qac; s.period = intra_period;
    s.levels = levels; s.acflag = acflag; s.mpm = mpm_mode; s.mvd = mvd; s.recon = recon;
    int period = intra_period > 0 ? intra_period : 1;
    s.njobs = (nframes + period - 1) / period;
    s.next_job = 0;
    pthread_mutex_init(&s.mu, 0);
    if (nthreads <= 1) seq_worker(&s);
    else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], 0, seq_worker, &s);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], 0);
        free(th);
    }
    pthread_mutex_destroy(&s.mu);
    return 0;
}
