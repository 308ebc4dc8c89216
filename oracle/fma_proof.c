/*
 * fma_proof.c — is a fused multiply-add bit-safe in the FIRST 1-D pass of the reference's 8x8 transforms?
 *
 * TEST INFRASTRUCTURE ONLY (built by oracle/Makefile into oracle/_ref/fma_proof where /root/reference exists; run by
 * tests/test_fma_proof.py).  Links oracle/_ref/libicsp_ref.so, i.e. the reference's own DCT_block / IDCT_block /
 * CDCT_block / CIDCT_block object code (ENC:2685-2749, 2825-2893, 4338-4419, 4687-4768), and compares them bit for bit
 * on random blocks with the arithmetic the HIP kernels use (icspcodec_amd/csrc/icsp_blk8.hip.inc):
 *
 *   forward  pass 1  tmp[v][u] = sum_x err[v][x]*cos[u][x]            FUSED:  s = fma(err, cos, s)
 *            pass 2  out[v][u] = sum_y tmp[y][u]*cos[v][y]            un-fused, as the reference
 *   inverse  pass 1  tmp[y][x] = sum_u (Cu*iq[y][u])*cos[u][x]        u = 0 as the reference, u >= 1 FUSED
 *            pass 2  out[y][x] = sum_v (Cv*tmp[v][x])*cos[v][y]       un-fused
 *
 * Why it must hold: the encoder's cosine table is 24-bit float literals, all multiples of 2^-26 (ENC.h:190-198).  Forward
 * pass 1: |err| <= 255, so every product and every partial sum (|s| < 2^11) is a multiple of 2^-26 with at most 37
 * significant bits -- nothing is ever rounded, fused or not.  Inverse pass 1, u >= 1: Cu = 1 and iq is an integer below
 * 2^20, so iq*cos is exact (<= 44 bits) and rn(s + rn(iq*cos)) == rn(s + iq*cos) == fma(iq, cos, s).  The second passes
 * multiply full 53-bit doubles, their products round: fusing THOSE changes results (mode 2 below demonstrates it), and so
 * does fusing anything with the DECODER's table, which is written as double literals (DEC.h:19-27; mode 3).
 *
 * usage: fma_proof <blocks> <seed>     prints one line per check, exit 0 iff every check that must hold holds.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void ref_costable(double out[64]);
double ref_irt2(void);
void ref_dct_block(const int in[64], double out[64]);
void ref_cdct_block(const int in[64], double out[64]);
void ref_idct_block(const int in[64], double out[64]);
void ref_cidct_block(const int in[64], double out[64]);
void ref_quant_luma(const double coef[64], int qdc, int qac, int q[64], int zz[64], int iq[64], int* acflag);
void ref_quant_chroma(const double coef[64], int qdc, int qac, int q[64], int zz[64], int iq[64], int* acflag);

static double g_cos[8][8], g_irt2;

static uint64_t rng_state;
static uint32_t rnd(void)
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}

/* fuse1: fuse pass 1 (the claim); fuse2: also fuse pass 2 (must break); fuse1 == 2: pass 1 FOLDED as the kernels do it
 * since round 2 -- cos[u][7-x] == (-1)^u cos[u][x] holds literally in the table, so
 *   tmp[v][u] = sum_{x<4} (err[v][x] +- err[v][7-x]) * cos[u][x]   (+ for even u, - for odd u),
 * four terms instead of eight.  Nothing in pass 1 ever rounds (above; |err +- err'| <= 510 is one more bit, 38 in all), so
 * both orders give the exact real value: the same double. */
static void dct_variant(const int in[64], double out[64], int fuse1, int fuse2)
{
    double tmp[8][8];
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++) {
            double s;
            if (fuse1 == 2) {
                const int sg = (u & 1) ? -1 : 1;
                s = (double)(in[v * 8] + sg * in[v * 8 + 7]) * g_cos[u][0];
                for (int x = 1; x < 4; x++) s = fma((double)(in[v * 8 + x] + sg * in[v * 8 + 7 - x]), g_cos[u][x], s);
            } else {
                s = (double)in[v * 8] * g_cos[u][0];
                for (int x = 1; x < 8; x++)
                    s = fuse1 ? fma((double)in[v * 8 + x], g_cos[u][x], s) : s + (double)in[v * 8 + x] * g_cos[u][x];
            }
            tmp[v][u] = s;
        }
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) {
            double s = 0;
            for (int y = 0; y < 8; y++) s = fuse2 ? fma(tmp[y][u], g_cos[v][y], s) : s + tmp[y][u] * g_cos[v][y];
            out[v * 8 + u] = s;
        }
    for (int i = 0; i < 8; i++) { out[i] *= g_irt2; out[i * 8] *= g_irt2; }
    for (int i = 0; i < 64; i++) out[i] *= 0.25;
}

static void idct_variant(const int in[64], double out[64], int fuse1, int fuse2, const double (*cs)[8])
{
    double tmp[8][8];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            double s = 0.0;
            s = s + (g_irt2 * (double)in[y * 8]) * cs[0][x];
            for (int u = 1; u < 8; u++)
                s = fuse1 ? fma((double)in[y * 8 + u], cs[u][x], s) : s + (1.0 * (double)in[y * 8 + u]) * cs[u][x];
            tmp[y][x] = s;
        }
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++) {
            double s = 0.0;
            for (int v = 0; v < 8; v++) {
                const double c = (v == 0) ? g_irt2 : 1.0;
                s = fuse2 ? fma(c * tmp[v][x], cs[v][y], s) : s + (c * tmp[v][x]) * cs[v][y];
            }
            out[y * 8 + x] = s * 0.25;
        }
}

static long diff_bits(const double* a, const double* b)
{
    long d = 0;
    for (int i = 0; i < 64; i++) d += memcmp(&a[i], &b[i], 8) != 0;
    return d;
}

int main(int argc, char** argv)
{
    const long nblk = argc > 1 ? atol(argv[1]) : 1000000;
    rng_state = argc > 2 ? strtoull(argv[2], 0, 10) : 1;
    double t[64];
    ref_costable(t);
    memcpy(g_cos, t, sizeof(t));
    g_irt2 = ref_irt2();
    /* the decoder's table: the same angles as 6-digit DOUBLE literals (DEC.h:19-27) */
    static const double dmag[8] = { 1.0, 0.980785, 0.92388, 0.83147, 0.707107, 0.55557, 0.382683, 0.19509 };
    double dcos[8][8];
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++) {
            int m = ((2 * x + 1) * u) % 32;
            if (m > 16) m = 32 - m;
            dcos[u][x] = (m > 8) ? -dmag[16 - m] : dmag[m];
        }
    long bad_f = 0, bad_fc = 0, bad_i = 0, bad_ic = 0, brk_f2 = 0, brk_i2 = 0, brk_dec = 0, bad_fold = 0, bad_foldc = 0;
    for (int u = 0; u < 8; u++)                       /* the symmetry the folded form rests on, literally */
        for (int x = 0; x < 8; x++)
            if (g_cos[u][7 - x] != ((u & 1) ? -g_cos[u][x] : g_cos[u][x])) { printf("table symmetry broken at [%d][%d]\n", u, x); return 1; }
    int in[64];
    double want[64], got[64], plain[64];
    for (long n = 0; n < nblk; n++) {
        /* forward: residuals in [-255, 255] (luma, P chroma) and raw pixels 0..255 (I chroma); sparse and flat cases mixed in */
        const uint32_t kind = rnd() % 8;
        for (int i = 0; i < 64; i++) {
            int v = (int)(rnd() % 511) - 255;
            if (kind == 1) v = (int)(rnd() % 256);
            if (kind == 2) v = (rnd() % 4) ? 0 : v;
            if (kind == 3) v = (rnd() & 1) ? 255 : -255;
            if (kind == 4) v = (int)(rnd() % 5) - 2;
            in[i] = v;
        }
        ref_dct_block(in, want);  dct_variant(in, got, 1, 0); bad_f += diff_bits(want, got) != 0;
        dct_variant(in, got, 2, 0); bad_fold += diff_bits(want, got) != 0;
        ref_cdct_block(in, want); dct_variant(in, got, 1, 0); bad_fc += diff_bits(want, got) != 0;
        dct_variant(in, got, 2, 0); bad_foldc += diff_bits(want, got) != 0;
        dct_variant(in, got, 1, 1); brk_f2 += diff_bits(want, got) != 0;
        /* inverse: dequantised coefficients level*q (+ DC predictor): |DC| up to 2*4080 + 1024, AC well inside +-2^15;
         * quantiser steps 1..255 make them multiples of q; sparse blocks as real streams have them */
        const int q = 1 + (int)(rnd() % 255), dense = (int)(rnd() % 4);
        for (int i = 0; i < 64; i++) {
            int lim = (i == 0) ? 9184 : 32767;
            int lv = ((int)(rnd() % (2 * (lim / q) + 1)) - lim / q) * q;
            if (dense == 0 && (rnd() % 8)) lv = 0;
            if (dense == 1 && i > 0 && (rnd() % 3)) lv = 0;
            in[i] = lv;
        }
        if (dense == 3) in[0] += (int)(rnd() % 2049) - 1024;
        ref_idct_block(in, want);  idct_variant(in, got, 1, 0, (const double (*)[8])g_cos); bad_i += diff_bits(want, got) != 0;
        ref_cidct_block(in, want); idct_variant(in, got, 1, 0, (const double (*)[8])g_cos); bad_ic += diff_bits(want, got) != 0;
        idct_variant(in, got, 1, 1, (const double (*)[8])g_cos); brk_i2 += diff_bits(want, got) != 0;
        idct_variant(in, plain, 0, 0, (const double (*)[8])dcos); idct_variant(in, got, 1, 0, (const double (*)[8])dcos);
        brk_dec += diff_bits(plain, got) != 0;
    }
    /* The quantiser for POWER-OF-TWO steps (round 5; icsp_blk8.hip.inc, quant_pow2): the reference computes
     *   luma    (int)(c + 0.5) / q           (ENC:2780: cast truncates, integer division truncates)
     *   chroma  (int)floor(c + 0.5) / q      (ENC:4642)
     * With x = rn(c + 0.5): trunc(trunc(x) / q) == trunc(x / q) for every integer q >= 1 and either sign; for q = 2^s the
     * scaling x * 2^-s is exact: trunc((c + 0.5) * 2^-s), add, multiply, truncating conversion (what the kernels do); it also
     * commutes with the rounding of the sum, rn(c + 0.5) * 2^-s == fma(c, 2^-s, 2^-(s+1)).  Chroma: trunc(floor(x) * 2^-s).
     * Checked against the reference's own Quantization_block / CQuantization_block on coefficients that sit on, just below
     * and just above every rounding and division boundary, and on random ones. */
    long bad_ql = 0, bad_qc = 0, nq = 0;
    {
        double coef[64];
        int lv[64], zz[64], iq[64], acf;
        for (long n = 0; n < nblk / 16 + 64; n++) {
            const int sdc = (int)(rnd() % 8), sac = (int)(rnd() % 8);
            const int qdc = 1 << sdc, qac = 1 << sac;
            for (int i = 0; i < 64; i++) {
                const int q = i ? qac : qdc;
                const uint32_t kind = rnd() % 6;
                double c;
                if (kind == 0) c = ((double)(rnd() % 4000001) - 2000000.0) / 1000.0 * 4.0;                    /* anywhere in +-8000 */
                else {
                    /* near k*q - 0.5 (the division boundary after rounding) or k - 0.5 (the rounding boundary) */
                    const long k = (long)(rnd() % 2001) - 1000;
                    const double b = (kind & 1) ? (double)(k * q) - 0.5 : (double)k - 0.5;
                    const int e = (int)(rnd() % 5) - 2;
                    c = b;
                    for (int t = 0; t < (e < 0 ? -e : e); t++) c = nextafter(c, e < 0 ? -1e9 : 1e9);
                    if (kind == 5) c = b + ((double)(rnd() % 2001) - 1000.0) * 1e-9;
                }
                coef[i] = c;
            }
            ref_quant_luma(coef, qdc, qac, lv, zz, iq, &acf);
            for (int i = 0; i < 64; i++) {
                const double inv = 1.0 / (double)(i ? qac : qdc);
                const int got = (int)fma(coef[i], inv, 0.5 * inv), got2 = (int)((coef[i] + 0.5) * inv);     /* the kernels use the second form */
                bad_ql += got != lv[i] || got2 != lv[i];
            }
            ref_quant_chroma(coef, qdc, qac, lv, zz, iq, &acf);
            for (int i = 0; i < 64; i++) {
                const double inv = 1.0 / (double)(i ? qac : qdc);
                const int got = (int)(floor(coef[i] + 0.5) * inv);
                bad_qc += got != lv[i];
            }
            nq++;
        }
    }
    printf("blocks %ld\n", nblk);
    printf("quantiser blocks %ld (power-of-two steps 1..128)\n", nq);
    printf("luma   quantiser (c+.5)*inv, fma vs Quantization_block: %ld levels differ (must be 0)\n", bad_ql);
    printf("chroma quantiser floor*inv vs CQuantization_block : %ld levels differ (must be 0)\n", bad_qc);
    printf("forward  pass-1 fused vs DCT_block   : %ld blocks differ (must be 0)\n", bad_f);
    printf("forward  pass-1 fused vs CDCT_block  : %ld blocks differ (must be 0)\n", bad_fc);
    printf("forward  pass-1 folded vs DCT_block  : %ld blocks differ (must be 0)\n", bad_fold);
    printf("forward  pass-1 folded vs CDCT_block : %ld blocks differ (must be 0)\n", bad_foldc);
    printf("inverse  pass-1 fused vs IDCT_block  : %ld blocks differ (must be 0)\n", bad_i);
    printf("inverse  pass-1 fused vs CIDCT_block : %ld blocks differ (must be 0)\n", bad_ic);
    printf("forward  pass-2 fused too            : %ld blocks differ (expected > 0: not bit-safe)\n", brk_f2);
    printf("inverse  pass-2 fused too            : %ld blocks differ (expected > 0: not bit-safe)\n", brk_i2);
    printf("decoder table, inverse pass-1 fused  : %ld blocks differ (expected > 0: not bit-safe)\n", brk_dec);
    return (bad_f || bad_fc || bad_i || bad_ic || bad_fold || bad_foldc || bad_ql || bad_qc) ? 1 : 0;
}
