// icsp_bitstream.cpp — host back end of the C ABI: header + per-macroblock syntax + the 13-category code + MSB-first
// bit packing, byte-exact to the reference's makebitstream (ENC:4849-4900), headerinit (ENC:4901-4922), allintraBody
// (ENC:4923-5031), intraBody (ENC:5032-5131), interBody (ENC:5132-5236), DCentropy/ACentropy/MVentropy
// (ENC:5417-5602, 5791-5989, 5990-6334).  ENC = /root/reference/source/encoder/ICSP_Codec_Encoder_source.cpp.
//
// Entropy coding is bit-serial and stays on the host (BASELINE.json north_star).  Differences from the reference, on
// purpose: 64-bit bit counters and a caller-sized buffer (the reference's `int` counters overflow beyond 256 MiB and
// its buffer is sized at 8 bits per luma pixel, ENC:4873-4875); the buffer is zero-initialised, which is what the
// reference's malloc returns in practice (fresh mmap) and what its final partial byte relies on.
#include <stdint.h>
#include <string.h>
#include "icsp_hip.h"

namespace {

struct BitWriter {
    uint8_t* buf; size_t cap; size_t nbits; bool overflow;
    // (buf[cnt++/8] <<= 1) |= bit  (ENC:4956): bits enter each byte MSB first; a final partial byte keeps its bits
    // right-aligned, exactly as the reference leaves it.
    inline void put(uint32_t value, int n)
    {
        for (int i = n - 1; i >= 0; i--) {
            size_t byte = nbits >> 3;
            if (byte >= cap) { overflow = true; return; }
            buf[byte] = (uint8_t)((buf[byte] << 1) | ((value >> i) & 1u));
            nbits++;
        }
    }
};

// One value of the shared DC / AC / MV code table (ENC:5417-5602): returns (code, length).
inline void code_value(int v, uint32_t& code, int& len)
{
    const uint32_t s = (v >= 0) ? 1u : 0u;
    uint32_t a = (uint32_t)(v >= 0 ? v : -v);
    if (a == 0) { code = 0; len = 2; return; }                       // 00
    if (a == 1) { code = (0x2u << 1) | s; len = 4; return; }          // 010 s
    if (a <= 31) {                                                   // 011/100/101/110, s, e bits of a-2^e
        int e = (a <= 3) ? 1 : (a <= 7) ? 2 : (a <= 15) ? 3 : 4;
        uint32_t prefix = 2u + (uint32_t)e;                          // 3,4,5,6 = 011,100,101,110
        code = (((prefix << 1) | s) << e) | (a - (1u << e));
        len = 4 + e;
        return;
    }
    int e = 5;
    while (e < 11 && a >= (2u << e)) e++;                            // 2^e <= a < 2^(e+1), capped at e = 11 (a >= 2048)
    uint32_t ones = (1u << (e - 2)) - 1u;                            // (e-2) one bits, then 0, then s
    uint32_t c = (a - (1u << e)) & ((1u << e) - 1u);                 // the reference emits the low e bits only
    code = (((ones << 1) << 1 | s) << e) | c;
    len = (e - 2) + 2 + e;
}

inline void put_value(BitWriter& w, int v)
{
    uint32_t c; int n;
    code_value(v, c, n);
    w.put(c, n);
}

// DC code, ACflag, then either 63 literal zero bits or the 63 AC codes (ENC:4958-4977)
inline void put_block(BitWriter& w, const int16_t* lv, int acflag)
{
    put_value(w, lv[0]);
    w.put((uint32_t)acflag & 1u, 1);
    if (acflag == 1) { w.put(0, 31); w.put(0, 32); }
    else for (int i = 1; i < 64; i++) put_value(w, lv[i]);
}

} // namespace

extern "C" {

size_t icsp_bitstream_bound(const icsp_params_t* p, int n)
{
    if (!p || n < 0) return 0;
    const size_t nmb = (size_t)(p->width / 16) * (p->height / 16);
    // per macroblock at most 1 + 2*22 (mv) + 4*2 (mode bits) + 6*(22 + 1 + 63*22) bits
    const size_t bits_per_mb = 1 + 44 + 8 + 6 * (22 + 1 + 63 * 22);
    return 14 + ((size_t)n * nmb * bits_per_mb) / 8 + 2;
}

static void write_header(const icsp_params_t* p, uint8_t* out)
{
    // packed struct of 14 bytes (ICSP_Codec_Encoder.h:201-212), little-endian shorts
    out[0] = 0; out[1] = 73; out[2] = 67; out[3] = 83; out[4] = 80;       // "\0ICSP" (ENC:4903)
    const uint16_t hh = (uint16_t)p->height, ww = (uint16_t)p->width;
    out[5] = (uint8_t)(hh & 0xff); out[6] = (uint8_t)(hh >> 8);
    out[7] = (uint8_t)(ww & 0xff); out[8] = (uint8_t)(ww >> 8);
    out[9] = (uint8_t)p->qp_dc; out[10] = (uint8_t)p->qp_ac; out[11] = 0;
    const uint16_t outro = (uint16_t)((p->intra_period & 0x3f) << 7);       // 6 bits of intraPeriod, then 7 zero bits (ENC:4910-4921)
    out[12] = (uint8_t)(outro & 0xff); out[13] = (uint8_t)(outro >> 8);
}

// One piece (an MSB-first bit string of `bits` bits, e.g. what icsp_pack_bits returned for one shard) into the body of a
// .bin image at bit offset `at` (body = image + 14).  The image must have been zeroed (icsp_bitstream_begin).  Pieces cover
// disjoint bit ranges, so different pieces may be placed by different threads at once: interior bytes are plain stores, the
// first and last byte of a range (which a neighbour may share) are OR-ed in atomically.
int icsp_bitstream_place(uint8_t* image, size_t cap, uint64_t at, const uint8_t* piece, uint64_t bits)
{
    if (!image || (!piece && bits)) return ICSP_ERR_UNENOUGH_PARAM;
    if (bits == 0) return ICSP_OK;
    if (cap < 14 + (size_t)((at + bits + 7) / 8)) return ICSP_ERR_RANGE;
    uint8_t* dst = image + 14 + (size_t)(at >> 3);
    const unsigned sh = (unsigned)(at & 7);
    const size_t nsrc = (size_t)((bits + 7) / 8);                     // source bytes; padding bits of the last one are ignored
    const size_t ndst = (size_t)((sh + bits + 7) / 8);                // destination bytes touched
    auto src_at = [&](size_t j) -> unsigned {                         // byte j of the source with the padding bits cleared
        if (j >= nsrc) return 0;
        unsigned v = piece[j];
        if (j == nsrc - 1 && (bits & 7)) v &= 0xff00u >> (bits & 7);
        return v;
    };
    auto out_byte = [&](size_t j) -> uint8_t {                        // destination byte j of the shifted string
        const unsigned hi = j ? src_at(j - 1) : 0, lo = src_at(j);
        return (uint8_t)(((hi << 8 | lo) >> sh) & 0xff);
    };
    __atomic_fetch_or(&dst[0], out_byte(0), __ATOMIC_RELAXED);
    if (ndst > 1) {
        size_t j = 1;
        if (sh == 0) { if (ndst > 2) memcpy(dst + 1, piece + 1, ndst - 2); j = ndst - 1; }
        else for (; j + 1 < ndst && j + 1 < nsrc; j++) dst[j] = (uint8_t)((piece[j - 1] << (8 - sh)) | (piece[j] >> sh));
        for (; j + 1 < ndst; j++) dst[j] = out_byte(j);              // (the source's last byte: padding bits masked)
        __atomic_fetch_or(&dst[ndst - 1], out_byte(ndst - 1), __ATOMIC_RELAXED);
    }
    return ICSP_OK;
}

// Zeroes the image for a body of total_bits bits and writes the 14-byte header; *out_bytes = 14 + total_bits/8 + 1.
int icsp_bitstream_begin(const icsp_params_t* p, uint64_t total_bits, uint8_t* image, size_t cap, size_t* out_bytes)
{
    if (!p || !image || !out_bytes) return ICSP_ERR_UNENOUGH_PARAM;
    const size_t body = (size_t)(total_bits / 8) + 1;                  // fwrite(..., cntbits/8 + 1, ...) (ENC:4895, 5029)
    if (cap < 14 + body + 1) return ICSP_ERR_RANGE;
    memset(image, 0, 14 + body + 1);
    write_header(p, image);
    *out_bytes = 14 + body;
    return ICSP_OK;
}

int icsp_bitstream_header(const icsp_params_t* p, uint64_t total_bits, uint8_t* image, size_t cap, size_t* out_bytes)
{
    if (!p || !image || !out_bytes) return ICSP_ERR_UNENOUGH_PARAM;
    const size_t body = (size_t)(total_bits / 8) + 1;
    if (cap < 14 + body + 1) return ICSP_ERR_RANGE;
    write_header(p, image);
    *out_bytes = 14 + body;
    return ICSP_OK;
}

// After every piece is placed: the reference shifts bits into each byte from the right, so a final partial byte holds its
// bits right-aligned (ENC:4956).
int icsp_bitstream_end(uint8_t* image, uint64_t total_bits)
{
    if (!image) return ICSP_ERR_UNENOUGH_PARAM;
    const unsigned r = (unsigned)(total_bits & 7);
    uint8_t* dst = image + 14;
    if (r) dst[total_bits >> 3] = (uint8_t)(dst[total_bits >> 3] >> (8 - r));
    return ICSP_OK;
}

int icsp_bitstream_assemble(const icsp_params_t* p, int npieces, const uint8_t* const* pieces,
                            const uint64_t* piece_bits, uint8_t* out, size_t cap, size_t* out_bytes)
{
    if (!p || !out || !out_bytes || npieces < 0 || (npieces > 0 && (!pieces || !piece_bits))) return ICSP_ERR_UNENOUGH_PARAM;
    uint64_t total = 0;
    for (int i = 0; i < npieces; i++) total += piece_bits[i];
    if (int rc = icsp_bitstream_begin(p, total, out, cap, out_bytes)) return rc;
    uint64_t at = 0;
    for (int i = 0; i < npieces; i++) {
        if (int rc = icsp_bitstream_place(out, cap, at, pieces[i], piece_bits[i])) return rc;
        at += piece_bits[i];
    }
    return icsp_bitstream_end(out, total);
}

// ---- the inverse: readHeader (DEC:14-37) and readBlockData (DEC:38-405) of the reference DECODER
// (DEC = /root/reference/source/decoder/ICSP_Codec_Decoder_source.cpp)
namespace {
// Reads through a 64-bit window: one unaligned load gives at least 57 valid bits at the cursor, enough for any code of the
// table (at most 3 + 7 + 1 + 11 = 22 bits), and a run of zero values ("00" each) is counted with one count-leading-zeros.
// Semantics of the bit-serial reader it replaces: MSB first (DEC:64-70), zeros past the end, `over` once a bit at or
// beyond nbits has been read.
struct BitReader {
    const uint8_t* p; uint64_t nbits, at; bool over;
    inline uint64_t peek() const                                       // bits at, at+1, ... from the top; >= 57 valid, zeros past the end
    {
        const uint64_t byte = at >> 3, nbytes = nbits >> 3;
        uint64_t w = 0;
        if (byte + 8 <= nbytes) { memcpy(&w, p + byte, 8); w = __builtin_bswap64(w); }
        else for (int k = 0; k < 8; k++) if (byte + k < nbytes) w |= (uint64_t)p[byte + k] << (56 - 8 * k);
        return w << (at & 7);
    }
    inline void skip(int n) { at += (uint64_t)n; if (at > nbits) over = true; }
    inline uint32_t bit() { const uint32_t v = (uint32_t)(peek() >> 63); skip(1); return v; }
    // one value of the DC / AC / MV code (DCientropy DEC:407-608, ACientropy DEC:810-1021, MVientropy DEC:2274-2653):
    // "00" = 0; else a 3-bit category: 010 -> e = 0, 011..110 -> e = 1..4, 111 -> e = 5 + the ones that follow (at most six;
    // the bit that ends them -- or, after six, one more bit whatever it is -- is consumed); then the sign, then e bits
    inline int value_from(uint64_t w, int& len)
    {
        const uint32_t c = (uint32_t)(w >> 61);
        if (c < 2) { len = 2; return 0; }
        int e, pre;
        if (c == 2) { e = 0; pre = 3; }
        else if (c < 7) { e = (int)c - 2; pre = 3; }
        else {
            const uint32_t six = (uint32_t)(w >> 55) & 63u;            // the six bits after the category
            const uint32_t inv = ~six & 63u;
            const int ones = inv ? __builtin_clz(inv) - 26 : 6;         // leading ones of the six
            e = 5 + ones; pre = 3 + ones + 1;
        }
        const uint32_t s = (uint32_t)(w >> (63 - pre)) & 1u;
        const uint32_t low = e ? (uint32_t)((w << (pre + 1)) >> (64 - e)) : 0u;
        len = pre + 1 + e;
        const int a = (1 << e) + (int)low;
        return s ? a : -a;
    }
    inline int value() { int len; const int v = value_from(peek(), len); skip(len); return v; }
    inline void block(int16_t* lv, uint8_t* acflag)
    {
        lv[0] = (int16_t)value();
        const uint32_t ac = bit();
        *acflag = (uint8_t)ac;
        if (ac) { at += 63; for (int i = 1; i < 64; i++) lv[i] = 0; return; }    // DEC:127-132
        for (int i = 1; i < 64;) {
            const uint64_t w = peek();
            // a run of zero values: every leading "00" pair of the window (28 pairs are always valid bits)
            int z = w ? (__builtin_clzll(w) >> 1) : 28;
            if (z > 28) z = 28;
            if (z > 64 - i) z = 64 - i;
            if (z) { for (int k = 0; k < z; k++) lv[i + k] = 0; i += z; skip(2 * z); continue; }
            int len;
            lv[i++] = (int16_t)value_from(w, len);
            skip(len);
        }
    }
};
} // namespace

int icsp_parse_header(const uint8_t* bin, size_t nbytes, icsp_params_t* out)
{
    if (!bin || !out) return ICSP_ERR_UNENOUGH_PARAM;
    if (nbytes < 14 || !(bin[0] == 0 && bin[1] == 73 && bin[2] == 67 && bin[3] == 83 && bin[4] == 80)) return ICSP_ERR_UNCORRECT_PARAM;
    out->height = bin[5] | (bin[6] << 8); out->width = bin[7] | (bin[8] << 8);
    out->qp_dc = bin[9]; out->qp_ac = bin[10];
    out->intra_period = ((bin[12] | (bin[13] << 8)) & 0x1F80) >> 7;      // DEC:29
    // the geometry icsp_create accepts; a crafted header must end in a status, not in a giant allocation by the caller
    if (out->width % 16 || out->height % 16 || out->width < 32 || out->width > 4096 || out->height < 16 || out->height > 2304 ||
        (out->width / 16) * (out->height / 16) > 8704 || out->qp_dc <= 0 || out->qp_ac <= 0)
        return ICSP_ERR_UNCORRECT_PARAM;
    return ICSP_OK;
}

int icsp_parse_bitstream(const uint8_t* bin, size_t nbytes, int n,
                         int16_t* levels, uint8_t* acflag, uint8_t* mpm_mode, int8_t* mvd)
{
    if (!levels || !acflag || !mpm_mode || !mvd) return ICSP_ERR_UNENOUGH_PARAM;
    icsp_params_t p;
    if (int rc = icsp_parse_header(bin, nbytes, &p)) return rc;
    if (n < 0) return ICSP_ERR_UNCORRECT_PARAM;
    const size_t nmb = (size_t)(p.width / 16) * (p.height / 16);
    BitReader r{ bin + 14, (uint64_t)(nbytes - 14) * 8, 0, false };
    for (int f = 0; f < n; f++) {
        const bool intra = p.intra_period <= 1 || f % p.intra_period == 0;          // DEC.h:293-306
        for (size_t mb = 0; mb < nmb; mb++) {
            const size_t o = (size_t)f * nmb + mb;
            if (intra) {
                mvd[o * 2] = mvd[o * 2 + 1] = 0;
                for (int k = 0; k < 4; k++) {
                    const uint32_t flag = r.bit(), mode = r.bit();
                    mpm_mode[o * 4 + k] = (uint8_t)(flag | (mode << 1));
                    r.block(levels + (o * 6 + k) * 64, acflag + o * 6 + k);
                }
            } else {
                r.bit();                                                             // MVmodeflag (DEC:286)
                mvd[o * 2] = (int8_t)r.value();
                mvd[o * 2 + 1] = (int8_t)r.value();
                for (int k = 0; k < 4; k++) { mpm_mode[o * 4 + k] = 0; r.block(levels + (o * 6 + k) * 64, acflag + o * 6 + k); }
            }
            r.block(levels + (o * 6 + 4) * 64, acflag + o * 6 + 4);
            r.block(levels + (o * 6 + 5) * 64, acflag + o * 6 + 5);
            // Running past the end is an error, except inside the stream's very last macroblock: the reference writes its
            // final partial byte right-aligned (ENC:4956) and reads it MSB-first (DEC:64-70), so the last few values of
            // a reference stream can parse as longer codes than were written; the reference reads its zeroed tail there.
            if ((r.over || r.at > r.nbits) && !(f == n - 1 && mb == nmb - 1)) return ICSP_ERR_RANGE;
        }
    }
    return ICSP_OK;
}

int icsp_write_bitstream(const icsp_params_t* p, int n, const int16_t* levels, const uint8_t* acflag,
                         const uint8_t* mpm_mode, const int8_t* mvd, uint8_t* out, size_t cap, size_t* out_bytes)
{
    if (!p || !levels || !acflag || !mpm_mode || !mvd || !out || !out_bytes) return ICSP_ERR_UNENOUGH_PARAM;
    if (cap < 15 || n < 0) return ICSP_ERR_UNCORRECT_PARAM;
    memset(out, 0, cap);
    write_header(p, out);

    BitWriter w{ out + 14, cap - 14, 0, false };
    const size_t nmb = (size_t)(p->width / 16) * (p->height / 16);
    for (int f = 0; f < n; f++) {
        const bool intra = (p->intra_period == 0) || (f % p->intra_period == 0);   // ENC:219, 4884
        for (size_t mb = 0; mb < nmb; mb++) {
            const size_t o = (size_t)f * nmb + mb;
            const int16_t* lv = levels + o * 384;
            const uint8_t* ac = acflag + o * 6;
            if (intra) {
                for (int k = 0; k < 4; k++) {                          // ENC:5052-5080
                    w.put(mpm_mode[o * 4 + k] & 1u, 1);               // MPMFlag
                    w.put((mpm_mode[o * 4 + k] >> 1) & 1u, 1);        // intraPredMode
                    put_block(w, lv + k * 64, ac[k]);
                }
            } else {
                w.put(1, 1);                                           // mv mode flag (ENC:5152)
                put_value(w, mvd[o * 2]);                              // differential mv x then y (ENC:5154-5157)
                put_value(w, mvd[o * 2 + 1]);
                for (int k = 0; k < 4; k++) put_block(w, lv + k * 64, ac[k]);
            }
            put_block(w, lv + 4 * 64, ac[4]);                          // Cb, Cr (ENC:5088-5128, 5195-5234)
            put_block(w, lv + 5 * 64, ac[5]);
            if (w.overflow) return ICSP_ERR_RANGE;
        }
    }
    const size_t body = w.nbits / 8 + 1;                               // fwrite(..., cntbits/8 + 1, ...) (ENC:4895, 5029)
    if (body > cap - 14) return ICSP_ERR_RANGE;
    *out_bytes = 14 + body;
    return ICSP_OK;
}

} // extern "C"
