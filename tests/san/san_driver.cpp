// Sanitizer driver (test infrastructure): the host half of the C ABI (icsp_bitstream.cpp) and the oracle's C sources, compiled
// together under -fsanitize=address,undefined by tests/test_sanitizers.py, exercised on the inputs that could walk off a
// buffer: truncated and garbage streams through the parser, the writer at every capacity around the true size, piece
// placement at random bit boundaries from several threads, headers with hostile geometry; plus one small encode + decode of
// the oracle itself (the reference relies on undefined behaviour at ENC:84-91 and ENC:4874; the restatements
// must not).  Exit 0 and no sanitizer report = pass.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "icsp_hip.h"
extern "C" {
#include "../../oracle/icsp_oracle.h"
}

static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }

#define CHECK(c) do { if (!(c)) { printf("CHECK failed line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main()
{
    const int W = 64, H = 48, nmb = (W / 16) * (H / 16), n = 5, period = 3;
    const size_t fsz = (size_t)W * H * 3 / 2;
    std::vector<uint8_t> clip(fsz * n);
    for (size_t i = 0; i < clip.size(); i++) clip[i] = (uint8_t)(96 + (i % W) + ((i / W) % 7) * 5 + (rnd() & 7) + (i / fsz) * 3);
    std::vector<int16_t> lv((size_t)n * nmb * 384);
    std::vector<uint8_t> ac((size_t)n * nmb * 6), mpm((size_t)n * nmb * 4), rec(fsz * n);
    std::vector<int8_t> mvd((size_t)n * nmb * 2);
    // ---- the oracle itself: encode with two threads, decode again
    CHECK(icsp_oracle_encode_sequence(clip.data(), n, W, H, 8, 8, period, 2, lv.data(), ac.data(), mpm.data(), mvd.data(), rec.data()) == 0);
    icsp_params_t p{ W, H, 8, 8, period };
    // ---- writer: exact size, every capacity around it
    const size_t bound = icsp_bitstream_bound(&p, n);
    std::vector<uint8_t> bs(bound);
    size_t nb = 0;
    CHECK(icsp_write_bitstream(&p, n, lv.data(), ac.data(), mpm.data(), mvd.data(), bs.data(), bs.size(), &nb) == ICSP_OK);
    for (size_t cap = 0; cap < nb + 3; cap += (cap < 32 || cap + 40 > nb) ? 1 : 97) {
        std::vector<uint8_t> small(cap ? cap : 1);
        size_t k = 0;
        const int rc = icsp_write_bitstream(&p, n, lv.data(), ac.data(), mpm.data(), mvd.data(), small.data(), cap, &k);
        CHECK(rc == ICSP_OK ? (cap >= nb && k == nb && memcmp(small.data(), bs.data(), nb) == 0) : (cap < nb));
    }
    // ---- parser: the real stream, every truncation of it, bit flips, pure garbage, hostile headers
    std::vector<int16_t> lv2(lv.size());
    std::vector<uint8_t> ac2(ac.size()), mpm2(mpm.size());
    std::vector<int8_t> mvd2(mvd.size());
    CHECK(icsp_parse_bitstream(bs.data(), nb, n, lv2.data(), ac2.data(), mpm2.data(), mvd2.data()) == ICSP_OK);
    CHECK(memcmp(lv.data(), lv2.data(), (lv.size() - 384) * 2) == 0);            // (the last macroblock may read the reference's garbled final byte)
    for (size_t cut = 0; cut < nb; cut += (cut < 40) ? 1 : 211) {
        std::vector<uint8_t> part(bs.begin(), bs.begin() + cut);
        part.push_back(0); part.pop_back();
        (void)icsp_parse_bitstream(part.data(), cut, n, lv2.data(), ac2.data(), mpm2.data(), mvd2.data());
    }
    for (int t = 0; t < 300; t++) {
        std::vector<uint8_t> bad(bs.begin(), bs.begin() + nb);
        for (int k = 0; k < 1 + (int)(rnd() % 40); k++) bad[14 + rnd() % (nb - 14)] ^= (uint8_t)(1u << (rnd() & 7));
        (void)icsp_parse_bitstream(bad.data(), bad.size(), n, lv2.data(), ac2.data(), mpm2.data(), mvd2.data());
        for (auto& b : bad) b = (uint8_t)rnd();
        memcpy(bad.data(), bs.data(), 14);
        (void)icsp_parse_bitstream(bad.data(), bad.size(), n, lv2.data(), ac2.data(), mpm2.data(), mvd2.data());
    }
    {
        uint8_t hdr[14];
        memcpy(hdr, bs.data(), 14);
        icsp_params_t q;
        const int dims[][2] = { { 65535, 65535 }, { 4096, 2304 }, { 16, 16 }, { 0, 0 }, { 4112, 16 }, { 352, 289 } };
        for (auto& d : dims) {
            hdr[5] = (uint8_t)(d[1] & 0xff); hdr[6] = (uint8_t)(d[1] >> 8); hdr[7] = (uint8_t)(d[0] & 0xff); hdr[8] = (uint8_t)(d[0] >> 8);
            CHECK(icsp_parse_header(hdr, 14, &q) != ICSP_OK);                    // none of these is a geometry icsp_create accepts
            CHECK(icsp_parse_bitstream(hdr, 14, 1, lv2.data(), ac2.data(), mpm2.data(), mvd2.data()) != ICSP_OK);
        }
        CHECK(icsp_parse_header(bs.data(), 13, &q) != ICSP_OK);
    }
    // ---- piece placement
    {
        // random byte strings as pieces at random bit lengths: placement from 4 threads == sequential assembly
        const int np = 9;
        std::vector<std::vector<uint8_t>> pc(np);
        std::vector<uint64_t> pb(np), at(np);
        uint64_t total = 0;
        for (int i = 0; i < np; i++) {
            pb[i] = (i == 4) ? 0 : 1 + rnd() % 3000;
            pc[i].resize((size_t)((pb[i] + 7) / 8) + 1);
            for (auto& b : pc[i]) b = (uint8_t)rnd();
            pc[i].resize((size_t)((pb[i] + 7) / 8));                             // exact size: an over-read is an ASan error
            at[i] = total; total += pb[i];
        }
        std::vector<const uint8_t*> ptr(np);
        for (int i = 0; i < np; i++) ptr[i] = pc[i].data();
        std::vector<uint8_t> seq(14 + total / 8 + 3), par(14 + total / 8 + 3);
        size_t k1 = 0, k2 = 0;
        CHECK(icsp_bitstream_assemble(&p, np, ptr.data(), pb.data(), seq.data(), seq.size(), &k1) == ICSP_OK);
        CHECK(icsp_bitstream_begin(&p, total, par.data(), par.size(), &k2) == ICSP_OK);
        std::vector<std::thread> th;
        for (int w = 0; w < 4; w++)
            th.emplace_back([&, w] { for (int i = w; i < np; i += 4) icsp_bitstream_place(par.data(), par.size(), at[i], pc[i].data(), pb[i]); });
        for (auto& t : th) t.join();
        CHECK(icsp_bitstream_end(par.data(), total) == ICSP_OK);
        CHECK(k1 == k2 && memcmp(seq.data(), par.data(), k1) == 0);
        CHECK(icsp_bitstream_place(par.data(), 20, 1000, pc[0].data(), pb[0]) == ICSP_ERR_RANGE);
    }
    // ---- decoder oracle on the parsed syntax
    {
        int w = 0, h = 0, qd = 0, qa = 0, per = 0;
        CHECK(icsp_oracle_parse_header(bs.data(), nb, &w, &h, &qd, &qa, &per) == 0 && w == W && h == H && per == period);
    }
    printf("sanitizer driver: ok\n");
    return 0;
}
