// Blake2b-512 (RFC 7693) with personalisation, host side.  Used by the Fiat-Shamir transcript (transcript.hip:
// "Halo2-Transcript") and by hash_to_curve's expand_message_xmd (params.hip: no personalisation).
#pragma once
#include <stdint.h>
#include <string.h>

namespace bzh {

struct Blake2b {
    uint64_t h[8];
    uint64_t t0 = 0, t1 = 0;
    uint8_t buf[128];
    size_t buflen = 0;
    static inline constexpr uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    void init(size_t outlen, const uint8_t personal[16]) {
        uint8_t p[64];
        memset(p, 0, sizeof(p));
        p[0] = (uint8_t)outlen;  // digest length
        p[1] = 0;                // key length
        p[2] = 1;                // fanout
        p[3] = 1;                // depth
        memcpy(p + 48, personal, 16);
        for (int i = 0; i < 8; i++) {
            uint64_t w;
            memcpy(&w, p + 8 * i, 8);
            h[i] = IV[i] ^ w;
        }
    }
    static inline uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void compress(const uint8_t block[128], bool last) {
        static const uint8_t S[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; i++) memcpy(&m[i], block + 8 * i, 8);
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[i + 8] = IV[i];
        }
        v[12] ^= t0;
        v[13] ^= t1;
        if (last) v[14] = ~v[14];
#define BZH_G(a, b, c, d, x, y)          \
    v[a] = v[a] + v[b] + x;              \
    v[d] = rotr(v[d] ^ v[a], 32);        \
    v[c] = v[c] + v[d];                  \
    v[b] = rotr(v[b] ^ v[c], 24);        \
    v[a] = v[a] + v[b] + y;              \
    v[d] = rotr(v[d] ^ v[a], 16);        \
    v[c] = v[c] + v[d];                  \
    v[b] = rotr(v[b] ^ v[c], 63);
        for (int r = 0; r < 12; r++) {
            const uint8_t* s = S[r];
            BZH_G(0, 4, 8, 12, m[s[0]], m[s[1]])
            BZH_G(1, 5, 9, 13, m[s[2]], m[s[3]])
            BZH_G(2, 6, 10, 14, m[s[4]], m[s[5]])
            BZH_G(3, 7, 11, 15, m[s[6]], m[s[7]])
            BZH_G(0, 5, 10, 15, m[s[8]], m[s[9]])
            BZH_G(1, 6, 11, 12, m[s[10]], m[s[11]])
            BZH_G(2, 7, 8, 13, m[s[12]], m[s[13]])
            BZH_G(3, 4, 9, 14, m[s[14]], m[s[15]])
        }
#undef BZH_G
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const uint8_t* in, size_t len) {
        while (len) {
            if (buflen == 128) {  // buffer full and more input follows: not the last block
                t0 += 128;
                if (t0 < 128) t1++;
                compress(buf, false);
                buflen = 0;
            }
            size_t take = 128 - buflen < len ? 128 - buflen : len;
            memcpy(buf + buflen, in, take);
            buflen += take;
            in += take;
            len -= take;
        }
    }
    void finalize(uint8_t out[64]) const {  // on a copy: the running state is kept
        Blake2b c = *this;
        c.t0 += c.buflen;
        if (c.t0 < c.buflen) c.t1++;
        memset(c.buf + c.buflen, 0, 128 - c.buflen);
        c.compress(c.buf, true);
        memcpy(out, c.h, 64);
    }
};

}  // namespace bzh
