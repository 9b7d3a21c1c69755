// Fiat-Shamir transcript of halo2_proofs 0.2.0 (`transcript::{Blake2bWrite, Challenge255}`,
// UPSTREAM / un-vendored: Cargo.lock:382-385), created by every caller of create_proof:
// benches/shot.rs:66-67, benches/board.rs:59-60, src/circuits/shot.rs:919-920,
// src/circuits/board.rs:911-912 (`Blake2bWrite::<_, vesta::Affine, Challenge255<_>>::init(vec![])`).
//
// Inherently sequential (a few hundred absorbs per proof), so it lives on the host: Blake2b-512
// (RFC 7693) with personalisation "Halo2-Transcript"; a point is absorbed as 0x01 || x || y
// (canonical little-endian), a scalar as 0x02 || repr, a challenge absorbs 0x00 and reduces the
// 64-byte digest of a CLONE of the state as a 512-bit little-endian integer.
#include <cstring>
#include <new>
#include <vector>

#include "curve.cuh"
#include "../../include/bzh2.h"

namespace {

struct Blake2b {
    uint64_t h[8];
    uint64_t t0 = 0, t1 = 0;
    uint8_t buf[128];
    size_t buflen = 0;
    static constexpr uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
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
constexpr uint64_t Blake2b::IV[8];

template <class P>
void reduce_wide(const uint8_t d[64], uint64_t out[4]) {
    using namespace bzh;
    Fe<P> lo, hi;
    for (int i = 0; i < 8; i++) {
        uint32_t a, b;
        memcpy(&a, d + 4 * i, 4);
        memcpy(&b, d + 32 + 4 * i, 4);
        lo.l[i] = a;
        hi.l[i] = b;
    }
    // x = lo + hi * 2^256 (mod p):  lo mod p = from_mont(to_mont(lo));  hi * R mod p = to_mont(hi)
    Fe<P> r = fe_add(fe_from_mont(fe_to_mont(lo)), fe_to_mont(hi));
    for (int i = 0; i < 4; i++) out[i] = (uint64_t)r.l[2 * i] | ((uint64_t)r.l[2 * i + 1] << 32);
}

}  // namespace

struct bzh_transcript {
    Blake2b state;
    int field;
    std::vector<uint8_t> proof;
};

extern "C" {

int bzh_transcript_new(int challenge_field, bzh_transcript** out) {
    if (!out || challenge_field < 0 || challenge_field > 3) return BZH_E_ARG;
    bzh_transcript* t = new (std::nothrow) bzh_transcript();
    if (!t) return BZH_E_OOM;
    t->field = challenge_field;
    t->state.init(64, reinterpret_cast<const uint8_t*>("Halo2-Transcript"));
    *out = t;
    return BZH_OK;
}
int bzh_transcript_free(bzh_transcript* t) {
    if (!t) return BZH_E_ARG;
    delete t;
    return BZH_OK;
}
int bzh_transcript_common_point(bzh_transcript* t, const uint64_t* xy_canonical) {
    if (!t || !xy_canonical) return BZH_E_ARG;
    uint8_t buf[65];
    buf[0] = 1;
    memcpy(buf + 1, xy_canonical, 64);
    t->state.update(buf, 65);
    return BZH_OK;
}
int bzh_transcript_common_scalar(bzh_transcript* t, const uint64_t* s_canonical) {
    if (!t || !s_canonical) return BZH_E_ARG;
    uint8_t buf[33];
    buf[0] = 2;
    memcpy(buf + 1, s_canonical, 32);
    t->state.update(buf, 33);
    return BZH_OK;
}
int bzh_transcript_write_point(bzh_transcript* t, int curve, const uint64_t* xy_canonical) {
    int rc = bzh_transcript_common_point(t, xy_canonical);
    if (rc) return rc;
    uint8_t c[32];
    rc = bzh_affine_compress(curve, xy_canonical, 1, BZH_FORM_CANONICAL, c);
    if (rc) return rc;
    t->proof.insert(t->proof.end(), c, c + 32);
    return BZH_OK;
}
int bzh_transcript_write_scalar(bzh_transcript* t, const uint64_t* s_canonical) {
    int rc = bzh_transcript_common_scalar(t, s_canonical);
    if (rc) return rc;
    const uint8_t* p = reinterpret_cast<const uint8_t*>(s_canonical);
    t->proof.insert(t->proof.end(), p, p + 32);
    return BZH_OK;
}
int bzh_transcript_squeeze_challenge(bzh_transcript* t, uint64_t* out_canonical) {
    if (!t || !out_canonical) return BZH_E_ARG;
    const uint8_t zero = 0;
    t->state.update(&zero, 1);
    uint8_t d[64];
    t->state.finalize(d);
    switch (t->field) {
        case BZH_FIELD_FP: reduce_wide<bzh::FpParams>(d, out_canonical); break;
        case BZH_FIELD_FQ: reduce_wide<bzh::FqParams>(d, out_canonical); break;
        case BZH_FIELD_BN254_FR: reduce_wide<bzh::BnFrParams>(d, out_canonical); break;
        case BZH_FIELD_BN254_FQ: reduce_wide<bzh::BnFqParams>(d, out_canonical); break;
    }
    return BZH_OK;
}
int bzh_transcript_proof(const bzh_transcript* t, const uint8_t** data, size_t* len) {
    if (!t || !data || !len) return BZH_E_ARG;
    *data = t->proof.data();
    *len = t->proof.size();
    return BZH_OK;
}

}  // extern "C"
