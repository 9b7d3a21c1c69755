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

#include "blake2b.hpp"
#include "curve.cuh"
#include "../../include/bzh2.h"

namespace {

using bzh::Blake2b;

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
