// Proof / IO record of the reference's wasm frontend behind the C ABI.
//
// src/wasm/circuit_wasm.rs:27-31 defines what crosses to JavaScript:
//     struct BattleZipsWASM { commitment: Vec<[u8; 32]>, proof: Vec<u8> }
// `commitment` = the public inputs as BinaryValue::from_fp(fp).to_repr() (32 little-endian bytes each, :75-83, :164-167),
// `proof` = Blake2bWrite::finalize(); verify_board / verify_shot (:86-116) read the same shape back through
// BinaryValue::from_repr(bin).to_fp(), which refuses a non-canonical element.  serde writes both fields as arrays of
// numbers.  Two forms here: that JSON text, and a fixed-stride binary record -- what the multi-GPU gather carries
// (one RCCL all_gather of equal-sized records) and what a batch client stores per proof.
//
// Host code only: no kernel, no ctx.
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/bzh2.h"

namespace {

constexpr size_t kHeader = BZH_RECORD_HEADER_BYTES;      // 16 bytes of header + 4 x 32 bytes of public inputs
constexpr size_t kMaxInputs = BZH_RECORD_MAX_INPUTS;

// Fp modulus (src/chips/bitify.rs:461), little-endian limbs
constexpr uint64_t kFp[4] = {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0000000000000000ull, 0x4000000000000000ull};

bool canonical(const uint8_t* repr) {
    uint64_t l[4];
    std::memcpy(l, repr, 32);
    for (int i = 3; i >= 0; --i) {
        if (l[i] < kFp[i]) return true;
        if (l[i] > kFp[i]) return false;
    }
    return false;   // p itself
}

void put32(uint8_t* p, uint32_t v) { p[0] = uint8_t(v); p[1] = uint8_t(v >> 8); p[2] = uint8_t(v >> 16); p[3] = uint8_t(v >> 24); }
uint32_t get32(const uint8_t* p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24; }

// checks shared by decode / to_json
int check_record(const uint8_t* rec, size_t stride, uint32_t* proof_len, uint32_t* n_inputs) {
    if (!rec || stride < kHeader) return BZH_E_ARG;
    uint32_t len = get32(rec), n = rec[4];
    if (n > kMaxInputs || len > stride - kHeader || rec[6] || rec[7] || get32(rec + 12)) return BZH_E_RANGE;
    for (uint32_t i = 0; i < n; ++i)
        if (!canonical(rec + 16 + 32 * i)) return BZH_E_RANGE;
    *proof_len = len;
    *n_inputs = n;
    return BZH_OK;
}

struct Cursor {
    const char* p;
    const char* end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool lit(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
    bool key(const char* name) {            // "name":
        ws();
        size_t n = std::strlen(name);
        if (size_t(end - p) < n + 2 || *p != '"' || std::memcmp(p + 1, name, n) != 0 || p[n + 1] != '"') return false;
        p += n + 2;
        return lit(':');
    }
    bool byte(uint8_t* out) {               // a JSON number 0..255, integers only
        ws();
        if (p >= end || *p < '0' || *p > '9') return false;
        unsigned v = 0;
        const char* s = p;
        while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + unsigned(*p - '0'); if (v > 255) return false; ++p; }
        if (p - s > 1 && *s == '0') return false;
        if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) return false;
        *out = uint8_t(v);
        return true;
    }
};

}  // namespace

extern "C" {

size_t bzh_record_stride(size_t proof_stride) { return kHeader + proof_stride; }

int bzh_record_encode(const uint64_t* public_inputs, size_t n_inputs, const uint8_t* proof, size_t proof_len, uint32_t kind, uint32_t index,
                      uint8_t* record, size_t record_stride) {
    if (!record || (n_inputs && !public_inputs) || (proof_len && !proof) || kind > 255) return BZH_E_ARG;
    if (n_inputs > kMaxInputs || record_stride < kHeader || proof_len > record_stride - kHeader) return BZH_E_RANGE;
    for (size_t i = 0; i < n_inputs; ++i)
        if (!canonical(reinterpret_cast<const uint8_t*>(public_inputs + 4 * i))) return BZH_E_RANGE;
    std::memset(record, 0, record_stride);
    put32(record, uint32_t(proof_len));
    record[4] = uint8_t(n_inputs);
    record[5] = uint8_t(kind);
    put32(record + 8, index);
    std::memcpy(record + 16, public_inputs, 32 * n_inputs);   // limbs are little-endian: the memory image IS to_repr
    if (proof_len) std::memcpy(record + kHeader, proof, proof_len);
    return BZH_OK;
}

int bzh_record_decode(const uint8_t* record, size_t record_stride, uint64_t* public_inputs, size_t* n_inputs, const uint8_t** proof,
                      size_t* proof_len, uint32_t* kind, uint32_t* index) {
    uint32_t len = 0, n = 0;
    int rc = check_record(record, record_stride, &len, &n);
    if (rc) return rc;
    if (public_inputs) std::memcpy(public_inputs, record + 16, 32 * n);
    if (n_inputs) *n_inputs = n;
    if (proof) *proof = record + kHeader;
    if (proof_len) *proof_len = len;
    if (kind) *kind = record[5];
    if (index) *index = get32(record + 8);
    return BZH_OK;
}

int bzh_record_to_json(const uint8_t* record, size_t record_stride, char* out, size_t cap, size_t* len) {
    uint32_t plen = 0, n = 0;
    int rc = check_record(record, record_stride, &plen, &n);
    if (rc) return rc;
    if (!len) return BZH_E_ARG;
    std::string s;
    s.reserve(64 + 140 * n + 4 * size_t(plen));
    char buf[8];
    s += "{\"commitment\":[";
    for (uint32_t i = 0; i < n; ++i) {
        s += i ? ",[" : "[";
        for (int b = 0; b < 32; ++b) {
            std::snprintf(buf, sizeof buf, b ? ",%u" : "%u", unsigned(record[16 + 32 * i + b]));
            s += buf;
        }
        s += "]";
    }
    s += "],\"proof\":[";
    for (uint32_t b = 0; b < plen; ++b) {
        std::snprintf(buf, sizeof buf, b ? ",%u" : "%u", unsigned(record[kHeader + b]));
        s += buf;
    }
    s += "]}";
    *len = s.size();
    if (!out) return BZH_OK;                 // size query
    if (cap < s.size() + 1) return BZH_E_RANGE;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return BZH_OK;
}

int bzh_record_from_json(const char* text, size_t text_len, uint32_t kind, uint32_t index, uint8_t* record, size_t record_stride) {
    if (!text || !record || kind > 255) return BZH_E_ARG;
    if (record_stride < kHeader) return BZH_E_RANGE;
    Cursor c{text, text + text_len};
    std::memset(record, 0, record_stride);
    bool have_c = false, have_p = false;
    uint32_t n = 0, plen = 0;
    if (!c.lit('{')) return BZH_E_ARG;
    for (int field = 0; field < 2; ++field) {
        if (field && !c.lit(',')) return BZH_E_ARG;
        Cursor save = c;
        if (!have_c && c.key("commitment")) {
            have_c = true;
            if (!c.lit('[')) return BZH_E_ARG;
            if (!c.lit(']')) {
                do {
                    if (n == kMaxInputs) return BZH_E_RANGE;
                    if (!c.lit('[')) return BZH_E_ARG;
                    for (int b = 0; b < 32; ++b) {
                        if (b && !c.lit(',')) return BZH_E_ARG;
                        if (!c.byte(record + 16 + 32 * n + b)) return BZH_E_ARG;
                    }
                    if (!c.lit(']')) return BZH_E_ARG;       // exactly 32 bytes: [u8; 32]
                    if (!canonical(record + 16 + 32 * n)) return BZH_E_RANGE;   // BinaryValue::from_repr(..).to_fp()
                    ++n;
                } while (c.lit(','));
                if (!c.lit(']')) return BZH_E_ARG;
            }
            continue;
        }
        c = save;
        if (!have_p && c.key("proof")) {
            have_p = true;
            if (!c.lit('[')) return BZH_E_ARG;
            if (!c.lit(']')) {
                do {
                    if (plen == record_stride - kHeader) return BZH_E_RANGE;
                    if (!c.byte(record + kHeader + plen)) return BZH_E_ARG;
                    ++plen;
                } while (c.lit(','));
                if (!c.lit(']')) return BZH_E_ARG;
            }
            continue;
        }
        return BZH_E_ARG;
    }
    if (!c.lit('}')) return BZH_E_ARG;
    c.ws();
    if (c.p != c.end || !have_c || !have_p) return BZH_E_ARG;
    put32(record, plen);
    record[4] = uint8_t(n);
    record[5] = uint8_t(kind);
    put32(record + 8, index);
    return BZH_OK;
}

}  // extern "C"
