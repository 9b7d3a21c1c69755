// Test program (not product code): the host's 64-bit-limb Montgomery product (csrc/field.cuh: fe_mul_host64) against the
// 32-bit CIOS twin it replaced, on every field of the library -- random operands, edge operands (0, 1, p - 1, R mod p, values
// with all-ones limbs), a Fermat inversion each.  Prints one line per field; tests/test_host_field_cpu.py asserts "OK".
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include "field.cuh"
using namespace bzh;

// the 32-bit CIOS, restated here so that both paths exist in one binary
template <class P>
static Fe<P> mul32(const Fe<P>& a, const Fe<P>& b) {
    uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t8 = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 8; j++) {
            const uint64_t x = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        uint64_t top = (uint64_t)t8 + c;
        const uint32_t m = t[0] * P::inv;
        c = ((uint64_t)m * P::mod(0) + t[0]) >> 32;
        for (int j = 1; j < 8; j++) {
            const uint64_t x = (uint64_t)m * P::mod(j) + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        top += c;
        t[7] = (uint32_t)top;
        t8 = (uint32_t)(top >> 32);
    }
    Fe<P> r;
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    fe_cond_sub_p(r, t8);
    return r;
}

template <class P>
static bool check(const char* name) {
    std::mt19937_64 rng(12345);
    auto rnd = [&]() {
        Fe<P> v;
        for (int i = 0; i < 8; i++) v.l[i] = (uint32_t)rng();
        v.l[7] &= 0x3fffffffu;          // below every modulus here (all of them exceed 2^253)
        return v;
    };
    std::vector<Fe<P>> edge;
    edge.push_back(fe_zero<P>());
    edge.push_back(fe_one<P>());
    Fe<P> one_raw = fe_zero<P>();
    one_raw.l[0] = 1;
    edge.push_back(one_raw);
    Fe<P> pm1;
    for (int i = 0; i < 8; i++) pm1.l[i] = P::mod(i);
    pm1.l[0] -= 1;
    edge.push_back(pm1);
    Fe<P> ones;
    for (int i = 0; i < 8; i++) ones.l[i] = 0xffffffffu;
    ones.l[7] = 0x1fffffffu;
    edge.push_back(ones);
    size_t bad = 0, n = 0;
    for (const auto& a : edge)
        for (const auto& b : edge) {
            n++;
            if (memcmp(fe_mul(a, b).l, mul32(a, b).l, 32)) bad++;
        }
    for (int i = 0; i < 20000; i++) {
        const Fe<P> a = rnd(), b = rnd();
        n++;
        if (memcmp(fe_mul(a, b).l, mul32(a, b).l, 32)) bad++;
    }
    const Fe<P> x = rnd(), xi = fe_inv(x);
    const bool inv_ok = !memcmp(fe_mul(x, xi).l, fe_one<P>().l, 32);
#if defined(BZH_HOST_MUL64)
    const char* path = "64-bit limbs";
#else
    const char* path = "32-bit limbs";
#endif
    printf("%s: %zu products, %zu mismatches, inverse %s, host path: %s -> %s\n", name, n, bad, inv_ok ? "ok" : "WRONG", path,
           (!bad && inv_ok) ? "OK" : "FAIL");
    return !bad && inv_ok;
}

int main() {
    bool ok = check<FpParams>("Fp") & check<FqParams>("Fq") & check<BnFrParams>("BN254 Fr") & check<BnFqParams>("BN254 Fq");
    return ok ? 0 : 1;
}
