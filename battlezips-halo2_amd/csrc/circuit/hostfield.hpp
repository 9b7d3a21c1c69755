// Host-side prime-field and Pallas-curve arithmetic for the circuit front end (witness synthesis, keygen
// constants).  4 x 64-bit limbs, Montgomery form with R = 2^256: the memory image of a value equals the device's
// 8 x 32-bit Montgomery limbs (csrc/field.cuh) and pasta_curves' in-memory Fp / Fq, so synthesised advice columns are
// uploaded without conversion.
//
// Stands where the reference reaches pasta_curves 0.4.1 (Cargo.lock:567-570, un-vendored) from its chips:
// src/chips/bitify.rs:117-123 (field adds / muls), src/chips/placement.rs:196 (lagrange_interpolate), and, through
// halo2_gadgets' ECC chip, the affine Pallas arithmetic of src/chips/pedersen.rs:104-134.
#pragma once
#include <stdint.h>
#include <string.h>

#include <array>
#include <vector>

namespace bzc {

typedef unsigned __int128 u128;

struct FpMod {  // Pallas base field = Vesta scalar field = the circuit field (modulus literal: src/chips/bitify.rs:461)
    static constexpr uint64_t m[4] = {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0ull, 0x4000000000000000ull};
    static constexpr uint64_t inv = 0x992d30ecffffffffull;  // -p^-1 mod 2^64
};
struct FqMod {  // Pallas scalar field
    static constexpr uint64_t m[4] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0ull, 0x4000000000000000ull};
    static constexpr uint64_t inv = 0x8c46eb20ffffffffull;
};

template <class M>
struct F {
    uint64_t l[4];

    static F zero() { return F{{0, 0, 0, 0}}; }
    static const F& one() {
        static const F o = from_raw_reduce({1, 0, 0, 0});
        return o;
    }
    static const F& r2() {  // R^2 mod p, by 512 modular doublings of 1
        static const F v = [] {
            F x{{1, 0, 0, 0}};
            for (int i = 0; i < 512; i++) x = add_raw(x, x);
            return x;
        }();
        return v;
    }
    // canonical integer (already < p) -> Montgomery
    static F from_raw_reduce(std::array<uint64_t, 4> c) {
        F x{{c[0], c[1], c[2], c[3]}};
        return mul(x, r2());
    }
    static F from_u64(uint64_t v) { return from_raw_reduce({v, 0, 0, 0}); }
    static F from_u128(u128 v) { return from_raw_reduce({(uint64_t)v, (uint64_t)(v >> 64), 0, 0}); }
    // 32 canonical little-endian bytes; false if >= p (from_repr(..) is None upstream)
    static bool from_repr(const uint8_t* b, F* out) {
        uint64_t c[4];
        memcpy(c, b, 32);
        if (!lt_mod(c)) return false;
        *out = from_raw_reduce({c[0], c[1], c[2], c[3]});
        return true;
    }
    static bool from_limbs(const uint64_t* c, F* out) { return from_repr((const uint8_t*)c, out); }
    void to_limbs(uint64_t* out) const {  // canonical
        F x = mont_reduce_only(*this);
        memcpy(out, x.l, 32);
    }
    void to_repr(uint8_t* out) const { to_limbs((uint64_t*)out); }
    std::array<uint64_t, 4> canon() const {
        std::array<uint64_t, 4> c;
        to_limbs(c.data());
        return c;
    }

    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const F& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
    bool operator!=(const F& o) const { return !(*this == o); }

    static bool lt_mod(const uint64_t* c) {
        for (int i = 3; i >= 0; i--) {
            if (c[i] < M::m[i]) return true;
            if (c[i] > M::m[i]) return false;
        }
        return false;
    }
    static F add_raw(const F& a, const F& b) {  // a + b mod p on raw residues < p
        F r;
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)a.l[i] + b.l[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
        if (!lt_mod(r.l)) {  // p < 2^255: no carry out of limb 3
            u128 br = 0;
            for (int i = 0; i < 4; i++) {
                u128 d = (u128)r.l[i] - M::m[i] - br;
                r.l[i] = (uint64_t)d;
                br = (d >> 64) & 1;
            }
        }
        return r;
    }
    friend F operator+(const F& a, const F& b) { return add_raw(a, b); }
    friend F operator-(const F& a, const F& b) {
        F r;
        u128 br = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)a.l[i] - b.l[i] - br;
            r.l[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
        if (br) {
            u128 c = 0;
            for (int i = 0; i < 4; i++) {
                c += (u128)r.l[i] + M::m[i];
                r.l[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    F operator-() const { return zero() - *this; }
    F dbl() const { return add_raw(*this, *this); }

    // Montgomery product (CIOS)
    static F mul(const F& a, const F& b) {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)a.l[j] * b.l[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            const uint64_t q = t[0] * M::inv;
            c = (u128)q * M::m[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; j++) {
                c += (u128)q * M::m[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        F r{{t[0], t[1], t[2], t[3]}};
        if (t[4] || !lt_mod(r.l)) {
            u128 br = 0;
            for (int i = 0; i < 4; i++) {
                u128 d = (u128)r.l[i] - M::m[i] - br;
                r.l[i] = (uint64_t)d;
                br = (d >> 64) & 1;
            }
        }
        return r;
    }
    friend F operator*(const F& a, const F& b) { return mul(a, b); }
    F sqr() const { return mul(*this, *this); }
    static F mont_reduce_only(const F& a) { return mul(a, F{{1, 0, 0, 0}}); }

    F pow(const uint64_t* e, int limbs) const {
        F acc = one();
        for (int i = limbs * 64 - 1; i >= 0; i--) {
            acc = acc.sqr();
            if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
        }
        return acc;
    }
    // inverse by Fermat (0 -> 0, like ff's invert().unwrap_or(0) call sites in the ECC witness code)
    F inv() const {
        uint64_t e[4] = {M::m[0] - 2, M::m[1], M::m[2], M::m[3]};
        return pow(e, 4);
    }
    bool is_odd() const { return canon()[0] & 1; }

    // Jacobi symbol (a / p) of the canonical value by the binary algorithm: +1 square, -1 non-square, 0 zero.
    int jacobi() const {
        uint64_t a[4], n[4];
        to_limbs(a);
        memcpy(n, M::m, 32);
        auto iszero = [](const uint64_t* x) { return (x[0] | x[1] | x[2] | x[3]) == 0; };
        auto shr = [](uint64_t* x, unsigned s) {
            while (s >= 64) {
                x[0] = x[1], x[1] = x[2], x[2] = x[3], x[3] = 0;
                s -= 64;
            }
            if (s) {
                x[0] = (x[0] >> s) | (x[1] << (64 - s));
                x[1] = (x[1] >> s) | (x[2] << (64 - s));
                x[2] = (x[2] >> s) | (x[3] << (64 - s));
                x[3] >>= s;
            }
        };
        auto ctz = [](const uint64_t* x) -> unsigned {
            unsigned s = 0;
            for (int i = 0; i < 4; i++) {
                if (x[i]) return s + (unsigned)__builtin_ctzll(x[i]);
                s += 64;
            }
            return s;
        };
        auto less = [](const uint64_t* x, const uint64_t* y) {
            for (int i = 3; i >= 0; i--) {
                if (x[i] != y[i]) return x[i] < y[i];
            }
            return false;
        };
        if (iszero(a)) return 0;
        int t = 1;
        while (!iszero(a)) {
            const unsigned z = ctz(a);
            if (z) {
                shr(a, z);
                const unsigned r = (unsigned)(n[0] & 7);
                if ((z & 1) && (r == 3 || r == 5)) t = -t;
            }
            if (less(a, n)) {
                for (int i = 0; i < 4; i++) {
                    uint64_t tmp = a[i];
                    a[i] = n[i];
                    n[i] = tmp;
                }
                if ((a[0] & 3) == 3 && (n[0] & 3) == 3) t = -t;
            }
            u128 br = 0;  // a -= n
            for (int i = 0; i < 4; i++) {
                u128 d = (u128)a[i] - n[i] - br;
                a[i] = (uint64_t)d;
                br = (d >> 64) & 1;
            }
        }
        return (n[0] == 1 && !(n[1] | n[2] | n[3])) ? t : 0;
    }

    // Square root with the root choice of pasta_curves 0.4.1's table-based `sqrt` (Sarkar): with p - 1 = 2^32 T,
    // g = 5^T the 2^32-th root of unity and t in [0, 2^32) even such that u^T g^t = 1, the result is
    // u^((T+1)/2) g^(t/2).  (This fixes WHICH of the two roots the fixed-base `u` tables hold; pinned by the sampled
    // U rows of tests/golden/fixed_bases.json.)  Returns false for a non-square.
    bool sqrt(F* out) const {
        if (is_zero()) {
            *out = zero();
            return true;
        }
        // T = (p - 1) >> 32 ; (T + 1) / 2 ; g = 5^T
        uint64_t pm1[4] = {M::m[0] - 1, M::m[1], M::m[2], M::m[3]};
        uint64_t T[4] = {(pm1[0] >> 32) | (pm1[1] << 32), (pm1[1] >> 32) | (pm1[2] << 32), (pm1[2] >> 32) | (pm1[3] << 32), pm1[3] >> 32};
        uint64_t Th[4];  // (T + 1) / 2 = (T >> 1) + 1 for odd T
        Th[0] = (T[0] >> 1) | (T[1] << 63), Th[1] = (T[1] >> 1) | (T[2] << 63), Th[2] = (T[2] >> 1) | (T[3] << 63), Th[3] = T[3] >> 1;
        for (int i = 0; i < 4 && ++Th[i] == 0; i++) {
        }
        static const F g = from_u64(5).pow(T, 4);
        static const std::vector<F> gpow = [] {  // gpow[i] = g^(2^i), i <= 32
            std::vector<F> v{g};
            for (int i = 0; i < 32; i++) v.push_back(v.back().sqr());
            return v;
        }();
        const F w = pow(Th, 4);          // u^((T+1)/2)
        F x = w.sqr() * inv();           // u^T, of order dividing 2^32
        // find t (bit by bit from the bottom) with x * g^t = 1; squares have even t
        uint64_t t = 0;
        F cur = x;
        for (int i = 0; i < 32; i++) {
            // cur^(2^(31-i)) is +1 or -1; if -1 the bit i of t is set
            F y = cur;
            for (int j = 0; j < 31 - i; j++) y = y.sqr();
            if (y != one()) {
                t |= (uint64_t)1 << i;
                cur = cur * gpow[i];
            }
        }
        if (t & 1) return false;
        F res = w;
        const uint64_t h = t >> 1;
        for (int i = 0; i < 32; i++) {
            if ((h >> i) & 1) res = res * gpow[i];
        }
        *out = res;
        return true;
    }
};
typedef F<FpMod> Fp;
typedef F<FqMod> Fq;

template <class T>
inline void batch_invert(std::vector<T>& v) {  // zeros stay zero
    std::vector<T> pre(v.size());
    T run = T::one();
    for (size_t i = 0; i < v.size(); i++) {
        pre[i] = run;
        if (!v[i].is_zero()) run = run * v[i];
    }
    T inv = run.inv();
    for (size_t i = v.size(); i-- > 0;) {
        if (v[i].is_zero()) continue;
        const T t = inv * pre[i];
        inv = inv * v[i];
        v[i] = t;
    }
}

// arithmetic::lagrange_interpolate (UPSTREAM halo2_proofs 0.2.0; call site src/chips/placement.rs:196): coefficients,
// low to high, of the polynomial of degree < n through (points[i], evals[i]).
template <class T>
inline std::vector<T> lagrange_interpolate(const std::vector<T>& points, const std::vector<T>& evals) {
    const size_t n = points.size();
    std::vector<T> res(n, T::zero());
    for (size_t j = 0; j < n; j++) {
        std::vector<T> num{T::one()};
        T den = T::one();
        for (size_t m = 0; m < n; m++) {
            if (m == j) continue;
            std::vector<T> nx(num.size() + 1, T::zero());
            for (size_t i = 0; i < num.size(); i++) {
                nx[i + 1] = nx[i + 1] + num[i];
                nx[i] = nx[i] - points[m] * num[i];
            }
            num.swap(nx);
            den = den * (points[j] - points[m]);
        }
        const T c = evals[j] * den.inv();
        for (size_t i = 0; i < num.size(); i++) res[i] = res[i] + c * num[i];
    }
    return res;
}

// ---------------------------------------------------------------------------------------------------------------
// Pallas: y^2 = x^3 + 5 over Fp.  Affine points; (0, 0) is the identity (halo2_gadgets' convention in the circuit).
// ---------------------------------------------------------------------------------------------------------------
struct Aff {
    Fp x, y;
    bool is_identity() const { return x.is_zero() && y.is_zero(); }
    bool operator==(const Aff& o) const { return x == o.x && y == o.y; }
};
struct Jac {
    Fp x, y, z;  // z = 0: identity
};
inline Jac jac_identity() { return Jac{Fp::zero(), Fp::one(), Fp::zero()}; }
inline Jac to_jac(const Aff& a) { return a.is_identity() ? jac_identity() : Jac{a.x, a.y, Fp::one()}; }
inline Jac jac_double(const Jac& p) {  // a = 0 curve
    if (p.z.is_zero()) return p;
    const Fp a = p.x.sqr(), b = p.y.sqr(), c = b.sqr();
    Fp d = (p.x + b).sqr() - a - c;
    d = d.dbl();
    const Fp e = a.dbl() + a, f = e.sqr();
    Jac r;
    r.x = f - d.dbl();
    r.y = e * (d - r.x) - c.dbl().dbl().dbl();
    r.z = (p.y * p.z).dbl();
    return r;
}
inline Jac jac_add(const Jac& p, const Jac& q) {
    if (p.z.is_zero()) return q;
    if (q.z.is_zero()) return p;
    const Fp z1z1 = p.z.sqr(), z2z2 = q.z.sqr();
    const Fp u1 = p.x * z2z2, u2 = q.x * z1z1;
    const Fp s1 = p.y * q.z * z2z2, s2 = q.y * p.z * z1z1;
    if (u1 == u2) {
        if (s1 == s2) return jac_double(p);
        return jac_identity();
    }
    const Fp h = u2 - u1, i = h.dbl().sqr(), j = h * i;
    const Fp rr = (s2 - s1).dbl(), v = u1 * i;
    Jac r;
    r.x = rr.sqr() - j - v.dbl();
    r.y = rr * (v - r.x) - (s1 * j).dbl();
    r.z = ((p.z + q.z).sqr() - z1z1 - z2z2) * h;
    return r;
}
// p + q for an affine, non-identity q (madd-2007-bl)
inline Jac jac_add_mixed(const Jac& p, const Aff& q) {
    if (p.z.is_zero()) return Jac{q.x, q.y, Fp::one()};
    const Fp z1z1 = p.z.sqr();
    const Fp u2 = q.x * z1z1, s2 = q.y * p.z * z1z1;
    if (u2 == p.x) {
        if (s2 == p.y) return jac_double(p);
        return jac_identity();
    }
    const Fp h = u2 - p.x, hh = h.sqr(), i = hh.dbl().dbl(), j = h * i;
    const Fp rr = (s2 - p.y).dbl(), v = p.x * i;
    Jac r;
    r.x = rr.sqr() - j - v.dbl();
    r.y = rr * (v - r.x) - (p.y * j).dbl();
    r.z = (p.z + h).sqr() - z1z1 - hh;
    return r;
}
inline Jac jac_neg(const Jac& p) { return Jac{p.x, -p.y, p.z}; }
inline std::vector<Aff> batch_normalize(const std::vector<Jac>& v) {
    std::vector<Fp> zs(v.size());
    for (size_t i = 0; i < v.size(); i++) zs[i] = v[i].z;
    batch_invert(zs);
    std::vector<Aff> out(v.size());
    for (size_t i = 0; i < v.size(); i++) {
        if (v[i].z.is_zero()) {
            out[i] = Aff{Fp::zero(), Fp::zero()};
            continue;
        }
        const Fp zi2 = zs[i].sqr();
        out[i] = Aff{v[i].x * zi2, v[i].y * zi2 * zs[i]};
    }
    return out;
}
inline Aff to_affine(const Jac& p) { return batch_normalize({p})[0]; }
// [s]P, s given as canonical little-endian limbs
inline Jac jac_mul(const Jac& p, const uint64_t* s, int limbs = 4) {
    Jac acc = jac_identity();
    for (int i = limbs * 64 - 1; i >= 0; i--) {
        acc = jac_double(acc);
        if ((s[i / 64] >> (i % 64)) & 1) acc = jac_add(acc, p);
    }
    return acc;
}

}  // namespace bzc
