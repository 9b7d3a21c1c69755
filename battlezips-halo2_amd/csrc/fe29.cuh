// Unsaturated field arithmetic for the MSM's bucket accumulation: 9 limbs of 29 bits in 32-bit words.
//
// The saturated product (field_mul_fips.inc) spends one v_addc_co_u32 on every v_mad_u64_u32: its column sums are 96 bits wide
// (96 multiply-adds + 96 carry instructions of the ~245 issue slots of a product).  With 29-bit limbs a column of the schoolbook
// product holds at most 9 products of 2^30 x 2^30 plus 5 reduction products of 2^29 x 2^29 -- < 2^64 -- so a column is a plain
// chain of v_mad_u64_u32 into ONE 64-bit accumulator, no carry instruction at all; the price is 81 + 45 multiply-adds instead of
// 64 + 32 and a shift / mask per column.  Additions are 9 independent v_add_u32 (no carry chain), subtractions add a multiple of
// p whose limbs dominate the subtrahend's and are followed by one parallel carry pass.
//
// Montgomery form with R' = 2^261 (9 x 29 bits).  Both Pasta moduli are 1 mod 2^32, hence 1 mod 2^29: -p^-1 = -1 mod 2^29 and the
// quotient digit of a column is the negated low limb, no multiplication; p has five non-zero 29-bit limbs besides limb 0
// (limbs 1..4 and 2^22 in limb 8).
//
// Bounds (values are NOT kept canonical):
//   fe29_mul / fe29_sqr   inputs: limbs < 2^30 (one lazy addition of carried values), a * b < 2^515;
//                         output: value < 2 p, limbs 0..7 < 2^29, limb 8 < 2^24
//   fe29_add              limb-wise, no carry: use at most once before a product
//   fe29_sub<K>           a - b + K p, carried; K p is written with limbs >= 2^30 - 2, so b's limbs may be anything < 2^30 - 2 and
//                         b's top limb at most (K p >> 232) - 2, i.e. b < K p
//   fe29_carry            one parallel pass: limbs 0..7 < 2^29 + 8
// Interface with the saturated 2^256-Montgomery world (what every table, bucket plane and kernel outside the accumulation loop
// holds): fe29_from_sat_x32 re-slices v * 2^5 (a value x 2^256 becomes x 2^261 = the R' form, unreduced: < 32 p, fine as ONE factor
// of a product whose other factor is < 2 p); fe29_to_sat multiplies by 2^256 (back to x 2^256), reduces fully and packs 8 words.
#pragma once
#include "field.cuh"

namespace bzh {

template <class P>
struct Fe29 {
    uint32_t l[9];
};

constexpr uint32_t kM29 = (1u << 29) - 1;

// limb i of the modulus in radix 2^29
template <class P>
BZH_HD constexpr uint32_t fe29_p(int i) {
    const int bit = 29 * i, w = bit >> 5, s = bit & 31;
    const uint64_t lo = w < 8 ? P::mod(w) : 0u, hi = w + 1 < 8 ? P::mod(w + 1) : 0u;
    return (uint32_t)(((lo | (hi << 32)) >> s) & kM29);
}
template <class P>
constexpr bool fe29_supported() {
    return (P::mod(0) & kM29) == 1u && fe29_p<P>(5) == 0u && fe29_p<P>(6) == 0u && fe29_p<P>(7) == 0u;
}

// limb i of K * p written with every limb but the top one raised by 2^30 (and the limb above lowered by 2): a subtrahend with
// limbs < 2^30 can be taken off limb-wise without a borrow
template <class P, int K>
BZH_HD constexpr uint32_t fe29_bias(int i) {
    // canonical digits of K * p
    uint64_t carry = 0;
    uint32_t d = 0;
    for (int j = 0; j <= i; j++) {
        const uint64_t v = (uint64_t)fe29_p<P>(j) * K + carry;
        d = j < 8 ? (uint32_t)(v & kM29) : (uint32_t)v;
        carry = j < 8 ? v >> 29 : 0;
    }
    if (i == 0) return d + (1u << 30);
    if (i < 8) return d + (1u << 30) - 2u;
    return d - 2u;
}

template <class P>
BZH_HD Fe29<P> fe29_zero() {
    Fe29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}

// (a * b + m * p) / 2^261.  Every column of the product gets its OWN 64-bit accumulator: the 81 limb products are independent
// multiply-adds (nothing but the column they land in orders them), and only the Montgomery pass -- carry in, quotient digit, the
// digit's five products with p's non-zero limbs dealt to the columns above -- is a dependent chain.  One running accumulator
// (the textbook product scanning) makes ALL 126 multiply-adds one chain: fine while two waves per SIMD alternate, but whenever the
// other wave waits for its table gather the lone wave issues a dependent v_mad_u64_u32 only every other slot -- measured in
// k_msm_accumulate as 0.8 instructions per slot against 1.0 for the saturated code, which ate the whole gain.
// p's limb 8 (2^22 for the Pasta fields) as a value the compiler cannot see through: m * 2^22 then stays ONE v_mad_u64_u32
// instead of becoming a 64-bit shift + a 64-bit add
template <class P>
BZH_HD uint32_t fe29_p8_opaque() {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t v;
    asm("s_mov_b32 %0, %1" : "=s"(v) : "i"(fe29_p<P>(8)));
    return v;
#else
    return fe29_p<P>(8);
#endif
}
// The columns below 9 arrive with 2^29 - 1 already added (fe29_mul / fe29_sqr start them there): with t = column + carry,
// the quotient digit is ~t mod 2^29 and the carry into the next column -- (t + digit) >> 29 -- is just t >> 29.
template <class P>
BZH_HD void fe29_montgomery_pass(uint64_t (&c)[17], Fe29<P>& r) {
    const uint32_t p8 = fe29_p8_opaque<P>();
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const uint64_t t = c[k] + carry;
        const uint32_t m = ~(uint32_t)t & kM29;   // p = 1 mod 2^29: the digit that clears the column's low limb
        carry = t >> 29;
#pragma unroll
        for (int l = 1; l < 8; l++) {
            if (fe29_p<P>(l) != 0u) c[k + l] += (uint64_t)m * fe29_p<P>(l);
        }
        c[k + 8] += (uint64_t)m * p8;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        const uint64_t t = c[k] + carry;
        r.l[k - 9] = (uint32_t)t & kM29;
        carry = t >> 29;
    }
    r.l[8] = (uint32_t)carry;
}
template <class P>
BZH_HD Fe29<P> fe29_mul(const Fe29<P>& a, const Fe29<P>& b) {
    static_assert(fe29_supported<P>(), "fe29: modulus must be 1 mod 2^29 with zero limbs 5..7 (the Pasta fields)");
    uint64_t c[17];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        c[k] = k < 9 ? (uint64_t)kM29 : 0;
#pragma unroll
        for (int j = (k > 8 ? k - 8 : 0); j <= (k < 8 ? k : 8); j++) c[k] += (uint64_t)a.l[j] * b.l[k - j];
    }
    Fe29<P> r;
    fe29_montgomery_pass<P>(c, r);
    return r;
}

// (a * b + c * d + m * p) / 2^261: two products into the same columns, ONE Montgomery pass (162 + 45 multiply-adds instead of
// 2 x 126).  Inputs: limb products A * B + C * D <= 1.8e18 in all (a column then stays below 2^64) and a b + c d < 2^515; output as
// fe29_mul's.  The quotient kernels' y-power glue -- ACC y^m + S inner, IN y + C -- is made of these.
template <class P>
BZH_HD Fe29<P> fe29_dot2(const Fe29<P>& a, const Fe29<P>& b, const Fe29<P>& c, const Fe29<P>& d) {
    static_assert(fe29_supported<P>(), "fe29: unsupported modulus");
    uint64_t col[17];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        col[k] = k < 9 ? (uint64_t)kM29 : 0;
#pragma unroll
        for (int j = (k > 8 ? k - 8 : 0); j <= (k < 8 ? k : 8); j++) col[k] += (uint64_t)a.l[j] * b.l[k - j];
#pragma unroll
        for (int j = (k > 8 ? k - 8 : 0); j <= (k < 8 ? k : 8); j++) col[k] += (uint64_t)c.l[j] * d.l[k - j];
    }
    Fe29<P> r;
    fe29_montgomery_pass<P>(col, r);
    return r;
}

template <class P>
BZH_HD Fe29<P> fe29_sqr(const Fe29<P>& a) {
    static_assert(fe29_supported<P>(), "fe29: unsupported modulus");
    uint64_t c[17];
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        c[k] = k < 9 ? (uint64_t)kM29 : 0;
#pragma unroll
        for (int j = (k > 8 ? k - 8 : 0); 2 * j < k; j++) c[k] += (uint64_t)d[j] * a.l[k - j];
        if ((k & 1) == 0) c[k] += (uint64_t)a.l[k >> 1] * a.l[k >> 1];
    }
    Fe29<P> r;
    fe29_montgomery_pass<P>(c, r);
    return r;
}

template <class P>
BZH_HD Fe29<P> fe29_add(const Fe29<P>& a, const Fe29<P>& b) {
    Fe29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// one parallel carry pass: every limb keeps its low 29 bits and takes the overflow of the limb below
template <class P>
BZH_HD Fe29<P> fe29_carry(const Fe29<P>& a) {
    Fe29<P> r;
    r.l[0] = a.l[0] & kM29;
#pragma unroll
    for (int i = 1; i < 8; i++) r.l[i] = (a.l[i] & kM29) + (a.l[i - 1] >> 29);
    r.l[8] = a.l[8] + (a.l[7] >> 29);
    return r;
}
// a - b + K p, carried.  b: limbs < 2^30 and value < (K p's top limb allows): K = 4 for b < 2 p, K = 8 for b < 4 p.
template <class P, int K>
BZH_HD Fe29<P> fe29_sub(const Fe29<P>& a, const Fe29<P>& b) {
    Fe29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (fe29_bias<P, K>(i) - b.l[i]);
    return fe29_carry(r);
}
// a - b + K p WITHOUT the carry pass, the bias written as J copies of (K / J) p: b's limbs 0..7 may be anything up to
// J (2^30 - 2) and its top limb up to K 2^22 - 2 J; the result's limbs grow by up to J (2^30 + 2^29) over a's.  For callers that
// track limb bounds themselves (the quotient generator, quotient_program.hpp: a carry pass only where a product or a
// subtrahend needs it).
template <class P, int K, int J>
BZH_HD Fe29<P> fe29_sub_lazy(const Fe29<P>& a, const Fe29<P>& b) {
    static_assert(J >= 1 && K % J == 0, "fe29_sub_lazy: K p is written as J copies of (K / J) p");
    Fe29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + ((uint32_t)J * fe29_bias<P, K / J>(i) - b.l[i]);
    return r;
}
// a - b - c - c + K p (the x3 of the mixed addition: R^2 - PPP - 2 Q), carried.  K p must dominate b + 2 c limb-wise: every bias
// limb is ~2^30 + digit, so b, c are taken in carried form (limbs < 2^29 + 8) and the bias is used TWICE (2 K p).
template <class P, int K>
BZH_HD Fe29<P> fe29_sub3(const Fe29<P>& a, const Fe29<P>& b, const Fe29<P>& c) {
    Fe29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (2u * fe29_bias<P, K>(i) - b.l[i] - 2u * c.l[i]);
    return fe29_carry(r);   // limbs stay below 2^32 (2^31 + 2^30 + 2^29): one pass leaves them under 2^29 + 8
}

// ---- saturated (8 x 32, 2^256-Montgomery, < p) <-> fe29 ----------------------------------------------------------
// v * 2^5 re-sliced into 9 limbs: bit 29 i of the result is bit 29 i - 5 of v
template <class P>
BZH_HD Fe29<P> fe29_from_sat_x32(const Fe<P>& v) {
    Fe29<P> r;
    r.l[0] = (v.l[0] << 5) & kM29;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        const int bit = 29 * i - 5, w = bit >> 5, s = bit & 31;
        const uint32_t lo = v.l[w] >> s, hi = (s && w + 1 < 8) ? v.l[w + 1] << (32 - s) : 0u;
        r.l[i] = (lo | hi) & kM29;
    }
    return r;
}
// v (saturated, canonical) -> v * 2^5 mod p in carried limbs, value < 2 p: the R' form of a table coordinate, fit to BE an
// accumulator coordinate.  32 v = top 2^254 + low and 2^254 = -c (mod p), c = p - 2^254 (limbs 0..4 of p): low + p - top c.
template <class P>
BZH_HD Fe29<P> fe29_from_sat_reduced(const Fe<P>& v) {
    Fe29<P> t = fe29_from_sat_x32(v);
    const uint32_t top = t.l[8] >> 22;           // < 64
    t.l[8] &= (1u << 22) - 1u;
    uint32_t e[6];
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        acc += (uint64_t)top * fe29_p<P>(i);
        e[i] = (uint32_t)acc & kM29;
        acc >>= 29;
    }
    e[5] = (uint32_t)acc;
    Fe29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = t.l[i] + (fe29_bias<P, 1>(i) - (i < 6 ? e[i] : 0u));
    return fe29_carry(r);
}
// carried value < 128 p -> the same residue below 2 p (carried): the bits above 2^254 fold back as -top c, plus one p so that
// nothing goes negative.  ~45 instructions; what keeps long chains of lazy additions / biased subtractions inside the product's
// input range.
template <class P>
BZH_HD Fe29<P> fe29_fold(const Fe29<P>& a) {
    // limbs 0..7 < 2^29 + 8, limb 8 < 2^29: top = bits from 2^254 up
    const uint32_t top = a.l[8] >> 22;            // < 128
    uint32_t e[6];
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        acc += (uint64_t)top * fe29_p<P>(i);
        e[i] = (uint32_t)acc & kM29;
        acc >>= 29;
    }
    e[5] = (uint32_t)acc;
    Fe29<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = a.l[i] + (fe29_bias<P, 1>(i) - (i < 6 ? e[i] : 0u));
    r.l[8] = (a.l[8] & ((1u << 22) - 1u)) + fe29_bias<P, 1>(8);
    return fe29_carry(r);
}
// a + b, carried (the quotient evaluator's addition: every value it holds keeps limbs below 2^29 + 8)
template <class P>
BZH_HD Fe29<P> fe29_add_c(const Fe29<P>& a, const Fe29<P>& b) {
    return fe29_carry(fe29_add(a, b));
}
// constants in fe29 form from a value given as 8 saturated words (raw integer, not shifted)
template <class P>
BZH_HD Fe29<P> fe29_from_raw(const uint32_t w8[8]) {
    Fe29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        const uint32_t lo = w < 8 ? w8[w] >> s : 0u, hi = (s && w + 1 < 8) ? w8[w + 1] << (32 - s) : 0u;
        r.l[i] = (lo | hi) & kM29;
    }
    return r;
}
// 2^k mod p as 8 raw words, by doubling from 1 with a conditional subtraction: constant-evaluated (fe29_k below)
template <class P>
struct Fe29Words {
    uint32_t w[8];
};
template <class P>
BZH_HD constexpr Fe29Words<P> fe29_pow2_mod_p(int k) {
    Fe29Words<P> r{{1, 0, 0, 0, 0, 0, 0, 0}};
    uint32_t* v = r.w;
    for (int s = 0; s < k; s++) {
        uint32_t c = 0;
        for (int i = 0; i < 8; i++) {
            const uint32_t n = (v[i] << 1) | c;
            c = v[i] >> 31;
            v[i] = n;
        }
        // v < 2 p < 2^256: subtract p if v >= p
        bool ge = true;
        for (int i = 7; i >= 0; i--) {
            if (v[i] != P::mod(i)) {
                ge = v[i] > P::mod(i);
                break;
            }
        }
        if (ge) {
            uint64_t br = 0;
            for (int i = 0; i < 8; i++) {
                const uint64_t d = (uint64_t)v[i] - P::mod(i) - br;
                v[i] = (uint32_t)d;
                br = (d >> 63) & 1;
            }
        }
    }
    return r;
}
// limb i (radix 2^29) of 2^k mod p, a compile-time constant
template <class P, int K>
BZH_HD constexpr uint32_t fe29_pow2_limb(int i) {
    constexpr Fe29Words<P> w = fe29_pow2_mod_p<P>(K);
    const int bit = 29 * i, wi = bit >> 5, s = bit & 31;
    const uint64_t lo = wi < 8 ? w.w[wi] : 0u, hi = wi + 1 < 8 ? w.w[wi + 1] : 0u;
    return (uint32_t)(((lo | (hi << 32)) >> s) & kM29);
}
// full reduction of a carried value < 4 p to the canonical representative, packed into 8 saturated words
template <class P>
BZH_HD Fe<P> fe29_pack_canonical(const Fe29<P>& a) {
    // exact digits: sequential carry
    uint32_t d[9];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t t = a.l[i] + c;
        d[i] = t & kM29;
        c = t >> 29;
    }
    d[8] = a.l[8] + c;
    // subtract p while >= p (at most three times for a value < 4 p)
#pragma unroll
    for (int rep = 0; rep < 3; rep++) {
        uint32_t t[9];
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const uint32_t s = d[i] - fe29_p<P>(i) - br;
            br = (i < 8) ? (s >> 31) : 0u;          // limb went negative: borrow from the limb above
            t[i] = (i < 8) ? (s & kM29) : s;
        }
        const bool neg = (int32_t)t[8] < 0;
        if (!neg) {
#pragma unroll
            for (int i = 0; i < 9; i++) d[i] = t[i];
        }
    }
    Fe<P> r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        // word w holds bits [32 w, 32 w + 32): from limbs floor(32 w / 29) ...
        const int bit = 32 * w, i = bit / 29, s = bit - 29 * i;
        uint64_t v = (uint64_t)d[i] >> s;
        if (i + 1 < 9) v |= (uint64_t)d[i + 1] << (29 - s);
        if (i + 2 < 9 && 58 - s < 32) v |= (uint64_t)d[i + 2] << (58 - s);
        r.l[w] = (uint32_t)v;
    }
    return r;
}
// R'-form (x 2^261, carried, < 128 p) -> saturated 2^256-Montgomery (< p) WITHOUT a product: the two forms differ by 2^5, and
// p = 1 mod 32, so  v / 32 = (u + ((-u) mod 32) p) >> 5  for the canonical u = v mod p -- exact digits, one fold of the bits above
// 2^254 (2^254 = -c mod p), one conditional subtraction, five small multiply-adds, a re-slice: ~170 instructions against ~370
// for fe29_to_sat.  This is what a finished bucket costs on its way back to the planes every other kernel reads.
template <class P>
BZH_HD Fe<P> fe29_to_sat_div32(const Fe29<P>& a) {
    // 1. exact digits
    uint32_t d[9], c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t t = a.l[i] + c;
        d[i] = t & kM29;
        c = t >> 29;
    }
    d[8] = a.l[8] + c;
    // 2. v = top 2^254 + low = low - top c (mod p):  w = low + p - top c  in (0, 2 p)
    const uint32_t top = d[8] >> 22;
    d[8] &= (1u << 22) - 1u;
    uint32_t e[6];
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        acc += (uint64_t)top * fe29_p<P>(i);
        e[i] = (uint32_t)acc & kM29;
        acc >>= 29;
    }
    e[5] = (uint32_t)acc;
    uint32_t w[9];
    c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t t = d[i] + (fe29_bias<P, 1>(i) - (i < 6 ? e[i] : 0u)) + c;
        w[i] = t & kM29;
        c = t >> 29;
    }
    w[8] = d[8] + fe29_bias<P, 1>(8) + c;
    // 3. u = w - p if that is not negative
    uint32_t u[9], br = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t t = w[i] - fe29_p<P>(i) - br;
        br = i < 8 ? t >> 31 : 0u;
        u[i] = i < 8 ? t & kM29 : t;
    }
    const bool neg = (int32_t)u[8] < 0;
#pragma unroll
    for (int i = 0; i < 9; i++) u[i] = neg ? w[i] : u[i];
    // 4. z = u + m p, m = -u mod 32: divisible by 32; z >> 5 < p
    const uint32_t m = (0u - u[0]) & 31u;
    uint32_t z[9];
    acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc += (uint64_t)u[i];
        if (fe29_p<P>(i) != 0u) acc += (uint64_t)m * fe29_p<P>(i);
        z[i] = i < 8 ? (uint32_t)acc & kM29 : (uint32_t)acc;
        acc >>= 29;
    }
    // 5. words of z >> 5: bit 32 w + 5 of z onwards
    Fe<P> r;
#pragma unroll
    for (int wd = 0; wd < 8; wd++) {
        const int bit = 32 * wd + 5, i = bit / 29, sh = bit - 29 * i;
        uint64_t v = (uint64_t)z[i] >> sh;
        if (i + 1 < 9) v |= (uint64_t)z[i + 1] << (29 - sh);
        if (i + 2 < 9 && 58 - sh < 32) v |= (uint64_t)z[i + 2] << (58 - sh);
        r.l[wd] = (uint32_t)v;
    }
    return r;
}
// R'-form (x 2^261, < 2 p or so) -> saturated 2^256-Montgomery (< p): one product with 2^256 mod p, full reduction, repack
template <class P>
BZH_HD Fe<P> fe29_to_sat(const Fe29<P>& a, const Fe29<P>& two256) {
    return fe29_pack_canonical(fe29_mul(a, two256));
}

// ---- fe29 values at rest: three planes per column of `size` elements -- limbs 0..3 (uint4), limbs 4..7 (uint4), limb 8 (u32);
//      9 size words per column.  Every plane is read coalesced by consecutive rows.  (The quotient evaluator's columns.)
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
template <class P>
__device__ __forceinline__ Fe29<P> fe29_load_planes(const uint32_t* __restrict__ col, size_t idx, size_t size) {
    const uint4 a = reinterpret_cast<const uint4*>(col)[idx];
    const uint4 b = reinterpret_cast<const uint4*>(col + 4 * size)[idx];
    Fe29<P> r;
    r.l[0] = a.x, r.l[1] = a.y, r.l[2] = a.z, r.l[3] = a.w;
    r.l[4] = b.x, r.l[5] = b.y, r.l[6] = b.z, r.l[7] = b.w;
    r.l[8] = col[8 * size + idx];
    return r;
}
template <class P>
__device__ __forceinline__ void fe29_store_planes(uint32_t* __restrict__ col, size_t idx, size_t size, const Fe29<P>& v) {
    reinterpret_cast<uint4*>(col)[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    reinterpret_cast<uint4*>(col + 4 * size)[idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    col[8 * size + idx] = v.l[8];
}
// The same load through a GLOBAL-address-space pointer with a 32-bit element index (the quotient kernels: their column pointers
// come out of a pointer table, so the compiler only knows them as generic -- flat_load with a 64-bit per-lane address, and a flat
// access also counts on lgkmcnt, so every wait for a scalar load waited for the prefetched columns too).  Uniform base in SGPRs +
// zero-extended 32-bit lane offset is the global_load saddr form: no per-lane 64-bit address arithmetic.  size * 16 < 2^32.
#define BZH_AS1 __attribute__((address_space(1)))
typedef const char BZH_AS1* fe29_gbytes;
typedef uint32_t fe29_u32x4 __attribute__((ext_vector_type(4)));
template <class P>
__device__ __forceinline__ Fe29<P> fe29_load_planes_g(fe29_gbytes col, uint32_t idx, size_t size) {
    const uint32_t off16 = idx << 4, off4 = idx << 2;
    const fe29_u32x4 a = *(const fe29_u32x4 BZH_AS1*)(col + off16);
    const fe29_u32x4 b = *(const fe29_u32x4 BZH_AS1*)(col + 16 * size + off16);
    Fe29<P> r;
    r.l[0] = a.x, r.l[1] = a.y, r.l[2] = a.z, r.l[3] = a.w;
    r.l[4] = b.x, r.l[5] = b.y, r.l[6] = b.z, r.l[7] = b.w;
    r.l[8] = *(const uint32_t BZH_AS1*)(col + 32 * size + off4);
    return r;
}
// a per-proof constant through a global-address-space pointer (for a callee that fetches its own operand)
template <class P>
__device__ __forceinline__ Fe29<P> fe29_load_const_g(fe29_gbytes p) {
    const fe29_u32x4 a = *(const fe29_u32x4 BZH_AS1*)p, b = *(const fe29_u32x4 BZH_AS1*)(p + 16);
    Fe29<P> r;
    r.l[0] = a.x, r.l[1] = a.y, r.l[2] = a.z, r.l[3] = a.w;
    r.l[4] = b.x, r.l[5] = b.y, r.l[6] = b.z, r.l[7] = b.w;
    r.l[8] = *(const uint32_t BZH_AS1*)(p + 32);
    return r;
}
// An Fe29 across a CALL: a 36-byte struct goes through scratch memory (byval / sret), a 9-lane vector travels in VGPRs.
typedef uint32_t fe29_vec __attribute__((ext_vector_type(9)));
template <class P>
__device__ __forceinline__ fe29_vec fe29_pack_vec(const Fe29<P>& a) {
    fe29_vec v;
#pragma unroll
    for (int i = 0; i < 9; i++) v[i] = a.l[i];
    return v;
}
template <class P>
__device__ __forceinline__ Fe29<P> fe29_unpack_vec(fe29_vec v) {
    Fe29<P> a;
#pragma unroll
    for (int i = 0; i < 9; i++) a.l[i] = v[i];
    return a;
}
// a per-proof constant: 12 words apart (9 used), 16-byte aligned
template <class P>
__device__ __forceinline__ Fe29<P> fe29_load_const(const uint32_t* __restrict__ p) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
    Fe29<P> r;
    r.l[0] = a.x, r.l[1] = a.y, r.l[2] = a.z, r.l[3] = a.w;
    r.l[4] = b.x, r.l[5] = b.y, r.l[6] = b.z, r.l[7] = b.w;
    r.l[8] = p[8];
    return r;
}
#endif

}  // namespace bzh
