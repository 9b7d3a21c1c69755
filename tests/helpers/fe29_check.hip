// Test program (not product code): the unsaturated 9 x 29-bit field arithmetic of the bucket accumulator (csrc/fe29.cuh,
// csrc/curve29.cuh) against the library's saturated host arithmetic, on the two Pasta fields -- products and squares of random and
// edge operands at the documented input bounds (lazily added, biased-subtracted, 32 p re-sliced inputs), the round trip with the
// saturated form, and chains of mixed additions (incl. the doubling and the inverse special cases) against curve.cuh's xyzz_madd
// as group elements.  Host code only.  Prints one line per field; tests/test_fe29_cpu.py asserts "OK".
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
#include "curve29.cuh"
using namespace bzh;

template <class P>
static Fe<P> rnd(std::mt19937_64& rng) {
    Fe<P> v;
    for (int i = 0; i < 8; i++) v.l[i] = (uint32_t)rng();
    v.l[7] &= 0x3fffffffu;   // < 2^254 < p
    return v;
}
template <class P>
static Fe<P> mul32x(const Fe<P>& a) {   // a * 32 mod p (saturated, canonical in / out)
    Fe<P> r = a;
    for (int i = 0; i < 5; i++) r = fe_dbl(r);
    return r;
}
// the value a fe29 operand stands for, as a saturated 2^256-Montgomery element: v29 = X * 2^261  ->  X * 2^256
template <class P>
static Fe<P> sat_of(const Fe29<P>& a, const Fe29Consts<P>& k) {
    return fe29_to_sat(a, k.two256);
}

template <class C>
static bool check(const char* name) {
    using P = typename C::Base;
    std::mt19937_64 rng(2929);
    const Fe29Consts<P> k = fe29_consts<P>();
    long fails = 0, cases = 0;
    auto expect = [&](bool ok, const char* what) {
        cases++;
        if (!ok && fails++ < 5) printf("  %s: FAIL %s\n", name, what);
    };
    std::vector<Fe<P>> vals;
    vals.push_back(fe_zero<P>());
    vals.push_back(fe_one<P>());
    Fe<P> pm1;
    for (int i = 0; i < 8; i++) pm1.l[i] = P::mod(i);
    pm1.l[0] -= 1;
    vals.push_back(pm1);
    {   // values whose x 32 has every possible top part: k * p / 64 + small
        for (int kk = 1; kk < 64; kk += 3) {
            Fe<P> v = fe_zero<P>();
            v.l[7] = (uint32_t)(((uint64_t)kk << 24) / 1u) & 0x3fffffffu;   // ~ kk * 2^248
            v.l[0] = (uint32_t)kk * 2654435761u;
            vals.push_back(v);
        }
    }
    for (int i = 0; i < 200; i++) vals.push_back(rnd<P>(rng));
    // round trip and products: from_sat_x32(A) stands for A's value in R' form
    for (size_t i = 0; i < vals.size(); i++) {
        const Fe<P> A = vals[i], B = vals[(i * 7 + 3) % vals.size()];
        const Fe29<P> a = fe29_from_sat_x32(A), b = fe29_mul(fe29_from_sat_x32(B), k.one);
        expect(fe_eq(sat_of(fe29_mul(a, k.one), k), A), "round trip");
        {   // the cheap reduction of a table coordinate: same value, below 2 p (top limb), carried limbs
            const Fe29<P> ared = fe29_from_sat_reduced(A);
            expect(fe_eq(sat_of(ared, k), A), "from_sat_reduced value");
            expect(ared.l[8] <= (2u << 22), "from_sat_reduced < 2 p");
            for (int j = 0; j < 8; j++) expect(ared.l[j] < (1u << 29) + 8u, "from_sat_reduced limb bound");
        }
        expect(fe_eq(sat_of(fe29_mul(a, b), k), fe_mul(A, B)), "mul (32 p x 2 p)");
        expect(fe_eq(fe29_to_sat_div32(fe29_mul(a, b)), fe_mul(A, B)), "to_sat_div32 of a product");
        expect(fe_eq(fe29_to_sat_div32(fe29_from_sat_x32(A)), A), "to_sat_div32 of a 32 p value");
        {   // fold: a 32 p and a 96 p value come back below 2 p, same residue, carried
            const Fe29<P> big = fe29_sub<P, 64>(fe29_from_sat_x32(A), b);     // < 32 p + 64 p
            const Fe29<P> f1 = fe29_fold(fe29_carry(fe29_from_sat_x32(A))), f2 = fe29_fold(big);
            expect(fe_eq(fe29_to_sat_div32(f1), A) && f1.l[8] <= (2u << 22), "fold of a 32 p value");
            expect(fe_eq(fe29_to_sat_div32(f2), fe_sub(A, B)) && f2.l[8] <= (2u << 22), "fold of a 96 p value");
            for (int j = 0; j < 8; j++) expect(f2.l[j] < (1u << 29) + 8u, "fold limb bound");
            expect(fe_eq(fe29_to_sat_div32(fe29_add_c(f1, fe29_add_c(f2, f2))), fe_add(A, fe_dbl(fe_sub(A, B)))), "add_c chain");
        }
        expect(fe_eq(fe29_to_sat_div32(fe29_sub<P, 16>(fe29_mul(a, k.one), b)), fe_sub(A, B)), "to_sat_div32 of an 18 p value");
        const Fe29<P> ar = fe29_mul(a, k.one);
        expect(fe_eq(sat_of(fe29_sqr(ar), k), fe_sqr(A)), "sqr");
        expect(fe_eq(sat_of(fe29_mul(fe29_add(ar, b), fe29_add(b, b)), k), fe_mul(fe_add(A, B), fe_dbl(B))), "mul of lazy sums");
        const Fe29<P> d16 = fe29_sub<P, 16>(ar, b), d4 = fe29_sub<P, 4>(b, ar);
        expect(fe_eq(sat_of(d16, k), fe_sub(A, B)), "sub<16>");
        expect(fe_eq(sat_of(d4, k), fe_sub(B, A)), "sub<4>");
        expect(fe_eq(sat_of(fe29_sqr(d16), k), fe_sqr(fe_sub(A, B))), "sqr of an 18 p operand");
        expect(fe_eq(sat_of(fe29_mul(d16, fe29_sub<P, 16>(b, ar)), k), fe_mul(fe_sub(A, B), fe_sub(B, A))), "18 p x 18 p");
        expect(fe_eq(sat_of(fe29_sub3<P, 4>(fe29_sqr(d16), ar, b), k), fe_sub(fe_sub(fe_sqr(fe_sub(A, B)), A), fe_dbl(B))), "sub3");
        expect(fe29_is_zero_mod_p(fe29_sub<P, 16>(ar, fe29_mul(a, k.one)), k), "a - a is a multiple of p");
        for (int j = 0; j < 8; j++) expect(d16.l[j] < (1u << 29) + 8u, "carried limb bound");
    }
    // chains of mixed additions against the saturated code, as group elements
    Affine<P> g;
    {   // a curve point: x = 1, 2, ... until x^3 + b is a square (p = 1 mod 4: Tonelli via exponent is heavy; use the generator)
        // (-1, 2) on Pasta: 4 = -1 + 5
        g.x = fe_neg(fe_one<P>());
        g.y = fe_from_u32<P>(2);
    }
    std::vector<Affine<P>> pts;
    {
        Xyzz<P> w = xyzz_from_affine(g);
        for (int i = 0; i < 40; i++) {
            pts.push_back(xyzz_to_affine(w));
            w = xyzz_dbl(w);
            xyzz_madd(w, g);
        }
    }
    for (int trial = 0; trial < 6; trial++) {
        Xyzz<P> ref = xyzz_identity<P>();
        Xyzz29<P> acc = xyzz29_identity<P>();
        for (int i = 0; i < 300; i++) {
            Affine<P> q = pts[rng() % pts.size()];
            if (trial == 1 && i == 1) q = pts[0], (void)0;
            if (trial >= 1 && i == 0) q = pts[0];
            if (trial == 1 && i == 1) q = pts[0];                         // acc == q: the doubling path
            if (trial == 2 && i == 1) { q = pts[0]; q.y = fe_neg(q.y); }  // acc == -q: back to the identity
            if (rng() & 1) q.y = fe_neg(q.y);
            if (trial == 1 && i == 1) q = pts[0];
            if (trial == 2 && i == 1) { q = pts[0]; q.y = fe_neg(q.y); }
            xyzz_madd(ref, q);
            xyzz29_madd(acc, q, k);
            if (i < 4 || i % 50 == 49) {
                const Affine<P> a = xyzz_to_affine(ref), b = xyzz_to_affine(xyzz29_to_sat(acc, k));
                expect(fe_eq(a.x, b.x) && fe_eq(a.y, b.y), "mixed-addition chain");
                const Xyzz<P> s1 = xyzz29_to_sat(acc, k), s2 = xyzz29_to_sat_fast(acc);
                expect(fe_eq(s1.x, s2.x) && fe_eq(s1.y, s2.y) && fe_eq(s1.zz, s2.zz) && fe_eq(s1.zzz, s2.zzz), "to_sat_fast == to_sat");
            }
        }
        {   // full additions and doublings against the saturated code, incl. acc == q and acc == -q
            Xyzz<P> r2 = xyzz_identity<P>(), other = xyzz_identity<P>();
            Xyzz29<P> a2 = xyzz29_identity<P>(), o2 = xyzz29_identity<P>();
            for (int i = 0; i < 12; i++) {
                xyzz_madd(other, pts[(trial * 5 + i) % pts.size()]);
                xyzz29_madd(o2, pts[(trial * 5 + i) % pts.size()], k);
                xyzz_add(r2, other);
                xyzz29_add(a2, o2);
                if (i % 3 == 2) {
                    r2 = xyzz_dbl(r2);
                    a2 = xyzz29_dbl(a2);
                }
                const Affine<P> a = xyzz_to_affine(r2), b = xyzz_to_affine(xyzz29_to_sat_fast(a2));
                expect(fe_eq(a.x, b.x) && fe_eq(a.y, b.y), "full-addition / doubling chain");
            }
            Xyzz<P> same = r2;
            Xyzz29<P> same29 = a2;
            xyzz_add(same, r2);
            xyzz29_add(same29, a2);                                   // acc == q: the doubling case
            const Affine<P> d1 = xyzz_to_affine(same), d2 = xyzz_to_affine(xyzz29_to_sat_fast(same29));
            expect(fe_eq(d1.x, d2.x) && fe_eq(d1.y, d2.y), "addition of equal points");
            Xyzz29<P> neg = a2;
            neg.y = fe29_sub<P, 8>(fe29_zero<P>(), a2.y);
            neg.y = fe29_fold(neg.y);
            Xyzz29<P> neg_b = neg;
            xyzz29_add(neg, a2);                                      // acc == -q
            expect(neg.id, "addition of opposite points gives the identity");
            // the call-free flavour (k_msm_chunksum): plain, equal and opposite operands
            Xyzz29<P> n1 = a2, n2 = a2;
            xyzz29_add_nocall(n1, o2);
            xyzz29_add(n2, o2);
            const Affine<P> e1 = xyzz_to_affine(xyzz29_to_sat_fast(n1)), e2 = xyzz_to_affine(xyzz29_to_sat_fast(n2));
            expect(fe_eq(e1.x, e2.x) && fe_eq(e1.y, e2.y), "call-free addition == addition");
            Xyzz29<P> n3 = a2;
            xyzz29_add_nocall(n3, a2);
            const Affine<P> e3 = xyzz_to_affine(xyzz29_to_sat_fast(n3));
            expect(fe_eq(d1.x, e3.x) && fe_eq(d1.y, e3.y), "call-free addition of equal points");
            xyzz29_add_nocall(neg_b, a2);
            expect(neg_b.id, "call-free addition of opposite points gives the identity");
            Xyzz29<P> n4 = xyzz29_identity<P>();
            xyzz29_add_nocall(n4, a2);
            xyzz29_add_nocall(n4, xyzz29_identity<P>());
            const Affine<P> e4 = xyzz_to_affine(xyzz29_to_sat_fast(n4)), e5 = xyzz_to_affine(xyzz29_to_sat_fast(a2));
            expect(fe_eq(e4.x, e5.x) && fe_eq(e4.y, e5.y), "call-free addition with the identity on either side");
            for (int j = 0; j < 8; j++) expect(a2.x.l[j] < (1u << 29) + 8u && a2.y.l[j] < (1u << 29) + 8u, "sum limb bound");
            expect(a2.x.l[8] < (12u << 22) && a2.y.l[8] < (8u << 22) && a2.zz.l[8] <= (2u << 22), "sum value bound");
        }
        // invariants of the representation
        if (!acc.id) {
            for (int j = 0; j < 8; j++) expect(acc.x.l[j] < (1u << 29) + 8u && acc.y.l[j] < (1u << 29) + 8u, "accumulator limb bound");
            expect(acc.x.l[8] < (12u << 22) && acc.y.l[8] < (8u << 22) && acc.zz.l[8] <= (2u << 22) && acc.zzz.l[8] <= (2u << 22), "accumulator value bound");
        }
    }
    printf("%s: %ld cases, %ld failures -> %s\n", name, cases, fails, fails ? "FAIL" : "OK");
    return fails == 0;
}

int main() {
    bool ok = check<PallasCurve>("fe29 over Fp (Pallas coordinates)");
    ok = check<VestaCurve>("fe29 over Fq (Vesta coordinates)") && ok;
    return ok ? 0 : 1;
}
