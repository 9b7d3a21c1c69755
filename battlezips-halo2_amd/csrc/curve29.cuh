// The bucket accumulator of k_msm_accumulate in unsaturated limbs (csrc/fe29.cuh): XYZZ mixed addition with the running sum
// held as 4 x 9 limbs of 29 bits in R' = 2^261 Montgomery form, NOT canonical -- invariants after every operation:
//     x < 11.5 p, y < 3.7 p (anything below 8 p is accepted), zz < 2 p, zzz < 2 p, every limb carried (< 2^29 + 8), identity kept as a flag.
// The table points arrive in the saturated 2^256 form every other kernel uses and are re-sliced on the fly (x 2^5 = the R' form,
// ~25 shifts per coordinate); a finished bucket goes back through one product per coordinate (fe29_to_sat).  Same group element
// as curve.cuh's xyzz_madd computes, hence the same MSM result and the same proof bytes.
#pragma once
#include "curve.cuh"
#include "fe29.cuh"

namespace bzh {

template <class P>
struct Xyzz29 {
    Fe29<P> x, y, zz, zzz;
    bool id;
};
// constants of the representation, built once per thread from compile-time words (host: at first use)
template <class P>
struct Fe29Consts {
    Fe29<P> one;      // 2^261 mod p: the R' form of 1, and the factor that reduces a re-sliced table coordinate below 2 p
    Fe29<P> two256;   // 2^256 mod p: R' form -> saturated form
};
template <class P>
BZH_HD Fe29Consts<P> fe29_consts() {
    Fe29Consts<P> c;
    static_for<9>([&](auto i) {
        constexpr uint32_t one_i = fe29_pow2_limb<P, 261>(decltype(i)::value), t_i = fe29_pow2_limb<P, 256>(decltype(i)::value);
        c.one.l[decltype(i)::value] = one_i;
        c.two256.l[decltype(i)::value] = t_i;
    });
    return c;
}

template <class P>
BZH_HD Xyzz29<P> xyzz29_identity() {
    Xyzz29<P> r;
    r.x = fe29_zero<P>();
    r.y = r.x;
    r.zz = r.x;
    r.zzz = r.x;
    r.id = true;
    return r;
}
// is the carried value a multiple of p?  (exact; rare path only)
template <class P>
BZH_HD bool fe29_is_zero_mod_p(const Fe29<P>& v, const Fe29Consts<P>& k) {
    return fe_is_zero(fe29_pack_canonical(fe29_mul(v, k.one)));
}
template <class P>
BZH_HD Xyzz<P> xyzz29_to_sat(const Xyzz29<P>& a, const Fe29Consts<P>& k) {
    if (a.id) return xyzz_identity<P>();
    Xyzz<P> r;
    r.x = fe29_to_sat(a.x, k.two256);
    r.y = fe29_to_sat(a.y, k.two256);
    r.zz = fe29_to_sat(a.zz, k.two256);
    r.zzz = fe29_to_sat(a.zzz, k.two256);
    return r;
}
// the same without products (fe29_to_sat_div32), branch-free: what the accumulate kernel's hand-over pass runs
template <class P>
BZH_HD Xyzz<P> xyzz29_to_sat_fast(const Xyzz29<P>& a) {
    Xyzz<P> r;
    r.x = fe29_to_sat_div32(a.x);
    r.y = fe29_to_sat_div32(a.y);
    r.zz = fe29_to_sat_div32(a.zz);
    r.zzz = fe29_to_sat_div32(a.zzz);
    const uint32_t keep = a.id ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.x.l[i] &= keep;
        r.y.l[i] &= keep;
        r.zz.l[i] &= keep;
        r.zzz.l[i] &= keep;
    }
    return r;
}
template <class P>
BZH_HD Xyzz29<P> xyzz29_from_sat(const Xyzz<P>& a, const Fe29Consts<P>& k) {
    if (xyzz_is_id(a)) return xyzz29_identity<P>();
    Xyzz29<P> r;
    r.x = fe29_mul(fe29_from_sat_x32(a.x), k.one);
    r.y = fe29_mul(fe29_from_sat_x32(a.y), k.one);
    r.zz = fe29_mul(fe29_from_sat_x32(a.zz), k.one);
    r.zzz = fe29_mul(fe29_from_sat_x32(a.zzz), k.one);
    r.id = false;
    return r;
}

// the special cases of the mixed addition, out of line: reached when P = u2 - x1 shows a small multiple of p in its low limb
// (14 chances in 2^29 for unrelated points).  Behind a call, so that the compiler cannot hoist the exact test -- a product and a
// canonical reduction -- into the main path; everything by VALUE: an accumulator whose address is passed to a function lives in
// scratch memory, and scratch accesses share the vector-memory counter with the table gathers (every wait for a spilled limb
// then also waits for the point prefetched for the next iteration: 1.8 x on the whole kernel, measured).
template <class P>
struct Madd29Special {
    Xyzz29<P> v;
    bool handled;
};
template <class P>
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__
#else
inline
#endif
Madd29Special<P> xyzz29_madd_special(const Fe29<P> qx, const Fe29<P> qy, const Fe29<P> pp_, const Fe29<P> r) {
    const Fe29Consts<P> k = fe29_consts<P>();
    Affine<P> q;   // the saturated point back from its limbs (here, not at the call: the compiler would hoist it into the main path)
    q.x = fe29_to_sat_div32(qx);
    q.y = fe29_to_sat_div32(qy);
    Madd29Special<P> out;
    out.v = xyzz29_identity<P>();                                                    // acc == -q
    out.handled = fe29_is_zero_mod_p(pp_, k);
    if (out.handled && fe29_is_zero_mod_p(r, k)) out.v = xyzz29_from_sat(xyzz_dbl_affine(q), k);   // acc == q: doubling, through the saturated code
    return out;
}

// acc += (qx, qy), the point given in the R' form below 2 p with carried limbs (a table's fe29 copy, or fe29_from_sat_reduced of
// its saturated coordinates), not the identity.  madd-2008-s, 8 M + 2 S, as curve.cuh's xyzz_madd.
template <class P>
BZH_HD void xyzz29_madd_q29(Xyzz29<P>& acc, const Fe29<P>& qx, const Fe29<P>& qy, const Fe29Consts<P>& k) {
    if (acc.id) {   // first point of a bucket: lanes reach this on different iterations, so it has to be free
        acc.x = qx;
        acc.y = qy;
        acc.zz = k.one;
        acc.zzz = k.one;
        acc.id = false;
        return;
    }
    const Fe29<P> u2 = fe29_mul(qx, acc.zz);                 // < 2 p
    const Fe29<P> s2 = fe29_mul(qy, acc.zzz);
    const Fe29<P> pp_ = fe29_sub<P, 16>(u2, acc.x);          // in (4.5 p, 18 p); = k p exactly when the x coordinates agree
    const Fe29<P> r = fe29_sub<P, 16>(s2, acc.y);
    // p = 1 mod 2^29 and the carried low limb is exact: a multiple k p of p, 4 < k <= 18, shows its k there
    if (pp_.l[0] - 5u <= 13u) {
        const Madd29Special<P> sp = xyzz29_madd_special<P>(qx, qy, pp_, r);
        if (sp.handled) {
            acc = sp.v;
            return;
        }
    }
    const Fe29<P> pp = fe29_sqr(pp_);                        // < 3.5 p
    const Fe29<P> ppp = fe29_mul(pp_, pp);                   // < 2 p
    const Fe29<P> qq = fe29_mul(acc.x, pp);                  // < 2 p
    const Fe29<P> x3 = fe29_sub3<P, 4>(fe29_sqr(r), ppp, qq);   // R^2 - PPP - 2 Q + 8 p < 11.5 p
    // Y3 = R (Q - X3) - Y1 PPP as ONE reduction: (8 p - Y1) PPP differs from -Y1 PPP by a multiple of p, so both products go into
    // the same columns (fe29_dot2; limb products 2.9e17 + 8.7e17 of the 1.8e18 a column holds; 324 p^2 + 16 p^2: below 3.7 p)
    acc.y = fe29_dot2(r, fe29_sub<P, 16>(qq, x3), fe29_sub_lazy<P, 8, 1>(fe29_zero<P>(), acc.y), ppp);
    acc.x = x3;
    acc.zz = fe29_mul(acc.zz, pp);
    acc.zzz = fe29_mul(acc.zzz, ppp);
}
// the same for a point in the saturated form
template <class P>
BZH_HD void xyzz29_madd(Xyzz29<P>& acc, const Affine<P>& q, const Fe29Consts<P>& k) {
    xyzz29_madd_q29(acc, fe29_from_sat_reduced(q.x), fe29_from_sat_reduced(q.y), k);
}

// ---- full additions and doublings in unsaturated limbs (the bucket reductions) -----------------------------------------
// invariants in and out: x < 11.5 p, y < 3.7 p (7.5 p accepted), zz, zzz < 2 p, limbs carried
// 2 p: dbl-2008-s-1 (a = 0).  y and x are folded below 2 p first: (2 y)^2 and x^2 would leave the product's input range.
template <class P>
BZH_HD Xyzz29<P> xyzz29_dbl(const Xyzz29<P>& p) {
    if (p.id) return p;
    const Fe29<P> yf = fe29_fold(p.y), xf = fe29_fold(p.x);
    const Fe29<P> u = fe29_add(yf, yf);                        // < 4 p, limbs < 2^30 + 16: one factor of a product
    const Fe29<P> v = fe29_sqr(u);
    const Fe29<P> w = fe29_mul(u, v);
    const Fe29<P> s = fe29_mul(xf, v);
    const Fe29<P> xx = fe29_sqr(xf);
    const Fe29<P> m = fe29_add_c(fe29_add_c(xx, xx), xx);      // < 6 p
    Xyzz29<P> r;
    r.x = fe29_sub3<P, 4>(fe29_sqr(m), fe29_zero<P>(), s);     // M^2 - 2 S + 8 p < 10 p
    r.y = fe29_dot2(m, fe29_sub<P, 16>(s, r.x), fe29_sub_lazy<P, 4, 1>(fe29_zero<P>(), yf), w);   // M (S - X3) - W Y, one reduction
    r.zz = fe29_mul(v, p.zz);
    r.zzz = fe29_mul(w, p.zzz);
    r.id = false;
    return r;
}
// acc += q, both XYZZ.  add-2008-s, 12 M + 2 S.  The equal-x cases go through the saturated code (rare).
template <class P>
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__
#else
__host__ __device__ inline
#endif
Madd29Special<P> xyzz29_add_special(const Xyzz29<P> acc, const Fe29<P> pp_, const Fe29<P> r) {
    const Fe29Consts<P> k = fe29_consts<P>();
    Madd29Special<P> out;
    out.v = xyzz29_identity<P>();                              // acc == -q
    out.handled = fe29_is_zero_mod_p(pp_, k);
    if (out.handled && fe29_is_zero_mod_p(r, k)) out.v = xyzz29_dbl(acc);   // acc == q
    return out;
}
template <class P>
BZH_HD void xyzz29_add(Xyzz29<P>& acc, const Xyzz29<P>& q) {
    if (q.id) return;
    if (acc.id) {
        acc = q;
        return;
    }
    const Fe29<P> u1 = fe29_mul(acc.x, q.zz), u2 = fe29_mul(q.x, acc.zz);
    const Fe29<P> s1 = fe29_mul(acc.y, q.zzz), s2 = fe29_mul(q.y, acc.zzz);
    const Fe29<P> pp_ = fe29_sub<P, 4>(u2, u1), r = fe29_sub<P, 4>(s2, s1);   // (2 p, 6 p): k p with 2 < k < 6 when the x agree
    if (pp_.l[0] - 3u <= 2u) {
        const Madd29Special<P> sp = xyzz29_add_special<P>(acc, pp_, r);
        if (sp.handled) {
            acc = sp.v;
            return;
        }
    }
    const Fe29<P> pp = fe29_sqr(pp_), ppp = fe29_mul(pp_, pp), qq = fe29_mul(u1, pp);
    const Fe29<P> x3 = fe29_sub3<P, 4>(fe29_sqr(r), ppp, qq);                  // < 10 p
    acc.y = fe29_dot2(r, fe29_sub<P, 16>(qq, x3), fe29_sub_lazy<P, 4, 1>(fe29_zero<P>(), s1), ppp);   // R (Q - X3) - S1 PPP, one reduction
    acc.x = x3;
    acc.zz = fe29_mul(fe29_mul(acc.zz, q.zz), pp);
    acc.zzz = fe29_mul(fe29_mul(acc.zzz, q.zzz), ppp);
}

// acc += q with the equal-x cases INLINE through the saturated addition (no call: a callee's registers count towards its
// kernel's, and xyzz29_add_special's 250 would halve k_msm_chunksum's occupancy).  Whatever the filter lets through -- equal
// points, opposite points, or the 3-in-2^29 coincidence -- takes the saturated path, which handles every case; the limbs are
// made opaque inside the branch so that none of its work is hoisted above the test.
template <class P>
BZH_HD void xyzz29_add_nocall(Xyzz29<P>& acc, const Xyzz29<P>& q) {
    if (q.id) return;
    if (acc.id) {
        acc = q;
        return;
    }
    const Fe29<P> u1 = fe29_mul(acc.x, q.zz), u2 = fe29_mul(q.x, acc.zz);
    const Fe29<P> s1 = fe29_mul(acc.y, q.zzz), s2 = fe29_mul(q.y, acc.zzz);
    const Fe29<P> pp_ = fe29_sub<P, 4>(u2, u1), r = fe29_sub<P, 4>(s2, s1);
    if (__builtin_expect(pp_.l[0] - 3u <= 2u, 0)) {
        Xyzz29<P> a = acc, b = q;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < 9; i++) {
            asm volatile("" : "+v"(a.x.l[i]), "+v"(a.y.l[i]), "+v"(a.zz.l[i]), "+v"(a.zzz.l[i]));
            asm volatile("" : "+v"(b.x.l[i]), "+v"(b.y.l[i]), "+v"(b.zz.l[i]), "+v"(b.zzz.l[i]));
        }
#endif
        Xyzz<P> sa = xyzz29_to_sat_fast(a);
        xyzz_add_inl(sa, xyzz29_to_sat_fast(b));
        acc.x = fe29_from_sat_reduced(sa.x), acc.y = fe29_from_sat_reduced(sa.y);
        acc.zz = fe29_from_sat_reduced(sa.zz), acc.zzz = fe29_from_sat_reduced(sa.zzz);
        acc.id = xyzz_is_id(sa);
        return;
    }
    const Fe29<P> pp = fe29_sqr(pp_), ppp = fe29_mul(pp_, pp), qq = fe29_mul(u1, pp);
    const Fe29<P> x3 = fe29_sub3<P, 4>(fe29_sqr(r), ppp, qq);
    acc.y = fe29_dot2(r, fe29_sub<P, 16>(qq, x3), fe29_sub_lazy<P, 4, 1>(fe29_zero<P>(), s1), ppp);   // R (Q - X3) - S1 PPP, one reduction
    acc.x = x3;
    acc.zz = fe29_mul(fe29_mul(acc.zz, q.zz), pp);
    acc.zzz = fe29_mul(fe29_mul(acc.zzz, q.zzz), ppp);
}

// ---- the same addition with the FOUR lanes of a quad on it (latency mode; see curve.cuh's xyzz_add_quad): every lane holds the same
//      acc and q, multiplies ONE of the up to four independent products of each of the four dependency levels and receives the
//      others by DPP quad broadcasts of the nine limbs.  4 products of 188 instructions per lane instead of 4 of 297.
template <class P>
__device__ __forceinline__ Fe29<P> fe29_quad_bcast(const Fe29<P>& v, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    Fe29<P> o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        o.l[i] = k == 0   ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], 0x00, 0xf, 0xf, true)
                 : k == 1 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], 0x55, 0xf, 0xf, true)
                 : k == 2 ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], 0xaa, 0xf, 0xf, true)
                          : (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], 0xff, 0xf, 0xf, true);
    }
    return o;
#else
    return v;   // (host pass of the compiler: never executed)
#endif
}
template <class P>
__device__ __forceinline__ Fe29<P> fe29_sel4(const QuadMasks& q, const Fe29<P>& a0, const Fe29<P>& a1, const Fe29<P>& a2, const Fe29<P>& a3) {
    Fe29<P> o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = (a0.l[i] & q.m0) | (a1.l[i] & q.m1) | (a2.l[i] & q.m2) | (a3.l[i] & q.m3);
    return o;
}
template <class P>
__device__ __forceinline__ void xyzz29_add_quad(Xyzz29<P>& acc, const Xyzz29<P>& q, int ql) {
    if (q.id) return;                     // (the same data in all four lanes: uniform inside the quad)
    if (acc.id) {
        acc = q;
        return;
    }
    const QuadMasks qm = quad_masks(ql);
    // level 1: u1 = x1 zz2, u2 = x2 zz1, s1 = y1 zzz2, s2 = y2 zzz1
    Fe29<P> t = fe29_mul(fe29_sel4(qm, acc.x, q.x, acc.y, q.y), fe29_sel4(qm, q.zz, acc.zz, q.zzz, acc.zzz));
    const Fe29<P> u1 = fe29_quad_bcast(t, 0), u2 = fe29_quad_bcast(t, 1), s1 = fe29_quad_bcast(t, 2), s2 = fe29_quad_bcast(t, 3);
    const Fe29<P> pp_ = fe29_sub<P, 4>(u2, u1), r = fe29_sub<P, 4>(s2, s1);
    if (pp_.l[0] - 3u <= 2u) {            // same x (rare): all four lanes take the plain path
        const Madd29Special<P> sp = xyzz29_add_special<P>(acc, pp_, r);
        if (sp.handled) {
            acc = sp.v;
            return;
        }
    }
    // level 2: pp = P^2, rr = R^2, zz12 = zz1 zz2, zzz12 = zzz1 zzz2
    t = fe29_mul(fe29_sel4(qm, pp_, r, acc.zz, acc.zzz), fe29_sel4(qm, pp_, r, q.zz, q.zzz));
    const Fe29<P> pp = fe29_quad_bcast(t, 0), rr = fe29_quad_bcast(t, 1), zz12 = fe29_quad_bcast(t, 2), zzz12 = fe29_quad_bcast(t, 3);
    // level 3: ppp = P pp, qq = u1 pp, zz3 = zz12 pp   (lane 3 repeats lane 0's product)
    t = fe29_mul(fe29_sel4(qm, pp_, u1, zz12, pp_), pp);
    const Fe29<P> ppp = fe29_quad_bcast(t, 0), qq = fe29_quad_bcast(t, 1), zz3 = fe29_quad_bcast(t, 2);
    const Fe29<P> x3 = fe29_sub3<P, 4>(rr, ppp, qq);
    // level 4: a = R (qq - x3), b = s1 ppp, zzz3 = zzz12 ppp
    t = fe29_mul(fe29_sel4(qm, r, s1, zzz12, s1), fe29_sel4(qm, fe29_sub<P, 16>(qq, x3), ppp, ppp, ppp));
    const Fe29<P> ya = fe29_quad_bcast(t, 0), yb = fe29_quad_bcast(t, 1), zzz3 = fe29_quad_bcast(t, 2);
    acc.x = x3;
    acc.y = fe29_sub<P, 4>(ya, yb);
    acc.zz = zz3;
    acc.zzz = zzz3;
}
// a saturated bucket -> Xyzz29 by the quad: lane ql converts coordinate ql, the four results are broadcast
template <class P>
__device__ __forceinline__ Xyzz29<P> xyzz29_from_sat_quad(const Xyzz<P>& v, int ql) {
    Xyzz29<P> r;
    r.id = xyzz_is_id(v);
    const QuadMasks qm = quad_masks(ql);
    const Fe29<P> c = fe29_from_sat_reduced(fe_sel4(qm, v.x, v.y, v.zz, v.zzz));
    r.x = fe29_quad_bcast(c, 0);
    r.y = fe29_quad_bcast(c, 1);
    r.zz = fe29_quad_bcast(c, 2);
    r.zzz = fe29_quad_bcast(c, 3);
    return r;
}
template <class P>
__device__ __forceinline__ Xyzz29<P> xyzz29_shfl_down(const Xyzz29<P>& v, int d) {
    Xyzz29<P> o;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int k = 0; k < 9; k++) {
        o.x.l[k] = (uint32_t)__shfl_down((int)v.x.l[k], d, 64);
        o.y.l[k] = (uint32_t)__shfl_down((int)v.y.l[k], d, 64);
        o.zz.l[k] = (uint32_t)__shfl_down((int)v.zz.l[k], d, 64);
        o.zzz.l[k] = (uint32_t)__shfl_down((int)v.zzz.l[k], d, 64);
    }
    o.id = __shfl_down((int)v.id, d, 64) != 0;
#else
    o = v;
#endif
    return o;
}

}  // namespace bzh
