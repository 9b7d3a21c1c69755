// Short-Weierstrass a = 0 curves (Vesta, Pallas, BN254 G1) in XYZZ coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity <=> ZZ = 0).  XYZZ keeps the
// bucket accumulator's mixed addition at 8M + 2S, the cheapest complete-enough
// form for Pippenger's inner loop.
//
// Device counterpart of pasta_curves 0.4.1 group arithmetic (UPSTREAM,
// un-vendored; Cargo.lock:567-570) as used by halo2_proofs::arithmetic::
// best_multiexp; reference call sites: benches/shot.rs:68 (create_proof),
// src/utils/pedersen.rs:27 (v*m + r*t).
#pragma once
#include "field.cuh"

namespace bzh {

struct VestaCurve {  // y^2 = x^3 + 5 over Fq, order p ; commitments of the IPA prover
    using Base = FqParams;
    static constexpr int id = 0;
    static constexpr uint32_t b = 5;
};
struct PallasCurve {  // y^2 = x^3 + 5 over Fp, order q ; Pedersen commitment (src/utils/pedersen.rs)
    using Base = FpParams;
    static constexpr int id = 1;
    static constexpr uint32_t b = 5;
};
struct Bn254Curve {  // y^2 = x^3 + 3 (config 5 microbench only; no reference)
    using Base = BnFqParams;
    static constexpr int id = 2;
    static constexpr uint32_t b = 3;
};

template <class P>
struct Affine {  // (0,0) <=> identity
    Fe<P> x, y;
};
template <class P>
struct Xyzz {
    Fe<P> x, y, zz, zzz;
};

template <class P>
BZH_HD bool aff_is_id(const Affine<P>& a) {
    return fe_is_zero(a.x) && fe_is_zero(a.y);
}
template <class P>
BZH_HD Xyzz<P> xyzz_identity() {
    Xyzz<P> r;
    r.x = fe_zero<P>();
    r.y = fe_zero<P>();
    r.zz = fe_zero<P>();
    r.zzz = fe_zero<P>();
    return r;
}
template <class P>
BZH_HD bool xyzz_is_id(const Xyzz<P>& p) {
    return fe_is_zero(p.zz);
}
template <class P>
BZH_HD Xyzz<P> xyzz_from_affine(const Affine<P>& a) {
    Xyzz<P> r;
    if (aff_is_id(a)) return xyzz_identity<P>();
    r.x = a.x;
    r.y = a.y;
    r.zz = fe_one<P>();
    r.zzz = fe_one<P>();
    return r;
}

// The rarely-executed group operations stay out of line on the device (the field multiply is
// inlined into them): the bucket loop's hot path is then xyzz_madd's main branch alone.
#if defined(__HIP_DEVICE_COMPILE__)
#define BZH_COLD __host__ __device__ __noinline__
#else
#define BZH_COLD BZH_HD
#endif

// dbl-2008-s-1 (a = 0): 6M + 3S... written as 7M + 2S with shared products.
// xyzz_dbl is out of line on the device (BZH_COLD); xyzz_dbl_inl is the same code inlined for the kernels whose time IS
// a chain of such operations (bucket reduction, final sums): a call passes its 32 + 32 limb arguments through scratch
// memory, ~700 B per lane of private-segment traffic per addition.
template <class P>
BZH_HD Xyzz<P> xyzz_dbl_inl(const Xyzz<P>& p) {
    if (xyzz_is_id(p)) return p;
    Fe<P> u = fe_dbl(p.y);
    Fe<P> v = fe_sqr(u);
    Fe<P> w = fe_mul(u, v);
    Fe<P> s = fe_mul(p.x, v);
    Fe<P> xx = fe_sqr(p.x);
    Fe<P> m = fe_add(fe_dbl(xx), xx);
    Xyzz<P> r;
    r.x = fe_sub(fe_sqr(m), fe_dbl(s));
    r.y = fe_sub(fe_mul(m, fe_sub(s, r.x)), fe_mul(w, p.y));
    r.zz = fe_mul(v, p.zz);
    r.zzz = fe_mul(w, p.zzz);
    return r;
}
template <class P>
BZH_COLD Xyzz<P> xyzz_dbl(const Xyzz<P> p) {
    return xyzz_dbl_inl(p);
}
// doubling of an affine point into XYZZ (mdbl-2008-s-1)
template <class P>
BZH_COLD Xyzz<P> xyzz_dbl_affine(const Affine<P> a) {
    Fe<P> u = fe_dbl(a.y);
    Fe<P> v = fe_sqr(u);
    Fe<P> w = fe_mul(u, v);
    Fe<P> s = fe_mul(a.x, v);
    Fe<P> xx = fe_sqr(a.x);
    Fe<P> m = fe_add(fe_dbl(xx), xx);
    Xyzz<P> r;
    r.x = fe_sub(fe_sqr(m), fe_dbl(s));
    r.y = fe_sub(fe_mul(m, fe_sub(s, r.x)), fe_mul(w, a.y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// acc += (qx, qy) affine, not the identity.  madd-2008-s: 8M + 2S.
// Handles acc == identity, acc == q (doubling) and acc == -q (identity).
template <class P>
BZH_HD void xyzz_madd(Xyzz<P>& acc, const Affine<P>& q) {
    if (xyzz_is_id(acc)) {
        acc.x = q.x;
        acc.y = q.y;
        acc.zz = fe_one<P>();
        acc.zzz = fe_one<P>();
        return;
    }
    Fe<P> u2 = fe_mul(q.x, acc.zz);
    Fe<P> s2 = fe_mul(q.y, acc.zzz);
    Fe<P> pp_ = fe_sub(u2, acc.x);
    Fe<P> r = fe_sub(s2, acc.y);
    if (fe_is_zero(pp_)) {
        if (fe_is_zero(r)) {
            acc = xyzz_dbl_affine(q);
        } else {
            acc = xyzz_identity<P>();
        }
        return;
    }
    Fe<P> pp = fe_sqr(pp_);
    Fe<P> ppp = fe_mul(pp_, pp);
    Fe<P> qq = fe_mul(acc.x, pp);
    Fe<P> x3 = fe_sub(fe_sub(fe_sqr(r), ppp), fe_dbl(qq));
    Fe<P> y3 = fe_sub(fe_mul(r, fe_sub(qq, x3)), fe_mul(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul(acc.zz, pp);
    acc.zzz = fe_mul(acc.zzz, ppp);
}

// acc += q (both XYZZ).  add-2008-s: 12M + 2S, all special cases handled.
template <class P>
BZH_COLD Xyzz<P> xyzz_add_impl(Xyzz<P> acc, const Xyzz<P> q);
template <class P>
BZH_HD void xyzz_add(Xyzz<P>& acc, const Xyzz<P>& q) {
    if (xyzz_is_id(q)) return;
    if (xyzz_is_id(acc)) {
        acc = q;
        return;
    }
    acc = xyzz_add_impl(acc, q);
}
template <class P>
BZH_HD Xyzz<P> xyzz_add_impl_inl(Xyzz<P> acc, const Xyzz<P>& q) {
    Fe<P> u1 = fe_mul(acc.x, q.zz);
    Fe<P> u2 = fe_mul(q.x, acc.zz);
    Fe<P> s1 = fe_mul(acc.y, q.zzz);
    Fe<P> s2 = fe_mul(q.y, acc.zzz);
    Fe<P> pp_ = fe_sub(u2, u1);
    Fe<P> r = fe_sub(s2, s1);
    if (fe_is_zero(pp_)) {
        if (fe_is_zero(r)) return xyzz_dbl(acc);   // never taken by sums of distinct buckets; stays out of line
        return xyzz_identity<P>();
    }
    Fe<P> pp = fe_sqr(pp_);
    Fe<P> ppp = fe_mul(pp_, pp);
    Fe<P> qq = fe_mul(u1, pp);
    Fe<P> x3 = fe_sub(fe_sub(fe_sqr(r), ppp), fe_dbl(qq));
    Fe<P> y3 = fe_sub(fe_mul(r, fe_sub(qq, x3)), fe_mul(s1, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul(fe_mul(acc.zz, q.zz), pp);
    acc.zzz = fe_mul(fe_mul(acc.zzz, q.zzz), ppp);
    return acc;
}
template <class P>
BZH_COLD Xyzz<P> xyzz_add_impl(Xyzz<P> acc, const Xyzz<P> q) {
    return xyzz_add_impl_inl(acc, q);
}
// acc += q, inlined (see xyzz_dbl_inl)
template <class P>
BZH_HD void xyzz_add_inl(Xyzz<P>& acc, const Xyzz<P>& q) {
    if (xyzz_is_id(q)) return;
    if (xyzz_is_id(acc)) {
        acc = q;
        return;
    }
    acc = xyzz_add_impl_inl(acc, q);
}

// ---------------------------------------------------------------------------
// acc += q with the FOUR lanes of a quad working on ONE addition (latency mode).  A wave64 issues one VALU instruction per
// four cycles whatever its lanes hold, so a dependent chain of additions on one wave costs ~3 700 issue slots per addition
// (6-8 us): the bucket reductions and final sums of a single proof are such chains.  Here every lane of a quad holds the same
// acc and q, computes ONE of the (up to four) independent products of each of the addition's four dependency levels and gets
// the others by quad broadcasts (DPP quad_perm, no LDS): 4 multiplications + ~190 selects + ~120 broadcasts per lane instead
// of 14 multiplications -- ~2.5 x shorter chains for a quarter of the lanes.  All four lanes return the same result.
// ---------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ Fe<P> fe_quad_bcast(const Fe<P>& v, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    Fe<P> o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
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
// operand of quad lane ql out of four candidates, by lane masks (m[k] = all ones in lane k of the quad): a chain of `?:` on the
// lane index is turned into a table in scratch memory by the compiler -- 64 scratch round trips per addition
struct QuadMasks {
    uint32_t m0, m1, m2, m3;
};
__device__ __forceinline__ QuadMasks quad_masks(int ql) {
    QuadMasks q;
    q.m0 = ql == 0 ? 0xffffffffu : 0u;
    q.m1 = ql == 1 ? 0xffffffffu : 0u;
    q.m2 = ql == 2 ? 0xffffffffu : 0u;
    q.m3 = ql == 3 ? 0xffffffffu : 0u;
    return q;
}
template <class P>
__device__ __forceinline__ Fe<P> fe_sel4(const QuadMasks& q, const Fe<P>& a0, const Fe<P>& a1, const Fe<P>& a2, const Fe<P>& a3) {
    Fe<P> o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.l[i] = (a0.l[i] & q.m0) | (a1.l[i] & q.m1) | (a2.l[i] & q.m2) | (a3.l[i] & q.m3);
    return o;
}
template <class P>
__device__ __forceinline__ void xyzz_add_quad(Xyzz<P>& acc, const Xyzz<P>& q, int ql) {
    if (xyzz_is_id(q)) return;            // (the same data in all four lanes: uniform inside the quad)
    if (xyzz_is_id(acc)) {
        acc = q;
        return;
    }
    const QuadMasks qm = quad_masks(ql);
    // level 1: u1 = x1 zz2, u2 = x2 zz1, s1 = y1 zzz2, s2 = y2 zzz1
    Fe<P> t = fe_mul(fe_sel4(qm, acc.x, q.x, acc.y, q.y), fe_sel4(qm, q.zz, acc.zz, q.zzz, acc.zzz));
    const Fe<P> u1 = fe_quad_bcast(t, 0), u2 = fe_quad_bcast(t, 1), s1 = fe_quad_bcast(t, 2), s2 = fe_quad_bcast(t, 3);
    const Fe<P> pp_ = fe_sub(u2, u1), r = fe_sub(s2, s1);
    if (fe_is_zero(pp_)) {                // same x: doubling or the inverse -- rare, all four lanes take the plain path
        if (fe_is_zero(r)) acc = xyzz_dbl(acc);
        else acc = xyzz_identity<P>();
        return;
    }
    // level 2: pp = P^2, rr = R^2, zz12 = zz1 zz2, zzz12 = zzz1 zzz2
    t = fe_mul(fe_sel4(qm, pp_, r, acc.zz, acc.zzz), fe_sel4(qm, pp_, r, q.zz, q.zzz));
    const Fe<P> pp = fe_quad_bcast(t, 0), rr = fe_quad_bcast(t, 1), zz12 = fe_quad_bcast(t, 2), zzz12 = fe_quad_bcast(t, 3);
    // level 3: ppp = P pp, qq = u1 pp, zz3 = zz12 pp   (lane 3 repeats lane 0's product)
    t = fe_mul(fe_sel4(qm, pp_, u1, zz12, pp_), pp);
    const Fe<P> ppp = fe_quad_bcast(t, 0), qq = fe_quad_bcast(t, 1), zz3 = fe_quad_bcast(t, 2);
    const Fe<P> x3 = fe_sub(fe_sub(rr, ppp), fe_dbl(qq));
    // level 4: a = R (qq - x3), b = s1 ppp, zzz3 = zzz12 ppp
    t = fe_mul(fe_sel4(qm, r, s1, zzz12, s1), fe_sel4(qm, fe_sub(qq, x3), ppp, ppp, ppp));
    const Fe<P> ya = fe_quad_bcast(t, 0), yb = fe_quad_bcast(t, 1), zzz3 = fe_quad_bcast(t, 2);
    acc.x = x3;
    acc.y = fe_sub(ya, yb);
    acc.zz = zz3;
    acc.zzz = zzz3;
}

// XYZZ -> Jacobian (X:Y:Z) with x = X/Z^2, y = Y/Z^3: Z = ZZZ/ZZ would need an
// inversion; instead scale: X' = X*ZZ... use (X*ZZZ^2... ) -- simplest exact
// map without inversion: Z = ZZZ * ZZ^-1 is avoided by X' = X * ZZ, Y' = Y * ZZZ,
// Z' = ZZ  since x = X/ZZ = X*ZZ/ZZ^2 and y = Y/ZZZ = Y*ZZZ/ZZZ^2 = Y*ZZZ/ZZ^3.
template <class P>
BZH_HD void xyzz_to_jacobian(const Xyzz<P>& p, Fe<P>& X, Fe<P>& Y, Fe<P>& Z) {
    if (xyzz_is_id(p)) {
        X = fe_zero<P>();
        Y = fe_zero<P>();
        Z = fe_zero<P>();
        return;
    }
    X = fe_mul(p.x, p.zz);
    Y = fe_mul(p.y, p.zzz);
    Z = p.zz;
}
template <class P>
BZH_HD Affine<P> xyzz_to_affine(const Xyzz<P>& p) {
    Affine<P> r;
    if (xyzz_is_id(p)) {
        r.x = fe_zero<P>();
        r.y = fe_zero<P>();
        return r;
    }
    // 1/ZZZ, then 1/ZZ = ZZZ^-2 * ZZ^2 ... cheaper: i = (ZZ*ZZZ)^-1; 1/ZZ = i*ZZZ; 1/ZZZ = i*ZZ
    Fe<P> i = fe_inv(fe_mul(p.zz, p.zzz));
    r.x = fe_mul(p.x, fe_mul(i, p.zzz));
    r.y = fe_mul(p.y, fe_mul(i, p.zz));
    return r;
}

}  // namespace bzh
