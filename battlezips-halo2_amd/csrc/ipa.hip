// Inner-product-argument opening on gfx950.
//
// Replaces halo2_proofs 0.2.0 `poly::commitment::{create_proof, verify_proof}` (UPSTREAM,
// un-vendored: Cargo.lock:382-385), step 9 of plonk::create_proof as called from
// benches/shot.rs:68, benches/board.rs:61-68, src/circuits/shot.rs:921-928,
// src/circuits/board.rs:913-920, and the verifier half used by benches/board.rs:80-86.
//
// MI355X-first restructuring: upstream halves the generator vector every round
// ("parallel_generator_collapse": n/2 variable-base 255-bit scalar multiplications, then MSMs on
// the folded, non-fixed bases).  Here the generators are NEVER folded.  After j rounds
//     g^(j)_i = sum_t s^(j)_t * G_(i + t*m),  m = n / 2^j,  s^(j+1)_(2t+beta) = s^(j)_t * u_j^beta,
// so L_j = <p_hi, g_lo> and R_j = <p_lo, g_hi> are MSMs over the ORIGINAL SRS with the scalar
// vectors  p_hi[i] * s_t  /  p_lo[i] * s_t  scattered to index i + t*m -- i.e. two more batched
// MSMs against the fixed window table already resident in HBM (bzh_bases_precompute), plus
// O(n) field multiplications for the scalars.  The [value*z]U and [rand]W terms ride in the same
// MSM (U and W are the last two entries of the table).  Output bytes are identical to the
// collapsing formulation (tests/test_gpu_ipa.py pins them against the oracle's restatement of
// upstream under a shared randomness stream).
#include <cstring>
#include <vector>

#include "ctx.hpp"
#include "curve.cuh"

namespace bzh {

// ---------------------------------------------------------------------------
// kernels (all Montgomery form)
// ---------------------------------------------------------------------------
// 64-byte RNG outputs -> field elements: 512-bit little-endian integer mod p (Field::random)
template <class P>
__global__ void __launch_bounds__(256) k_reduce_wide(const uint32_t* __restrict__ in, size_t count, uint32_t* __restrict__ out) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= count) return;
    Fe<P> lo = fe_load<P>(in + g * 16), hi = fe_load<P>(in + g * 16 + 8);
    // value = lo + hi * 2^256; Montgomery image = lo*R + hi*R*R = mul(lo, R2) + mul(mul(hi, R2), R2)
    Fe<P> r2 = fe_r2<P>();
    fe_store(out + g * 8, fe_add(fe_mul(lo, r2), fe_mul(fe_mul(hi, r2), r2)));
}

// out[i] = x^i (binary method per thread)
template <class P>
__global__ void __launch_bounds__(256) k_powers(const uint32_t* __restrict__ x, size_t n, uint32_t* __restrict__ out) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= n) return;
    Fe<P> acc = fe_one<P>(), pw = fe_load<P>(x);
    for (size_t e = g; e; e >>= 1) {
        if (e & 1) acc = fe_mul(acc, pw);
        pw = fe_sqr(pw);
    }
    fe_store(out + g * 8, acc);
}

// v[0] -= *s
template <class P>
__global__ void k_sub_at0(uint32_t* __restrict__ v, const uint32_t* __restrict__ s) {
    if (threadIdx.x == 0 && blockIdx.x == 0) fe_store(v, fe_sub(fe_load<P>(v), fe_load<P>(s)));
}

// Round scalars: for idx < n with r = idx mod m, t = idx div m:
//   L[idx] = r <  m/2 ? p[r + m/2] * s[t] : 0        R[idx] = r >= m/2 ? p[r - m/2] * s[t] : 0
//   L[n] = vl * z, L[n+1] = l_rand                   R[n] = vr * z, R[n+1] = r_rand
template <class P>
__global__ void __launch_bounds__(256) k_ipa_round_vectors(const uint32_t* __restrict__ p, const uint32_t* __restrict__ s, size_t n,
                                                             unsigned log_m, const uint32_t* __restrict__ vlr,
                                                             const uint32_t* __restrict__ z, const uint32_t* __restrict__ rands,
                                                             uint32_t* __restrict__ lr) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint32_t* L = lr;
    uint32_t* R = lr + (n + 2) * 8;
    if (g < n) {
        const size_t m = (size_t)1 << log_m, half = m >> 1, r = g & (m - 1), t = g >> log_m;
        const Fe<P> st = fe_load<P>(s + t * 8);
        const Fe<P> zero = fe_zero<P>();
        if (r < half) {
            fe_store(L + g * 8, fe_mul(fe_load<P>(p + (r + half) * 8), st));
            fe_store(R + g * 8, zero);
        } else {
            fe_store(L + g * 8, zero);
            fe_store(R + g * 8, fe_mul(fe_load<P>(p + (r - half) * 8), st));
        }
    } else if (g == n) {
        const Fe<P> zz = fe_load<P>(z);
        fe_store(L + n * 8, fe_mul(fe_load<P>(vlr), zz));
        fe_store(R + n * 8, fe_mul(fe_load<P>(vlr + 8), zz));
        fe_store(L + (n + 1) * 8, fe_load<P>(rands));
        fe_store(R + (n + 1) * 8, fe_load<P>(rands + 8));
    }
}

// s_new[2t + beta] = s[t] * u^beta
template <class P>
__global__ void __launch_bounds__(256) k_ipa_s_update(const uint32_t* __restrict__ s, size_t count, const uint32_t* __restrict__ u,
                                                        uint32_t* __restrict__ s_new) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= count) return;
    const Fe<P> v = fe_load<P>(s + g * 8);
    fe_store(s_new + (2 * g) * 8, v);
    fe_store(s_new + (2 * g + 1) * 8, fe_mul(v, fe_load<P>(u)));
}

// ---------------------------------------------------------------------------
// host-side field / point helpers (portable fe_mul)
// ---------------------------------------------------------------------------
template <class P>
static Fe<P> h_load(const uint64_t* p) {
    Fe<P> v;
    for (int i = 0; i < 4; i++) {
        v.l[2 * i] = (uint32_t)p[i];
        v.l[2 * i + 1] = (uint32_t)(p[i] >> 32);
    }
    return v;
}
template <class P>
static void h_store(uint64_t* p, const Fe<P>& v) {
    for (int i = 0; i < 4; i++) p[i] = (uint64_t)v.l[2 * i] | ((uint64_t)v.l[2 * i + 1] << 32);
}
// Jacobian (Montgomery limbs as produced by msm_run) -> affine canonical x||y
template <class PB>
static void h_jac_to_affine_canonical(const uint64_t* xyz, uint64_t* xy) {
    Fe<PB> X = h_load<PB>(xyz), Y = h_load<PB>(xyz + 4), Z = h_load<PB>(xyz + 8);
    if (fe_is_zero(Z)) {
        memset(xy, 0, 64);
        return;
    }
    Fe<PB> zi = fe_inv(Z), zi2 = fe_sqr(zi), zi3 = fe_mul(zi2, zi);
    h_store<PB>(xy, fe_from_mont(fe_mul(X, zi2)));
    h_store<PB>(xy + 4, fe_from_mont(fe_mul(Y, zi3)));
}
// square root in the base field (Tonelli-Shanks; Montgomery in/out); returns false if non-residue
template <class PB>
static bool h_sqrt(const Fe<PB>& a, unsigned S, uint32_t gen, Fe<PB>& out) {
    if (fe_is_zero(a)) {
        out = a;
        return true;
    }
    uint32_t pm1[8], tt[8], t1h[8];
    uint64_t br = 1;
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)PB::mod(i) - br;
        pm1[i] = (uint32_t)d;
        br = (d >> 63) & 1;
    }
    auto shr = [](const uint32_t* in, unsigned s, uint32_t* o) {
        for (int i = 0; i < 8; i++) {
            unsigned src = i + s / 32;
            uint64_t lo = src < 8 ? in[src] : 0, hi = src + 1 < 8 ? in[src + 1] : 0;
            o[i] = (s % 32) ? (uint32_t)(((lo | (hi << 32)) >> (s % 32)) & 0xffffffffu) : (uint32_t)lo;
        }
    };
    shr(pm1, S, tt);  // t = (p-1) / 2^S (odd)
    // (t + 1) / 2
    uint32_t tp1[8];
    uint64_t c = 1;
    for (int i = 0; i < 8; i++) {
        c += tt[i];
        tp1[i] = (uint32_t)c;
        c >>= 32;
    }
    shr(tp1, 1, t1h);
    Fe<PB> zgen = fe_pow(fe_from_u32<PB>(gen), tt);  // generator of the 2-Sylow subgroup
    Fe<PB> x = fe_pow(a, t1h), b = fe_pow(a, tt);
    const Fe<PB> one = fe_one<PB>();
    unsigned m = S;
    while (!fe_eq(b, one)) {
        unsigned i = 0;
        Fe<PB> b2 = b;
        while (!fe_eq(b2, one)) {
            b2 = fe_sqr(b2);
            i++;
            if (i >= m) return false;  // not a square
        }
        Fe<PB> w = zgen;
        for (unsigned k = 0; k + i + 1 < m; k++) w = fe_sqr(w);
        zgen = fe_sqr(w);
        x = fe_mul(x, w);
        b = fe_mul(b, zgen);
        m = i;
    }
    out = x;
    return fe_eq(fe_sqr(x), a);
}

template <class C>
struct CurveMeta;
template <>
struct CurveMeta<VestaCurve> {
    using SF = FpParams;
    static constexpr int scalar_field = BZH_FIELD_FP;
    static constexpr unsigned base_S = 32;
    static constexpr uint32_t base_gen = 5;
};
template <>
struct CurveMeta<PallasCurve> {
    using SF = FqParams;
    static constexpr int scalar_field = BZH_FIELD_FQ;
    static constexpr unsigned base_S = 32;
    static constexpr uint32_t base_gen = 5;
};
template <>
struct CurveMeta<Bn254Curve> {
    using SF = BnFrParams;
    static constexpr int scalar_field = BZH_FIELD_BN254_FR;
    static constexpr unsigned base_S = 1;
    static constexpr uint32_t base_gen = 3;
};

// pasta_curves from_bytes: x little-endian, bit 255 = parity of y; zeros = identity
template <class C>
static bool h_decompress(const uint8_t* in, uint64_t* xy_canonical) {
    using PB = typename C::Base;
    uint8_t raw[32];
    memcpy(raw, in, 32);
    const unsigned ysign = raw[31] >> 7;
    raw[31] &= 0x7f;
    uint64_t xl[4];
    memcpy(xl, raw, 32);
    if (!(xl[0] | xl[1] | xl[2] | xl[3])) {
        if (ysign) return false;
        memset(xy_canonical, 0, 64);
        return true;
    }
    Fe<PB> x = h_load<PB>(xl);
    // canonical check x < p
    {
        Fe<PB> t = x;
        fe_cond_sub_p(t, 0);
        if (!fe_eq(t, x)) return false;
    }
    Fe<PB> xm = fe_to_mont(x);
    Fe<PB> rhs = fe_add(fe_mul(fe_sqr(xm), xm), fe_from_u32<PB>(C::b));
    Fe<PB> y;
    if (!h_sqrt<PB>(rhs, CurveMeta<C>::base_S, CurveMeta<C>::base_gen, y)) return false;
    Fe<PB> yc = fe_from_mont(y);
    if ((yc.l[0] & 1u) != ysign) yc = fe_from_mont(fe_neg(y));
    h_store<PB>(xy_canonical, x);
    h_store<PB>(xy_canonical + 4, yc);
    return true;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

#define IPA_TRY(expr)             \
    do {                          \
        int rc__ = (expr);        \
        if (rc__) return rc__;    \
    } while (0)

// ---------------------------------------------------------------------------
// prover
// ---------------------------------------------------------------------------
template <class C>
static int ipa_open_t(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_poly_in, int poly_on_device, const uint64_t* blind,
                      const uint64_t* x3, const uint8_t* rng_bytes, bzh_transcript* tr, uint64_t* out_v) {
    using SF = typename CurveMeta<C>::SF;
    using PB = typename C::Base;
    const int field = CurveMeta<C>::scalar_field;
    const size_t n = bases->n - 2;
    unsigned k = 0;
    while (((size_t)1 << k) < n) k++;
    if (((size_t)1 << k) != n) return BZH_E_ARG;
    const size_t nrand = n + 1 + 2 * (size_t)k;
    hipStream_t st = ctx->stream;

    // device arena: raw rng | rand scalars | s_poly | pp(2n: [poly | s_poly] then folded) | b | svec x2 | LR(2(n+2)) | small
    const size_t words = nrand * 16 + nrand * 8 + 2 * n * 8 + n * 8 + 2 * n * 8 + n * 8 + 2 * (n + 2) * 8 + 64 * 8 + (n + 2) * 8;
    DevBuf arena;
    BZH_HIP_TRY(ctx, hipMalloc(&arena.p, words * 4));
    uint32_t* cur = (uint32_t*)arena.p;
    auto take = [&](size_t w) {
        uint32_t* r = cur;
        cur += w;
        return r;
    };
    uint32_t* d_raw = take(nrand * 16);
    uint32_t* d_rand = take(nrand * 8);
    uint32_t* d_pp = take(2 * n * 8);   // [poly | s_poly] -> p' (first n)
    uint32_t* d_ppb = take(n * 8);      // fold ping-pong
    uint32_t* d_b = take(n * 8);
    uint32_t* d_bb = take(n * 8);
    uint32_t* d_s0 = take(n * 8);
    uint32_t* d_lr = take(2 * (n + 2) * 8);
    uint32_t* d_small = take(64 * 8);   // [0] x3, [1] s(x3)/v, [2] xi, [3] z, [4..5] vl vr, [6] u, [7] u_inv
    uint32_t* d_commit = take((n + 2) * 8);
    void* d_out = nullptr;
    IPA_TRY(ws_ensure(ctx, 3, 4 * 96, &d_out));

    auto up = [&](uint32_t* dst, const Fe<SF>& v) -> int {  // Montgomery element to device
        BZH_HIP_TRY(ctx, hipMemcpyAsync(dst, v.l, 32, hipMemcpyHostToDevice, st));
        BZH_HIP_TRY(ctx, hipStreamSynchronize(st));  // v is a stack temporary
        return BZH_OK;
    };
    auto down = [&](const uint32_t* src, Fe<SF>& v) -> int {
        BZH_HIP_TRY(ctx, hipMemcpyAsync(v.l, src, 32, hipMemcpyDeviceToHost, st));
        BZH_HIP_TRY(ctx, hipStreamSynchronize(st));
        return BZH_OK;
    };
    const unsigned g256 = 256;
    auto blocks = [&](size_t c) { return dim3((unsigned)((c + g256 - 1) / g256)); };

    // randomness: upstream draw order = n coefficients of s(X), s_blind, then (l_j, r_j) per round
    BZH_HIP_TRY(ctx, hipMemcpyAsync(d_raw, rng_bytes, nrand * 64, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((k_reduce_wide<SF>), blocks(nrand), dim3(g256), 0, st, d_raw, nrand, d_rand);
    uint32_t* d_spoly = d_pp + n * 8;
    BZH_HIP_TRY(ctx, hipMemcpyAsync(d_spoly, d_rand, n * 32, hipMemcpyDeviceToDevice, st));
    if (poly_on_device) {
        BZH_HIP_TRY(ctx, hipMemcpyAsync(d_pp, d_poly_in, n * 32, hipMemcpyDeviceToDevice, st));
    }  // else the caller staged it into d_poly_in == nullptr path below
    const Fe<SF> x3m = fe_to_mont(h_load<SF>(x3));
    IPA_TRY(up(d_small, x3m));
    // s(X) -= s(x3)
    IPA_TRY(poly_eval(ctx, field, d_spoly, n, 1, d_small, 0, d_small + 8));
    hipLaunchKernelGGL((k_sub_at0<SF>), dim3(1), dim3(64), 0, st, d_spoly, d_small + 8);
    // S = commit(s, s_blind): scalars [s..., 0, s_blind]
    BZH_HIP_TRY(ctx, hipMemcpyAsync(d_commit, d_spoly, n * 32, hipMemcpyDeviceToDevice, st));
    BZH_HIP_TRY(ctx, hipMemsetAsync(d_commit + n * 8, 0, 32, st));
    BZH_HIP_TRY(ctx, hipMemcpyAsync(d_commit + (n + 1) * 8, d_rand + n * 8, 32, hipMemcpyDeviceToDevice, st));
    IPA_TRY(msm_run(ctx, bases, d_commit, n + 2, 1, BZH_FORM_MONTGOMERY, (uint32_t*)d_out));
    uint64_t jac[24], xy[8];
    BZH_HIP_TRY(ctx, hipMemcpyAsync(jac, d_out, 96, hipMemcpyDeviceToHost, st));
    Fe<SF> s_blind;
    IPA_TRY(down(d_rand + n * 8, s_blind));
    h_jac_to_affine_canonical<PB>(jac, xy);
    IPA_TRY(bzh_transcript_write_point(tr, C::id, xy));
    uint64_t ch[4];
    IPA_TRY(bzh_transcript_squeeze_challenge(tr, ch));
    const Fe<SF> xi = fe_to_mont(h_load<SF>(ch));
    IPA_TRY(bzh_transcript_squeeze_challenge(tr, ch));
    const Fe<SF> z = fe_to_mont(h_load<SF>(ch));
    IPA_TRY(up(d_small + 16, xi));
    IPA_TRY(up(d_small + 24, z));
    // p' = poly + xi * s_poly   ([poly | s_poly] folded with xi), then p'[0] -= v, v = p'(x3)
    IPA_TRY(poly_fold(ctx, field, d_pp, n, 1, d_small + 16, 0, d_ppb));
    uint32_t* p_cur = d_ppb;
    uint32_t* p_nxt = d_pp;
    IPA_TRY(poly_eval(ctx, field, p_cur, n, 1, d_small, 0, d_small + 8));
    hipLaunchKernelGGL((k_sub_at0<SF>), dim3(1), dim3(64), 0, st, p_cur, d_small + 8);
    Fe<SF> vm;
    IPA_TRY(down(d_small + 8, vm));
    h_store<SF>(out_v, fe_from_mont(vm));
    Fe<SF> f = fe_add(fe_mul(s_blind, xi), fe_to_mont(h_load<SF>(blind)));
    // b = powers of x3 ; s = [1]
    hipLaunchKernelGGL((k_powers<SF>), blocks(n), dim3(g256), 0, st, d_small, n, d_b);
    uint32_t* b_cur = d_b;
    uint32_t* b_nxt = d_bb;
    uint32_t* s_cur = d_s0;
    uint32_t* s_nxt = d_commit;  // n elements are enough (d_commit is free after S)
    IPA_TRY(up(s_cur, fe_one<SF>()));
    BZH_HIP_TRY(ctx, hipGetLastError());

    for (unsigned j = 0; j < k; j++) {
        const size_t m = n >> j, half = m >> 1;
        // value_l = <p_hi, b_lo>, value_r = <p_lo, b_hi>
        IPA_TRY(poly_inner_product(ctx, field, p_cur + half * 8, b_cur, half, 1, d_small + 32));
        IPA_TRY(poly_inner_product(ctx, field, p_cur, b_cur + half * 8, half, 1, d_small + 40));
        hipLaunchKernelGGL((k_ipa_round_vectors<SF>), blocks(n + 1), dim3(g256), 0, st, p_cur, s_cur, n, k - j, d_small + 32,
                           d_small + 24, d_rand + (n + 1 + 2 * (size_t)j) * 8, d_lr);
        BZH_HIP_TRY(ctx, hipGetLastError());
        IPA_TRY(msm_run(ctx, bases, d_lr, n + 2, 2, BZH_FORM_MONTGOMERY, (uint32_t*)d_out));
        BZH_HIP_TRY(ctx, hipMemcpyAsync(jac, d_out, 192, hipMemcpyDeviceToHost, st));
        Fe<SF> lr[2];
        BZH_HIP_TRY(ctx, hipMemcpyAsync(lr, d_rand + (n + 1 + 2 * (size_t)j) * 8, 64, hipMemcpyDeviceToHost, st));
        BZH_HIP_TRY(ctx, hipStreamSynchronize(st));
        h_jac_to_affine_canonical<PB>(jac, xy);
        IPA_TRY(bzh_transcript_write_point(tr, C::id, xy));
        h_jac_to_affine_canonical<PB>(jac + 12, xy);
        IPA_TRY(bzh_transcript_write_point(tr, C::id, xy));
        IPA_TRY(bzh_transcript_squeeze_challenge(tr, ch));
        const Fe<SF> u = fe_to_mont(h_load<SF>(ch));
        const Fe<SF> u_inv = fe_inv(u);
        Fe<SF> uu[2] = {u, u_inv};
        BZH_HIP_TRY(ctx, hipMemcpyAsync(d_small + 48, uu, 64, hipMemcpyHostToDevice, st));
        BZH_HIP_TRY(ctx, hipStreamSynchronize(st));
        // p' <- p_lo + u^-1 p_hi ; b <- b_lo + u b_hi ; s <- s (x) (1, u)
        IPA_TRY(poly_fold(ctx, field, p_cur, half, 1, d_small + 56, 0, p_nxt));
        IPA_TRY(poly_fold(ctx, field, b_cur, half, 1, d_small + 48, 0, b_nxt));
        hipLaunchKernelGGL((k_ipa_s_update<SF>), blocks((size_t)1 << j), dim3(g256), 0, st, s_cur, (size_t)1 << j, d_small + 48, s_nxt);
        BZH_HIP_TRY(ctx, hipGetLastError());
        std::swap(p_cur, p_nxt);
        std::swap(b_cur, b_nxt);
        std::swap(s_cur, s_nxt);
        f = fe_add(f, fe_add(fe_mul(lr[0], u_inv), fe_mul(lr[1], u)));
    }
    Fe<SF> c;
    IPA_TRY(down(p_cur, c));
    uint64_t sc[4];
    h_store<SF>(sc, fe_from_mont(c));
    IPA_TRY(bzh_transcript_write_scalar(tr, sc));
    h_store<SF>(sc, fe_from_mont(f));
    IPA_TRY(bzh_transcript_write_scalar(tr, sc));
    return BZH_OK;
}

// ---------------------------------------------------------------------------
// verifier:  sum_j (u_j^-1 L_j + u_j R_j) + P - [v]G_0 + [xi]S  ==  [c]G'_0 + [c b_0 z]U + [f]W
// The n-term G'_0 = <s, G> runs on the GPU against the SRS table with s started at c; the
// (2k+5)-term left side is a second, small MSM on an ad-hoc table.
// ---------------------------------------------------------------------------
template <class C>
static int ipa_verify_t(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* commitment_xy, const uint64_t* x3, const uint64_t* v,
                        const uint8_t* proof, size_t proof_len, bzh_transcript* tr, const uint64_t* g0_u_w_xy) {
    using SF = typename CurveMeta<C>::SF;
    const size_t n = bases->n - 2;
    unsigned k = 0;
    while (((size_t)1 << k) < n) k++;
    if (((size_t)1 << k) != n) return BZH_E_ARG;
    if (proof_len != 32 * (1 + 2 * (size_t)k + 2)) return BZH_E_VERIFY;
    hipStream_t st = ctx->stream;
    const size_t nl = 2 * (size_t)k + 5;
    std::vector<uint64_t> pts(nl * 8), scal(nl * 4);
    uint64_t ch[4];
    // S
    uint64_t S[8];
    if (!h_decompress<C>(proof, S)) return BZH_E_VERIFY;
    IPA_TRY(bzh_transcript_common_point(tr, S));
    IPA_TRY(bzh_transcript_squeeze_challenge(tr, ch));
    const Fe<SF> xi = fe_to_mont(h_load<SF>(ch));
    IPA_TRY(bzh_transcript_squeeze_challenge(tr, ch));
    const Fe<SF> z = fe_to_mont(h_load<SF>(ch));
    std::vector<Fe<SF>> us(k);
    for (unsigned j = 0; j < k; j++) {
        uint64_t* L = &pts[(2 * j) * 8];
        uint64_t* R = &pts[(2 * j + 1) * 8];
        if (!h_decompress<C>(proof + 32 + 64 * j, L) || !h_decompress<C>(proof + 64 + 64 * j, R)) return BZH_E_VERIFY;
        IPA_TRY(bzh_transcript_common_point(tr, L));
        IPA_TRY(bzh_transcript_common_point(tr, R));
        IPA_TRY(bzh_transcript_squeeze_challenge(tr, ch));
        us[j] = fe_to_mont(h_load<SF>(ch));
        if (fe_is_zero(us[j])) return BZH_E_VERIFY;
        h_store<SF>(&scal[(2 * j) * 4], fe_from_mont(fe_inv(us[j])));
        h_store<SF>(&scal[(2 * j + 1) * 4], fe_from_mont(us[j]));
    }
    uint64_t cl[4], fl[4];
    memcpy(cl, proof + 32 + 64 * k, 32);
    memcpy(fl, proof + 64 + 64 * k, 32);
    Fe<SF> c = h_load<SF>(cl), f = h_load<SF>(fl);
    {
        Fe<SF> t = c;
        fe_cond_sub_p(t, 0);
        Fe<SF> t2 = f;
        fe_cond_sub_p(t2, 0);
        if (!fe_eq(t, c) || !fe_eq(t2, f)) return BZH_E_VERIFY;  // non-canonical scalars
    }
    const Fe<SF> cm = fe_to_mont(c), fm = fe_to_mont(f), x3m = fe_to_mont(h_load<SF>(x3)), vm = fe_to_mont(h_load<SF>(v));
    // b_0 = prod_j (1 + u_j x3^(2^(k-1-j)))
    std::vector<Fe<SF>> xp(k ? k : 1);
    if (k) {
        xp[0] = x3m;
        for (unsigned i = 1; i < k; i++) xp[i] = fe_sqr(xp[i - 1]);
    }
    Fe<SF> b0 = fe_one<SF>();
    for (unsigned j = 0; j < k; j++) b0 = fe_mul(b0, fe_add(fe_one<SF>(), fe_mul(us[j], xp[k - 1 - j])));
    // left-side table tail: P (1), G_0 (-v), S (xi), U (-c b0 z), W (-f)
    size_t o = 2 * (size_t)k;
    memcpy(&pts[o * 8], commitment_xy, 64);
    h_store<SF>(&scal[o * 4], fe_from_mont(fe_one<SF>()));
    o++;
    memcpy(&pts[o * 8], g0_u_w_xy, 64);
    h_store<SF>(&scal[o * 4], fe_from_mont(fe_neg(vm)));
    o++;
    memcpy(&pts[o * 8], S, 64);
    h_store<SF>(&scal[o * 4], fe_from_mont(xi));
    o++;
    memcpy(&pts[o * 8], g0_u_w_xy + 8, 64);
    h_store<SF>(&scal[o * 4], fe_from_mont(fe_neg(fe_mul(fe_mul(cm, b0), z))));
    o++;
    memcpy(&pts[o * 8], g0_u_w_xy + 16, 64);
    h_store<SF>(&scal[o * 4], fe_from_mont(fe_neg(fm)));

    // device: s vector (started at c) and the two MSMs
    DevBuf arena;
    const size_t words = 2 * (n + 2) * 8 + nl * 16 + nl * 8 + 64;
    BZH_HIP_TRY(ctx, hipMalloc(&arena.p, words * 4));
    uint32_t* d_s0 = (uint32_t*)arena.p;
    uint32_t* d_s1 = d_s0 + (n + 2) * 8;
    uint32_t* d_pts = d_s1 + (n + 2) * 8;
    uint32_t* d_scal = d_pts + nl * 16;
    uint32_t* d_u = d_scal + nl * 8;
    BZH_HIP_TRY(ctx, hipMemcpyAsync(d_s0, cm.l, 32, hipMemcpyHostToDevice, st));
    BZH_HIP_TRY(ctx, hipStreamSynchronize(st));
    uint32_t* s_cur = d_s0;
    uint32_t* s_nxt = d_s1;
    for (unsigned j = 0; j < k; j++) {
        BZH_HIP_TRY(ctx, hipMemcpyAsync(d_u, us[j].l, 32, hipMemcpyHostToDevice, st));
        const size_t cnt = (size_t)1 << j;
        hipLaunchKernelGGL((k_ipa_s_update<SF>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, s_cur, cnt, d_u, s_nxt);
        BZH_HIP_TRY(ctx, hipGetLastError());
        BZH_HIP_TRY(ctx, hipStreamSynchronize(st));
        std::swap(s_cur, s_nxt);
    }
    BZH_HIP_TRY(ctx, hipMemsetAsync(s_cur + n * 8, 0, 64, st));  // no U / W contribution on the right
    void* d_out = nullptr;
    IPA_TRY(ws_ensure(ctx, 3, 4 * 96, &d_out));
    IPA_TRY(msm_run(ctx, bases, s_cur, n + 2, 1, BZH_FORM_MONTGOMERY, (uint32_t*)d_out));
    uint64_t jac[24];
    BZH_HIP_TRY(ctx, hipMemcpyAsync(jac, d_out, 96, hipMemcpyDeviceToHost, st));
    // left side on an ad-hoc table (canonical in)
    BZH_HIP_TRY(ctx, hipMemcpyAsync(d_pts, pts.data(), nl * 64, hipMemcpyHostToDevice, st));
    BZH_HIP_TRY(ctx, hipMemcpyAsync(d_scal, scal.data(), nl * 32, hipMemcpyHostToDevice, st));
    IPA_TRY(bases_to_montgomery(ctx, C::id, d_pts, nl));
    bzh_bases tmp;
    tmp.curve = C::id;
    tmp.n = nl;
    tmp.d_xy = d_pts;
    tmp.device = ctx->device;
    IPA_TRY(msm_run(ctx, &tmp, d_scal, nl, 1, BZH_FORM_CANONICAL, (uint32_t*)d_out + 24));
    BZH_HIP_TRY(ctx, hipMemcpyAsync(jac + 12, (uint32_t*)d_out + 24, 96, hipMemcpyDeviceToHost, st));
    BZH_HIP_TRY(ctx, hipStreamSynchronize(st));
    uint64_t rhs[8], lhs[8];
    h_jac_to_affine_canonical<typename C::Base>(jac, rhs);
    // the second MSM ran with canonical form: its output limbs are canonical, convert for the helper
    {
        using PB = typename C::Base;
        uint64_t jm[12];
        for (int q = 0; q < 3; q++) h_store<PB>(jm + 4 * q, fe_to_mont(h_load<PB>(jac + 12 + 4 * q)));
        h_jac_to_affine_canonical<PB>(jm, lhs);
    }
    return memcmp(lhs, rhs, 64) == 0 ? BZH_OK : BZH_E_VERIFY;
}

int random_field(bzh_ctx* ctx, int field, const uint32_t* d_raw, size_t count, uint32_t* d_out) {
    if (!count) return BZH_OK;
    const dim3 grid((unsigned)((count + 255) / 256)), block(256);
    switch (field) {
        case BZH_FIELD_FP: hipLaunchKernelGGL((k_reduce_wide<FpParams>), grid, block, 0, ctx->stream, d_raw, count, d_out); break;
        case BZH_FIELD_FQ: hipLaunchKernelGGL((k_reduce_wide<FqParams>), grid, block, 0, ctx->stream, d_raw, count, d_out); break;
        case BZH_FIELD_BN254_FR: hipLaunchKernelGGL((k_reduce_wide<BnFrParams>), grid, block, 0, ctx->stream, d_raw, count, d_out); break;
        case BZH_FIELD_BN254_FQ: hipLaunchKernelGGL((k_reduce_wide<BnFqParams>), grid, block, 0, ctx->stream, d_raw, count, d_out); break;
        default: return BZH_E_ARG;
    }
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

int ipa_open(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_poly, const uint64_t* blind, const uint64_t* x3,
             const uint8_t* rng_bytes, bzh_transcript* tr, uint64_t* out_v) {
    switch (bases->curve) {
        case BZH_CURVE_VESTA: return ipa_open_t<VestaCurve>(ctx, bases, d_poly, 1, blind, x3, rng_bytes, tr, out_v);
        case BZH_CURVE_PALLAS: return ipa_open_t<PallasCurve>(ctx, bases, d_poly, 1, blind, x3, rng_bytes, tr, out_v);
        case BZH_CURVE_BN254: return ipa_open_t<Bn254Curve>(ctx, bases, d_poly, 1, blind, x3, rng_bytes, tr, out_v);
    }
    return BZH_E_ARG;
}
int ipa_verify(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* commitment_xy, const uint64_t* x3, const uint64_t* v,
               const uint8_t* proof, size_t proof_len, bzh_transcript* tr, const uint64_t* g0_u_w_xy) {
    switch (bases->curve) {
        case BZH_CURVE_VESTA: return ipa_verify_t<VestaCurve>(ctx, bases, commitment_xy, x3, v, proof, proof_len, tr, g0_u_w_xy);
        case BZH_CURVE_PALLAS: return ipa_verify_t<PallasCurve>(ctx, bases, commitment_xy, x3, v, proof, proof_len, tr, g0_u_w_xy);
        case BZH_CURVE_BN254: return ipa_verify_t<Bn254Curve>(ctx, bases, commitment_xy, x3, v, proof, proof_len, tr, g0_u_w_xy);
    }
    return BZH_E_ARG;
}

}  // namespace bzh
