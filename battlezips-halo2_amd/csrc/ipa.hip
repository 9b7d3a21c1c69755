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
//
// Round 3: ONE collapse instead of k.  Every round over the original table costs nwin * n bucket additions whatever the
// round, 14 x 24 n at k = 14 -- a third of a proof's additions.  At round j* the folded generators
//     G'[i] = sum_{t < 2^j*} s_t G_(i + t m),  m = n / 2^j*
// are materialised ONCE per proof through the same window table (msm_collapse_table, csrc/msm.hip: the scalars are shared by
// all i, so the bucket method runs with the level as the outer loop and one output per lane: (2^j* nwin + 2 levels) additions
// per output instead of a 255-bit double-and-add per term) and given their own small window table; the remaining rounds are
// the same paired MSMs on m points.  Additions per opening: 24 n j* + ~48 n + 150 m + 26 m (k - j*) instead of 24 n k
// (k = 14, j* = 4: 168 n against 336 n).  The group elements L_j, R_j are the same, hence the proof bytes.  Used when the
// batch is large enough to be throughput-bound (the collapse is two long per-lane chains: it would lengthen a single
// proof); BZH_IPA_COLLAPSE = 0 disables it, = j forces round j at any batch size; BZH_IPA_TAIL_C sets the tail window.
#include <cstring>
#include <vector>

#include "block.cuh"
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
// prover, batched: `batch` independent openings advance in lockstep so that every round is ONE
// MSM launch of 2*batch vectors, one device->host copy and one host<-device challenge upload.
// Per-proof vectors are compact: p, b are (batch, m) and s is (batch, n/m) in round j (m = n >> j).
// ---------------------------------------------------------------------------
// host-written constants per proof (d_hc): [0] x3, [1] xi, [2] z, [3] u, [4] u_inv   (stride kHc)
static constexpr size_t kHc = 8;

template <class P>
__global__ void k_sub_at0_batch(uint32_t* __restrict__ v, size_t stride, const uint32_t* __restrict__ s, size_t batch) {
    const size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (b >= batch) return;
    uint32_t* e = v + b * stride * 8;
    fe_store(e, fe_sub(fe_load<P>(e), fe_load<P>(s + b * 8)));
}

// commit scalars of S: out[b] = [s_poly_b (n) | 0 | s_blind_b]
template <class P>
__global__ void __launch_bounds__(256) k_ipa_commit_scalars(const uint32_t* __restrict__ spoly, const uint32_t* __restrict__ rands,
                                                              size_t n, size_t nrand, uint32_t* __restrict__ out) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x, b = blockIdx.y;
    if (g >= n + 2) return;
    Fe<P> v = fe_zero<P>();
    if (g < n) v = fe_load<P>(spoly + (b * n + g) * 8);
    else if (g == n + 1) v = fe_load<P>(rands + (b * nrand + n) * 8);
    fe_store(out + (b * (n + 2) + g) * 8, v);
}

// p'[b][i] = poly[b][i] + xi_b * s[b][i]
template <class P>
__global__ void __launch_bounds__(256) k_ipa_axpy(const uint32_t* __restrict__ poly, const uint32_t* __restrict__ spoly,
                                                    const uint32_t* __restrict__ hc, size_t n, uint32_t* __restrict__ out) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x, b = blockIdx.y;
    if (g >= n) return;
    const Fe<P> xi = fe_load<P>(hc + (b * kHc + 1) * 8);
    const size_t o = (b * n + g) * 8;
    fe_store(out + o, fe_add(fe_load<P>(poly + o), fe_mul(xi, fe_load<P>(spoly + o))));
}

// bvec[b][i] = x3_b^i, s[b][0] = 1
template <class P>
__global__ void __launch_bounds__(256) k_ipa_init_b_s(const uint32_t* __restrict__ hc, size_t n, uint32_t* __restrict__ bvec,
                                                        uint32_t* __restrict__ s) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x, b = blockIdx.y;
    if (g >= n) return;
    Fe<P> acc = fe_one<P>(), pw = fe_load<P>(hc + b * kHc * 8);
    if (g == 0) fe_store(s + b * 8, acc);
    for (size_t e = g; e; e >>= 1) {
        if (e & 1) acc = fe_mul(acc, pw);
        pw = fe_sqr(pw);
    }
    fe_store(bvec + (b * n + g) * 8, acc);
}

// out[b][0] = <p_hi, b_lo>, out[b][1] = <p_lo, b_hi>; grid (batch, 2)
template <class P>
__global__ void __launch_bounds__(256) k_ipa_inner2(const uint32_t* __restrict__ p, const uint32_t* __restrict__ bv, size_t half,
                                                      uint32_t* __restrict__ out) {
    __shared__ __align__(16) uint4 sh[2 * 256];
    const size_t b = blockIdx.x, side = blockIdx.y, m = 2 * half;
    const uint32_t* pa = p + (b * m + (side ? 0 : half)) * 8;
    const uint32_t* ba = bv + (b * m + (side ? half : 0)) * 8;
    Fe<P> acc = fe_zero<P>();
    for (size_t i = threadIdx.x; i < half; i += 256) acc = fe_add(acc, fe_mul(fe_load<P>(pa + i * 8), fe_load<P>(ba + i * 8)));
    acc = block_sum(acc, sh, 256);
    if (threadIdx.x == 0) fe_store(out + (b * 2 + side) * 8, acc);
}

// Round scalars of proof b: for idx < n with r = idx mod m, t = idx div m:
//   L[idx] = r <  m/2 ? p[r + m/2] * s[t] : 0        R[idx] = r >= m/2 ? p[r - m/2] * s[t] : 0
//   L[n] = vl * z, L[n+1] = l_rand                   R[n] = vr * z, R[n+1] = r_rand
template <class P>
__global__ void __launch_bounds__(256) k_ipa_round_vectors(const uint32_t* __restrict__ p, const uint32_t* __restrict__ s, size_t n,
                                                             unsigned log_m, const uint32_t* __restrict__ vlr,
                                                             const uint32_t* __restrict__ hc, const uint32_t* __restrict__ rands,
                                                             size_t nrand, size_t rand_n, unsigned round, uint32_t* __restrict__ lr) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x, b = blockIdx.y;
    const size_t m = (size_t)1 << log_m, half = m >> 1;
    const uint32_t* pb = p + b * m * 8;
    const uint32_t* sb = s + b * (n >> log_m) * 8;
    uint32_t* L = lr + b * 2 * (n + 2) * 8;
    uint32_t* R = L + (n + 2) * 8;
    if (g < n) {
        const size_t r = g & (m - 1), t = g >> log_m;
        const Fe<P> st = fe_load<P>(sb + t * 8);
        const Fe<P> zero = fe_zero<P>();
        if (r < half) {
            fe_store(L + g * 8, fe_mul(fe_load<P>(pb + (r + half) * 8), st));
            fe_store(R + g * 8, zero);
        } else {
            fe_store(L + g * 8, zero);
            fe_store(R + g * 8, fe_mul(fe_load<P>(pb + (r - half) * 8), st));
        }
    } else if (g == n) {
        const Fe<P> zz = fe_load<P>(hc + (b * kHc + 2) * 8);
        const uint32_t* rd = rands + (b * nrand + rand_n + 1 + 2 * (size_t)round) * 8;   // rand_n: the opening's full length
        fe_store(L + n * 8, fe_mul(fe_load<P>(vlr + b * 16), zz));
        fe_store(R + n * 8, fe_mul(fe_load<P>(vlr + b * 16 + 8), zz));
        fe_store(L + (n + 1) * 8, fe_load<P>(rd));
        fe_store(R + (n + 1) * 8, fe_load<P>(rd + 8));
    }
}

// The same scalars as ONE dense vector per proof for msm_run_paired (L and R have disjoint supports):
//   LR[idx] = p[(r + m/2) mod m] * s[t]  (class = r >= m/2),  LR[n..n+1] = (vl z, l_rand),  LR[n+2..n+3] = (vr z, r_rand)
template <class P>
__global__ void __launch_bounds__(256) k_ipa_round_vectors_paired(const uint32_t* __restrict__ p, const uint32_t* __restrict__ s,
                                                                    size_t n, unsigned log_m, const uint32_t* __restrict__ vlr,
                                                                    const uint32_t* __restrict__ hc,
                                                                    const uint32_t* __restrict__ rands, size_t nrand, size_t rand_n,
                                                                    unsigned round, uint32_t* __restrict__ lr) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x, b = blockIdx.y;
    const size_t m = (size_t)1 << log_m, half = m >> 1;
    uint32_t* V = lr + b * (n + 4) * 8;
    if (g < n) {
        const size_t r = g & (m - 1), t = g >> log_m;
        const size_t src = r < half ? r + half : r - half;
        fe_store(V + g * 8, fe_mul(fe_load<P>(p + (b * m + src) * 8), fe_load<P>(s + (b * (n >> log_m) + t) * 8)));
    } else if (g == n) {
        const Fe<P> zz = fe_load<P>(hc + (b * kHc + 2) * 8);
        const uint32_t* rd = rands + (b * nrand + rand_n + 1 + 2 * (size_t)round) * 8;
        fe_store(V + n * 8, fe_mul(fe_load<P>(vlr + b * 16), zz));
        fe_store(V + (n + 1) * 8, fe_load<P>(rd));
        fe_store(V + (n + 2) * 8, fe_mul(fe_load<P>(vlr + b * 16 + 8), zz));
        fe_store(V + (n + 3) * 8, fe_load<P>(rd + 8));
    }
}

// one launch per round: p' = p_lo + u^-1 p_hi, b' = b_lo + u b_hi (i < half), s'[2t + beta] = s[t] u^beta (t < cnt)
template <class P>
__global__ void __launch_bounds__(256) k_ipa_round_fold(const uint32_t* __restrict__ p, const uint32_t* __restrict__ bv,
                                                          const uint32_t* __restrict__ s, size_t half, size_t cnt,
                                                          const uint32_t* __restrict__ hc, uint32_t* __restrict__ p_out,
                                                          uint32_t* __restrict__ b_out, uint32_t* __restrict__ s_out) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x, b = blockIdx.y;
    const Fe<P> u = fe_load<P>(hc + (b * kHc + 3) * 8);
    if (g < half) {
        const Fe<P> ui = fe_load<P>(hc + (b * kHc + 4) * 8);
        const uint32_t* pb = p + b * 2 * half * 8;
        const uint32_t* bb = bv + b * 2 * half * 8;
        fe_store(p_out + (b * half + g) * 8, fe_add(fe_load<P>(pb + g * 8), fe_mul(ui, fe_load<P>(pb + (half + g) * 8))));
        fe_store(b_out + (b * half + g) * 8, fe_add(fe_load<P>(bb + g * 8), fe_mul(u, fe_load<P>(bb + (half + g) * 8))));
    }
    if (g < cnt) {
        const Fe<P> v = fe_load<P>(s + (b * cnt + g) * 8);
        fe_store(s_out + (b * 2 * cnt + 2 * g) * 8, v);
        fe_store(s_out + (b * 2 * cnt + 2 * g + 1) * 8, fe_mul(v, u));
    }
}

// Jacobian (Montgomery limbs as produced by msm_run) -> affine canonical x||y for `cnt` points with ONE inversion
template <class PB>
static void h_jac_batch_to_affine_canonical(const uint64_t* xyz, size_t cnt, uint64_t* xy) {
    std::vector<Fe<PB>> pre(cnt + 1);
    pre[0] = fe_one<PB>();
    for (size_t i = 0; i < cnt; i++) {
        const Fe<PB> Z = h_load<PB>(xyz + i * 12 + 8);
        pre[i + 1] = fe_is_zero(Z) ? pre[i] : fe_mul(pre[i], Z);
    }
    Fe<PB> inv = fe_inv(pre[cnt]);
    for (size_t i = cnt; i-- > 0;) {
        const Fe<PB> Z = h_load<PB>(xyz + i * 12 + 8);
        if (fe_is_zero(Z)) {
            memset(xy + i * 8, 0, 64);
            continue;
        }
        const Fe<PB> zi = fe_mul(inv, pre[i]);
        inv = fe_mul(inv, Z);
        const Fe<PB> zi2 = fe_sqr(zi), zi3 = fe_mul(zi2, zi);
        h_store<PB>(xy + i * 8, fe_from_mont(fe_mul(h_load<PB>(xyz + i * 12), zi2)));
        h_store<PB>(xy + i * 8 + 4, fe_from_mont(fe_mul(h_load<PB>(xyz + i * 12 + 4), zi3)));
    }
}

template <class C>
static int ipa_open_t(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_polys, size_t batch, const uint64_t* blinds,
                      const uint64_t* x3s, const uint8_t* rng_bytes, size_t rng_stride, bzh_transcript* const* trs,
                      uint64_t* out_v, const uint32_t* d_raw_in) {
    using SF = typename CurveMeta<C>::SF;
    using PB = typename C::Base;
    const int field = CurveMeta<C>::scalar_field;
    const size_t n = bases->n - 2, B = batch;
    unsigned k = 0;
    while (((size_t)1 << k) < n) k++;
    if (((size_t)1 << k) != n || !B) return BZH_E_ARG;
    const size_t nrand = n + 1 + 2 * (size_t)k, ntail = 1 + 2 * (size_t)k;
    hipStream_t st = ctx->stream;

    // The generator collapse: at round jstar the folded generators get their own tables and the rounds continue on m points
    // (header).  jstar minimises  24 n j + 48 n + 150 m + 26 m (k - j)  over the rounds that leave m >= 64.
    unsigned jstar = 0;
    int tail_c = 9;
    {
        static const int env_j = [] {
            const char* e = getenv("BZH_IPA_COLLAPSE");
            return e ? atoi(e) : -1;
        }();
        static const int env_c = [] {
            const char* e = getenv("BZH_IPA_TAIL_C");
            return e ? atoi(e) : 0;
        }();
        if (env_c >= 4 && env_c <= 13) tail_c = env_c;
        if (bases->pre_c != 0 && !bases->vec_col_stride && env_j != 0) {
            if (env_j > 0) {
                if ((unsigned)env_j + 1 < k) jstar = (unsigned)env_j;   // forced (tests): any batch size, m >= 2
            } else if (B >= 8 && k >= 9) {
                double best = 24.0 * (double)n * k * 0.9;               // at least 10 % fewer additions than no collapse
                for (unsigned j = 1; j + 6 <= k; j++) {
                    const double m = (double)(n >> j);
                    const double cost = 24.0 * n * j + 48.0 * n + 150.0 * m + 26.0 * m * (k - j);
                    if (cost < best) {
                        best = cost;
                        jstar = j;
                    }
                }
            }
            if (jstar && (size_t)(2 * ((size_t)1 << jstar) * bases->pre_nwin) * 4 > 48 * 1024) jstar = 0;   // item lists must fit LDS
        }
    }
    const size_t tail_m = jstar ? n >> jstar : 0;
    const int tail_nwin = (256 + tail_c - 1) / tail_c;
    const size_t tail_scratch = jstar ? msm_collapse_scratch_bytes(bases, (size_t)1 << jstar, B, tail_c) : 0;
    // device arena (workspace slot 4), per proof: raw rng | rand scalars | s_poly | p x2 | b x2 | s x2 | LR | S scalars | small
    const size_t per = nrand * 16 + nrand * 8 + n * 8 + 2 * n * 8 + 2 * n * 8 + 2 * n * 8 + 2 * (n + 4) * 8 + (n + 2) * 8 +
                       (kHc + 4) * 8;
    const size_t tail_words = jstar ? B * (tail_m + 2) * (size_t)tail_nwin * (16 + 20) + (tail_scratch + 3) / 4 + 64 : 0;   // table + its fe29 copy
    void* arena = nullptr;
    IPA_TRY(ws_ensure(ctx, 4, (B * per + tail_words) * 4 + 256 + 16 * 16, &arena));   // (+ the roundings of take())
    uint32_t* cur = (uint32_t*)arena;
    auto take = [&](size_t w) {   // every piece starts 16-byte aligned (uint4 loads / stores): word counts rounded up to 4
        uint32_t* r = cur;
        cur += (w + 3) & ~(size_t)3;
        return r;
    };
    uint32_t* d_raw = take(B * nrand * 16);
    uint32_t* d_rand = take(B * nrand * 8);
    uint32_t* d_spoly = take(B * n * 8);
    uint32_t* p_cur = take(B * n * 8);
    uint32_t* p_nxt = take(B * n * 8);
    uint32_t* b_cur = take(B * n * 8);
    uint32_t* b_nxt = take(B * n * 8);
    uint32_t* s_cur = take(B * n * 8);
    uint32_t* s_nxt = take(B * n * 8);
    uint32_t* d_lr = take(B * 2 * (n + 4) * 8);
    const bool paired = bases->pre_c != 0 && k >= 1;  // window table: L_j and R_j of a proof share one dense vector
    uint32_t* d_commit = take(B * (n + 2) * 8);
    uint32_t* d_tail_table = jstar ? take(B * (tail_m + 2) * (size_t)tail_nwin * 16) : nullptr;
    uint32_t* d_tail_table29 = jstar ? take(B * (tail_m + 2) * (size_t)tail_nwin * 20) : nullptr;
    void* d_tail_scratch = jstar ? (void*)take((tail_scratch + 3) / 4) : nullptr;
    uint32_t* d_hc = take(B * kHc * 8);
    uint32_t* d_dv = take(B * 8);       // s(x3) / v per proof
    uint32_t* d_vlr = take(B * 2 * 8);  // value_l, value_r per proof
    void* d_out = nullptr;
    IPA_TRY(ws_ensure(ctx, 5, 2 * B * 96, &d_out));

    const unsigned g256 = 256;
    auto grid2 = [&](size_t c) { return dim3((unsigned)((c + g256 - 1) / g256), (unsigned)B); };
    if (B > 65535) return BZH_E_ARG;

    // randomness: upstream draw order = n coefficients of s(X), s_blind, then (l_j, r_j) per round
    if (d_raw_in) {  // the 64-byte draws are already on the device (batch x nrand, proof-major)
        BZH_HIP_TRY(ctx, hipMemcpyAsync(d_raw, d_raw_in, B * nrand * 64, hipMemcpyDeviceToDevice, st));
    } else {
        for (size_t b = 0; b < B; b++)
            IPA_TRY(h2d_small(ctx, d_raw + b * nrand * 16, rng_bytes + b * rng_stride, nrand * 64));
    }
    hipLaunchKernelGGL((k_reduce_wide<SF>), dim3((unsigned)((B * nrand + g256 - 1) / g256)), dim3(g256), 0, st, d_raw, B * nrand,
                       d_rand);
    BZH_HIP_TRY(ctx, hipMemcpy2DAsync(d_spoly, n * 32, d_rand, nrand * 32, n * 32, B, hipMemcpyDeviceToDevice, st));
    // the scalars the host needs (s_blind and the round blinds) come back in one strided copy
    std::vector<Fe<SF>> tail(B * ntail);
    for (size_t b = 0; b < B; b++) IPA_TRY(d2h_async(ctx, &tail[b * ntail], d_rand + (b * nrand + n) * 8, ntail * 32));

    std::vector<Fe<SF>> hc(B * kHc, fe_zero<SF>());
    for (size_t b = 0; b < B; b++) hc[b * kHc] = fe_to_mont(h_load<SF>(x3s + 4 * b));
    auto push_hc = [&]() -> int {
        return h2d_small(ctx, d_hc, hc.data(), B * kHc * 32);
    };
    IPA_TRY(push_hc());
    // s(X) -= s(x3)
    IPA_TRY(poly_eval(ctx, field, d_spoly, n, B, d_hc, kHc, d_dv));
    hipLaunchKernelGGL((k_sub_at0_batch<SF>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, d_spoly, n, d_dv, B);
    // S = commit(s, s_blind): scalars [s..., 0, s_blind]
    hipLaunchKernelGGL((k_ipa_commit_scalars<SF>), grid2(n + 2), dim3(g256), 0, st, d_spoly, d_rand, n, nrand, d_commit);
    BZH_HIP_TRY(ctx, hipGetLastError());
    IPA_TRY(msm_run(ctx, bases, d_commit, n + 2, B, BZH_FORM_MONTGOMERY, (uint32_t*)d_out));
    std::vector<uint64_t> jac(2 * B * 12), xy(2 * B * 8);
    IPA_TRY(d2h_async(ctx, jac.data(), d_out, B * 96));
    IPA_TRY(d2h_finish(ctx));
    h_jac_batch_to_affine_canonical<PB>(jac.data(), B, xy.data());
    uint64_t ch[4];
    std::vector<Fe<SF>> f(B);
    for (size_t b = 0; b < B; b++) {
        IPA_TRY(bzh_transcript_write_point(trs[b], C::id, xy.data() + b * 8));
        IPA_TRY(bzh_transcript_squeeze_challenge(trs[b], ch));
        const Fe<SF> xi = fe_to_mont(h_load<SF>(ch));
        IPA_TRY(bzh_transcript_squeeze_challenge(trs[b], ch));
        hc[b * kHc + 1] = xi;
        hc[b * kHc + 2] = fe_to_mont(h_load<SF>(ch));
        f[b] = fe_add(fe_mul(tail[b * ntail], xi), fe_to_mont(h_load<SF>(blinds + 4 * b)));
    }
    IPA_TRY(push_hc());
    // p' = poly + xi * s_poly, then p'[0] -= v with v = p'(x3)
    hipLaunchKernelGGL((k_ipa_axpy<SF>), grid2(n), dim3(g256), 0, st, d_polys, d_spoly, d_hc, n, p_cur);
    IPA_TRY(poly_eval(ctx, field, p_cur, n, B, d_hc, kHc, d_dv));
    hipLaunchKernelGGL((k_sub_at0_batch<SF>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, p_cur, n, d_dv, B);
    std::vector<Fe<SF>> vm(B);
    IPA_TRY(d2h_async(ctx, vm.data(), d_dv, B * 32));  // lands at the next d2h_finish
    hipLaunchKernelGGL((k_ipa_init_b_s<SF>), grid2(n), dim3(g256), 0, st, d_hc, n, b_cur, s_cur);
    BZH_HIP_TRY(ctx, hipGetLastError());

    std::vector<Fe<SF>> us(B), pre(B + 1);
    // the instance the rounds run on: the SRS and n points, or (from round jstar on) the per-proof tables and tail_m points
    const bzh_bases* rb = bases;
    bzh_bases tail_bases;
    size_t rn = n;          // points of the current instance
    unsigned j0 = 0;        // the round its s vector restarted at
    for (unsigned j = 0; j < k; j++) {
        if (jstar && j == jstar) {
            IPA_TRY(msm_collapse_table(ctx, bases, s_cur, (size_t)1 << j, B, tail_c, d_tail_table29, d_tail_table, d_tail_scratch, &tail_bases));
            rb = &tail_bases;
            rn = tail_m;
            j0 = j;
            // s restarts at (1): the folded generators ARE the instance now
            std::vector<Fe<SF>> ones(B, fe_one<SF>());
            IPA_TRY(h2d_small(ctx, s_cur, ones.data(), B * 32));
        }
        const size_t m = n >> j, half = m >> 1, cnt = (size_t)1 << (j - j0);
        hipLaunchKernelGGL((k_ipa_inner2<SF>), dim3((unsigned)B, 2), dim3(256), 0, st, p_cur, b_cur, half, d_vlr);
        if (paired) {
            hipLaunchKernelGGL((k_ipa_round_vectors_paired<SF>), grid2(rn + 1), dim3(g256), 0, st, p_cur, s_cur, rn, k - j, d_vlr, d_hc,
                               d_rand, nrand, n, j, d_lr);
            BZH_HIP_TRY(ctx, hipGetLastError());
            IPA_TRY(msm_run_paired(ctx, rb, d_lr, rn, k - j, B, BZH_FORM_MONTGOMERY, (uint32_t*)d_out));
        } else {
            hipLaunchKernelGGL((k_ipa_round_vectors<SF>), grid2(n + 1), dim3(g256), 0, st, p_cur, s_cur, n, k - j, d_vlr, d_hc,
                               d_rand, nrand, n, j, d_lr);
            BZH_HIP_TRY(ctx, hipGetLastError());
            IPA_TRY(msm_run(ctx, bases, d_lr, n + 2, 2 * B, BZH_FORM_MONTGOMERY, (uint32_t*)d_out));
        }
        IPA_TRY(d2h_async(ctx, jac.data(), d_out, 2 * B * 96));
        IPA_TRY(d2h_finish(ctx));
        h_jac_batch_to_affine_canonical<PB>(jac.data(), 2 * B, xy.data());
        pre[0] = fe_one<SF>();
        for (size_t b = 0; b < B; b++) {
            IPA_TRY(bzh_transcript_write_point(trs[b], C::id, xy.data() + (2 * b) * 8));
            IPA_TRY(bzh_transcript_write_point(trs[b], C::id, xy.data() + (2 * b + 1) * 8));
            IPA_TRY(bzh_transcript_squeeze_challenge(trs[b], ch));
            us[b] = fe_to_mont(h_load<SF>(ch));
            if (fe_is_zero(us[b])) return BZH_E_ARG;  // a zero challenge has no inverse (probability 2^-255)
            pre[b + 1] = fe_mul(pre[b], us[b]);
        }
        Fe<SF> inv = fe_inv(pre[B]);
        for (size_t b = B; b-- > 0;) {
            const Fe<SF> u_inv = fe_mul(inv, pre[b]);
            inv = fe_mul(inv, us[b]);
            hc[b * kHc + 3] = us[b];
            hc[b * kHc + 4] = u_inv;
            f[b] = fe_add(f[b], fe_add(fe_mul(tail[b * ntail + 1 + 2 * j], u_inv), fe_mul(tail[b * ntail + 2 + 2 * j], us[b])));
        }
        IPA_TRY(push_hc());
        // p' <- p_lo + u^-1 p_hi ; b <- b_lo + u b_hi ; s <- s (x) (1, u)
        hipLaunchKernelGGL((k_ipa_round_fold<SF>), grid2(half > cnt ? half : cnt), dim3(g256), 0, st, p_cur, b_cur, s_cur, half, cnt,
                           d_hc, p_nxt, b_nxt, s_nxt);
        BZH_HIP_TRY(ctx, hipGetLastError());
        std::swap(p_cur, p_nxt);
        std::swap(b_cur, b_nxt);
        std::swap(s_cur, s_nxt);
    }
    std::vector<Fe<SF>> c(B);
    IPA_TRY(d2h_async(ctx, c.data(), p_cur, B * 32));
    IPA_TRY(d2h_finish(ctx));
    for (size_t b = 0; b < B; b++) {
        h_store<SF>(out_v + 4 * b, fe_from_mont(vm[b]));
        uint64_t sc[4];
        h_store<SF>(sc, fe_from_mont(c[b]));
        IPA_TRY(bzh_transcript_write_scalar(trs[b], sc));
        h_store<SF>(sc, fe_from_mont(f[b]));
        IPA_TRY(bzh_transcript_write_scalar(trs[b], sc));
    }
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
    // identity points cannot enter the transcript upstream (Blake2bRead::common_point errors): reject them
    auto is_identity = [](const uint64_t* p) {
        uint64_t any = 0;
        for (int i = 0; i < 8; i++) any |= p[i];
        return any == 0;
    };
    if (!h_decompress<C>(proof, S) || is_identity(S)) return BZH_E_VERIFY;
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
        if (is_identity(L) || is_identity(R)) return BZH_E_VERIFY;
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

// s[b][i] = c_b * prod_{j : bit (k-1-j) of i} u_(b,j)  for i < n;  s[b][n], s[b][n+1] = 0   (the verifier's G'_0 scalars)
template <class P>
__global__ void __launch_bounds__(256) k_ipa_verify_s(const uint32_t* __restrict__ cu, size_t n, unsigned k, uint32_t* __restrict__ s) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= n + 2) return;
    Fe<P> acc = fe_zero<P>();
    if (i < n) {
        const uint32_t* row = cu + b * (size_t)(k + 1) * 8;  // [c, u_0 .. u_(k-1)]
        acc = fe_load<P>(row);
        for (unsigned j = 0; j < k; j++)
            if ((i >> (k - 1 - j)) & 1) acc = fe_mul(acc, fe_load<P>(row + (size_t)(j + 1) * 8));
    }
    fe_store(s + (b * (n + 2) + i) * 8, acc);
}

// One workgroup per opening: sum_i scal[b][i] * pts[b][i] over nl (~100) ad-hoc points, one term per lane by
// double-and-add (scalars canonical, points affine Montgomery, (0,0) = identity padding), then an LDS tree over the
// lanes.  The 255 dependent doublings of a fresh point (~5 us each on a SIMD of its own) are the floor of any schedule
// for this sum, ~1.5 ms; work and workspace are linear in the batch (a shared Pippenger table over batch * nl points
// was quadratic in it).
static constexpr int kSmallMsmThreads = 128;
template <class C>
__global__ void __launch_bounds__(kSmallMsmThreads) k_ipa_small_msm(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ scal,
                                                                     size_t nl, uint32_t* __restrict__ out_xyz) {
    using P = typename C::Base;
    constexpr int T = kSmallMsmThreads;
    __shared__ Xyzz<P> sh[T];
    const int lane = threadIdx.x;
    const size_t b = blockIdx.x;
    Xyzz<P> sum = xyzz_identity<P>();
    for (size_t i = lane; i < nl; i += T) {
        Affine<P> q;
        q.x = fe_load<P>(pts + (b * nl + i) * 16);
        q.y = fe_load<P>(pts + (b * nl + i) * 16 + 8);
        if (aff_is_id(q)) continue;
        const uint32_t* sc = scal + (b * nl + i) * 8;
        int top = 255;
        while (top >= 0 && !((sc[top >> 5] >> (top & 31)) & 1)) top--;
        if (top < 0) continue;
        Xyzz<P> acc = xyzz_from_affine(q);
        for (int bit = top - 1; bit >= 0; bit--) {
            acc = xyzz_dbl_inl(acc);
            if ((sc[bit >> 5] >> (bit & 31)) & 1) xyzz_madd(acc, q);
        }
        xyzz_add(sum, acc);
    }
    sh[lane] = sum;
    __syncthreads();
    for (int d = T >> 1; d >= 1; d >>= 1) {
        if (lane < d) {
            Xyzz<P> o = sh[lane + d];
            xyzz_add(sum, o);
            sh[lane] = sum;
        }
        __syncthreads();
    }
    if (lane == 0) {
        Fe<P> X, Y, Z;
        xyzz_to_jacobian(sum, X, Y, Z);
        uint32_t* o = out_xyz + b * 24;
        fe_store(o, X);
        fe_store(o + 8, Y);
        fe_store(o + 16, Z);
    }
}

// For every opening b:  sum_i lc_scal[b][i] * lc_pts[b][i]  ==  <c_b * s(u_b), G>  ?   (the IPA verification equation with
// everything but the n-term G'_0 moved to the left: L_j, R_j, S, the opened commitment expanded into the proof's own
// commitments, G_0, U, W).  lc_pts: batch x nl affine canonical points ((0,0) = identity padding), lc_scal: canonical.
template <class C>
static int ipa_check_batch_t(bzh_ctx* ctx, const bzh_bases* bases, size_t batch, size_t nl, const uint64_t* lc_pts,
                             const uint64_t* lc_scal, const uint64_t* cu /* batch x (k+1) canonical: c, u_j */, int* ok) {
    using SF = typename CurveMeta<C>::SF;
    using PB = typename C::Base;
    const size_t n = bases->n - 2, B = batch;
    unsigned k = 0;
    while (((size_t)1 << k) < n) k++;
    if (((size_t)1 << k) != n || !B || !nl) return BZH_E_ARG;
    hipStream_t st = ctx->stream;
    const size_t na = B * nl;  // every opening's own nl points and scalars, side by side
    const size_t words = B * (n + 2) * 8 + B * (k + 1) * 8 + na * 16 + na * 8 + 2 * B * 24 + 256;
    void* arena = nullptr;
    IPA_TRY(ws_ensure(ctx, 4, words * 4, &arena));
    uint32_t* d_s = (uint32_t*)arena;
    uint32_t* d_cu = d_s + B * (n + 2) * 8;
    uint32_t* d_pts = d_cu + B * (k + 1) * 8;
    uint32_t* d_scal = d_pts + na * 16;
    uint32_t* d_out = d_scal + na * 8;
    // right side
    std::vector<Fe<SF>> cum(B * (k + 1));
    for (size_t i = 0; i < cum.size(); i++) cum[i] = fe_to_mont(h_load<SF>(cu + 4 * i));
    IPA_TRY(h2d_small(ctx, d_cu, cum.data(), cum.size() * 32));
    hipLaunchKernelGGL((k_ipa_verify_s<SF>), dim3((unsigned)((n + 2 + 255) / 256), (unsigned)B), dim3(256), 0, st, d_cu, n, k, d_s);
    BZH_HIP_TRY(ctx, hipGetLastError());
    IPA_TRY(msm_run(ctx, bases, d_s, n + 2, B, BZH_FORM_MONTGOMERY, d_out));
    // left side: B independent nl-term sums
    IPA_TRY(h2d_small(ctx, d_pts, lc_pts, na * 64));
    IPA_TRY(bases_to_montgomery(ctx, C::id, d_pts, na));
    IPA_TRY(h2d_small(ctx, d_scal, lc_scal, na * 32));
    hipLaunchKernelGGL((k_ipa_small_msm<C>), dim3((unsigned)B), dim3(kSmallMsmThreads), 0, st, d_pts, d_scal, nl, d_out + B * 24);
    BZH_HIP_TRY(ctx, hipGetLastError());
    std::vector<uint64_t> jac(2 * B * 12), lhs(B * 8), rhs(B * 8);
    IPA_TRY(d2h_async(ctx, jac.data(), d_out, 2 * B * 96));
    IPA_TRY(d2h_finish(ctx));
    h_jac_batch_to_affine_canonical<PB>(jac.data(), B, rhs.data());
    h_jac_batch_to_affine_canonical<PB>(jac.data() + B * 12, B, lhs.data());
    for (size_t b = 0; b < B; b++) ok[b] = memcmp(&lhs[b * 8], &rhs[b * 8], 64) == 0;
    return BZH_OK;
}

int ipa_check_batch(bzh_ctx* ctx, const bzh_bases* bases, size_t batch, size_t nl, const uint64_t* lc_pts, const uint64_t* lc_scal,
                    const uint64_t* cu, int* ok) {
    switch (bases->curve) {
        case BZH_CURVE_VESTA: return ipa_check_batch_t<VestaCurve>(ctx, bases, batch, nl, lc_pts, lc_scal, cu, ok);
        case BZH_CURVE_PALLAS: return ipa_check_batch_t<PallasCurve>(ctx, bases, batch, nl, lc_pts, lc_scal, cu, ok);
        case BZH_CURVE_BN254: return ipa_check_batch_t<Bn254Curve>(ctx, bases, batch, nl, lc_pts, lc_scal, cu, ok);
    }
    return BZH_E_ARG;
}

// pasta_curves from_bytes for one compressed point (host): false = not on the curve / non-canonical
bool point_decompress(int curve, const uint8_t* in, uint64_t* xy_canonical) {
    switch (curve) {
        case BZH_CURVE_VESTA: return h_decompress<VestaCurve>(in, xy_canonical);
        case BZH_CURVE_PALLAS: return h_decompress<PallasCurve>(in, xy_canonical);
        case BZH_CURVE_BN254: return h_decompress<Bn254Curve>(in, xy_canonical);
    }
    return false;
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

int ipa_open(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_polys, size_t batch, const uint64_t* blinds,
             const uint64_t* x3s, const uint8_t* rng_bytes, size_t rng_stride, bzh_transcript* const* trs, uint64_t* out_v,
             const uint32_t* d_raw_in) {
    switch (bases->curve) {
        case BZH_CURVE_VESTA: return ipa_open_t<VestaCurve>(ctx, bases, d_polys, batch, blinds, x3s, rng_bytes, rng_stride, trs, out_v, d_raw_in);
        case BZH_CURVE_PALLAS: return ipa_open_t<PallasCurve>(ctx, bases, d_polys, batch, blinds, x3s, rng_bytes, rng_stride, trs, out_v, d_raw_in);
        case BZH_CURVE_BN254: return ipa_open_t<Bn254Curve>(ctx, bases, d_polys, batch, blinds, x3s, rng_bytes, rng_stride, trs, out_v, d_raw_in);
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
