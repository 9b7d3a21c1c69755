// Params::new(k) for the reference's commitment scheme:
//   let params: Params<vesta::Affine> = Params::new(K);        benches/shot.rs:58, benches/board.rs:51,
//                                                               src/circuits/shot.rs:915, src/circuits/board.rs:907,
//                                                               src/wasm/circuit_wasm.rs:57,97,145,180 (on EVERY call)
// halo2_proofs 0.2.0 `poly::commitment::Params::new` (UPSTREAM, un-vendored) is a pure function of k:
//   g[i]       = hash_to_curve("Halo2-Parameters")(0u8 || i as u32 LE)          i < n = 2^k
//   g_lagrange = inverse group FFT of g (the commitments' Lagrange basis: g_lagrange[i] = sum_j L_i-coefficient_j g[j])
//   w = hash_to_curve(..)([1]),  u = hash_to_curve(..)([2])
// hash_to_curve is pasta_curves 0.4.1's: expand_message_xmd over BLAKE2b, simplified SWU on the 3-isogenous curve,
// the isogeny back (restated from RFC 9380 + Velu's formulas; the same construction on Pallas reproduces the
// reference's two `generator` known answers, src/utils/constants/fixed_bases/board_commit_{v,r}.rs:2941-2948, which
// tests/test_params_cpu.py checks through this very code).
//
// Host C++ for the hashing (setup work, threads over i), device kernels for the group FFT (n/2 log n scalar
// multiplications of points by roots of unity), results cached on disk keyed by (curve, k).
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "blake2b.hpp"
#include "circuit/hostfield.hpp"
#include "ctx.hpp"
#include "curve.cuh"

namespace {

using bzc::Fp;
using bzc::Fq;

// ---- hash_to_curve (host) -----------------------------------------------------------------------------------------
// iso-curve y^2 = x^3 + A x + 1265 and the x-coordinate x0 of the isogeny's kernel (the rational root of the
// 3-division polynomial 3x^4 + 6A x^2 + 12B x - A^2); checked at first use: psi3(x0) = 0, the Velu image has j = 0 and
// (1/3)^6 B' = 5.
template <class F>
struct IsoParams;
template <>
struct IsoParams<Fp> {   // Pallas (base field Fp)
    static constexpr const char* curve_id = "pallas";
    static constexpr uint64_t A[4] = {0x92bb4b0b657a014bull, 0xb74134581a27a59full, 0x49be2d7258370742ull, 0x18354a2eb0ea8c9cull};
    static constexpr uint64_t X0[4] = {0x6a57031b4ba19471ull, 0x4301a71d1ff0c7cdull, 0x52cfc0198fdb5ac3ull, 0x115468c111fb3180ull};
};
template <>
struct IsoParams<Fq> {   // Vesta (base field Fq)
    static constexpr const char* curve_id = "vesta";
    static constexpr uint64_t A[4] = {0xc515ad7242eaa6b1ull, 0x9673928c7d01b212ull, 0x81639c4d96f78773ull, 0x267f9b2ee592271aull};
    static constexpr uint64_t X0[4] = {0xea8f4dd1286f2e8cull, 0xbf4c98bd6fef5204ull, 0x75d5c33ad251d4a6ull, 0x1ae90dbd54bf6d15ull};
};

template <class F>
struct Iso {
    F a, b, z, x0, t, u, s2, s3;   // SWU Z = -13; Velu constants t, u; scaling s = 1/3
    bool ok = false;
    Iso() {
        F::from_limbs(IsoParams<F>::A, &a);
        F::from_limbs(IsoParams<F>::X0, &x0);
        b = F::from_u64(1265);
        z = -F::from_u64(13);
        const F x0sq = x0.sqr();
        // psi3(x0) = 3 x0^4 + 6 A x0^2 + 12 B x0 - A^2
        const F psi = F::from_u64(3) * x0sq.sqr() + F::from_u64(6) * a * x0sq + F::from_u64(12) * b * x0 - a.sqr();
        const F y0sq = x0sq * x0 + a * x0 + b;
        t = F::from_u64(6) * x0sq + a.dbl();
        u = y0sq.dbl().dbl();
        const F w = u + x0 * t;
        const F a_img = a - F::from_u64(5) * t, b_img = b - F::from_u64(7) * w;
        const F s = F::from_u64(3).inv();
        s2 = s.sqr();
        s3 = s2 * s;
        ok = psi.is_zero() && a_img.is_zero() && (s3.sqr() * b_img == F::from_u64(5));
    }
};
template <class F>
static const Iso<F>& iso() {
    static const Iso<F> v;
    return v;
}

template <class F>
struct Pt {
    F x, y;
    bool inf = false;
};
template <class F>
static Pt<F> iso_add(const Pt<F>& p, const Pt<F>& q) {   // on y^2 = x^3 + a x + b
    if (p.inf) return q;
    if (q.inf) return p;
    F lambda;
    if (p.x == q.x) {
        if (p.y != q.y || p.y.is_zero()) return Pt<F>{F::zero(), F::zero(), true};
        lambda = (F::from_u64(3) * p.x.sqr() + iso<F>().a) * p.y.dbl().inv();
    } else {
        lambda = (q.y - p.y) * (q.x - p.x).inv();
    }
    Pt<F> r;
    r.x = lambda.sqr() - p.x - q.x;
    r.y = lambda * (p.x - r.x) - p.y;
    return r;
}
// simplified SWU (RFC 9380 6.6.2, AB != 0) onto the iso-curve
template <class F>
static Pt<F> map_to_curve_simple_swu(const F& uu) {
    const Iso<F>& I = iso<F>();
    const F zu2 = I.z * uu.sqr();
    const F ta = zu2.sqr() + zu2;
    F x1;
    if (ta.is_zero()) x1 = I.b * (I.z * I.a).inv();
    else x1 = (-I.b) * I.a.inv() * (F::one() + ta.inv());
    const F gx1 = x1.sqr() * x1 + I.a * x1 + I.b;
    Pt<F> r;
    F y;
    if (gx1.sqrt(&y)) {
        r.x = x1;
    } else {
        r.x = zu2 * x1;
        const F gx2 = r.x.sqr() * r.x + I.a * r.x + I.b;
        if (!gx2.sqrt(&y)) y = F::zero();   // cannot happen: one of gx1, gx2 is a square
    }
    if (uu.is_odd() != y.is_odd()) y = -y;   // sgn0(u) == sgn0(y)
    r.y = y;
    return r;
}
// the normalised 3-isogeny (Velu, then (X, Y) -> (X / 9, Y / 27)) from the iso-curve to y^2 = x^3 + 5
template <class F>
static Pt<F> iso_map(const Pt<F>& p) {
    if (p.inf) return p;
    const Iso<F>& I = iso<F>();
    const F d = p.x - I.x0;
    if (d.is_zero()) return Pt<F>{F::zero(), F::zero(), true};
    const F di = d.inv(), di2 = di.sqr();
    const F X = p.x + I.t * di + I.u * di2;
    const F Y = p.y * (F::one() - I.t * di2 - I.u.dbl() * di2 * di);
    return Pt<F>{I.s2 * X, I.s3 * Y, false};
}
template <class F>
static F field_from_be64(const uint8_t* b) {   // OS2IP of 64 big-endian bytes, mod p
    uint64_t lo[4], hi[4];
    for (int i = 0; i < 4; i++) {
        uint64_t l = 0, h = 0;
        for (int j = 0; j < 8; j++) {
            l |= (uint64_t)b[63 - (8 * i + j)] << (8 * j);
            h |= (uint64_t)b[31 - (8 * i + j)] << (8 * j);
        }
        lo[i] = l, hi[i] = h;
    }
    // any 256-bit x: mul(x, R^2) = x R mod p (Montgomery form of x mod p)
    const F lom = F::mul(F{{lo[0], lo[1], lo[2], lo[3]}}, F::r2());
    const F him = F::mul(F::mul(F{{hi[0], hi[1], hi[2], hi[3]}}, F::r2()), F::r2());   // (hi 2^256) R
    return lom + him;
}
// CurveExt::hash_to_curve(domain_prefix)(message), affine; false only for the (never observed) identity result
template <class F>
static bool hash_to_curve_t(const std::string& domain_prefix, const uint8_t* msg, size_t len, F* ox, F* oy) {
    const std::string dst = domain_prefix + "-" + IsoParams<F>::curve_id + "_XMD:BLAKE2b_SSWU_RO_";
    std::vector<uint8_t> dst_prime(dst.begin(), dst.end());
    dst_prime.push_back((uint8_t)dst.size());
    const uint8_t nopersonal[16] = {0};
    uint8_t b0[64], b1[64], b2[64];
    {
        bzh::Blake2b h;
        h.init(64, nopersonal);
        const uint8_t zpad[128] = {0};
        h.update(zpad, 128);
        h.update(msg, len);
        const uint8_t lib[3] = {0, 128, 0};
        h.update(lib, 3);
        h.update(dst_prime.data(), dst_prime.size());
        h.finalize(b0);
    }
    {
        bzh::Blake2b h;
        h.init(64, nopersonal);
        h.update(b0, 64);
        const uint8_t one = 1;
        h.update(&one, 1);
        h.update(dst_prime.data(), dst_prime.size());
        h.finalize(b1);
    }
    {
        bzh::Blake2b h;
        h.init(64, nopersonal);
        uint8_t x[64];
        for (int i = 0; i < 64; i++) x[i] = b0[i] ^ b1[i];
        h.update(x, 64);
        const uint8_t two = 2;
        h.update(&two, 1);
        h.update(dst_prime.data(), dst_prime.size());
        h.finalize(b2);
    }
    const F u0 = field_from_be64<F>(b1), u1 = field_from_be64<F>(b2);
    const Pt<F> r = iso_map(iso_add(map_to_curve_simple_swu(u0), map_to_curve_simple_swu(u1)));
    if (r.inf) return false;
    *ox = r.x;
    *oy = r.y;
    return true;
}

// ---- group FFT (device) ------------------------------------------------------------------------------------------
using namespace bzh;

template <class C>
__device__ Xyzz<typename C::Base> point_scalar_mul(const Xyzz<typename C::Base>& p, const uint32_t* k /* 8 canonical limbs */) {
    using P = typename C::Base;
    Xyzz<P> acc = xyzz_identity<P>();
    bool started = false;
    for (int bit = 255; bit >= 0; bit--) {
        if (started) acc = xyzz_dbl(acc);
        if ((k[bit >> 5] >> (bit & 31)) & 1u) {
            if (started) xyzz_add(acc, p);
            else acc = p, started = true;
        }
    }
    return acc;
}
template <class P>
__device__ __forceinline__ Xyzz<P> xyzz_ld(const uint32_t* a) {
    Xyzz<P> v;
    v.x = fe_load<P>(a), v.y = fe_load<P>(a + 8), v.zz = fe_load<P>(a + 16), v.zzz = fe_load<P>(a + 24);
    return v;
}
template <class P>
__device__ __forceinline__ void xyzz_st(uint32_t* a, const Xyzz<P>& v) {
    fe_store(a, v.x), fe_store(a + 8, v.y), fe_store(a + 16, v.zz), fe_store(a + 24, v.zzz);
}
// out[bitrev(i)] = g[i] (affine Montgomery -> XYZZ)
template <class C>
__global__ void __launch_bounds__(256) k_gfft_load(const uint32_t* __restrict__ g_xy, uint32_t* __restrict__ work, unsigned log_n) {
    using P = typename C::Base;
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x, n = (size_t)1 << log_n;
    if (i >= n) return;
    const size_t r = log_n ? (size_t)(__brevll((unsigned long long)i) >> (64 - log_n)) : 0;
    Affine<P> a;
    a.x = fe_load<P>(g_xy + i * 16), a.y = fe_load<P>(g_xy + i * 16 + 8);
    xyzz_st<P>(work + r * 32, xyzz_from_affine(a));
}
// one radix-2 decimation-in-time stage: butterflies (i0, i0 + m), twiddle tw[j * stride] (canonical limbs)
template <class C>
__global__ void __launch_bounds__(64) k_gfft_stage(uint32_t* __restrict__ work, size_t n, size_t m, const uint32_t* __restrict__ tw, size_t stride) {
    using P = typename C::Base;
    const size_t idx = blockIdx.x * (size_t)64 + threadIdx.x;
    if (idx >= n / 2) return;
    const size_t grp = idx / m, j = idx - grp * m, i0 = grp * 2 * m + j, i1 = i0 + m;
    const Xyzz<P> a = xyzz_ld<P>(work + i0 * 32);
    Xyzz<P> t = xyzz_ld<P>(work + i1 * 32);
    if (j) t = point_scalar_mul<C>(t, tw + j * stride * 8);
    Xyzz<P> s = a, d = a;
    xyzz_add(s, t);
    Xyzz<P> nt = t;
    nt.y = fe_neg(t.y);
    xyzz_add(d, nt);
    xyzz_st<P>(work + i0 * 32, s);
    xyzz_st<P>(work + i1 * 32, d);
}
// out_xy[i] = affine([scale] work[i])
template <class C>
__global__ void __launch_bounds__(64) k_gfft_finish(const uint32_t* __restrict__ work, size_t n, const uint32_t* __restrict__ scale,
                                                     uint32_t* __restrict__ out_xy) {
    using P = typename C::Base;
    const size_t i = blockIdx.x * (size_t)64 + threadIdx.x;
    if (i >= n) return;
    const Xyzz<P> v = point_scalar_mul<C>(xyzz_ld<P>(work + i * 32), scale);
    const Affine<P> a = xyzz_to_affine(v);
    fe_store(out_xy + i * 16, a.x);
    fe_store(out_xy + i * 16 + 8, a.y);
}

static std::string default_cache_dir() {
    const char* env = getenv("BZH_CACHE_DIR");
    if (env && *env) return env;
    Dl_info info;
    if (dladdr((void*)&default_cache_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t s = p.rfind('/');
        return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/.bzh2_cache";
    }
    return ".bzh2_cache";
}

}  // namespace

struct bzh_params {
    unsigned k = 0;
    size_t n = 0;
    std::vector<uint64_t> g, g_lagrange;   // n x 8 canonical limbs (x || y)
    uint64_t w[8], u[8];
    bzh_bases *bases = nullptr, *bases_lagrange = nullptr;   // (g | u | w) and (g_lagrange | u | w), window tables
    bool from_cache = false;
};

extern "C" {

int bzh_hash_to_curve(int curve, const char* domain_prefix, const uint8_t* msg, size_t len, uint64_t* out_xy) {
    if (!domain_prefix || (!msg && len) || !out_xy) return BZH_E_ARG;
    if (curve == BZH_CURVE_PALLAS) {
        if (!iso<Fp>().ok) return BZH_E_HIP;
        Fp x, y;
        if (!hash_to_curve_t<Fp>(domain_prefix, msg, len, &x, &y)) return BZH_E_RANGE;
        x.to_limbs(out_xy), y.to_limbs(out_xy + 4);
        return BZH_OK;
    }
    if (curve == BZH_CURVE_VESTA) {
        if (!iso<Fq>().ok) return BZH_E_HIP;
        Fq x, y;
        if (!hash_to_curve_t<Fq>(domain_prefix, msg, len, &x, &y)) return BZH_E_RANGE;
        x.to_limbs(out_xy), y.to_limbs(out_xy + 4);
        return BZH_OK;
    }
    return BZH_E_ARG;
}

// g, w, u of Params::new(k) on the host (no device): g_xy = n x 8 canonical limbs
int bzh_params_generators(unsigned k, uint64_t* g_xy, uint64_t* w_xy, uint64_t* u_xy, unsigned threads) {
    if (k > 24 || (!g_xy && !w_xy && !u_xy)) return BZH_E_ARG;
    if (!iso<Fq>().ok) return BZH_E_HIP;
    const size_t n = (size_t)1 << k;
    const std::string domain = "Halo2-Parameters";
    std::atomic<int> bad{0};
    if (g_xy) {
        if (!threads) threads = std::max(1u, std::min(32u, host_thread_budget()));
        threads = (unsigned)std::min<size_t>(threads, n);
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (;;) {
                const size_t lo = next.fetch_add(256);
                if (lo >= n) return;
                for (size_t i = lo; i < std::min(n, lo + 256); i++) {
                    const uint8_t msg[5] = {0, (uint8_t)i, (uint8_t)(i >> 8), (uint8_t)(i >> 16), (uint8_t)(i >> 24)};
                    Fq x, y;
                    if (!hash_to_curve_t<Fq>(domain, msg, 5, &x, &y)) {
                        bad = 1;
                        continue;
                    }
                    x.to_limbs(g_xy + 8 * i), y.to_limbs(g_xy + 8 * i + 4);
                }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
    }
    for (int which = 1; which <= 2; which++) {
        uint64_t* out = which == 1 ? w_xy : u_xy;
        if (!out) continue;
        const uint8_t msg[1] = {(uint8_t)which};
        Fq x, y;
        if (!hash_to_curve_t<Fq>(domain, msg, 1, &x, &y)) return BZH_E_RANGE;
        x.to_limbs(out), y.to_limbs(out + 4);
    }
    return bad ? BZH_E_RANGE : BZH_OK;
}

// g_lagrange = inverse group FFT of g, on the device: g_xy / out_xy host arrays of n x 8 canonical limbs
int bzh_group_ifft(bzh_ctx* ctx, int curve, const uint64_t* g_xy, unsigned k, uint64_t* out_xy) {
    if (!ctx || !g_xy || !out_xy || curve != BZH_CURVE_VESTA || k > 24) return BZH_E_ARG;
    using C = VestaCurve;
    const size_t n = (size_t)1 << k;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // twiddles omega^-j (j < n / 2) and the scale n^-1, canonical, computed on the host in the scalar field (Fp)
    uint64_t wl[4];
    int rc = bzh_field_omega(BZH_FIELD_FP, k, BZH_FORM_CANONICAL, wl);
    if (rc) return rc;
    Fp omega, pw = Fp::one();
    Fp::from_limbs(wl, &omega);
    const Fp omega_inv = omega.inv();
    std::vector<uint64_t> tw(std::max<size_t>(n / 2, 1) * 4 + 4);
    for (size_t j = 0; j < n / 2; j++) {
        pw.to_limbs(&tw[4 * j]);
        pw = pw * omega_inv;
    }
    Fp::from_u64((uint64_t)n).inv().to_limbs(&tw[std::max<size_t>(n / 2, 1) * 4]);
    uint32_t *d_g = nullptr, *d_work = nullptr, *d_tw = nullptr, *d_out = nullptr;
    auto cleanup = [&] {
        if (d_g) (void)hipFree(d_g);
        if (d_work) (void)hipFree(d_work);
        if (d_tw) (void)hipFree(d_tw);
        if (d_out) (void)hipFree(d_out);
    };
    hipError_t e = hipMalloc((void**)&d_g, n * 64);
    if (e == hipSuccess) e = hipMalloc((void**)&d_work, n * 128);
    if (e == hipSuccess) e = hipMalloc((void**)&d_tw, tw.size() * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, n * 64);
    if (e != hipSuccess) {
        cleanup();
        return BZH_E_OOM;
    }
    e = hipMemcpyAsync(d_g, g_xy, n * 64, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        cleanup();
        return BZH_E_HIP;
    }
    rc = bases_to_montgomery(ctx, curve, d_g, n);
    if (rc) {
        cleanup();
        return rc;
    }
    hipLaunchKernelGGL((k_gfft_load<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_g, d_work, k);
    for (size_t m = 1; m < n; m <<= 1) {
        const size_t stride = n / (2 * m);
        hipLaunchKernelGGL((k_gfft_stage<C>), dim3((unsigned)((n / 2 + 63) / 64)), dim3(64), 0, ctx->stream, d_work, n, m, d_tw, stride);
    }
    hipLaunchKernelGGL((k_gfft_finish<C>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, d_work, n,
                       d_tw + std::max<size_t>(n / 2, 1) * 8, d_out);
    e = hipGetLastError();
    if (e == hipSuccess) {
        rc = field_convert(ctx, BZH_FIELD_FQ, d_out, n * 2, 0);   // coordinates: Montgomery -> canonical
        if (rc) {
            cleanup();
            return rc;
        }
        e = hipMemcpyAsync(out_xy, d_out, n * 64, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) {
        ctx->last_error = std::string("bzh_group_ifft: ") + hipGetErrorString(e);
        return BZH_E_HIP;
    }
    return BZH_OK;
}

int bzh_params_free(bzh_ctx* ctx, bzh_params* p) {
    if (!p) return BZH_OK;
    if (ctx) {
        if (p->bases) bzh_bases_free(ctx, p->bases);
        if (p->bases_lagrange) bzh_bases_free(ctx, p->bases_lagrange);
    }
    delete p;
    return BZH_OK;
}

// Params::new(k), cached under cache_dir (NULL: next to the library, or $BZH_CACHE_DIR; "" disables the cache);
// uploads (g | u | w) and (g_lagrange | u | w) with window tables of `window_bits` (0: the planner's).
int bzh_params_create(bzh_ctx* ctx, unsigned k, const char* cache_dir, int window_bits, bzh_params** out) {
    if (!ctx || !out || k < 1 || k > 24) return BZH_E_ARG;
    std::unique_ptr<bzh_params> p(new bzh_params());
    p->k = k;
    p->n = (size_t)1 << k;
    const size_t n = p->n;
    p->g.assign(n * 8, 0);
    p->g_lagrange.assign(n * 8, 0);
    const std::string dir = cache_dir ? std::string(cache_dir) : default_cache_dir();
    const std::string path = dir + "/params_vesta_k" + std::to_string(k) + "_v1.bin";
    bool loaded = false;
    if (!dir.empty()) {
        if (FILE* f = fopen(path.c_str(), "rb")) {
            uint64_t hdr[2] = {0, 0};
            loaded = fread(hdr, 8, 2, f) == 2 && hdr[0] == 0x315352535a42ull /* "BZSRS1" */ && hdr[1] == k &&
                     fread(p->g.data(), 8, n * 8, f) == n * 8 && fread(p->g_lagrange.data(), 8, n * 8, f) == n * 8 &&
                     fread(p->w, 8, 8, f) == 8 && fread(p->u, 8, 8, f) == 8;
            fclose(f);
        }
    }
    if (!loaded) {
        int rc = bzh_params_generators(k, p->g.data(), p->w, p->u, 0);
        if (rc) return rc;
        rc = bzh_group_ifft(ctx, BZH_CURVE_VESTA, p->g.data(), k, p->g_lagrange.data());
        if (rc) return rc;
        if (!dir.empty()) {
            mkdir(dir.c_str(), 0755);
            const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
            if (FILE* f = fopen(tmp.c_str(), "wb")) {
                const uint64_t hdr[2] = {0x315352535a42ull, k};
                const bool ok = fwrite(hdr, 8, 2, f) == 2 && fwrite(p->g.data(), 8, n * 8, f) == n * 8 &&
                                fwrite(p->g_lagrange.data(), 8, n * 8, f) == n * 8 && fwrite(p->w, 8, 8, f) == 8 && fwrite(p->u, 8, 8, f) == 8;
                fclose(f);
                if (ok) rename(tmp.c_str(), path.c_str());
                else remove(tmp.c_str());
            }
        }
    }
    p->from_cache = loaded;
    // (g | u | w) and (g_lagrange | u | w | g_0): sum_r g_lagrange[r] = g_0 (the Lagrange polynomials sum to 1), so a column with
    // a long constant stretch is committed as (column - c) + c * g_0 -- the grand products of the permutation argument stay at
    // their final value from the last copy constraint to the blinding rows (csrc/prove.hip: commit, shift_row)
    std::vector<uint64_t> tbl((n + 3) * 8);
    for (int which = 0; which < 2; which++) {
        memcpy(tbl.data(), which ? p->g_lagrange.data() : p->g.data(), n * 64);
        memcpy(&tbl[n * 8], p->u, 64);
        memcpy(&tbl[(n + 1) * 8], p->w, 64);
        memcpy(&tbl[(n + 2) * 8], p->g.data(), 64);
        bzh_bases* h = nullptr;
        int rc = bzh_bases_upload(ctx, BZH_CURVE_VESTA, tbl.data(), n + 2 + (which ? 1 : 0), BZH_FORM_CANONICAL, BZH_MEM_HOST, &h);
        if (!rc) rc = bzh_bases_precompute(ctx, h, window_bits);
        if (rc) {
            if (h) bzh_bases_free(ctx, h);
            bzh_params_free(ctx, p.release());
            return rc;
        }
        (which ? p->bases_lagrange : p->bases) = h;
    }
    *out = p.release();
    return BZH_OK;
}

int bzh_params_bases(const bzh_params* p, bzh_bases** g, bzh_bases** g_lagrange) {
    if (!p) return BZH_E_ARG;
    if (g) *g = p->bases;
    if (g_lagrange) *g_lagrange = p->bases_lagrange;
    return BZH_OK;
}
// host copies (canonical affine): any of the outputs may be NULL; g_xy / g_lagrange_xy hold n x 8 limbs
int bzh_params_points(const bzh_params* p, uint64_t* g_xy, uint64_t* g_lagrange_xy, uint64_t* w_xy, uint64_t* u_xy, int* from_cache) {
    if (!p) return BZH_E_ARG;
    if (g_xy) memcpy(g_xy, p->g.data(), p->n * 64);
    if (g_lagrange_xy) memcpy(g_lagrange_xy, p->g_lagrange.data(), p->n * 64);
    if (w_xy) memcpy(w_xy, p->w, 64);
    if (u_xy) memcpy(u_xy, p->u, 64);
    if (from_cache) *from_cache = p->from_cache ? 1 : 0;
    return BZH_OK;
}

}  // extern "C"
