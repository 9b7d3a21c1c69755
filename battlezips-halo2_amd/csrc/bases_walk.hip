// bzh_bases_walk: the synthetic base set of the config-5 microbench ("bases = hash_to_curve-free cheap generator
// walk G_i = [i+1] G computed once"), made ON the device -- 2^24 points are 1 GB; producing them through n single-point MSMs and a
// host normalisation took minutes.  Thread t owns the 16 consecutive multiples [16t + 1 .. 16t + 16] G: its first one by a
// double-and-add over the index, the rest by mixed additions of G, then ONE inversion for the 16 (Montgomery's trick along the
// thread's run) -- ~70 field multiplications per point.  The oracle's orc_point_walk makes the same set on the host.
// bzh_bases_points reads points of any table back (canonical affine): what the tests compare.
#include "ctx.hpp"
#include "curve.cuh"

using namespace bzh;

namespace {

constexpr int WALK_L = 16;

template <class C>
__global__ void __launch_bounds__(64) k_bases_walk(uint32_t* __restrict__ out, const uint32_t* __restrict__ g_mont, size_t n) {
    using P = typename C::Base;
    const size_t t = blockIdx.x * (size_t)64 + threadIdx.x, first = t * WALK_L;
    if (first >= n) return;
    const int cnt = (int)min((size_t)WALK_L, n - first);
    Affine<P> g;
    g.x = fe_load<P>(g_mont);
    g.y = fe_load<P>(g_mont + 8);
    // [first + 1] G, most significant bit first
    const unsigned long long s = (unsigned long long)first + 1;
    Xyzz<P> acc = xyzz_identity<P>();
    for (int bit = 63 - __clzll((long long)s); bit >= 0; bit--) {
        acc = xyzz_dbl(acc);
        if ((s >> bit) & 1) xyzz_madd(acc, g);
    }
    Xyzz<P> pts[WALK_L];
    Fe<P> pref[WALK_L];
    Fe<P> run = fe_one<P>();
    for (int i = 0; i < cnt; i++) {
        if (i) xyzz_madd(acc, g);
        pts[i] = acc;
        pref[i] = run;
        if (!xyzz_is_id(acc)) run = fe_mul(run, fe_mul(acc.zz, acc.zzz));   // (identity only if the group order divides the index)
    }
    Fe<P> inv = fe_inv(run);
    for (int i = cnt - 1; i >= 0; i--) {
        uint32_t* o = out + (first + (size_t)i) * 16;
        if (xyzz_is_id(pts[i])) {
            fe_store<P>(o, fe_zero<P>());
            fe_store<P>(o + 8, fe_zero<P>());
            continue;
        }
        const Fe<P> di = fe_mul(inv, pref[i]);                      // 1 / (zz zzz)
        inv = fe_mul(inv, fe_mul(pts[i].zz, pts[i].zzz));
        fe_store<P>(o, fe_mul(pts[i].x, fe_mul(di, pts[i].zzz)));     // x / zz
        fe_store<P>(o + 8, fe_mul(pts[i].y, fe_mul(di, pts[i].zz)));  // y / zzz
    }
}

template <class P>
__global__ void __launch_bounds__(256) k_points_canonical(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, size_t elems) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < elems) fe_store<P>(out + i * 8, fe_from_mont(fe_load<P>(in + i * 8)));
}

template <class C>
int walk_t(bzh_ctx* ctx, const uint64_t* g_xy, int form, size_t n, uint32_t* d_out) {
    using P = typename C::Base;
    uint32_t g[16];
    for (int c = 0; c < 2; c++) {
        Fe<P> v;
        memcpy(v.l, g_xy + 4 * c, 32);
        if (form == BZH_FORM_CANONICAL) v = fe_to_mont(v);
        memcpy(g + 8 * c, v.l, 32);
    }
    void* d_g = nullptr;
    int rc = ws_ensure(ctx, 0, 64, &d_g);
    if (rc) return rc;
    if ((rc = h2d_small(ctx, d_g, g, 64))) return rc;
    const size_t threads = (n + WALK_L - 1) / WALK_L;
    hipLaunchKernelGGL((k_bases_walk<C>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, ctx->stream, d_out, (const uint32_t*)d_g, n);
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

}  // namespace

extern "C" {

int bzh_bases_walk(bzh_ctx* ctx, int curve, const uint64_t* g_xy, int form, size_t n, bzh_bases** out) {
    if (!ctx || !g_xy || !out || !n || n > ((size_t)1 << 28) || curve < 0 || curve > 2 || (form != BZH_FORM_CANONICAL && form != BZH_FORM_MONTGOMERY))
        return BZH_E_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    bzh_bases* b = new (std::nothrow) bzh_bases();
    if (!b) return BZH_E_OOM;
    b->curve = curve;
    b->n = n;
    b->device = ctx->device;
    if (hipMalloc((void**)&b->d_xy, n * 64) != hipSuccess) {
        delete b;
        ctx->last_error = "hipMalloc(bases walk)";
        return BZH_E_OOM;
    }
    int rc = curve == BZH_CURVE_VESTA ? walk_t<VestaCurve>(ctx, g_xy, form, n, b->d_xy)
             : curve == BZH_CURVE_PALLAS ? walk_t<PallasCurve>(ctx, g_xy, form, n, b->d_xy)
                                         : walk_t<Bn254Curve>(ctx, g_xy, form, n, b->d_xy);
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) {
        ctx->last_error = "k_bases_walk failed";
        rc = BZH_E_HIP;
    }
    if (rc) {
        (void)hipFree(b->d_xy);
        delete b;
        return rc;
    }
    *out = b;
    return BZH_OK;
}

int bzh_bases_points(bzh_ctx* ctx, const bzh_bases* bases, size_t first, size_t count, uint64_t* out_xy) {
    if (!ctx || !bases || !out_xy || bases->device != ctx->device) return BZH_E_ARG;
    const size_t total = bases->pre_c ? (bases->row_stride ? bases->row_stride : bases->n) * (size_t)bases->pre_nwin : bases->n;
    if (first > total || count > total - first) return BZH_E_RANGE;   // (a window table reads on through its rows)
    if (!count) return BZH_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* tmp = nullptr;
    int rc = ws_ensure(ctx, 1, count * 64, &tmp);
    if (rc) return rc;
    const size_t elems = count * 2;
    const uint32_t* src = bases->d_xy + first * 16;
    const dim3 grid((unsigned)((elems + 255) / 256));
    switch (bases->curve) {
        case BZH_CURVE_VESTA: hipLaunchKernelGGL((k_points_canonical<VestaCurve::Base>), grid, dim3(256), 0, ctx->stream, (uint32_t*)tmp, src, elems); break;
        case BZH_CURVE_PALLAS: hipLaunchKernelGGL((k_points_canonical<PallasCurve::Base>), grid, dim3(256), 0, ctx->stream, (uint32_t*)tmp, src, elems); break;
        default: hipLaunchKernelGGL((k_points_canonical<Bn254Curve::Base>), grid, dim3(256), 0, ctx->stream, (uint32_t*)tmp, src, elems); break;
    }
    BZH_HIP_TRY(ctx, hipGetLastError());
    BZH_HIP_TRY(ctx, hipMemcpyAsync(out_xy, tmp, count * 64, hipMemcpyDeviceToHost, ctx->stream));
    BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BZH_OK;
}

}  // extern "C"
