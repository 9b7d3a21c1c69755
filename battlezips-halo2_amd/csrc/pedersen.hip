// Batched native Pedersen commitment on the device: the reference's
//     pedersen_commit(message, trapdoor) = V * Fq::from_repr(message.to_repr()) + R * trapdoor     src/utils/pedersen.rs:17-28
// (V, R = hash_to_curve("battlezips:hash2curve")(b"v" / b"r") on Pallas, recomputed by the reference on EVERY call; called by
// every frontend and again inside ShotChip::synthesize, src/chips/shot.rs:319) for n (message, trapdoor) pairs in one launch.
//
// Both bases are fixed, so the sum is read off a ctx-owned direct-lookup table instead of being computed by a bucket method:
//     T[b][w][d - 1] = d * 2^(8w) * G_b        b in {V, R}, w < 32, 1 <= d <= 128       (8 192 affine points, 512 KB: L2-resident)
// A scalar is recoded into 32 signed 8-bit digits (|d| <= 128: the top window of a 255-bit scalar holds 7 bits + a carry), and a
// commitment is 64 XYZZ mixed additions of table entries (negated for d < 0) -- no doublings -- followed by one inversion for the
// affine result.  One lane per commitment: n = 2 816 (BASELINE configs[3]'s batch) is 44 waves.
// Integer VALU work like everything else here; algorithmic bytes: 64 B in (two scalars) + 64 B out per commitment.
#include "ctx.hpp"
#include "curve.cuh"
#include "pedersen_generators.hpp"

using namespace bzh;

namespace {

using PB = PallasCurve::Base;       // Fp: coordinates
constexpr int PED_C = 8, PED_NWIN = 32, PED_ENT = 128;

// one thread per table entry: d * 2^(8w) * G_b by doublings and a left-to-right double-and-add over d's 8 bits, then its own
// inversion.  Runs once per ctx (8 192 threads, ~ 1 ms).
__global__ void __launch_bounds__(64) k_pedersen_table(uint32_t* __restrict__ tbl, const uint32_t* __restrict__ gens /* 2 x (x, y) Montgomery */) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= 2 * PED_NWIN * PED_ENT) return;
    const int b = t / (PED_NWIN * PED_ENT), w = (t / PED_ENT) % PED_NWIN, d = t % PED_ENT + 1;
    Affine<PB> g;
    g.x = fe_load<PB>(gens + b * 16);
    g.y = fe_load<PB>(gens + b * 16 + 8);
    Xyzz<PB> base = xyzz_from_affine(g);
    for (int i = 0; i < PED_C * w; i++) base = xyzz_dbl(base);
    Xyzz<PB> acc = xyzz_identity<PB>();
    for (int bit = 7; bit >= 0; bit--) {
        acc = xyzz_dbl(acc);
        if ((d >> bit) & 1) xyzz_add(acc, base);
    }
    const Affine<PB> a = xyzz_to_affine(acc);
    fe_store<PB>(tbl + (size_t)t * 16, a.x);
    fe_store<PB>(tbl + (size_t)t * 16 + 8, a.y);
}

// signed 8-bit digit w of the canonical scalar s (8 limbs), carry threaded by the caller
__device__ __forceinline__ int ped_digit(const uint32_t* s, int w, int& carry) {
    int d = (int)((s[w >> 2] >> ((w & 3) * 8)) & 0xffu) + carry;
    carry = d > 128;
    return carry ? d - 256 : d;
}

__global__ void __launch_bounds__(64) k_pedersen_commit(const uint32_t* __restrict__ tbl, const uint32_t* __restrict__ scalars /* n x (m, t) canonical */,
                                                       size_t n, uint32_t* __restrict__ out /* n x (x, y) canonical */) {
    const size_t i = blockIdx.x * (size_t)64 + threadIdx.x;
    if (i >= n) return;
    Xyzz<PB> acc = xyzz_identity<PB>();
    for (int b = 0; b < 2; b++) {
        uint32_t s[8];
        const uint4 lo = *(const uint4*)(scalars + (i * 2 + b) * 8), hi = *(const uint4*)(scalars + (i * 2 + b) * 8 + 4);
        s[0] = lo.x, s[1] = lo.y, s[2] = lo.z, s[3] = lo.w, s[4] = hi.x, s[5] = hi.y, s[6] = hi.z, s[7] = hi.w;
        int carry = 0;
        for (int w = 0; w < PED_NWIN; w++) {
            const int d = ped_digit(s, w, carry);
            if (d == 0) continue;
            const int mag = d < 0 ? -d : d;
            const uint32_t* e = tbl + ((size_t)(b * PED_NWIN + w) * PED_ENT + (mag - 1)) * 16;
            Affine<PB> q;
            q.x = fe_load<PB>(e);
            q.y = fe_load<PB>(e + 8);
            if (d < 0) q.y = fe_neg(q.y);
            xyzz_madd(acc, q);
        }
        // scalars are < q < 2^255: the last window's carry is always absorbed (|d| <= 128 there)
    }
    const Affine<PB> a = xyzz_to_affine(acc);       // identity -> (0, 0)
    fe_store<PB>(out + i * 16, fe_from_mont(a.x));
    fe_store<PB>(out + i * 16 + 8, fe_from_mont(a.y));
}

static bool lt_modulus_fq(const uint64_t* v) {
    uint32_t w[8];
    memcpy(w, v, 32);
    for (int i = 7; i >= 0; i--) {
        if (w[i] < FqParams::mod(i)) return true;
        if (w[i] > FqParams::mod(i)) return false;
    }
    return false;
}

static int ensure_table(bzh_ctx* ctx) {
    if (ctx->ped_tbl) return BZH_OK;
    uint32_t gens[32];
    const uint64_t* src[2] = {PEDERSEN_GEN_V, PEDERSEN_GEN_R};
    for (int b = 0; b < 2; b++)
        for (int c = 0; c < 2; c++) {
            Fe<PB> v;
            memcpy(v.l, src[b] + 4 * c, 32);
            v = fe_to_mont(v);
            memcpy(gens + b * 16 + c * 8, v.l, 32);
        }
    void* d_gens = nullptr;
    int rc = ws_ensure(ctx, 0, sizeof(gens), &d_gens);
    if (rc) return rc;
    if ((rc = h2d_small(ctx, d_gens, gens, sizeof(gens)))) return rc;
    uint32_t* tbl = nullptr;
    const size_t entries = (size_t)2 * PED_NWIN * PED_ENT;
    BZH_HIP_TRY(ctx, hipMalloc((void**)&tbl, entries * 64));
    hipLaunchKernelGGL(k_pedersen_table, dim3((unsigned)(entries / 64)), dim3(64), 0, ctx->stream, tbl, (const uint32_t*)d_gens);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        (void)hipFree(tbl);
        ctx->last_error = std::string("k_pedersen_table: ") + hipGetErrorString(e);
        return BZH_E_HIP;
    }
    ctx->ped_tbl = tbl;
    return BZH_OK;
}

}  // namespace

extern "C" int bzh_pedersen_commit_batch(bzh_ctx* ctx, const uint64_t* messages, const uint64_t* trapdoors, size_t n, uint64_t* out_xy) {
    if (!ctx || !messages || !trapdoors || !out_xy || !n || n > ((size_t)1 << 24)) return BZH_E_ARG;
    for (size_t i = 0; i < n; i++) {
        // Fq::from_repr(message.to_repr()).unwrap() / a trapdoor that is an Fq: non-canonical reprs are refused (panic upstream)
        if (!lt_modulus_fq(messages + 4 * i) || !lt_modulus_fq(trapdoors + 4 * i)) return BZH_E_RANGE;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_table(ctx);
    if (rc) return rc;
    void* ws = nullptr;
    if ((rc = ws_ensure(ctx, 1, n * 128, &ws))) return rc;      // n x (64 B scalars in | 64 B point out)
    uint32_t *d_sc = (uint32_t*)ws, *d_out = (uint32_t*)ws + n * 16;
    char* slot = nullptr;
    if ((rc = h2d_stage(ctx, n * 64, &slot))) return rc;
    for (size_t i = 0; i < n; i++) {
        memcpy(slot + i * 64, messages + 4 * i, 32);
        memcpy(slot + i * 64 + 32, trapdoors + 4 * i, 32);
    }
    if ((rc = h2d_commit(ctx, d_sc, slot, n * 64))) return rc;
    {
        ScopedTimer t(ctx, BZH_T_POLY);
        hipLaunchKernelGGL(k_pedersen_commit, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->ped_tbl,
                           (const uint32_t*)d_sc, n, d_out);
    }
    BZH_HIP_TRY(ctx, hipGetLastError());
    if ((rc = d2h_async(ctx, out_xy, d_out, n * 64))) return rc;
    return d2h_finish(ctx);
}
