// Radix-2 NTT / iNTT / coset NTT on gfx950, multi-pass with LDS-resident tiles.
//
// Replaces halo2_proofs 0.2.0 `arithmetic::best_fft` and the EvaluationDomain
// wrappers `ifft`, `coeff_to_extended`, `extended_to_coeff` (UPSTREAM,
// un-vendored: Cargo.lock:382-385) on the create_proof path entered at
// benches/shot.rs:68 / benches/board.rs:61-68: ~17 iNTT(n) + ~18 coset NTT(8n)
// + 1 extended iNTT per proof.
//
// N = 2^k is factored as A * R * B per pass (A = radices already done, R = this
// pass, B = radices still to do).  In place, element (a, r, j) lives at
// a*R*B + r*B + j.  Pass p transforms the r axis inside an LDS tile of R x W
// elements (W adjacent j, i.e. W*32-byte contiguous runs), multiplies by the
// inter-pass twiddle omega_N^(A*k_r*j) and writes back in place.  The last pass
// (B = 1) reads W rows whose digit-reversed indices are consecutive and writes
// X[digitrev(a) + A*k_r], which makes the final order natural with W*32-byte
// contiguous runs on both sides.  Modular-integer work, no MFMA.
//
// Inside a tile the radix-2 stages run three at a time on 8 rows held in registers (stage_group): one LDS
// round trip and one barrier per three stages, and the twiddles equal to 1 in the first group are skipped.
// The inter-pass twiddles omega_N^(A*row*j) come from a per-pass R x B table read like the data (one multiply
// per element instead of two; domains up to 2^20), the two-level power table above that.  coeff_to_extended
// (ntt_run_padded) reads only the 2^k coefficients: the first log2(8) stages of a zero-padded vector only
// replicate values, so pass 0 skips them along with 7/8 of its loads and the padded copy.
// Measured (tools/ubench_ntt.py, profiles/r02_ubench_ntt.txt): the kernel runs at ~70 % of the VALU bound its
// own multiply/add counts give at two waves per SIMD.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ntt_common.cuh"

namespace bzh {

// the in-wave pass kernel lives in csrc/ntt_wave.cuh, instantiated per field in csrc/ntt_wave_{fp,fq,bnfr}.hip
template <class P>
void launch_ntt_pass_wave(const NttPassArgs& aa, unsigned tiles, unsigned nb, hipStream_t stream);
extern template void launch_ntt_pass_wave<FpParams>(const NttPassArgs&, unsigned, unsigned, hipStream_t);
extern template void launch_ntt_pass_wave<FqParams>(const NttPassArgs&, unsigned, unsigned, hipStream_t);
extern template void launch_ntt_pass_wave<BnFrParams>(const NttPassArgs&, unsigned, unsigned, hipStream_t);

// C consecutive radix-2 DIT stages (s0 .. s0+C-1) on 2^C tile rows held in registers: one LDS read and one
// LDS write per element for the C stages.  Rows of a group: (hi << (s0+C)) | (m << s0) | lo, m < 2^C.
// FIRST (s0 == 0): the stage twiddles are the compile-time powers of omega_2^C, and the ones equal to 1 are skipped.
template <class P, int C, bool FIRST>
__device__ __forceinline__ void stage_group(uint4* tile, const NttPassArgs& g, int s0, int items, int tid) {
    constexpr int E = 1 << C;
    const int r = g.r, logW = g.logW, W = 1 << logW;
    const int ngroups = items >> C;
    const int stride = 1 << (s0 + logW);
    for (int gi = tid; gi < ngroups; gi += kNttThreads) {
        const int col = gi & (W - 1), gg = gi >> logW;
        const int lo = gg & ((1 << s0) - 1), hi = gg >> s0;
        const int base = ((((hi << C) << s0) | lo) << logW) + col;
        Fe<P> x[E];
#pragma unroll
        for (int m = 0; m < E; m++) x[m] = tile_get<P>(tile, base + m * stride);
#pragma unroll
        for (int t = 0; t < C; t++) {
            const int sh = r - 1 - (s0 + t);  // stage s = s0 + t uses omega_R^(j << sh), j < 2^s
#pragma unroll
            for (int b = 0; b < E / 2; b++) {
                const int jm = b & ((1 << t) - 1), top = ((b >> t) << (t + 1)) + jm, bot = top + (1 << t);
                Fe<P> v = x[bot];
                if (!(FIRST && jm == 0)) {
                    const int j = FIRST ? jm : (lo | (jm << s0));
                    v = fe_mul(v, fe_load<P>(g.sub_tw + ((size_t)j << sh) * 8));
                }
                x[bot] = fe_sub(x[top], v);
                x[top] = fe_add(x[top], v);
            }
            // keep the next stage's twiddle loads behind this stage: hoisting all of them costs ~90 VGPRs and spills
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < E; m++) tile_put(tile, base + m * stride, x[m]);
    }
}

template <class P>
__global__ void __launch_bounds__(kNttThreads) __attribute__((amdgpu_waves_per_eu(2))) k_ntt_pass(NttPassArgs g) {
    __shared__ __align__(16) uint4 tile[2 * kTileElems];
    const int tid = threadIdx.x, T = kNttThreads;
    const int r = g.r, R = 1 << r, logW = g.logW, W = 1 << logW;
    const size_t N = (size_t)1 << g.log_n;
    const uint32_t* vin = g.src + (size_t)blockIdx.y * N * 8;
    uint32_t* vec = g.dst + (size_t)blockIdx.y * N * 8;
    const size_t tile_id = blockIdx.x;
    const int items = R << logW;

    if (!g.last) {
        const size_t B = (size_t)1 << g.logB;
        const size_t tiles_per_a = B >> logW;
        const size_t a = tile_id / tiles_per_a, j0 = (tile_id - a * tiles_per_a) << logW;
        const uint32_t* base = vin + ((a << (r + g.logB)) + j0) * 8;
        if (g.nz) {  // A == 1, a == 0
            const uint32_t* sv = g.src + (size_t)blockIdx.y * (N >> g.nz) * 8 + j0 * 8;
            const int live = items >> g.nz;
            for (int it = tid; it < live; it += T) {
                const int col = it & (W - 1), row = it >> logW;
                Fe<P> v = fe_load<P>(sv + ((size_t)row * B + col) * 8);
                if (g.pre_lo) v = fe_mul(v, pow_table<P>(g.pre_lo, g.pre_hi, g.h, (size_t)row * B + j0 + col));
                if (g.cube_pre) {
                    const unsigned c3 = (unsigned)(((size_t)row * B + j0 + col) % 3);
                    if (c3) v = fe_mul(v, cube_const<P>(g, c3));
                }
                const int p0 = (int)(bitrev((uint32_t)row, r) << logW) + col;  // low nz bits of the tile row are zero
                for (int m = 0; m < (1 << g.nz); m++) tile_put(tile, p0 + (m << logW), v);
            }
        } else
        for (int it = tid; it < items; it += T) {
            const int col = it & (W - 1), row = it >> logW;
            Fe<P> v = fe_load<P>(base + ((size_t)row * B + col) * 8);
            if (g.pre_lo) v = fe_mul(v, pow_table<P>(g.pre_lo, g.pre_hi, g.h, (size_t)row * B + j0 + col));
            if (g.cube_pre) {
                const unsigned c3 = (unsigned)(((size_t)row * B + j0 + col) % 3);
                if (c3) v = fe_mul(v, cube_const<P>(g, c3));
            }
            tile_put(tile, (int)(bitrev((uint32_t)row, r) << logW) + col, v);
        }
    } else {
        // rows: d = tile_id*W + i  ->  a = digit-unreversed(d); element (a, rr) at a*R + rr
        for (int it = tid; it < items; it += T) {
            const int rr = it & (R - 1), i = it >> r;
            size_t d = (tile_id << logW) + i, a = 0;
            for (int q = 0; q < g.nprev; q++) {
                a = (a << g.prev_bits[q]) | (d & (((size_t)1 << g.prev_bits[q]) - 1));
                d >>= g.prev_bits[q];
            }
            Fe<P> v = fe_load<P>(vin + ((a << r) + rr) * 8);
            if (g.pre_lo) v = fe_mul(v, pow_table<P>(g.pre_lo, g.pre_hi, g.h, (size_t)rr));  // single-pass only
            if (g.cube_pre && (rr % 3)) v = fe_mul(v, cube_const<P>(g, (unsigned)(rr % 3)));
            tile_put(tile, (int)(bitrev((uint32_t)rr, r) << logW) + i, v);
        }
    }
    // radix-2 DIT stages over the tile rows, up to three stages per LDS round trip (stage_group)
    {
        int s0 = g.nz ? g.nz : (r < 3 ? r : 3);
        if (!g.nz) {
            __syncthreads();
            switch (s0) {
                case 1: stage_group<P, 1, true>(tile, g, 0, items, tid); break;
                case 2: stage_group<P, 2, true>(tile, g, 0, items, tid); break;
                default: stage_group<P, 3, true>(tile, g, 0, items, tid); break;
            }
        }
        int ngr = (r - s0 + 2) / 3;
        for (; ngr > 0; ngr--) {
            const int c = (r - s0 + ngr - 1) / ngr;  // split what is left as evenly as possible
            __syncthreads();
            switch (c) {
                case 1: stage_group<P, 1, false>(tile, g, s0, items, tid); break;
                case 2: stage_group<P, 2, false>(tile, g, s0, items, tid); break;
                default: stage_group<P, 3, false>(tile, g, s0, items, tid); break;
            }
            s0 += c;
        }
    }
    __syncthreads();
    if (!g.last) {
        const size_t B = (size_t)1 << g.logB;
        const size_t tiles_per_a = B >> logW;
        const size_t a = tile_id / tiles_per_a, j0 = (tile_id - a * tiles_per_a) << logW;
        uint32_t* base = vec + ((a << (r + g.logB)) + j0) * 8;
        for (int it = tid; it < items; it += T) {
            const int col = it & (W - 1), row = it >> logW;
            Fe<P> v = tile_get<P>(tile, it);
            if (g.tw_direct) {
                v = fe_mul(v, fe_load<P>(g.tw_direct + ((size_t)row * B + j0 + col) * 8));
            } else {
                const size_t e = ((size_t)row * (j0 + col)) << g.logA;
                if (e || g.tw_always) v = fe_mul(v, pow_table<P>(g.tw_lo, g.tw_hi, g.h, e));
            }
            fe_store(base + ((size_t)row * B + col) * 8, v);
        }
    } else {
        for (int it = tid; it < items; it += T) {
            const int i = it & (W - 1), kr = it >> logW;
            Fe<P> v = tile_get<P>(tile, it);
            const size_t k = ((size_t)kr << g.logA) + (tile_id << logW) + i;
            if (g.post_lo) v = fe_mul(v, pow_table<P>(g.post_lo, g.post_hi, g.h, k));
            if (g.cube_post) v = fe_mul(v, cube_const<P>(g, (unsigned)(k % 3)));
            ntt_store_out<P>(g, vec, (size_t)blockIdx.y, (size_t)1 << g.log_n, k, v);
        }
    }
}

// ---------------------------------------------------------------------------
// host: domain tables (cached per ctx) and pass planning
// ---------------------------------------------------------------------------
struct NttDomain {
    int field;
    unsigned log_n;
    int inverse;
    int has_shift;
    uint32_t omega[8];  // Montgomery, as given (before inversion)
    uint32_t shift[8];
    int h;
    uint32_t* d_tables = nullptr;  // one allocation
    size_t off_tw_lo, off_tw_hi, off_sc_lo, off_sc_hi;
    size_t off_tw_hi_scaled;  // omega^(e << h) * n^-1 (inverse domains): pass 0 of a multi-pass plain inverse
    int shift_is_cube;        // shift^3 == 1, shift != 1
    uint32_t cube[3][8];      // forward: 1, s, s^2 ; inverse: n^-1, n^-1 s^-1, n^-1 s^-2 (s = 1 without a shift)
    size_t off_sub[12];  // sub-NTT twiddles for radix bits 1..11
    size_t off_direct[4];  // per non-final pass: omega^(A*row*j) (x n^-1 on pass 0 of a plain inverse), R x B; 0 = none
};

struct NttCache {
    std::vector<NttDomain> doms;
};
static std::mutex g_cache_mu;
static std::vector<std::pair<bzh_ctx*, NttCache*>> g_caches;

static NttCache* cache_for(bzh_ctx* ctx) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (auto& p : g_caches)
        if (p.first == ctx) return p.second;
    g_caches.emplace_back(ctx, new NttCache());
    return g_caches.back().second;
}
void ntt_cache_drop(bzh_ctx* ctx) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (size_t i = 0; i < g_caches.size(); i++)
        if (g_caches[i].first == ctx) {
            for (auto& d : g_caches[i].second->doms)
                if (d.d_tables) (void)hipFree(d.d_tables);
            delete g_caches[i].second;
            g_caches.erase(g_caches.begin() + i);
            return;
        }
}

template <class P>
static void fill_pows(std::vector<uint32_t>& out, size_t off, const Fe<P>& base, const Fe<P>& first, size_t count) {
    Fe<P> acc = first;
    for (size_t i = 0; i < count; i++) {
        for (int k = 0; k < 8; k++) out[off + i * 8 + k] = acc.l[k];
        acc = fe_mul(acc, base);
    }
}
template <class P>
static Fe<P> pow2k(Fe<P> x, int k) {
    for (int i = 0; i < k; i++) x = fe_sqr(x);
    return x;
}

// Pass plan: one pass up to 2^11, else ceil(k/9) passes of near-equal radix bits.
static int plan_passes(unsigned log_n, int* bits) {
    if (log_n <= 11) {
        bits[0] = (int)log_n;
        return 1;
    }
    const int np = (int)((log_n + 8) / 9);
    const int base = (int)log_n / np, extra = (int)log_n % np;
    for (int i = 0; i < np; i++) bits[i] = base + (i < extra ? 1 : 0);
    return np;
}
static constexpr unsigned kDirectTwiddleMaxLog = 20;  // 32 MiB per table at 2^20; larger domains use the two-level table

template <class P>
static int build_domain(bzh_ctx* ctx, NttDomain& d, const Fe<P>& omega_m, const Fe<P>* shift_m) {
    const unsigned k = d.log_n;
    const size_t N = (size_t)1 << k;
    d.h = (int)((k + 1) / 2);
    const size_t nlo = (size_t)1 << d.h, nhi = (size_t)1 << (k - d.h);
    Fe<P> w = d.inverse ? fe_inv(omega_m) : omega_m;
    size_t words = 0;
    d.off_tw_lo = words; words += nlo * 8;
    d.off_tw_hi = words; words += nhi * 8;
    d.off_sc_lo = words; words += nlo * 8;
    d.off_sc_hi = words; words += nhi * 8;
    d.off_tw_hi_scaled = words; words += nhi * 8;
    for (int rb = 1; rb <= 11; rb++) {
        d.off_sub[rb] = words;
        if ((unsigned)rb <= k) words += ((size_t)1 << (rb - 1)) * 8;
    }
    int bits[5];
    const int np = plan_passes(k, bits);
    for (int p = 0; p < 4; p++) d.off_direct[p] = 0;
    if (np >= 2 && k <= kDirectTwiddleMaxLog) {
        int logA = 0;
        for (int p = 0; p + 1 < np; p++) {
            d.off_direct[p] = words;
            words += (N >> logA) * 8;
            logA += bits[p];
        }
    }
    std::vector<uint32_t> host(words, 0u);
    fill_pows<P>(host, d.off_tw_lo, w, fe_one<P>(), nlo);
    fill_pows<P>(host, d.off_tw_hi, pow2k(w, d.h), fe_one<P>(), nhi);
    // scale tables: forward = shift^n ; inverse = n^-1 * shift^-k (or n^-1 alone)
    Fe<P> sbase = fe_one<P>(), first_hi = fe_one<P>();
    if (d.has_shift) sbase = d.inverse ? fe_inv(*shift_m) : *shift_m;
    if (d.inverse) {
        Fe<P> nn = fe_zero<P>();
        nn.l[0] = (uint32_t)(N & 0xffffffffu);
        nn.l[1] = (uint32_t)((uint64_t)N >> 32);
        first_hi = fe_inv(fe_to_mont(nn));
    }
    fill_pows<P>(host, d.off_sc_lo, sbase, fe_one<P>(), nlo);
    fill_pows<P>(host, d.off_sc_hi, pow2k(sbase, d.h), first_hi, nhi);
    fill_pows<P>(host, d.off_tw_hi_scaled, pow2k(w, d.h), first_hi, nhi);
    {
        const Fe<P> s2 = fe_sqr(sbase), s3 = fe_mul(s2, sbase);
        d.shift_is_cube = d.has_shift && fe_eq(s3, fe_one<P>()) && !fe_eq(sbase, fe_one<P>());
        const Fe<P> c0 = first_hi, c1 = fe_mul(first_hi, sbase), c2 = fe_mul(first_hi, s2);
        for (int q = 0; q < 8; q++) {
            d.cube[0][q] = c0.l[q];
            d.cube[1][q] = c1.l[q];
            d.cube[2][q] = c2.l[q];
        }
    }
    for (int rb = 1; rb <= 11 && (unsigned)rb <= k; rb++) {
        Fe<P> wr = pow2k(w, (int)k - rb);  // omega_N^(N/R)
        fill_pows<P>(host, d.off_sub[rb], wr, fe_one<P>(), (size_t)1 << (rb - 1));
    }
    if (d.off_direct[0]) {
        int logA = 0;
        for (int p = 0; p + 1 < np; p++) {
            const size_t R = (size_t)1 << bits[p], B = N >> (logA + bits[p]);
            const Fe<P> wa = pow2k(w, logA);  // omega^A
            const Fe<P> first = (p == 0 && d.inverse && !d.has_shift) ? first_hi : fe_one<P>();
            Fe<P> wrow = fe_one<P>();         // omega^(A*row)
            for (size_t row = 0; row < R; row++) {
                fill_pows<P>(host, d.off_direct[p] + row * B * 8, wrow, first, B);
                wrow = fe_mul(wrow, wa);
            }
            logA += bits[p];
        }
    }
    BZH_HIP_TRY(ctx, hipMalloc((void**)&d.d_tables, words * 4));
    BZH_HIP_TRY(ctx, hipMemcpyAsync(d.d_tables, host.data(), words * 4, hipMemcpyHostToDevice, ctx->stream));
    BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // host vector goes out of scope
    return BZH_OK;
}

template <class P>
static int ntt_run_t(bzh_ctx* ctx, uint32_t* d_data, unsigned log_n, size_t batch, const uint64_t* omega,
                     const uint64_t* coset_shift, int inverse, int form, const uint32_t* d_src = nullptr, unsigned src_log = 0,
                     uint32_t* d_out29 = nullptr) {
    if (log_n == 0) return d_out29 ? BZH_E_ARG : BZH_OK;  // size-1 transform is the identity (n^-1 = 1, shift^0 = 1)
    Fe<P> w, sh = fe_one<P>();
    for (int i = 0; i < 4; i++) {
        w.l[2 * i] = (uint32_t)omega[i];
        w.l[2 * i + 1] = (uint32_t)(omega[i] >> 32);
        if (coset_shift) {
            sh.l[2 * i] = (uint32_t)coset_shift[i];
            sh.l[2 * i + 1] = (uint32_t)(coset_shift[i] >> 32);
        }
    }
    if (form == BZH_FORM_CANONICAL) {
        w = fe_to_mont(w);
        if (coset_shift) sh = fe_to_mont(sh);
    }
    NttCache* cache = cache_for(ctx);
    NttDomain* dom = nullptr;
    for (auto& d : cache->doms) {
        if (d.field == P::id && d.log_n == log_n && d.inverse == (inverse != 0) && d.has_shift == (coset_shift != nullptr) &&
            memcmp(d.omega, w.l, 32) == 0 && (!coset_shift || memcmp(d.shift, sh.l, 32) == 0)) {
            dom = &d;
            break;
        }
    }
    if (!dom) {
        NttDomain d;
        d.field = P::id;
        d.log_n = log_n;
        d.inverse = inverse != 0;
        d.has_shift = coset_shift != nullptr;
        memcpy(d.omega, w.l, 32);
        memcpy(d.shift, sh.l, 32);
        int rc = build_domain<P>(ctx, d, w, coset_shift ? &sh : nullptr);
        if (rc) return rc;
        cache->doms.push_back(d);
        dom = &cache->doms.back();
    }
    // canonical input -> Montgomery (and back at the end)
    const size_t total = batch << log_n;
    if (form == BZH_FORM_CANONICAL) {
        int rc = field_convert(ctx, P::id, d_data, total, 1);
        if (rc) return rc;
    }
    // plan passes
    int bits[5];
    const int np = plan_passes(log_n, bits);
    // zero-padded source (coeff_to_extended): 2^src_log coefficients per vector, read straight from d_src by pass 0
    int nz = 0;
    if (d_src) {
        if (src_log > log_n || form != BZH_FORM_MONTGOMERY) return BZH_E_ARG;
        nz = (int)(log_n - src_log);
        if ((np < 2 || nz > bits[0] || nz == 0) && !d_data) return BZH_E_ARG;   // (planes-only output needs the multi-pass path)
        if (np < 2 || nz > bits[0] || nz == 0) {  // single-pass sizes: pad in memory and run the plain transform
            BZH_HIP_TRY(ctx, hipMemsetAsync(d_data, 0, total * 32, ctx->stream));
            BZH_HIP_TRY(ctx, hipMemcpy2DAsync(d_data, ((size_t)32) << log_n, d_src, ((size_t)32) << src_log, ((size_t)32) << src_log, batch,
                                              hipMemcpyDeviceToDevice, ctx->stream));
            nz = 0;
            d_src = nullptr;
        }
    }
    // np >= 2: pass 0 data -> scratch, middle passes in scratch, last pass scratch -> data
    uint32_t* d_scratch = nullptr;
    if (np >= 2) {
        void* sp = nullptr;
        int rc = ws_ensure(ctx, 0, total * 32, &sp);
        if (rc) return rc;
        d_scratch = (uint32_t*)sp;
    }
    const bool pre = (!inverse && coset_shift);
    const bool post = (inverse != 0);
    int logA = 0;
    for (int p = 0; p < np; p++) {
        NttPassArgs a;
        a.src = (p == 0) ? (nz ? d_src : d_data) : d_scratch;
        a.nz = (p == 0) ? nz : 0;
        a.dst = (p == np - 1) ? d_data : d_scratch;
        a.log_n = log_n;
        a.r = bits[p];
        a.logA = logA;
        a.logB = (int)log_n - logA - bits[p];
        a.last = (p == np - 1);
        int logW = 11 - a.r;  // R * W = 2048
        if (!a.last && logW > a.logB) logW = a.logB;
        if (a.last && logW > logA) logW = logA;
        a.logW = logW;
        a.nprev = p;
        for (int q = 0; q < 4; q++) a.prev_bits[q] = q < p ? bits[q] : 0;
        a.sub_tw = dom->d_tables + dom->off_sub[a.r];
        a.tw_lo = dom->d_tables + dom->off_tw_lo;
        a.tw_hi = dom->d_tables + dom->off_tw_hi;
        a.tw_direct = (p + 1 < np && dom->off_direct[p]) ? dom->d_tables + dom->off_direct[p] : nullptr;
        a.h = dom->h;
        a.pre_lo = (pre && p == 0) ? dom->d_tables + dom->off_sc_lo : nullptr;
        a.pre_hi = dom->d_tables + dom->off_sc_hi;
        a.post_lo = (post && a.last) ? dom->d_tables + dom->off_sc_lo : nullptr;
        a.post_hi = dom->d_tables + dom->off_sc_hi;
        a.cube_pre = a.cube_post = a.tw_always = 0;
        a.dst29 = nullptr;
        memcpy(a.cube, dom->cube, sizeof(a.cube));
        if (pre && dom->shift_is_cube) {  // coeff_to_extended: shift = ZETA
            if (p == 0) a.cube_pre = 1;
            a.pre_lo = nullptr;
        }
        if (post) {
            if (coset_shift && dom->shift_is_cube) {  // extended_to_coeff: n^-1 * ZETA^-k
                a.post_lo = nullptr;
                a.cube_post = a.last ? 1 : 0;
            } else if (!coset_shift) {  // plain ifft: the constant n^-1
                a.post_lo = nullptr;
                if (np == 1) {
                    a.cube_post = 1;  // cube[0..2] all equal n^-1
                } else if (p == 0) {
                    a.tw_hi = dom->d_tables + dom->off_tw_hi_scaled;  // folded into the first inter-pass twiddle
                    a.tw_always = 1;
                }
            }
        }
        const size_t tiles = ((size_t)1 << log_n) >> (a.r + logW);
        {
            ScopedTimer t(ctx, BZH_T_NTT);
            if (ctx->profiling && p == 0)
                ctx->alg_bytes[BZH_T_NTT] += (32.0 + (nz ? 32.0 / (double)(1 << nz) : 32.0)) * (double)batch * (double)((size_t)1 << log_n);
            for (size_t b0 = 0; b0 < batch; b0 += 65535) {
                size_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
                NttPassArgs aa = a;
                aa.src = a.src + (b0 << (log_n - (unsigned)a.nz)) * 8;
                aa.dst = a.dst ? a.dst + (b0 << log_n) * 8 : nullptr;
                aa.dst29 = (a.last && d_out29) ? d_out29 + (b0 << log_n) * 9 : nullptr;
                // in-wave butterflies when the tile is a full 2048 elements of 6..9 radix bits; BZH_NTT_LDS=1 keeps the LDS tile
                static const bool lds_only = getenv("BZH_NTT_LDS") != nullptr;
                const dim3 grid((unsigned)tiles, (unsigned)nb), blk(kNttThreads);
                // (BN254's base field has 2-adicity 1: no transform of it reaches the wave kernel's sizes, none is instantiated)
                constexpr bool wave_ok = P::id != BZH_FIELD_BN254_FQ;
                if (wave_ok && !lds_only && a.r >= 6 && a.r <= 9 && logW == 11 - a.r && a.nz <= 3) {
                    if constexpr (wave_ok) launch_ntt_pass_wave<P>(aa, (unsigned)tiles, (unsigned)nb, ctx->stream);
                } else {
                    hipLaunchKernelGGL((k_ntt_pass<P>), grid, blk, 0, ctx->stream, aa);
                }
            }
        }
        BZH_HIP_TRY(ctx, hipGetLastError());
        logA += bits[p];
    }
    if (form == BZH_FORM_CANONICAL) {
        int rc = field_convert(ctx, P::id, d_data, total, 0);
        if (rc) return rc;
    }
    return BZH_OK;
}

template <class P>
__global__ void __launch_bounds__(256) k_field_convert(uint32_t* data, size_t count, int to_mont) {
    size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= count) return;
    Fe<P> v = fe_load<P>(data + gid * 8);
    fe_store(data + gid * 8, to_mont ? fe_to_mont(v) : fe_from_mont(v));
}

int field_convert(bzh_ctx* ctx, int field, uint32_t* d, size_t count, int to_mont) {
    if (!count) return BZH_OK;
    dim3 grid((unsigned)((count + 255) / 256)), block(256);
    switch (field) {
        case BZH_FIELD_FP: hipLaunchKernelGGL((k_field_convert<FpParams>), grid, block, 0, ctx->stream, d, count, to_mont); break;
        case BZH_FIELD_FQ: hipLaunchKernelGGL((k_field_convert<FqParams>), grid, block, 0, ctx->stream, d, count, to_mont); break;
        case BZH_FIELD_BN254_FR: hipLaunchKernelGGL((k_field_convert<BnFrParams>), grid, block, 0, ctx->stream, d, count, to_mont); break;
        case BZH_FIELD_BN254_FQ: hipLaunchKernelGGL((k_field_convert<BnFqParams>), grid, block, 0, ctx->stream, d, count, to_mont); break;
        default: return BZH_E_ARG;
    }
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

int ntt_run(bzh_ctx* ctx, int field, uint32_t* d_data, unsigned log_n, size_t batch, const uint64_t* omega,
            const uint64_t* coset_shift, int inverse, int form) {
    switch (field) {
        case BZH_FIELD_FP: return ntt_run_t<FpParams>(ctx, d_data, log_n, batch, omega, coset_shift, inverse, form);
        case BZH_FIELD_FQ: return ntt_run_t<FqParams>(ctx, d_data, log_n, batch, omega, coset_shift, inverse, form);
        case BZH_FIELD_BN254_FR: return ntt_run_t<BnFrParams>(ctx, d_data, log_n, batch, omega, coset_shift, inverse, form);
        case BZH_FIELD_BN254_FQ: return ntt_run_t<BnFqParams>(ctx, d_data, log_n, batch, omega, coset_shift, inverse, form);
    }
    return BZH_E_ARG;
}

// coeff_to_extended without the padded copy: `batch` polynomials of 2^src_log coefficients at d_src (pitch 2^src_log)
// -> their evaluations over the 2^log_n coset at d_dst (pitch 2^log_n).  Montgomery form.
int ntt_run_padded(bzh_ctx* ctx, int field, uint32_t* d_dst, const uint32_t* d_src, unsigned src_log, unsigned log_n, size_t batch,
                   const uint64_t* omega, const uint64_t* coset_shift, uint32_t* d_out29) {
    const int f = BZH_FORM_MONTGOMERY;
    if (d_out29 && field != BZH_FIELD_FP && field != BZH_FIELD_FQ) return BZH_E_ARG;
    switch (field) {
        case BZH_FIELD_FP: return ntt_run_t<FpParams>(ctx, d_dst, log_n, batch, omega, coset_shift, 0, f, d_src, src_log, d_out29);
        case BZH_FIELD_FQ: return ntt_run_t<FqParams>(ctx, d_dst, log_n, batch, omega, coset_shift, 0, f, d_src, src_log, d_out29);
        case BZH_FIELD_BN254_FR: return ntt_run_t<BnFrParams>(ctx, d_dst, log_n, batch, omega, coset_shift, 0, f, d_src, src_log);
        case BZH_FIELD_BN254_FQ: return ntt_run_t<BnFqParams>(ctx, d_dst, log_n, batch, omega, coset_shift, 0, f, d_src, src_log);
    }
    return BZH_E_ARG;
}

}  // namespace bzh
