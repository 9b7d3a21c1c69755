// Prover-stage vector primitives on gfx950 (N4, N5, N6).
//
// Device counterparts of halo2_proofs 0.2.0 helpers (UPSTREAM, un-vendored: Cargo.lock:382-385)
// that create_proof (benches/shot.rs:68, src/circuits/board.rs:913-920) runs between its MSMs
// and FFTs:
//   ff::BatchInvert::batch_invert            permutation / lookup grand-product denominators
//   grand-product running product z(X)       permutation::Argument::commit, lookup commit_product
//   arithmetic::eval_polynomial              the ~80-100 openings written after challenge x
//   arithmetic::compute_inner_product        IPA rounds (value_l / value_r)
//   lo + u * hi                              IPA round folding of p' and b
// All kernels work on Montgomery-form elements (8 x u32), strided so that a wave touches
// 64 consecutive 32-byte elements; modular-integer VALU work, no MFMA.
#include "block.cuh"
#include "ctx.hpp"
#include "field.cuh"

namespace bzh {

static constexpr int kVecThreads = 256;

// ---------------------------------------------------------------------------
// batch inversion: thread t owns elements t, t+T, t+2T, ... (coalesced); Montgomery's
// trick over its chain with one Fermat inversion per thread.  tmp holds the running
// prefixes (same size as data).  Zeros stay zero and do not poison the chain.
// ---------------------------------------------------------------------------
template <class P>
__global__ void __launch_bounds__(kVecThreads) k_batch_invert(uint32_t* __restrict__ data, uint32_t* __restrict__ tmp,
                                                                size_t count, size_t nthreads) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    Fe<P> acc = fe_one<P>();
    for (size_t i = t; i < count; i += nthreads) {
        Fe<P> v = fe_load<P>(data + i * 8);
        fe_store(tmp + i * 8, acc);
        if (!fe_is_zero(v)) acc = fe_mul(acc, v);
    }
    Fe<P> inv = fe_inv(acc);
    size_t last = t + ((count - 1 - t) / nthreads) * nthreads;  // largest index of this thread's chain
    for (size_t i = last;; i -= nthreads) {
        Fe<P> v = fe_load<P>(data + i * 8);
        if (!fe_is_zero(v)) {
            Fe<P> pre = fe_load<P>(tmp + i * 8);
            fe_store(data + i * 8, fe_mul(inv, pre));
            inv = fe_mul(inv, v);
        }
        if (i < nthreads) break;
    }
}

// ---------------------------------------------------------------------------
// exclusive running product, three phases over tiles of kScanTile elements:
//   A: tile-local exclusive scan in place + tile totals
//   B: one thread per vector scans its tile totals (<= n/2048 multiplications)
//   C: multiply every tile by its offset
// ---------------------------------------------------------------------------
static constexpr int kScanTile = 2048;  // 256 threads x 8 consecutive elements

template <class P>
struct MulOp {
    static __device__ __forceinline__ Fe<P> id() { return fe_one<P>(); }
    static __device__ __forceinline__ Fe<P> op(const Fe<P>& a, const Fe<P>& b) { return fe_mul(a, b); }
};
template <class P>
struct AddOp {
    static __device__ __forceinline__ Fe<P> id() { return fe_zero<P>(); }
    static __device__ __forceinline__ Fe<P> op(const Fe<P>& a, const Fe<P>& b) { return fe_add(a, b); }
};

template <class P, class Op>
__global__ void __launch_bounds__(256) k_scan_tiles(uint32_t* __restrict__ data, size_t n, size_t tiles_per_vec,
                                                      uint32_t* __restrict__ totals) {
    __shared__ __align__(16) uint4 sh[2][2 * 256];
    const int tid = threadIdx.x;
    const size_t vec = blockIdx.y, tile = blockIdx.x;
    uint32_t* base = data + (vec * n + tile * kScanTile) * 8;
    const size_t lim = min((size_t)kScanTile, n - tile * kScanTile);
    // thread-local products over 8 consecutive elements
    Fe<P> loc[8];
    Fe<P> run = Op::id();
#pragma unroll
    for (int k = 0; k < 8; k++) {
        size_t i = (size_t)tid * 8 + k;
        loc[k] = i < lim ? fe_load<P>(base + i * 8) : Op::id();
    }
    Fe<P> tot = loc[0];
#pragma unroll
    for (int k = 1; k < 8; k++) tot = Op::op(tot, loc[k]);
    // inclusive Hillis-Steele scan of the 256 thread totals in LDS (planes layout)
    int cur = 0;
    Fe<P> incl = tot;
    for (int d = 1; d < 256; d <<= 1) {
        sh[cur][tid] = make_uint4(incl.l[0], incl.l[1], incl.l[2], incl.l[3]);
        sh[cur][256 + tid] = make_uint4(incl.l[4], incl.l[5], incl.l[6], incl.l[7]);
        __syncthreads();
        if (tid >= d) {
            uint4 a = sh[cur][tid - d], b = sh[cur][256 + tid - d];
            Fe<P> o;
            o.l[0] = a.x; o.l[1] = a.y; o.l[2] = a.z; o.l[3] = a.w;
            o.l[4] = b.x; o.l[5] = b.y; o.l[6] = b.z; o.l[7] = b.w;
            incl = Op::op(incl, o);
        }
        cur ^= 1;
    }
    // exclusive prefix of this thread = inclusive of the previous thread
    sh[cur][tid] = make_uint4(incl.l[0], incl.l[1], incl.l[2], incl.l[3]);
    sh[cur][256 + tid] = make_uint4(incl.l[4], incl.l[5], incl.l[6], incl.l[7]);
    __syncthreads();
    if (tid > 0) {
        uint4 a = sh[cur][tid - 1], b = sh[cur][256 + tid - 1];
        run.l[0] = a.x; run.l[1] = a.y; run.l[2] = a.z; run.l[3] = a.w;
        run.l[4] = b.x; run.l[5] = b.y; run.l[6] = b.z; run.l[7] = b.w;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        size_t i = (size_t)tid * 8 + k;
        if (i < lim) fe_store(base + i * 8, run);
        run = Op::op(run, loc[k]);
    }
    if (tid == 255) fe_store(totals + (vec * tiles_per_vec + tile) * 8, incl);
}

template <class P, class Op>
__global__ void __launch_bounds__(64) k_scan_totals(uint32_t* __restrict__ totals, size_t tiles_per_vec, size_t batch) {
    const size_t vec = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (vec >= batch) return;
    Fe<P> run = Op::id();
    uint32_t* t = totals + vec * tiles_per_vec * 8;
    for (size_t k = 0; k < tiles_per_vec; k++) {
        Fe<P> v = fe_load<P>(t + k * 8);
        fe_store(t + k * 8, run);
        run = Op::op(run, v);
    }
}

template <class P, class Op>
__global__ void __launch_bounds__(256) k_scan_apply(uint32_t* __restrict__ data, size_t n, size_t tiles_per_vec,
                                                      const uint32_t* __restrict__ totals) {
    const size_t vec = blockIdx.y, tile = blockIdx.x;
    if (tile == 0) return;  // offset 1
    const Fe<P> off = fe_load<P>(totals + (vec * tiles_per_vec + tile) * 8);
    uint32_t* base = data + (vec * n + tile * kScanTile) * 8;
    const size_t lim = min((size_t)kScanTile, n - tile * kScanTile);
    for (size_t i = threadIdx.x; i < lim; i += 256) fe_store(base + i * 8, Op::op(fe_load<P>(base + i * 8), off));
}

// ---------------------------------------------------------------------------
// polynomial evaluation: one workgroup per (polynomial, point).  Thread t sums
// c[t + kT] y^k with y = x^T by Horner (coalesced reads), then the block adds x^t * partial_t.
// ---------------------------------------------------------------------------
// p(x) of the n coefficients at c by the whole workgroup (kVecThreads threads); the sum is valid in thread 0
template <class P>
__device__ __forceinline__ Fe<P> block_eval_poly(const uint32_t* __restrict__ c, size_t n, const Fe<P>& x, uint4* sh) {
    constexpr int T = kVecThreads;
    const int tid = threadIdx.x;
    // x^t for this thread (binary method over the 8 bits of t) and y = x^T
    Fe<P> xt = fe_one<P>(), pw = x;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        if ((tid >> bit) & 1) xt = fe_mul(xt, pw);
        pw = fe_sqr(pw);
    }
    const Fe<P> y = pw;  // x^256
    Fe<P> acc = fe_zero<P>();
    if ((size_t)tid < n) {
        size_t last = tid + ((n - 1 - tid) / T) * T;
        for (size_t i = last;; i -= T) {
            acc = fe_add(fe_mul(acc, y), fe_load<P>(c + i * 8));
            if (i < (size_t)T) break;
        }
        acc = fe_mul(acc, xt);
    }
    return block_sum(acc, sh, T);
}
template <class P>
__global__ void __launch_bounds__(kVecThreads) k_eval_poly(const uint32_t* __restrict__ coeffs, size_t n,
                                                             const uint32_t* __restrict__ xs, size_t x_stride,
                                                             uint32_t* __restrict__ out) {
    __shared__ __align__(16) uint4 sh[2 * kVecThreads];
    const size_t b = blockIdx.x;
    const Fe<P> acc = block_eval_poly<P>(coeffs + b * n * 8, n, fe_load<P>(xs + b * x_stride * 8), sh);
    if (threadIdx.x == 0) fe_store(out + b * 8, acc);
}

// inner product sum_i a_i b_i per vector
template <class P>
__global__ void __launch_bounds__(kVecThreads) k_inner_product(const uint32_t* __restrict__ a, const uint32_t* __restrict__ bv,
                                                                 size_t n, uint32_t* __restrict__ out) {
    __shared__ __align__(16) uint4 sh[2 * kVecThreads];
    constexpr int T = kVecThreads;
    const size_t vec = blockIdx.x;
    Fe<P> acc = fe_zero<P>();
    for (size_t i = threadIdx.x; i < n; i += T)
        acc = fe_add(acc, fe_mul(fe_load<P>(a + (vec * n + i) * 8), fe_load<P>(bv + (vec * n + i) * 8)));
    acc = block_sum(acc, sh, T);
    if (threadIdx.x == 0) fe_store(out + vec * 8, acc);
}

// out[i] = lo[i] + u * hi[i], i < half, for `batch` vectors of length 2*half laid out [lo | hi]
template <class P>
__global__ void __launch_bounds__(kVecThreads) k_fold(const uint32_t* __restrict__ in, size_t half, size_t batch,
                                                        const uint32_t* __restrict__ u, size_t u_stride,
                                                        uint32_t* __restrict__ out) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= half * batch) return;
    const size_t vec = g / half, i = g - vec * half;
    const Fe<P> uu = fe_load<P>(u + vec * u_stride * 8);
    const uint32_t* v = in + vec * 2 * half * 8;
    fe_store(out + (vec * half + i) * 8, fe_add(fe_load<P>(v + i * 8), fe_mul(uu, fe_load<P>(v + (half + i) * 8))));
}

// elementwise product a[i] *= b[i]
template <class P>
__global__ void __launch_bounds__(kVecThreads) k_vec_mul(uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t count) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g >= count) return;
    fe_store(a + g * 8, fe_mul(fe_load<P>(a + g * 8), fe_load<P>(b + g * 8)));
}

// ---------------------------------------------------------------------------
// host drivers (device pointers, Montgomery form)
// ---------------------------------------------------------------------------
template <class P>
static int batch_invert_t(bzh_ctx* ctx, uint32_t* d, size_t count) {
    if (!count) return BZH_OK;
    void* tmp = nullptr;
    int rc = ws_ensure(ctx, 0, count * 32, &tmp);
    if (rc) return rc;
    // chains of ~16 elements, at least one wave, at most ~64k threads
    size_t nthreads = (count + 15) / 16;
    if (nthreads < 64) nthreads = count < 64 ? count : 64;
    if (nthreads > 65536) nthreads = 65536;
    ScopedTimer t(ctx, BZH_T_POLY);
    hipLaunchKernelGGL((k_batch_invert<P>), dim3((unsigned)((nthreads + kVecThreads - 1) / kVecThreads)), dim3(kVecThreads), 0,
                       ctx->stream, d, (uint32_t*)tmp, count, nthreads);
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

template <class P, class Op>
static int prefix_scan_t(bzh_ctx* ctx, uint32_t* d, size_t n, size_t batch) {
    if (!n || !batch) return BZH_OK;
    const size_t tiles = (n + kScanTile - 1) / kScanTile;
    void* tot = nullptr;
    int rc = ws_ensure(ctx, 2, tiles * batch * 32, &tot);
    if (rc) return rc;
    ScopedTimer t(ctx, BZH_T_POLY);
    for (size_t b0 = 0; b0 < batch; b0 += 65535) {
        const size_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
        uint32_t* dd = d + b0 * n * 8;
        uint32_t* tt = (uint32_t*)tot + b0 * tiles * 8;
        hipLaunchKernelGGL((k_scan_tiles<P, Op>), dim3((unsigned)tiles, (unsigned)nb), dim3(256), 0, ctx->stream, dd, n, tiles, tt);
        if (tiles > 1) {
            hipLaunchKernelGGL((k_scan_totals<P, Op>), dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, ctx->stream, tt, tiles, nb);
            hipLaunchKernelGGL((k_scan_apply<P, Op>), dim3((unsigned)tiles, (unsigned)nb), dim3(256), 0, ctx->stream, dd, n, tiles, tt);
        }
    }
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

template <class P>
static int eval_poly_t(bzh_ctx* ctx, const uint32_t* coeffs, size_t n, size_t batch, const uint32_t* xs, size_t x_stride,
                       uint32_t* out) {
    if (!batch) return BZH_OK;
    ScopedTimer t(ctx, BZH_T_POLY);
    hipLaunchKernelGGL((k_eval_poly<P>), dim3((unsigned)batch), dim3(kVecThreads), 0, ctx->stream, coeffs, n, xs, x_stride, out);
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

#define BZH_FIELD_SWITCH(field, CALL)                     \
    switch (field) {                                      \
        case BZH_FIELD_FP: return CALL(FpParams);         \
        case BZH_FIELD_FQ: return CALL(FqParams);         \
        case BZH_FIELD_BN254_FR: return CALL(BnFrParams); \
        case BZH_FIELD_BN254_FQ: return CALL(BnFqParams); \
    }                                                     \
    return BZH_E_ARG;

int poly_batch_invert(bzh_ctx* ctx, int field, uint32_t* d, size_t count) {
#define CALL(PP) batch_invert_t<PP>(ctx, d, count)
    BZH_FIELD_SWITCH(field, CALL)
#undef CALL
}
int poly_prefix_product(bzh_ctx* ctx, int field, uint32_t* d, size_t n, size_t batch) {
#define CALL(PP) prefix_scan_t<PP, MulOp<PP>>(ctx, d, n, batch)
    BZH_FIELD_SWITCH(field, CALL)
#undef CALL
}
int poly_eval(bzh_ctx* ctx, int field, const uint32_t* coeffs, size_t n, size_t batch, const uint32_t* xs, size_t x_stride,
              uint32_t* out) {
#define CALL(PP) eval_poly_t<PP>(ctx, coeffs, n, batch, xs, x_stride, out)
    BZH_FIELD_SWITCH(field, CALL)
#undef CALL
}

// ---------------------------------------------------------------------------
// kate_division: quotient of p(X) by (X - x).  With d_t = c_(n-1-t) the quotient obeys the Horner recurrence
// r_k = x r_(k-1) + d_k, q_(n-2-k) = r_k (the remainder r_(n-1) = p(x) is dropped).  One workgroup per polynomial span: every
// thread runs the recurrence over its own contiguous segment of L indices from zero (one multiplication per element),
// the segment results are combined by a scan of R_t = x^L R_(t-1) + loc_t through LDS (log2(threads) steps), and a
// second walk over the segment from the carried-in value writes the quotient -- two multiplications per coefficient and no
// x^-1.  (The round-2 version raised x^-1 and x to the element's index per element: ~56 multiplications per coefficient.)
// ---------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ Fe<P> fe_pow_u64(Fe<P> base, size_t e) {
    Fe<P> acc = fe_one<P>();
    for (; e; e >>= 1) {
        if (e & 1) acc = fe_mul(acc, base);
        base = fe_sqr(base);
    }
    return acc;
}
static constexpr int kKateMaxThreads = 1024;
// Few polynomials (a single proof) would leave the chip to one workgroup each, so a polynomial may be cut into S spans of
// G = threads * L indices, one workgroup per span: k_kate_span_totals evaluates every span but the last from zero
// (span s covers coefficients c[n-(s+1)G .. n-1-sG], its value at x is E_s), and span s of k_kate_rows starts from the
// carried-in value C_s = sum_(s'<s) x^(G(s-1-s')) E_s' (Horner over the preceding totals).
template <class P>
__global__ void __launch_bounds__(kVecThreads) k_kate_span_totals(const uint32_t* __restrict__ c, size_t n, const uint32_t* __restrict__ xs,
                                                                    size_t G, uint32_t* __restrict__ tot) {
    __shared__ __align__(16) uint4 sh[2 * kVecThreads];
    const size_t s = blockIdx.x, v = blockIdx.y, S1 = gridDim.x;  // spans 0 .. S-2: all of them full
    const Fe<P> acc = block_eval_poly<P>(c + (v * n + n - (s + 1) * G) * 8, G, fe_load<P>(xs + v * 8), sh);
    if (threadIdx.x == 0) fe_store(tot + (v * S1 + s) * 8, acc);
}
// x of vector v sits at xs + v*8; vector v's n coefficients at c + v*n*8, its n-1 quotient coefficients at q + v*(n-1)*8;
// grid (S, batch), blockDim.x * L = G; tot: the S-1 span totals per vector (unused when S = 1)
template <class P>
__global__ void __launch_bounds__(kKateMaxThreads) k_kate_rows(const uint32_t* __restrict__ c, size_t n, const uint32_t* __restrict__ xs,
                                                                 size_t L, const uint32_t* __restrict__ tot, uint32_t* __restrict__ q) {
    __shared__ uint32_t lds[8 * kKateMaxThreads];  // word w of thread t's value at lds[w * T + t]
    const unsigned t = threadIdx.x, T = blockDim.x;
    const size_t s = blockIdx.x, S = gridDim.x, v = blockIdx.y, m = n - 1, G = (size_t)T * L;
    const size_t kb = s * G + (size_t)t * L, k0 = kb < m ? kb : m, k1 = k0 + L < m ? k0 + L : m;
    const Fe<P> x = fe_load<P>(xs + v * 8);
    const uint32_t* cv = c + v * n * 8;
    uint32_t* qv = q + v * m * 8;
    Fe<P> M = fe_pow_u64(x, L);
    Fe<P> carry = fe_zero<P>();
    if (s && t == 0) {  // only thread 0 starts from the carried-in value
        Fe<P> xg = M;
        for (unsigned w = 1; w < T; w <<= 1) xg = fe_sqr(xg);  // x^G, T a power of two
        for (size_t sp = 0; sp < s; sp++) carry = fe_add(fe_mul(carry, xg), fe_load<P>(tot + (v * (S - 1) + sp) * 8));
    }
    Fe<P> R = t ? fe_zero<P>() : carry;
    for (size_t k = k0; k < k1; k++) R = fe_add(fe_mul(R, x), fe_load<P>(cv + (n - 1 - k) * 8));
    auto put = [&](const Fe<P>& a) {
#pragma unroll
        for (int w = 0; w < 8; w++) lds[w * T + t] = a.l[w];
    };
    auto get = [&](unsigned from) {
        Fe<P> a;
#pragma unroll
        for (int w = 0; w < 8; w++) a.l[w] = lds[w * T + from];
        return a;
    };
    for (unsigned st = 1; st < T; st <<= 1) {
        put(R);
        __syncthreads();
        if (t >= st) R = fe_add(R, fe_mul(M, get(t - st)));
        __syncthreads();
        M = fe_sqr(M);
    }
    put(R);
    __syncthreads();
    Fe<P> r = t ? get(t - 1) : carry;
    for (size_t k = k0; k < k1; k++) {
        r = fe_add(fe_mul(r, x), fe_load<P>(cv + (n - 1 - k) * 8));
        fe_store(qv + (m - 1 - k) * 8, r);
    }
}

template <class P>
static int kate_t(bzh_ctx* ctx, const uint32_t* d_c, size_t n, size_t batch, const uint32_t* d_xs, uint32_t* d_q) {
    if (n < 2 || !batch) return BZH_OK;
    if (batch > 65535) return BZH_E_ARG;
    const size_t m = n - 1;
    // 256-thread workgroups (64 for tiny inputs), about 256 workgroups per launch: S = 256 / batch spans per polynomial,
    // at least 4 coefficients per thread and at most 32 spans.  Per coefficient the kernel spends 2 multiplications on the
    // recurrence and (3 log2(threads) + log2(L) + S) / L on the stitching, so short chains are for small batches only.
    unsigned threads = 64;
    while (threads < 256u && (size_t)threads * 4 < m) threads <<= 1;
    size_t S = (256 + batch - 1) / batch;
    if (S > 32) S = 32;
    size_t L = (m + (size_t)threads * S - 1) / ((size_t)threads * S);
    if (L < 4) L = 4;
    S = (m + (size_t)threads * L - 1) / ((size_t)threads * L);
    void* tot = nullptr;
    if (S > 1) {
        int rc = ws_ensure(ctx, 2, (S - 1) * batch * 32, &tot);
        if (rc) return rc;
    }
    ScopedTimer t(ctx, BZH_T_POLY);
    if (S > 1)
        hipLaunchKernelGGL((k_kate_span_totals<P>), dim3((unsigned)(S - 1), (unsigned)batch), dim3(kVecThreads), 0, ctx->stream, d_c, n, d_xs,
                           (size_t)threads * L, (uint32_t*)tot);
    hipLaunchKernelGGL((k_kate_rows<P>), dim3((unsigned)S, (unsigned)batch), dim3(threads), 0, ctx->stream, d_c, n, d_xs, L,
                       (const uint32_t*)tot, d_q);
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
int poly_kate_division(bzh_ctx* ctx, int field, const uint32_t* d_c, size_t n, size_t batch, const uint32_t* d_xs, uint32_t* d_q) {
#define CALL(PP) kate_t<PP>(ctx, d_c, n, batch, d_xs, d_q)
    BZH_FIELD_SWITCH(field, CALL)
#undef CALL
}

template <class P>
static int inner_t(bzh_ctx* ctx, const uint32_t* a, const uint32_t* b, size_t n, size_t batch, uint32_t* out) {
    if (!batch) return BZH_OK;
    ScopedTimer t(ctx, BZH_T_POLY);
    hipLaunchKernelGGL((k_inner_product<P>), dim3((unsigned)batch), dim3(kVecThreads), 0, ctx->stream, a, b, n, out);
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
int poly_inner_product(bzh_ctx* ctx, int field, const uint32_t* a, const uint32_t* b, size_t n, size_t batch, uint32_t* out) {
#define CALL(PP) inner_t<PP>(ctx, a, b, n, batch, out)
    BZH_FIELD_SWITCH(field, CALL)
#undef CALL
}

template <class P>
static int fold_t(bzh_ctx* ctx, const uint32_t* in, size_t half, size_t batch, const uint32_t* u, size_t u_stride, uint32_t* out) {
    const size_t total = half * batch;
    if (!total) return BZH_OK;
    ScopedTimer t(ctx, BZH_T_POLY);
    hipLaunchKernelGGL((k_fold<P>), dim3((unsigned)((total + kVecThreads - 1) / kVecThreads)), dim3(kVecThreads), 0, ctx->stream, in,
                       half, batch, u, u_stride, out);
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
int poly_fold(bzh_ctx* ctx, int field, const uint32_t* in, size_t half, size_t batch, const uint32_t* u, size_t u_stride,
              uint32_t* out) {
#define CALL(PP) fold_t<PP>(ctx, in, half, batch, u, u_stride, out)
    BZH_FIELD_SWITCH(field, CALL)
#undef CALL
}

template <class P>
static int vec_mul_t(bzh_ctx* ctx, uint32_t* a, const uint32_t* b, size_t count) {
    if (!count) return BZH_OK;
    ScopedTimer t(ctx, BZH_T_POLY);
    hipLaunchKernelGGL((k_vec_mul<P>), dim3((unsigned)((count + kVecThreads - 1) / kVecThreads)), dim3(kVecThreads), 0, ctx->stream,
                       a, b, count);
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
int poly_vec_mul(bzh_ctx* ctx, int field, uint32_t* a, const uint32_t* b, size_t count) {
#define CALL(PP) vec_mul_t<PP>(ctx, a, b, count)
    BZH_FIELD_SWITCH(field, CALL)
#undef CALL
}

}  // namespace bzh
