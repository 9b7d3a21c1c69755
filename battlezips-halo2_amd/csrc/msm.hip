// Multi-scalar multiplication on gfx950: signed-digit Pippenger with the
// per-(vector, chunk, window) counting sort and bucket lists staged in LDS.
//
// Replaces halo2_proofs 0.2.0 `arithmetic::best_multiexp` / `Params::commit*`
// (UPSTREAM, un-vendored: Cargo.lock:382-385) on the create_proof path entered at
// benches/shot.rs:68, benches/board.rs:61-68, src/circuits/shot.rs:921-928,
// src/circuits/board.rs:913-920.  ~28 full-size MSMs per proof, all against the
// same SRS bases, hence the batched entry point.
//
// Pipeline (all on the ctx's stream):
//   k_msm_digits      scalars -> signed c-bit digits, u16 [vector][window][point]
//   k_msm_accumulate  one workgroup (256 threads) per (chunk, window, vector): LDS histogram ->
//                     LDS scan -> LDS scatter (bucket-sorted point list, u16) -> equal slices of that
//                     list per thread, XYZZ mixed additions, points gathered from the HBM/L2-resident
//                     base table (64 B per point), buckets cut by slice boundaries stitched
//   k_msm_chunksum    window-table MSMs with enough vectors: bucket-wise sum of a vector's chunks
//   k_msm_reduce      per segment: sum_m m*B_m via sliced running sums, an LDS suffix scan and LDS
//                     tree sums; writes the Jacobian result itself when the segment is the result
//   k_msm_finalize    otherwise, per vector: sum chunks per window, Horner over windows
// Modular-integer work (v_mad_u64_u32), no MFMA.
#include "ctx.hpp"
#include "curve29.cuh"

namespace bzh {

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
struct MsmPlan {
    int c;           // window bits
    int nwin;        // ceil(256 / c)
    int M;           // buckets per window = 2^(c-1), magnitudes 1..M
    size_t chunk;    // points per chunk (<= 32768: u16 local index + sign bit)
    size_t nchunks;
    int threads;     // accumulate workgroup size
};

static constexpr size_t kMaxChunk = 32768;
static constexpr int kAccThreadsPlan = 256;

// window size for a precomputed table of n points: every window becomes a separate
// row of the table (2^(cw) G_i), so one MSM is a single-window problem over nwin*n points.
static int plan_precompute_c(size_t n) {
    double best = 1e300;
    int best_c = 9;
    for (int c = 6; c <= 13; c++) {
        int nwin = (256 + c - 1) / c;
        double E = (double)nwin * (double)n, M = (double)(1u << (c - 1));
        double chunks = ceil(E / (double)kMaxChunk);
        double cost = E * 10.0 + chunks * (M * 38.0 + 3000.0);
        if (cost < best) {
            best = cost;
            best_c = c;
        }
    }
    return best_c;
}

static MsmPlan plan_msm(size_t n) {
    MsmPlan p;
    p.chunk = n < kMaxChunk ? (n ? n : 1) : kMaxChunk;
    p.nchunks = n ? (n + p.chunk - 1) / p.chunk : 1;
    // cost model (units: field multiplications per segment): balanced accumulation
    // (10 per point, one lost slot per bucket start) + 2 full additions per bucket in
    // the reduction + fixed per-segment overhead (sort, stitch, tree sums).
    double best = 1e300;
    int best_c = 4;
    for (int c = 3; c <= 15; c++) {
        int nwin = (256 + c - 1) / c;
        double M = (double)(1u << (c - 1));
        double cost = nwin * ((double)p.chunk * 10.0 + M * (10.0 + 28.0) + 3000.0);
        if (cost < best) {
            best = cost;
            best_c = c;
        }
    }
    p.c = best_c;
    p.nwin = (256 + p.c - 1) / p.c;
    p.M = 1 << (p.c - 1);
    p.threads = kAccThreadsPlan;
    return p;
}

struct DigitOffset {
    uint32_t l[8];
};

// ---------------------------------------------------------------------------
// k_msm_digits: one thread per scalar.  s' = s + sum_{w < nwin-1} 2^(c-1) 2^(cw);
// digit_w = ((s' >> cw) & (2^c - 1)) - 2^(c-1) for w < nwin-1 and the top window
// keeps the unsigned remainder (<= 2^(c-1) because c*nwin >= 256 > bits(s)).
// u16 encoding: bit 15 = sign, bits 0..14 = magnitude (0 = skip).
// ---------------------------------------------------------------------------
// Paired vectors (MsmPair, the IPA rounds): entry i of a vector belongs to result class 0 or 1
// (class = bit (log_m - 1) of i for i < n_pair; past that, the last two entries are class 1); class-1
// magnitudes are shifted up by M so that the accumulate kernel files them in a second bucket set.
struct MsmPair {
    unsigned log_m;
    size_t n_pair;
    uint32_t M;
};
template <class SF>
__global__ void __launch_bounds__(256) k_msm_digits(const uint32_t* __restrict__ scalars, size_t n, size_t total,
                                                      int form, int c, int nwin, DigitOffset off,
                                                      uint16_t* __restrict__ digits, MsmPair pair) {
    size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= total) return;
    size_t b = gid / n, i = gid - b * n;
    uint32_t cls_off = 0;
    if (pair.M) {
        const bool hi = i < pair.n_pair ? ((i >> (pair.log_m - 1)) & 1) != 0 : (i >= pair.n_pair + 2);
        cls_off = hi ? pair.M : 0u;
    }
    Fe<SF> s = fe_load<SF>(scalars + gid * 8);
    if (form == BZH_FORM_MONTGOMERY) s = fe_from_mont(s);
    uint32_t l[8];
    uint64_t cy = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        cy += (uint64_t)s.l[k] + off.l[k];
        l[k] = (uint32_t)cy;
        cy >>= 32;
    }
    const uint32_t mask = (1u << c) - 1u, H = 1u << (c - 1);
    uint16_t* out = digits + (b * (size_t)nwin) * n + i;
    uint64_t acc = 0;
    int bits = 0, w = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        acc |= (uint64_t)l[k] << bits;
        bits += 32;
        while (bits >= c && w < nwin - 1) {
            int d = (int)((uint32_t)acc & mask) - (int)H;
            uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
            if (mag) mag += cls_off;
            out[(size_t)w * n] = (uint16_t)(mag | (d < 0 ? 0x8000u : 0u));
            acc >>= c;
            bits -= c;
            w++;
        }
    }
    {
        uint32_t mag = (uint32_t)acc & 0x7fffu;  // top window, unsigned
        if (mag) mag += cls_off;
        out[(size_t)w * n] = (uint16_t)mag;
    }
}

// ---------------------------------------------------------------------------
// LDS / global "plane" layout for XYZZ values: 8 planes of uint4 (x lo, x hi,
// y lo, y hi, zz lo, zz hi, zzz lo, zzz hi); element t of plane p sits at
// (p * stride + t) * 16 B, so a wave's access is 64 consecutive 16-byte slots
// (conflict-free ds_*_b128, fully coalesced global dwordx4).
// ---------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ void planes_put(uint4* buf, size_t stride, size_t t, const Xyzz<P>& v) {
    const Fe<P>* f[4] = {&v.x, &v.y, &v.zz, &v.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        buf[(2 * k) * stride + t] = make_uint4(f[k]->l[0], f[k]->l[1], f[k]->l[2], f[k]->l[3]);
        buf[(2 * k + 1) * stride + t] = make_uint4(f[k]->l[4], f[k]->l[5], f[k]->l[6], f[k]->l[7]);
    }
}
template <class P>
__device__ __forceinline__ Xyzz<P> planes_get(const uint4* buf, size_t stride, size_t t) {
    Xyzz<P> v;
    Fe<P>* f[4] = {&v.x, &v.y, &v.zz, &v.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint4 a = buf[(2 * k) * stride + t], b = buf[(2 * k + 1) * stride + t];
        f[k]->l[0] = a.x; f[k]->l[1] = a.y; f[k]->l[2] = a.z; f[k]->l[3] = a.w;
        f[k]->l[4] = b.x; f[k]->l[5] = b.y; f[k]->l[6] = b.z; f[k]->l[7] = b.w;
    }
    return v;
}

template <class P>
__device__ __forceinline__ Affine<P> affine_load(const uint32_t* p) {
    Affine<P> a;
    a.x = fe_load<P>(p);
    a.y = fe_load<P>(p + 8);
    return a;
}

// exclusive scan of a[0..len) in LDS by the whole workgroup; scratch >= 17 words
__device__ __forceinline__ void block_exclusive_scan(uint32_t* a, int len, uint32_t* scratch) {
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (T + 63) >> 6;
    const int per = (len + T - 1) / T;
    const int lo = tid * per, hi = min(lo + per, len);
    uint32_t sum = 0;
    for (int k = lo; k < hi; k++) sum += a[k];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < nw ? scratch[lane] : 0u, wi = w;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            uint32_t v = __shfl_up(wi, d, 64);
            if (lane >= d) wi += v;
        }
        if (lane < nw) scratch[lane] = wi - w;
    }
    __syncthreads();
    uint32_t base = scratch[wave] + incl - sum;
    for (int k = lo; k < hi; k++) {
        uint32_t v = a[k];
        a[k] = base;
        base += v;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// k_msm_accumulate: grid (nchunks, nwin, batch), T threads.  Dynamic LDS:
//   cnt[M + 2] u32 | scratch[32] u32 | idB[T] u32 | sorted[chunk] u16
//
// After the LDS counting sort the bucket-sorted point list is cut into T equal slices, one
// per thread, whatever the bucket boundaries are: every lane performs the same number of
// mixed additions, so skewed digits (the top window holds only 0/1/2, adversarial inputs put
// every point in one bucket) cost nothing extra.  A bucket that lies inside one slice is
// summed and stored by that thread.  A bucket cut by slice boundaries leaves per-thread
// partial sums: the thread where it starts keeps a tail partial A, every later thread a head
// partial B; B is suffix-reduced per bucket across threads (Hillis-Steele over a global
// scratch pair, log2(span) steps) and the starting thread stores A + B'.
// ---------------------------------------------------------------------------
// U29: the running sum of a slice is kept in unsaturated 9 x 29-bit limbs (csrc/fe29.cuh, curve29.cuh: no carry instruction in
// the products, 1.13 x the mixed additions per second at this kernel's two waves per SIMD); table points are re-sliced as they
// arrive and a finished bucket is converted back, so every byte outside the loop -- tables, bucket planes, stitch buffers -- is
// what the saturated variant reads and writes.  The Pasta curves only (p = 1 mod 2^29 with three zero limbs).
template <class P>
__device__ __forceinline__ Xyzz<P> acc29_to_sat(const Xyzz29<P>& a) {
    return xyzz29_to_sat_fast(a);   // product-free (x 2^-5 exactly), inline: no call, no copy of the argument through scratch
}
// A finished running sum leaves the loop RAW: its 36 limb words as they are, limbs 0..7 of the four coordinates in the eight planes
// a saturated bucket occupies and the four ninth limbs in one extra uint4 (identity: all zero).  Lanes end buckets on different
// iterations, so whatever the hand-over costs is paid by the whole wave on nearly every iteration: a conversion there (four
// products and canonical reductions) made the kernel 1.6 x slower than the saturated one.  Raw buckets are converted in place
// by a uniform pass at the end of the kernel.
template <class P>
__device__ __forceinline__ void raw29_put(uint4* planes, size_t stride, size_t t, uint4* ninth, const Xyzz29<P>& v) {
    const Fe29<P>* f[4] = {&v.x, &v.y, &v.zz, &v.zzz};
    const uint32_t keep = v.id ? 0u : 0xffffffffu;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        planes[(2 * k) * stride + t] = make_uint4(f[k]->l[0] & keep, f[k]->l[1] & keep, f[k]->l[2] & keep, f[k]->l[3] & keep);
        planes[(2 * k + 1) * stride + t] = make_uint4(f[k]->l[4] & keep, f[k]->l[5] & keep, f[k]->l[6] & keep, f[k]->l[7] & keep);
    }
    ninth[t] = make_uint4(v.x.l[8] & keep, v.y.l[8] & keep, v.zz.l[8] & keep, v.zzz.l[8] & keep);
}
template <class P>
__device__ __forceinline__ Xyzz29<P> raw29_get(const uint4* planes, size_t stride, size_t t, const uint4* ninth) {
    Xyzz29<P> v;
    Fe29<P>* f[4] = {&v.x, &v.y, &v.zz, &v.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint4 a = planes[(2 * k) * stride + t], b = planes[(2 * k + 1) * stride + t];
        f[k]->l[0] = a.x; f[k]->l[1] = a.y; f[k]->l[2] = a.z; f[k]->l[3] = a.w;
        f[k]->l[4] = b.x; f[k]->l[5] = b.y; f[k]->l[6] = b.z; f[k]->l[7] = b.w;
    }
    const uint4 n = ninth[t];
    v.x.l[8] = n.x, v.y.l[8] = n.y, v.zz.l[8] = n.z, v.zzz.l[8] = n.w;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) any |= v.zz.l[i];
    v.id = any == 0u;     // zz = Z^2 is never 0 mod p for a point, and a raw identity is stored as zeros
    return v;
}
template <class C, int T, bool U29>
__device__ __forceinline__ void msm_accumulate_body(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ bases29,
                                                                  const uint16_t* __restrict__ digits, size_t n, int nwin,
                                                                  int M, size_t chunk, uint4* __restrict__ buckets,
                                                                  uint4* __restrict__ partials, size_t row_len,
                                                                  size_t row_stride, size_t dup_from,
                                                                  unsigned long long* __restrict__ add_counter,
                                                                  size_t vec_col_stride, size_t vec0, uint32_t xcd_vecs,
                                                                  uint32_t xcd_chunks) {
    using P = typename C::Base;
    extern __shared__ __align__(16) uint32_t lds[];
    uint32_t* cnt = lds;  // index m in [0, M]; cnt[0] stays 0
    uint32_t* scratch = lds + (M + 2);
    uint32_t* idB = scratch + 32;
    uint16_t* sorted = reinterpret_cast<uint16_t*>(idB + T);

    const int tid = threadIdx.x;
    // vector index fastest: workgroups dispatched together (and dealt round-robin to the XCDs) work on the SAME chunk of
    // different vectors, i.e. gather from the same 2 MB window of the base table, which then lives in every XCD's L2
    size_t b = blockIdx.x, w = blockIdx.y, ck = blockIdx.z, nchunks = gridDim.z;
    if (xcd_chunks) {
        // XCD-aware order (1-D grid, window tables): workgroups go round-robin to the 8 XCDs by linear id, so id & 7 names the
        // XCD; the (chunk, vector) pairs in chunk-major order are cut into 8 equal runs, one per XCD, so that the workgroups
        // resident on an XCD gather from one or two ~2 MB windows of the table (its L2 holds 4 MB) instead of eight of them
        const size_t lin = blockIdx.x, xcd = lin & 7, i = lin >> 3, total = (size_t)xcd_vecs * xcd_chunks, per = (total + 7) / 8;
        const size_t item = xcd * per + i;
        if (item >= total) return;  // padding of the runs (whole workgroup, before any barrier)
        ck = item / xcd_vecs;
        b = item - ck * xcd_vecs;
        w = 0;
        nchunks = xcd_chunks;
    }
    const size_t c0 = ck * chunk;
    const int len = (int)min(chunk, n - c0);
    const uint16_t* dg = digits + (b * (size_t)nwin + w) * n + c0;

    for (int i = tid; i < M + 2; i += T) cnt[i] = 0;
    if (tid == 0) scratch[31] = 0;  // max bucket population
    __syncthreads();
    // digits are read 8 at a time (one dwordx4 per lane) when the row is 16-byte aligned
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dg) & 15u) == 0;
    const int nvec = vec_ok ? (len >> 3) : 0;
    for (int v = tid; v < nvec; v += T) {
        const uint4 d4 = reinterpret_cast<const uint4*>(dg)[v];
        const uint32_t w4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t m0 = w4[k] & 0x7fffu, m1 = (w4[k] >> 16) & 0x7fffu;
            if (m0) atomicAdd(&cnt[m0], 1u);
            if (m1) atomicAdd(&cnt[m1], 1u);
        }
    }
    for (int i = nvec * 8 + tid; i < len; i += T) {
        uint32_t m = dg[i] & 0x7fffu;
        if (m) atomicAdd(&cnt[m], 1u);
    }
    __syncthreads();
    {
        uint32_t mx = 0;
        for (int i = 1 + tid; i <= M; i += T) mx = max(mx, cnt[i]);
        for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
        if ((tid & 63) == 0) atomicMax(&scratch[31], mx);
    }
    block_exclusive_scan(cnt, M + 1, scratch);
    for (int v = tid; v < nvec; v += T) {
        const uint4 d4 = reinterpret_cast<const uint4*>(dg)[v];
        const uint32_t w4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t d = (w4[k >> 1] >> ((k & 1) * 16)) & 0xffffu, m = d & 0x7fffu;
            if (m) {
                uint32_t pos = atomicAdd(&cnt[m], 1u);
                sorted[pos] = (uint16_t)((uint32_t)(v * 8 + k) | (d & 0x8000u));
            }
        }
    }
    for (int i = nvec * 8 + tid; i < len; i += T) {
        uint32_t d = dg[i], m = d & 0x7fffu;
        if (m) {
            uint32_t pos = atomicAdd(&cnt[m], 1u);
            sorted[pos] = (uint16_t)((uint32_t)i | (d & 0x8000u));
        }
    }
    __syncthreads();
    // bucket m owns sorted[cnt[m-1] .. cnt[m]); cnt[M] = number of non-zero digits
    const size_t segi = (b * (size_t)nwin + w) * nchunks + ck;
    uint4* seg = buckets + segi * (size_t)M * 8;
    // per segment: two stitch buffers of T x 8 planes; U29: + the ninth limbs of raw buckets [M] and of raw head partials [T]
    const size_t part_stride = (size_t)(2 * T) * 8 + (U29 ? (size_t)M + (size_t)T : (size_t)0);
    uint4* pbuf0 = partials + segi * part_stride;
    uint4* pbuf1 = pbuf0 + (size_t)T * 8;
    uint4* raw9 = pbuf1 + (size_t)T * 8;
    uint4* raw9_head = raw9 + (size_t)M;
    (void)raw9_head;
    // Precomputed tables: item e of the flattened [window][point] digit array addresses row
    // e / row_len, column e % row_len of a table whose rows are row_stride points apart
    // (row_len < row_stride when a prefix of the table is used).  row_len == 0: plain bases.
    const bool remap = row_len != 0 && row_len != row_stride;
    // (per-proof tables side by side: vector vec0 + b reads its own columns)
    const uint32_t* base0 = bases + ((vec0 + b) * vec_col_stride + c0) * 16;
    const uint32_t total = cnt[M], maxpop = scratch[31];
    if (add_counter && tid == 0 && total) atomicAdd(add_counter, (unsigned long long)total);   // profiling: bucket additions actually made
    const uint32_t L = (total + T - 1) / T;
    const uint32_t start = min((uint32_t)tid * L, total), end = min(start + L, total);

    // empty buckets are the identity
    for (int m = 1 + tid; m <= M; m += T)
        if (cnt[m] == cnt[m - 1]) planes_put(seg, (size_t)M, (size_t)(m - 1), xyzz_identity<P>());

    // what the slice leaves for the stitch below
    Xyzz<P> acc = xyzz_identity<P>();   // the slice's last partial sum (start of a cut bucket), saturated
    uint32_t head_id = 0, tail_id = 0;  // tail_id: bucket whose sum continues in later threads (acc = A)
    bool tail_through = false;          // ... and that had also begun before this slice (A is the identity)
    // The slice, in one of two arithmetics chosen PER WORKGROUP (uniform): unsaturated limbs when the chunk is dense (on average
    // 8 or more points per bucket: the n-term commitments of h, f, s, the random polynomial, the opening's rounds), saturated
    // when it is sparse (witness columns in the Lagrange basis leave one or two points per bucket: there a bucket's conversion
    // back to the saturated form would cost as much as the additions it holds -- measured: 63.6 ms per batch of 64 with the
    // unsaturated loop everywhere against 39.9 saturated).
    auto slice = [&](auto u29_tag) {
    constexpr bool U = decltype(u29_tag)::value;
    // the slice's running sum: saturated XYZZ, or (U) the unsaturated form
    using Acc = typename std::conditional<U, Xyzz29<P>, Xyzz<P>>::type;
    Acc run;
    Fe29Consts<P> k29;
    if constexpr (U) {
        run = xyzz29_identity<P>();
        k29 = fe29_consts<P>();
    } else {
        run = xyzz_identity<P>();
    }
    auto run_value = [&]() -> Xyzz<P> {
        if constexpr (U) return acc29_to_sat<P>(run);
        else return run;
    };
    auto run_clear = [&] {
        if constexpr (U) run = xyzz29_identity<P>();
        else run = xyzz_identity<P>();
    };
    // hand-over of a finished sum INSIDE the loop: raw (U29) or as it is
    auto put_bucket = [&](uint32_t m) {
        if constexpr (U) raw29_put<P>(seg, (size_t)M, (size_t)(m - 1), raw9, run);
        else planes_put(seg, (size_t)M, (size_t)(m - 1), run);
    };
    bool head_raw = false;   // U29: the head partial in pbuf0 is still raw
    (void)head_raw;
    auto put_head = [&] {
        if constexpr (U) {
            raw29_put<P>(pbuf0, (size_t)T, (size_t)tid, raw9_head, run);
            head_raw = true;
        } else {
            planes_put(pbuf0, (size_t)T, (size_t)tid, run);
        }
    };
    if (start < end) {
        // first bucket with cnt[m] > start
        uint32_t lo = 1, hi = (uint32_t)M;
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (cnt[mid] > start) hi = mid; else lo = mid + 1;
        }
        uint32_t m = lo, bbeg = cnt[m - 1], bend = cnt[m];
        // software pipeline: the gather of item j+1 is in flight while item j is added
        // row / column of the chunk's first item, once per thread: an item is then at most chunk / row_len rows further on, found
        // by subtraction instead of a 64-bit division per item
        const size_t row0 = remap ? c0 / row_len : 0, col0 = remap ? c0 - row0 * row_len : 0;
        auto point_index = [&](uint32_t e) -> size_t {
            size_t pidx = (size_t)(e & 0x7fffu);
            if (remap) {
                size_t row = row0, col = col0 + pidx;
                while (col >= row_len) {
                    col -= row_len;
                    row++;
                }
                if (dup_from && col >= dup_from) col -= 2;  // the last two table columns appear twice (paired vectors)
                pidx = row * row_stride + col - c0;
            }
            return pidx;
        };
        auto bucket_step = [&](uint32_t j) {   // bucket bookkeeping of item j (both arithmetics)
            if (j == bend) {  // bucket m ended inside this slice
                if (bbeg < start) {  // head partial: parked in the stitch buffer, not in registers
                    put_head();
                    head_id = m;
                } else {
                    put_bucket(m);
                }
                run_clear();
                do { m++; } while (cnt[m] <= j);
                bbeg = cnt[m - 1];
                bend = cnt[m];
            }
        };
        if constexpr (U) {
            // operands straight from the table's fe29 copy (20 words per point: x, y in carried limbs below 2 p): nothing to
            // re-slice or fold per addition; the gather of item j+1 is in flight while item j is added
            const uint32_t* base29 = bases29 + ((vec0 + b) * vec_col_stride + c0) * 20;
            auto load29 = [&](uint32_t e, Fe29<P>& x, Fe29<P>& y) {
                const uint4* q = reinterpret_cast<const uint4*>(base29 + point_index(e) * 20);
                const uint4 a = q[0], bb = q[1], c = q[2], d = q[3], f = q[4];
                x.l[0] = a.x, x.l[1] = a.y, x.l[2] = a.z, x.l[3] = a.w, x.l[4] = bb.x, x.l[5] = bb.y, x.l[6] = bb.z, x.l[7] = bb.w, x.l[8] = c.x;
                y.l[0] = c.y, y.l[1] = c.z, y.l[2] = c.w, y.l[3] = d.x, y.l[4] = d.y, y.l[5] = d.z, y.l[6] = d.w, y.l[7] = f.x, y.l[8] = f.y;
            };
            uint32_t e_next = sorted[start];
            Fe29<P> x_next, y_next;
            load29(e_next, x_next, y_next);
            for (uint32_t j = start; j < end; j++) {
                bucket_step(j);
                const uint32_t e = e_next;
                const Fe29<P> qx = x_next;
                Fe29<P> qy = y_next;
                if (j + 1 < end) {
                    e_next = sorted[j + 1];
                    load29(e_next, x_next, y_next);
                }
                uint32_t any = 0;   // the identity is stored as zeros
#pragma unroll
                for (int i = 0; i < 9; i++) any |= qx.l[i] | qy.l[i];
                if (any) {
                    if (e & 0x8000u) qy = fe29_sub<P, 4>(fe29_zero<P>(), qy);   // -y = 4 p - y
                    xyzz29_madd_q29(run, qx, qy, k29);
                }
            }
        } else {
            uint32_t e_next = sorted[start];
            Affine<P> q_next = affine_load<P>(base0 + point_index(e_next) * 16);
            for (uint32_t j = start; j < end; j++) {
                bucket_step(j);
                const uint32_t e = e_next;
                Affine<P> q = q_next;
                if (j + 1 < end) {   // (a second point in flight -- prefetch two additions ahead -- measured no better: 39.7 vs 38.7 ms per batch)
                    e_next = sorted[j + 1];
                    q_next = affine_load<P>(base0 + point_index(e_next) * 16);
                }
                if (!aff_is_id(q)) {
                    if (e & 0x8000u) q.y = fe_neg(q.y);
                    xyzz_madd(run, q);
                }
            }
        }
        // last segment of the slice: bucket m, items [max(bbeg,start), min(bend,end))
        if (bend > end) {
            tail_id = m;
            if (bbeg < start) {  // neither begins nor ends here
                put_head();
                head_id = m;
                tail_through = true;
            } else {
                acc = run_value();   // the start of a cut bucket: the stitch continues with it (once per thread, outside the loop)
            }
        } else if (bbeg < start) {
            put_head();
            head_id = m;
        } else {
            put_bucket(m);
        }
    }
    if constexpr (U) {
        // raw -> saturated, uniformly: every thread its own head partial (nearly every slice begins inside a bucket) ...
        if (head_raw) planes_put(pbuf0, (size_t)T, (size_t)tid, acc29_to_sat<P>(raw29_get<P>(pbuf0, (size_t)T, (size_t)tid, raw9_head)));
        // ... and the buckets that lie inside ONE slice (those were put raw; cut buckets are written by the stitch below, empty ones
        // above), M / T each whoever summed them
        __syncthreads();
        for (int m = 1 + tid; m <= M; m += T) {
            const uint32_t b0 = cnt[m - 1], b1 = cnt[m];
            if (b1 > b0 && b0 / L == (b1 - 1) / L)
                planes_put(seg, (size_t)M, (size_t)(m - 1), acc29_to_sat<P>(raw29_get<P>(seg, (size_t)M, (size_t)(m - 1), raw9)));
        }
    }
    };
    if constexpr (U29) {
        if (bases29 && total >= 8u * (uint32_t)M) slice(std::true_type{});
        else slice(std::false_type{});
    } else {
        slice(std::false_type{});
    }
    // ---- stitch buckets cut by slice boundaries (uniform control flow from here) ----
    idB[tid] = head_id;
    __syncthreads();
    uint32_t span = L ? (maxpop + L - 1) / L + 1 : 1;  // a bucket touches at most this many slices
    if (span > (uint32_t)T) span = (uint32_t)T;
    if (span > 1) {
        uint4* cur = pbuf0;
        uint4* nxt = pbuf1;
        Xyzz<P> headv = xyzz_identity<P>();
        if (head_id) headv = planes_get<P>(cur, (size_t)T, (size_t)tid);  // own store, same thread
        for (uint32_t d = 1; d < span; d <<= 1) {
            __syncthreads();
            if (head_id && tid + d < (uint32_t)T && idB[tid + d] == head_id) {
                Xyzz<P> o = planes_get<P>(cur, (size_t)T, (size_t)(tid + d));
                xyzz_add(headv, o);
            }
            planes_put(nxt, (size_t)T, (size_t)tid, headv);
            uint4* tmp = cur;
            cur = nxt;
            nxt = tmp;
        }
        __syncthreads();
        if (tail_id && !tail_through) {  // this thread holds the start of a cut bucket: finish it
            if (tid + 1 < T && idB[tid + 1] == tail_id) {
                Xyzz<P> o = planes_get<P>(cur, (size_t)T, (size_t)(tid + 1));
                xyzz_add(acc, o);
            }
            planes_put(seg, (size_t)M, (size_t)(tail_id - 1), acc);
        }
    }
}

#define BZH_ACC_PARAMS                                                                                                       \
    const uint32_t *__restrict__ bases, const uint32_t *__restrict__ bases29, const uint16_t *__restrict__ digits, size_t n, int nwin, int M, size_t chunk,              \
        uint4 *__restrict__ buckets, uint4 *__restrict__ partials, size_t row_len, size_t row_stride, size_t dup_from,            \
        unsigned long long *__restrict__ add_counter, size_t vec_col_stride, size_t vec0, uint32_t xcd_vecs, uint32_t xcd_chunks
#define BZH_ACC_ARGS \
    bases, bases29, digits, n, nwin, M, chunk, buckets, partials, row_len, row_stride, dup_from, add_counter, vec_col_stride, vec0, xcd_vecs, xcd_chunks
template <class C, int T, bool U29>
__global__ void __launch_bounds__(T) k_msm_accumulate(BZH_ACC_PARAMS) {
    msm_accumulate_body<C, T, U29>(BZH_ACC_ARGS);
}
#undef BZH_ACC_PARAMS
#undef BZH_ACC_ARGS

// ---------------------------------------------------------------------------
// k_msm_reduce: one workgroup per segment computes sum_{m=1..M} m * B_m.
// Thread t owns slice (tL, (t+1)L]: S_t = sum B_m, W_t = sum (m - tL) B_m by a
// downward running sum.  Then total = sum_t W_t + L * sum_{t>=1} Suffix_t with
// Suffix_t = sum_{u>=t} S_u (Hillis-Steele suffix scan in LDS, tree sums).
// Dynamic LDS: 2 * T * 128 B.
// ---------------------------------------------------------------------------
// result of one reduced segment: the window sum for k_msm_finalize, or (out_xyz != null: the segment IS the result)
// the Jacobian point in the caller's form
template <class P>
__device__ __forceinline__ void reduce_emit(const Xyzz<P>& v, uint4* winsums, size_t nseg, size_t segi, int form, uint32_t* out_xyz) {
    if (!out_xyz) {
        planes_put(winsums, nseg, segi, v);
        return;
    }
    Fe<P> X, Y, Z;
    xyzz_to_jacobian(v, X, Y, Z);
    if (form == BZH_FORM_CANONICAL) {
        X = fe_from_mont(X);
        Y = fe_from_mont(Y);
        Z = fe_from_mont(Z);
    }
    uint32_t* o = out_xyz + segi * 24;
    fe_store(o, X);
    fe_store(o + 8, Y);
    fe_store(o + 16, Z);
}

template <class P>
__device__ __forceinline__ Xyzz<P> block_tree_sum(Xyzz<P> v, uint4* buf, int T) {
    const int tid = threadIdx.x;
    for (int s = T >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        if (tid >= s && tid < 2 * s) planes_put(buf, (size_t)T, (size_t)tid, v);
        __syncthreads();
        if (tid < s) {
            Xyzz<P> o = planes_get<P>(buf, (size_t)T, (size_t)(tid + s));
            xyzz_add_inl(v, o);
        }
    }
    return v;  // valid in thread 0
}

template <class C>
__global__ void __launch_bounds__(256) k_msm_reduce(const uint4* __restrict__ buckets, int M, int nclass, size_t spv,
                                                      size_t aseg_mult, uint4* __restrict__ winsums, int form,
                                                      uint32_t* __restrict__ out_xyz) {
    using P = typename C::Base;
    extern __shared__ __align__(16) uint32_t lds[];
    uint4* bufA = reinterpret_cast<uint4*>(lds);
    const int T = blockDim.x, tid = threadIdx.x;
    uint4* bufB = bufA + (size_t)T * 8;
    // accumulate segment blockIdx.x / nclass holds nclass bucket sets side by side in each plane
    // (aseg_mult > 1: the chunks of a vector were pre-summed into its first segment, only that one is reduced)
    const size_t aseg = blockIdx.x / nclass, cls = blockIdx.x - aseg * nclass, MS = (size_t)M * nclass;
    const size_t segi = ((aseg / spv) * nclass + cls) * spv + aseg % spv;  // result vectors are class-major
    const uint4* seg = buckets + aseg * aseg_mult * MS * 8 + cls * M;
    const int L = M / T;  // host guarantees T <= M, both powers of two

    Xyzz<P> S = xyzz_identity<P>(), W = xyzz_identity<P>();
    for (int k = L; k >= 1; k--) {
        Xyzz<P> bkt = planes_get<P>(seg, MS, (size_t)(tid * L + k - 1));
        xyzz_add_inl(S, bkt);
        xyzz_add_inl(W, S);
    }
    // suffix scan of S over threads
    Xyzz<P> suf = S;
    uint4* cur = bufA;
    uint4* nxt = bufB;
    for (int d = 1; d < T; d <<= 1) {
        planes_put(cur, (size_t)T, (size_t)tid, suf);
        __syncthreads();
        if (tid + d < T) {
            Xyzz<P> o = planes_get<P>(cur, (size_t)T, (size_t)(tid + d));
            xyzz_add_inl(suf, o);
        }
        uint4* tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
    // V_t = W_t + L * Suf_t for t >= 1 (log2 L doublings per thread), then ONE tree sum
    if (tid >= 1) {
        for (int k = L; k > 1; k >>= 1) suf = xyzz_dbl_inl(suf);
        xyzz_add_inl(W, suf);
    }
    Xyzz<P> lo = block_tree_sum(W, bufA, T);
    if (tid == 0) reduce_emit<P>(lo, winsums, (size_t)gridDim.x, segi, form, out_xyz);
}

// ---------------------------------------------------------------------------
// k_msm_reduce_wave: the same sum, ONE wave per segment and no LDS: lane t owns L = M/64
// buckets (2L additions), the suffix scan and the final sum run over cross-lane shuffles
// (6 + 6 steps).  2L + 16 additions per wave instead of (2M/256 + 24) x 4 waves: 2.7x less issued
// work at M = 1024, which is what counts once there are more segments than SIMDs.
// ---------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ Xyzz<P> xyzz_shfl_down(const Xyzz<P>& v, int d) {
    Xyzz<P> o;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        o.x.l[k] = (uint32_t)__shfl_down((int)v.x.l[k], d, 64);
        o.y.l[k] = (uint32_t)__shfl_down((int)v.y.l[k], d, 64);
        o.zz.l[k] = (uint32_t)__shfl_down((int)v.zz.l[k], d, 64);
        o.zzz.l[k] = (uint32_t)__shfl_down((int)v.zzz.l[k], d, 64);
    }
    return o;
}

template <class C>
__global__ void __launch_bounds__(64) k_msm_reduce_wave(const uint4* __restrict__ buckets, int M, int nclass, size_t spv,
                                                         size_t aseg_mult, uint4* __restrict__ winsums, int form,
                                                         uint32_t* __restrict__ out_xyz) {
    using P = typename C::Base;
    const int lane = threadIdx.x;
    const size_t aseg = blockIdx.x / nclass, cls = blockIdx.x - aseg * nclass, MS = (size_t)M * nclass;
    const size_t segi = ((aseg / spv) * nclass + cls) * spv + aseg % spv;
    const uint4* seg = buckets + aseg * aseg_mult * MS * 8 + cls * M;
    const int L = M >> 6;  // host guarantees M >= 64, a power of two

    Xyzz<P> S = xyzz_identity<P>(), W = xyzz_identity<P>();
    for (int k = L; k >= 1; k--) {
        Xyzz<P> bkt = planes_get<P>(seg, MS, (size_t)(lane * L + k - 1));
        xyzz_add_inl(S, bkt);
        xyzz_add_inl(W, S);
    }
    // inclusive suffix sums of S across lanes
    Xyzz<P> suf = S;
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
        Xyzz<P> o = xyzz_shfl_down(suf, d);
        if (lane + d < 64) xyzz_add_inl(suf, o);
    }
    // V_t = W_t + L * Suf_t (t >= 1), then sum V over lanes
    if (lane >= 1) {
        for (int k = L; k > 1; k >>= 1) suf = xyzz_dbl_inl(suf);
        xyzz_add_inl(W, suf);
    }
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
        Xyzz<P> o = xyzz_shfl_down(W, d);
        if (lane < d) xyzz_add_inl(W, o);
    }
    if (lane == 0) reduce_emit<P>(W, winsums, (size_t)gridDim.x, segi, form, out_xyz);
}

// ---------------------------------------------------------------------------
// Latency mode (a handful of segments: a single proof's MSMs).  The same reduction with FOUR lanes per logical lane
// (xyzz_add_quad, csrc/curve.cuh): 16 logical lanes per wave, lane t owns L = M/16 buckets, the suffix scan and the final sum
// run over 4 + 4 shuffle steps of 4 * d lanes.  2L + 8 quad additions of ~1 500 issue slots instead of 2(M/64) + 12 of ~3 700.
// ---------------------------------------------------------------------------
// The arithmetic of the quad reductions as a policy: saturated XYZZ, or (Pasta curves) unsaturated limbs end to end -- buckets are
// converted as they are loaded (lane ql of a quad converts coordinate ql, broadcasts), the running sums, their shuffles and their
// LDS exchanges stay in 9 x 29-bit limbs, one product-free conversion at the very end.  Four products of 188 instructions per lane
// and addition instead of four of 297.
template <class P, bool U>
struct QuadRep;
template <class P>
struct QuadRep<P, false> {
    using X = Xyzz<P>;
    static constexpr int kPlanes = 8;
    static __device__ __forceinline__ X identity() { return xyzz_identity<P>(); }
    static __device__ __forceinline__ X load(const uint4* seg, size_t stride, size_t idx, int) { return planes_get<P>(seg, stride, idx); }
    static __device__ __forceinline__ void add(X& a, const X& b, int ql) { xyzz_add_quad(a, b, ql); }
    static __device__ __forceinline__ X dbl(const X& a) { return xyzz_dbl_inl(a); }
    static __device__ __forceinline__ X shfl_down(const X& a, int d) { return xyzz_shfl_down(a, d); }
    static __device__ __forceinline__ void lds_put(uint4* buf, int TL, int lt, const X& v) { planes_put(buf, (size_t)TL, (size_t)lt, v); }
    static __device__ __forceinline__ X lds_get(const uint4* buf, int TL, int lt) { return planes_get<P>(buf, (size_t)TL, (size_t)lt); }
    static __device__ __forceinline__ Xyzz<P> sat(const X& v) { return v; }
};
template <class P>
struct QuadRep<P, true> {
    using X = Xyzz29<P>;
    static constexpr int kPlanes = 9;
    static __device__ __forceinline__ X identity() { return xyzz29_identity<P>(); }
    static __device__ __forceinline__ X load(const uint4* seg, size_t stride, size_t idx, int ql) {
        return xyzz29_from_sat_quad(planes_get<P>(seg, stride, idx), ql);
    }
    static __device__ __forceinline__ void add(X& a, const X& b, int ql) { xyzz29_add_quad(a, b, ql); }
    static __device__ __forceinline__ X dbl(const X& a) { return xyzz29_dbl(a); }
    static __device__ __forceinline__ X shfl_down(const X& a, int d) { return xyzz29_shfl_down(a, d); }
    static __device__ __forceinline__ void lds_put(uint4* buf, int TL, int lt, const X& v) { raw29_put<P>(buf, (size_t)TL, (size_t)lt, buf + (size_t)8 * TL, v); }
    static __device__ __forceinline__ X lds_get(const uint4* buf, int TL, int lt) { return raw29_get<P>(buf, (size_t)TL, (size_t)lt, buf + (size_t)8 * TL); }
    static __device__ __forceinline__ Xyzz<P> sat(const X& v) { return xyzz29_to_sat_fast(v); }
};

template <class C, bool U>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_msm_reduce_quad(const uint4* __restrict__ buckets, int M, int nclass, size_t spv,
                                                         size_t aseg_mult, uint4* __restrict__ winsums, int form,
                                                         uint32_t* __restrict__ out_xyz) {
    using P = typename C::Base;
    using Q = QuadRep<P, U>;
    using X = typename Q::X;
    const int lane = threadIdx.x, ll = lane >> 2, ql = lane & 3;
    const size_t aseg = blockIdx.x / nclass, cls = blockIdx.x - aseg * nclass, MS = (size_t)M * nclass;
    const size_t segi = ((aseg / spv) * nclass + cls) * spv + aseg % spv;
    const uint4* seg = buckets + aseg * aseg_mult * MS * 8 + cls * M;
    const int L = M >> 4;  // host guarantees M >= 16, a power of two

    X S = Q::identity(), W = Q::identity();
    for (int k = L; k >= 1; k--) {
        const X bkt = Q::load(seg, MS, (size_t)(ll * L + k - 1), ql);   // the four lanes of a quad read the same bucket
        Q::add(S, bkt, ql);
        Q::add(W, S, ql);
    }
    X suf = S;
#pragma unroll 1
    for (int d = 1; d < 16; d <<= 1) {
        const X o = Q::shfl_down(suf, 4 * d);
        if (ll + d < 16) Q::add(suf, o, ql);
    }
    if (ll >= 1) {
        for (int k = L; k > 1; k >>= 1) suf = Q::dbl(suf);
        Q::add(W, suf, ql);
    }
#pragma unroll 1
    for (int d = 8; d >= 1; d >>= 1) {
        const X o = Q::shfl_down(W, 4 * d);
        if (ll < d) Q::add(W, o, ql);
    }
    if (lane == 0) reduce_emit<P>(Q::sat(W), winsums, (size_t)gridDim.x, segi, form, out_xyz);
}

// The same with a whole workgroup: 256 threads = 64 logical lanes (lt = tid / 4), L = M/64 buckets each; the suffix scan and
// the final tree run over LDS planes written by lane 0 of every quad (2 * T_l * kPlanes * 16 B of dynamic LDS): 2L + 12 quad additions.
template <class P>
__device__ __forceinline__ Xyzz<P> block_tree_sum_quad(Xyzz<P> v, uint4* buf, int TL, int lt, int ql) {
    for (int s = TL >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        if (lt >= s && lt < 2 * s && ql == 0) planes_put(buf, (size_t)TL, (size_t)lt, v);
        __syncthreads();
        if (lt < s) {
            const Xyzz<P> o = planes_get<P>(buf, (size_t)TL, (size_t)(lt + s));
            xyzz_add_quad(v, o, ql);
        }
    }
    return v;  // valid in logical lane 0
}
template <class Q>
__device__ __forceinline__ typename Q::X block_tree_sum_quad_rep(typename Q::X v, uint4* buf, int TL, int lt, int ql) {
    for (int s = TL >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        if (lt >= s && lt < 2 * s && ql == 0) Q::lds_put(buf, TL, lt, v);
        __syncthreads();
        if (lt < s) {
            const typename Q::X o = Q::lds_get(buf, TL, lt + s);
            Q::add(v, o, ql);
        }
    }
    return v;  // valid in logical lane 0
}
template <class C, bool U>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_msm_reduce_quad_wg(
    const uint4* __restrict__ buckets, int M, int nclass, size_t spv, size_t aseg_mult, uint4* __restrict__ winsums, int form,
    uint32_t* __restrict__ out_xyz) {
    using P = typename C::Base;
    using Q = QuadRep<P, U>;
    using X = typename Q::X;
    extern __shared__ __align__(16) uint32_t lds[];
    const int tid = threadIdx.x, lt = tid >> 2, ql = tid & 3, TL = blockDim.x >> 2;
    uint4* bufA = reinterpret_cast<uint4*>(lds);
    uint4* bufB = bufA + (size_t)TL * Q::kPlanes;
    const size_t aseg = blockIdx.x / nclass, cls = blockIdx.x - aseg * nclass, MS = (size_t)M * nclass;
    const size_t segi = ((aseg / spv) * nclass + cls) * spv + aseg % spv;
    const uint4* seg = buckets + aseg * aseg_mult * MS * 8 + cls * M;
    const int L = M / TL;  // host guarantees TL <= M, both powers of two

    X S = Q::identity(), W = Q::identity();
    for (int k = L; k >= 1; k--) {
        const X bkt = Q::load(seg, MS, (size_t)(lt * L + k - 1), ql);
        Q::add(S, bkt, ql);
        Q::add(W, S, ql);
    }
    X suf = S;
    uint4* cur = bufA;
    uint4* nxt = bufB;
    for (int d = 1; d < TL; d <<= 1) {
        if (ql == 0) Q::lds_put(cur, TL, lt, suf);
        __syncthreads();
        if (lt + d < TL) {
            const X o = Q::lds_get(cur, TL, lt + d);
            Q::add(suf, o, ql);
        }
        uint4* tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
    if (lt >= 1) {
        for (int k = L; k > 1; k >>= 1) suf = Q::dbl(suf);
        Q::add(W, suf, ql);
    }
    const X lo = block_tree_sum_quad_rep<Q>(W, bufA, TL, lt, ql);
    if (tid == 0) reduce_emit<P>(Q::sat(lo), winsums, (size_t)gridDim.x, segi, form, out_xyz);
}

// k_msm_finalize in latency mode: per window the chunk results are summed by 64 logical lanes (quads of a 256-thread
// workgroup) and a 6-step LDS tree; the Horner pass over the windows runs on quad 0.
template <class C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_msm_finalize_quad(
    const uint4* __restrict__ winsums, size_t nseg, int nwin, size_t nchunks, int c, int form, uint32_t* __restrict__ out_xyz) {
    using P = typename C::Base;
    __shared__ __align__(16) uint4 buf[64 * 8];
    __shared__ __align__(16) uint4 wbuf[64 * 8];  // per-window sums, nwin <= 64
    const int tid = threadIdx.x, lt = tid >> 2, ql = tid & 3;   // 64 logical lanes
    const size_t b = blockIdx.x;
    for (int w = 0; w < nwin; w++) {
        Xyzz<P> acc = xyzz_identity<P>();
        for (size_t ck = lt; ck < nchunks; ck += 64) {
            const Xyzz<P> v = planes_get<P>(winsums, nseg, (b * (size_t)nwin + w) * nchunks + ck);
            xyzz_add_quad(acc, v, ql);
        }
        if (nchunks > 1) acc = block_tree_sum_quad(acc, buf, 64, lt, ql);
        if (tid == 0) planes_put(wbuf, 64, (size_t)w, acc);
    }
    __syncthreads();
    if (lt == 0) {
        Xyzz<P> acc = xyzz_identity<P>();
        for (int w = nwin - 1; w >= 0; w--) {
            for (int k = 0; k < c; k++) acc = xyzz_dbl_inl(acc);
            const Xyzz<P> v = planes_get<P>(wbuf, 64, (size_t)w);
            xyzz_add_quad(acc, v, ql);
        }
        if (tid == 0) {
            Fe<P> X, Y, Z;
            xyzz_to_jacobian(acc, X, Y, Z);
            if (form == BZH_FORM_CANONICAL) {
                X = fe_from_mont(X);
                Y = fe_from_mont(Y);
                Z = fe_from_mont(Z);
            }
            uint32_t* o = out_xyz + b * 24;
            fe_store(o, X);
            fe_store(o + 8, Y);
            fe_store(o + 16, Z);
        }
    }
}

// ---------------------------------------------------------------------------
// k_msm_chunksum: bucket-wise sum of a vector's chunk segments into its first segment.  All lanes do useful
// additions, unlike the running-sum reduction, whose per-segment cost this removes for every chunk but one:
// (nchunks - 1) additions per bucket here against 2 per bucket AND per chunk there, plus its scan overhead.
// Grid: (buckets / 64) x vectors; the four waves of a workgroup share 64 buckets and each sums a quarter of the
// chunks (next segment's load in flight during the addition), then two LDS combining steps: a dependent chain of
// nchunks / 4 + 2 additions (~4.5 us each at this occupancy) instead of nchunks - 1.
// ---------------------------------------------------------------------------
// The element the sums run on: the saturated XYZZ point, or (U) its unsaturated-limb form -- converted once per loaded bucket
// (4 re-slices) and once per stored sum, 14 products of 188 instead of 297 instructions per addition in between.
template <class P, bool U>
struct SumRep;
template <class P>
struct SumRep<P, false> {
    using X = Xyzz<P>;
    static constexpr int kPlanes = 8;
    static __device__ __forceinline__ X identity() { return xyzz_identity<P>(); }
    static __device__ __forceinline__ X from_sat(const Xyzz<P>& v) { return v; }
    static __device__ __forceinline__ void add(X& a, const X& b) { xyzz_add_inl(a, b); }
    static __device__ __forceinline__ void lds_put(uint4* buf, int lane, const X& v) { planes_put(buf, (size_t)64, (size_t)lane, v); }
    static __device__ __forceinline__ X lds_get(const uint4* buf, int lane) { return planes_get<P>(buf, (size_t)64, (size_t)lane); }
    static __device__ __forceinline__ Xyzz<P> sat(const X& v) { return v; }
};
template <class P>
struct SumRep<P, true> {
    using X = Xyzz29<P>;
    static constexpr int kPlanes = 9;
    static __device__ __forceinline__ X identity() { return xyzz29_identity<P>(); }
    static __device__ __forceinline__ X from_sat(const Xyzz<P>& v) {   // (re-slice + fold per coordinate, no product)
        X r;
        r.x = fe29_from_sat_reduced(v.x), r.y = fe29_from_sat_reduced(v.y), r.zz = fe29_from_sat_reduced(v.zz), r.zzz = fe29_from_sat_reduced(v.zzz);
        r.id = xyzz_is_id(v);
        return r;
    }
    static __device__ __forceinline__ void add(X& a, const X& b) { xyzz29_add_nocall(a, b); }
    static __device__ __forceinline__ void lds_put(uint4* buf, int lane, const X& v) { raw29_put<P>(buf, (size_t)64, (size_t)lane, buf + 8 * 64, v); }
    static __device__ __forceinline__ X lds_get(const uint4* buf, int lane) { return raw29_get<P>(buf, (size_t)64, (size_t)lane, buf + 8 * 64); }
    static __device__ __forceinline__ Xyzz<P> sat(const X& v) { return xyzz29_to_sat_fast(v); }
};
template <class C, bool U>
__global__ void __launch_bounds__(256) k_msm_chunksum(uint4* __restrict__ buckets, int MS, size_t nchunks) {
    using P = typename C::Base;
    using R = SumRep<P, U>;
    using X = typename R::X;
    constexpr int KP = R::kPlanes;
    __shared__ __align__(16) uint4 part[3 * 64 * KP];  // partial sums of waves 1..3, plane layout (stride 64)
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const size_t m = blockIdx.x * (size_t)64 + lane, v = blockIdx.y;
    const bool live = m < (size_t)MS;
    uint4* seg0 = buckets + v * nchunks * (size_t)MS * 8;
    // wave q sums chunks [c0, c1)
    const size_t per = (nchunks + 3) / 4, c0 = min((size_t)q * per, nchunks), c1 = min(c0 + per, nchunks);
    X acc = R::identity();
    if (live && c0 < c1) {
        Xyzz<P> nxt = planes_get<P>(seg0 + c0 * (size_t)MS * 8, (size_t)MS, m);
        acc = R::from_sat(nxt);
        if (c0 + 1 < c1) nxt = planes_get<P>(seg0 + (c0 + 1) * (size_t)MS * 8, (size_t)MS, m);
        for (size_t ck = c0 + 1; ck < c1; ck++) {
            const X o = R::from_sat(nxt);
            if (ck + 1 < c1) nxt = planes_get<P>(seg0 + (ck + 1) * (size_t)MS * 8, (size_t)MS, m);
            R::add(acc, o);
        }
    }
    // waves 1..3 park their sums; 0 += 1 and 2 += 3, then 0 += 2 (ONE addition site in a two-step loop: three inlined copies of
    // the unsaturated addition cost the kernel its second wave per SIMD)
#pragma unroll 1
    for (int step = 0; step < 2; step++) {
        const bool put = step == 0 ? q != 0 : q == 2, take = step == 0 ? (q == 0 || q == 2) : q == 0;
        const int put_slot = step == 0 ? q - 1 : 1, take_slot = step == 0 ? q : 1;
        if (put) R::lds_put(part + (size_t)put_slot * 64 * KP, lane, acc);
        __syncthreads();
        if (take) {
            const X o = R::lds_get(part + (size_t)take_slot * 64 * KP, lane);
            R::add(acc, o);
        }
        __syncthreads();
    }
    if (q == 0 && live) planes_put(seg0, (size_t)MS, m, R::sat(acc));
}

// ---------------------------------------------------------------------------
// k_msm_finalize: one wave per vector.  winsums planes are indexed by
// seg = (b * nwin + w) * nchunks + ck with stride nseg.
// ---------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(64) k_msm_finalize(const uint4* __restrict__ winsums, size_t nseg, int nwin,
                                                       size_t nchunks, int c, int form, uint32_t* __restrict__ out_xyz) {
    using P = typename C::Base;
    __shared__ __align__(16) uint4 buf[64 * 8];
    __shared__ __align__(16) uint4 wbuf[64 * 8];  // per-window sums, nwin <= 64
    const int lane = threadIdx.x;
    const size_t b = blockIdx.x;
    for (int w = 0; w < nwin; w++) {
        Xyzz<P> acc = xyzz_identity<P>();
        for (size_t ck = lane; ck < nchunks; ck += 64) {
            Xyzz<P> v = planes_get<P>(winsums, nseg, (b * (size_t)nwin + w) * nchunks + ck);
            xyzz_add_inl(acc, v);
        }
        if (nchunks > 1) acc = block_tree_sum(acc, buf, 64);
        if (lane == 0) planes_put(wbuf, 64, (size_t)w, acc);
    }
    __syncthreads();
    if (lane == 0) {
        Xyzz<P> acc = xyzz_identity<P>();
        for (int w = nwin - 1; w >= 0; w--) {
            for (int k = 0; k < c; k++) acc = xyzz_dbl_inl(acc);
            Xyzz<P> v = planes_get<P>(wbuf, 64, (size_t)w);
            xyzz_add_inl(acc, v);
        }
        Fe<P> X, Y, Z;
        xyzz_to_jacobian(acc, X, Y, Z);
        if (form == BZH_FORM_CANONICAL) {
            X = fe_from_mont(X);
            Y = fe_from_mont(Y);
            Z = fe_from_mont(Z);
        }
        uint32_t* o = out_xyz + b * 24;
        fe_store(o, X);
        fe_store(o + 8, Y);
        fe_store(o + 16, Z);
    }
}

template <class P>
__global__ void __launch_bounds__(256) k_to_montgomery(uint32_t* data, size_t count) {
    size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= count) return;
    Fe<P> v = fe_load<P>(data + gid * 8);
    fe_store(data + gid * 8, fe_to_mont(v));
}

// ---------------------------------------------------------------------------
// k_expand_bases: table[w*n + i] = 2^(c*w) * G_i, affine, for w = 1..nwin-1 (row 0 is the
// input).  One thread per point walks the rows: c doublings in XYZZ, then back to affine.
// ---------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(256) k_expand_bases(uint32_t* __restrict__ table, size_t n, int c, int nwin) {
    using P = typename C::Base;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<P> p = affine_load<P>(table + i * 16);
    for (int w = 1; w < nwin; w++) {
        if (!aff_is_id(p)) {
            Xyzz<P> q = xyzz_dbl_affine(p);
            for (int k = 1; k < c; k++) q = xyzz_dbl(q);
            p = xyzz_to_affine(q);
        }
        uint32_t* o = table + ((size_t)w * n + i) * 16;
        fe_store(o, p.x);
        fe_store(o + 8, p.y);
    }
}

// ---------------------------------------------------------------------------
// Global-sort path for window-table MSMs in the many-vector regime (msm_run_t: use_gs).
// The LDS-chunk kernel above keeps one bucket set PER CHUNK and pays for it in the reduction (chunk pre-sum +
// running sums: as expensive as the accumulation once c > 11).  Here a vector's items are sorted by bucket ONCE,
// across chunks (histogram per chunk -> offsets per (bucket, chunk) -> scatter), so that there is one bucket set per
// vector, the reduction shrinks by the chunk count and wider windows (fewer table rows = fewer additions) pay off.
//   k_gs_hist     per (chunk, vector): LDS histogram of the chunk's digits                 -> hist[b][ck][m]
//   k_gs_scan     per vector: bucket ends E[m], offsets hist[b][ck][m] := start of (m, ck), identity for empty buckets
//   k_gs_scatter  per (chunk, vector): item index | sign << 31 to sorted[b][position]
//   k_gs_accumulate per (region, vector): the equal-slice accumulation of k_msm_accumulate over its region of the
//                 sorted list; buckets cut by REGION boundaries leave a head / tail partial per workgroup
//   k_gs_stitch   per vector: joins those workgroup partials (a handful per vector)
// ---------------------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(T) k_gs_hist(const uint16_t* __restrict__ digits, size_t n_eff, size_t chunk, int M,
                                                uint32_t* __restrict__ hist) {
    extern __shared__ __align__(16) uint32_t lds[];
    uint32_t* cnt = lds;  // [0, M]
    const int tid = threadIdx.x;
    const size_t ck = blockIdx.x, b = blockIdx.y, nchunks = gridDim.x;
    const size_t c0 = ck * chunk;
    const int len = (int)min(chunk, n_eff - c0);
    const uint16_t* dg = digits + b * n_eff + c0;
    for (int i = tid; i <= M; i += T) cnt[i] = 0;
    __syncthreads();
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dg) & 15u) == 0;
    const int nvec = vec_ok ? (len >> 3) : 0;
    for (int v = tid; v < nvec; v += T) {
        const uint4 d4 = reinterpret_cast<const uint4*>(dg)[v];
        const uint32_t w4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t m0 = w4[k] & 0x7fffu, m1 = (w4[k] >> 16) & 0x7fffu;
            if (m0) atomicAdd(&cnt[m0], 1u);
            if (m1) atomicAdd(&cnt[m1], 1u);
        }
    }
    for (int i = nvec * 8 + tid; i < len; i += T) {
        const uint32_t m = dg[i] & 0x7fffu;
        if (m) atomicAdd(&cnt[m], 1u);
    }
    __syncthreads();
    uint32_t* out = hist + (b * nchunks + ck) * (size_t)(M + 1);
    for (int i = tid; i <= M; i += T) out[i] = cnt[i];
}

// one workgroup per vector.  ends[b][m] = end of bucket m in the sorted list (ends[b][0] = 0, ends[b][M] = item count);
// hist[b][ck][m] becomes the first position of chunk ck's share of bucket m; maxpop[b] = largest bucket; empty buckets
// are set to the identity here (k_gs_accumulate only writes buckets that hold items).
template <class C>
__global__ void __launch_bounds__(1024) k_gs_scan(uint32_t* __restrict__ hist, size_t nchunks, int M, uint32_t* __restrict__ ends,
                                                   uint32_t* __restrict__ maxpop, uint4* __restrict__ buckets) {
    using P = typename C::Base;
    extern __shared__ __align__(16) uint32_t lds[];
    uint32_t* tot = lds;  // [0, M]: bucket populations, then (exclusive scan) bucket starts
    uint32_t* scratch = lds + (M + 2);
    const int tid = threadIdx.x, T = blockDim.x;
    const size_t b = blockIdx.x, hs = (size_t)(M + 1);
    uint32_t* h = hist + b * nchunks * hs;
    if (tid == 0) scratch[31] = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (int m = tid; m <= M; m += T) {
        uint32_t run = 0;
        for (size_t ck = 0; ck < nchunks; ck++) {
            const uint32_t v = h[ck * hs + m];
            h[ck * hs + m] = run;  // exclusive prefix over the chunks
            run += v;
        }
        tot[m] = run;
        if (m == M) scratch[30] = run;
        if (m) mx = max(mx, run);
    }
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if ((tid & 63) == 0) atomicMax(&scratch[31], mx);
    __syncthreads();
    uint4* bk = buckets + b * (size_t)M * 8;
    for (int m = 1 + tid; m <= M; m += T)
        if (tot[m] == 0) planes_put(bk, (size_t)M, (size_t)(m - 1), xyzz_identity<P>());
    const uint32_t maxp = scratch[31], last_pop = scratch[30];
    __syncthreads();
    block_exclusive_scan(tot, M + 1, scratch);  // tot[m] = first position of bucket m (bucket 0 holds nothing)
    uint32_t* e = ends + b * hs;
    for (int m = tid; m <= M; m += T) {
        const uint32_t base = tot[m];
        for (size_t ck = 0; ck < nchunks; ck++) h[ck * hs + m] += base;
        e[m] = m < M ? tot[m + 1] : base + last_pop;
    }
    if (tid == 0) maxpop[b] = maxp;
}

template <int T>
__global__ void __launch_bounds__(T) k_gs_scatter(const uint16_t* __restrict__ digits, size_t n_eff, size_t chunk, int M,
                                                   const uint32_t* __restrict__ hist, uint32_t* __restrict__ sorted) {
    extern __shared__ __align__(16) uint32_t lds[];
    uint32_t* cnt = lds;  // next free position of every bucket for this chunk
    const int tid = threadIdx.x;
    const size_t ck = blockIdx.x, b = blockIdx.y, nchunks = gridDim.x;
    const size_t c0 = ck * chunk;
    const int len = (int)min(chunk, n_eff - c0);
    const uint16_t* dg = digits + b * n_eff + c0;
    const uint32_t* off = hist + (b * nchunks + ck) * (size_t)(M + 1);
    for (int i = tid; i <= M; i += T) cnt[i] = off[i];
    __syncthreads();
    uint32_t* srt = sorted + b * n_eff;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dg) & 15u) == 0;
    const int nvec = vec_ok ? (len >> 3) : 0;
    for (int v = tid; v < nvec; v += T) {
        const uint4 d4 = reinterpret_cast<const uint4*>(dg)[v];
        const uint32_t w4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t d = (w4[k >> 1] >> ((k & 1) * 16)) & 0xffffu, m = d & 0x7fffu;
            if (m) {
                const uint32_t pos = atomicAdd(&cnt[m], 1u);
                srt[pos] = (uint32_t)(c0 + (size_t)(v * 8 + k)) | ((d & 0x8000u) << 16);
            }
        }
    }
    for (int i = nvec * 8 + tid; i < len; i += T) {
        const uint32_t d = dg[i], m = d & 0x7fffu;
        if (m) {
            const uint32_t pos = atomicAdd(&cnt[m], 1u);
            srt[pos] = (uint32_t)(c0 + (size_t)i) | ((d & 0x8000u) << 16);
        }
    }
}

// grid (regions, vectors).  Region w of a vector covers sorted positions [w * Lw, (w+1) * Lw); inside it the T threads
// take equal slices exactly as in k_msm_accumulate.  wg_head / wg_tail: one XYZZ value per (vector, region) each
// (planes of stride regions * vectors), ids in wg_ids[(b * regions + w) * 2 + {0: head, 1: tail}] (0 = none).
template <class C, int T>
__global__ void __launch_bounds__(T) k_gs_accumulate(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                      const uint32_t* __restrict__ ends, const uint32_t* __restrict__ maxpops,
                                                      size_t n_eff, int M, uint4* __restrict__ buckets, uint4* __restrict__ partials,
                                                      uint4* __restrict__ wg_head, uint4* __restrict__ wg_tail,
                                                      uint32_t* __restrict__ wg_ids, size_t row_len, size_t row_stride,
                                                      size_t dup_from) {
    using P = typename C::Base;
    __shared__ uint32_t idB[T];
    extern __shared__ __align__(16) uint32_t sl[];  // the region's slice of the sorted list (coalesced load, strided use)
    const int tid = threadIdx.x;
    const size_t w = blockIdx.x, b = blockIdx.y, nreg = gridDim.x, nwg_total = (size_t)gridDim.x * gridDim.y;
    const size_t wgi = b * nreg + w;
    const uint32_t* cnt = ends + b * (size_t)(M + 1);  // cnt[m] = end of bucket m, cnt[0] = 0
    const uint32_t* srt = sorted + b * n_eff;
    uint4* seg = buckets + b * (size_t)M * 8;
    uint4* pbuf0 = partials + wgi * (size_t)(2 * T) * 8;
    uint4* pbuf1 = pbuf0 + (size_t)T * 8;
    const uint32_t total = cnt[M], maxpop = maxpops[b];
    const uint32_t Lw = (uint32_t)((total + nreg - 1) / nreg);
    const uint32_t W0 = min((uint32_t)(w * Lw), total), W1 = min(W0 + Lw, total);
    const uint32_t L = (W1 - W0 + T - 1) / T;
    const uint32_t start = min(W0 + (uint32_t)tid * L, W1), end = min(start + L, W1);
    for (uint32_t i = tid; i < W1 - W0; i += T) sl[i] = srt[W0 + i];
    __syncthreads();

    Xyzz<P> acc = xyzz_identity<P>();
    uint32_t head_id = 0, tail_id = 0;
    bool tail_through = false;
    if (start < end) {
        uint32_t lo = 1, hi = (uint32_t)M;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cnt[mid] > start) hi = mid; else lo = mid + 1;
        }
        uint32_t m = lo, bbeg = cnt[m - 1], bend = cnt[m];
        auto point_addr = [&](uint32_t e) -> const uint32_t* {
            const size_t g = (size_t)(e & 0x7fffffffu), row = g / row_len;
            size_t col = g - row * row_len;
            if (dup_from && col >= dup_from) col -= 2;
            return bases + (row * row_stride + col) * 16;
        };
        uint32_t e_next = sl[start - W0];
        Affine<P> q_next = affine_load<P>(point_addr(e_next));
        for (uint32_t j = start; j < end; j++) {
            if (j == bend) {  // bucket m ended inside this slice
                if (bbeg < start) {
                    planes_put(pbuf0, (size_t)T, (size_t)tid, acc);
                    head_id = m;
                } else {
                    planes_put(seg, (size_t)M, (size_t)(m - 1), acc);
                }
                acc = xyzz_identity<P>();
                do { m++; } while (cnt[m] <= j);
                bbeg = cnt[m - 1];
                bend = cnt[m];
            }
            const uint32_t e = e_next;
            Affine<P> q = q_next;
            if (j + 1 < end) {
                e_next = sl[j + 1 - W0];
                q_next = affine_load<P>(point_addr(e_next));
            }
            if (!aff_is_id(q)) {
                if (e & 0x80000000u) q.y = fe_neg(q.y);
                xyzz_madd(acc, q);
            }
        }
        if (bend > end) {
            tail_id = m;
            if (bbeg < start) {  // neither begins nor ends here
                planes_put(pbuf0, (size_t)T, (size_t)tid, acc);
                head_id = m;
                acc = xyzz_identity<P>();
                tail_through = true;
            }
        } else if (bbeg < start) {
            planes_put(pbuf0, (size_t)T, (size_t)tid, acc);
            head_id = m;
        } else {
            planes_put(seg, (size_t)M, (size_t)(m - 1), acc);
        }
    }
    // ---- stitch buckets cut by slice boundaries inside the region (uniform control flow from here) ----
    idB[tid] = head_id;
    __syncthreads();
    uint32_t span = L ? (maxpop + L - 1) / L + 1 : 1;
    if (span > (uint32_t)T) span = (uint32_t)T;
    uint4* cur = pbuf0;
    uint4* nxt = pbuf1;
    Xyzz<P> headv = xyzz_identity<P>();
    if (head_id) headv = planes_get<P>(cur, (size_t)T, (size_t)tid);
    for (uint32_t d = 1; d < span; d <<= 1) {
        __syncthreads();
        if (head_id && tid + d < (uint32_t)T && idB[tid + d] == head_id) {
            const Xyzz<P> o = planes_get<P>(cur, (size_t)T, (size_t)(tid + d));
            xyzz_add(headv, o);
        }
        planes_put(nxt, (size_t)T, (size_t)tid, headv);
        uint4* tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
    __syncthreads();
    // thread 0's run of head partials belongs to a bucket that began in an earlier region: hand it to k_gs_stitch
    if (tid == 0) {
        uint32_t hid = 0;
        if (head_id && cnt[head_id - 1] < W0) {
            hid = head_id;
            planes_put(wg_head, nwg_total, wgi, headv);
        }
        wg_ids[wgi * 2] = hid;
        wg_ids[wgi * 2 + 1] = 0;
    }
    __syncthreads();
    if (tail_id && !tail_through) {  // this thread holds the start of a cut bucket: finish it
        if (tid + 1 < T && idB[tid + 1] == tail_id) {
            const Xyzz<P> o = planes_get<P>(cur, (size_t)T, (size_t)(tid + 1));
            xyzz_add(acc, o);
        }
        if (cnt[tail_id] > W1) {  // ... unless it continues in the next region
            planes_put(wg_tail, nwg_total, wgi, acc);
            wg_ids[wgi * 2 + 1] = tail_id;
        } else {
            planes_put(seg, (size_t)M, (size_t)(tail_id - 1), acc);
        }
    }
}

// per vector: a bucket that starts in region w (its tail partial) and runs on through later regions (their head partials)
template <class C>
__global__ void __launch_bounds__(64) k_gs_stitch(const uint4* __restrict__ wg_head, const uint4* __restrict__ wg_tail,
                                                   const uint32_t* __restrict__ wg_ids, size_t nreg, size_t nwg_total, int M,
                                                   uint4* __restrict__ buckets) {
    using P = typename C::Base;
    const size_t b = blockIdx.x;
    uint4* seg = buckets + b * (size_t)M * 8;
    for (size_t w = threadIdx.x; w < nreg; w += blockDim.x) {
        const uint32_t m = wg_ids[(b * nreg + w) * 2 + 1];
        if (!m) continue;
        Xyzz<P> acc = planes_get<P>(wg_tail, nwg_total, b * nreg + w);
        for (size_t w2 = w + 1; w2 < nreg && wg_ids[(b * nreg + w2) * 2] == m; w2++) {
            const Xyzz<P> o = planes_get<P>(wg_head, nwg_total, b * nreg + w2);
            xyzz_add(acc, o);
        }
        planes_put(seg, (size_t)M, (size_t)(m - 1), acc);
    }
}

// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------
template <class C, class SF>
static int msm_run_t(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_scalars, size_t n, size_t batch, int form,
                     uint32_t* d_out, const MsmPair* pair_in) {
    // paired vectors: `n` counts the dense entries (n_pair + 4, the two tail columns of the table addressed twice);
    // every input vector yields two results (class 0, class 1), d_out holds 2 * batch points
    const int nclass = pair_in ? 2 : 1;
    if (pair_in && (!bases->pre_c || n != pair_in->n_pair + 4 || n - 2 > bases->n)) return BZH_E_ARG;
    if (n == 0) {
        BZH_HIP_TRY(ctx, hipMemsetAsync(d_out, 0, batch * 96, ctx->stream));
        return BZH_OK;
    }
    MsmPlan p = plan_msm(n);
    const bool pre = bases->pre_c != 0;
    size_t n_eff = n, row_len = 0, row_stride = 0;
    int acc_nwin = 0;
    if (pre) {  // one window over the flattened [window][point] item space
        p.c = bases->pre_c;
        p.nwin = bases->pre_nwin;
        p.M = 1 << (p.c - 1);
        n_eff = (size_t)p.nwin * n;
        // Chunk count: at least ceil(n_eff / 32768) (u16 local index), more when that fills the chip's
        // CUs more evenly -- the launch's makespan is ceil(workgroups / CUs) rounds of `chunk` additions
        // plus, per extra chunk, one more bucket set to reduce (2 full additions per bucket).
        static const size_t max_chunk = [] {
            const char* e = getenv("BZH_ACC_CHUNK");  // tuning knob: items per accumulate workgroup (<= 32768)
            const size_t v = e ? (size_t)atol(e) : kMaxChunk;
            return v >= 1024 && v <= kMaxChunk ? v : kMaxChunk;
        }();
        const size_t cmin = (n_eff + max_chunk - 1) / max_chunk;
        const double cus = (double)(ctx->num_cu > 0 ? ctx->num_cu : 256);
        double best = 1e300;
        size_t best_nc = cmin;
        static const size_t split_max = [] {
            const char* e = getenv("BZH_ACC_SPLIT");   // tuning knob: at most this many times the minimal chunk count
            const long v = e ? atol(e) : 16;
            return (size_t)(v >= 1 && v <= 64 ? v : 16);
        }();
        for (size_t nc = cmin; nc <= cmin * split_max && nc <= n_eff; nc++) {
            const double chunk = ceil((double)n_eff / (double)nc);
            const double rounds = ceil((double)(nc * batch) / cus);
            const double t = rounds * (chunk * 10.0 + (double)p.M * 38.0 + 6000.0);
            if (t < best * 0.999) {
                best = t;
                best_nc = nc;
            }
        }
        p.nchunks = best_nc;
        p.chunk = (n_eff + p.nchunks - 1) / p.nchunks;
        p.chunk = (p.chunk + 63) & ~(size_t)63;
        p.nchunks = (n_eff + p.chunk - 1) / p.chunk;
        row_len = n;
        row_stride = bases->row_stride ? bases->row_stride : bases->n;
    }
    acc_nwin = pre ? 1 : p.nwin;
    MsmPair pair{0, 0, 0};
    if (pair_in) {
        if (2 * (size_t)p.M > 0x7fffu) return BZH_E_ARG;
        pair = *pair_in;
        pair.M = (uint32_t)p.M;
    }
    const int M_acc = p.M * nclass;  // buckets per accumulate segment
    // global-sort path (one bucket set per vector), opt-in with BZH_MSM_GS=1.  Measured on the k = 14 proof batches
    // (16 proofs, window bits in brackets): accumulate 22.1 ms [13] against 24.5 ms [11] for the LDS-chunk kernel, sort
    // 3.9 ms against 0.4 ms, reduction 10.9 ms against 9.4 ms -> 405 against 397 proofs/s with four batches in flight:
    // the additions scale with the table rows either way and big buckets (c <= 12) cost more stitching here, so the
    // default stays on the chunk kernel.
    static const int gs_env = [] {
        const char* e = getenv("BZH_MSM_GS");
        return e ? atoi(e) : 0;
    }();
    const bool use_gs = pre && gs_env != 0 && n_eff < ((size_t)1 << 31) && n_eff >= 4096 && !bases->vec_col_stride;
    constexpr int GT = 512;  // threads per region workgroup
    size_t gs_nreg = 1;
    if (use_gs) {
        p.chunk = kMaxChunk;
        p.nchunks = (n_eff + p.chunk - 1) / p.chunk;
        static const int gs_items = [] {
            const char* e = getenv("BZH_GS_ITEMS");
            return e ? atoi(e) : 32;
        }();  // sorted items per thread and region (region slice staged in LDS: 4 B per item)
        gs_nreg = (n_eff + (size_t)GT * gs_items - 1) / ((size_t)GT * gs_items);
        const size_t want = (256 + batch - 1) / batch;  // fill the CUs when there are few vectors
        if (gs_nreg < want) gs_nreg = want;
        const size_t cap = n_eff / ((size_t)GT * 8) ? n_eff / ((size_t)GT * 8) : 1;
        if (gs_nreg > cap) gs_nreg = cap;
        const size_t floor_reg = (n_eff + 28671) / 28672;  // region slice in LDS: at most 112 KB
        if (gs_nreg < floor_reg) gs_nreg = floor_reg;
    }
    // slice the batch so the bucket workspace stays bounded
    const size_t seg_bytes = (size_t)M_acc * 128;
    const size_t segs_per_vec = (size_t)acc_nwin * p.nchunks;
    const size_t budget = (size_t)2 << 30;
    const size_t gs_vec_bytes = n_eff * 4 + p.nchunks * (size_t)(M_acc + 1) * 4 + (size_t)(M_acc + 1) * 4 + 64 + seg_bytes +
                                gs_nreg * ((size_t)2 * GT * 128 + 2 * 128 + 8) + 1024;
    size_t slice = use_gs ? budget / gs_vec_bytes : budget / (segs_per_vec * (seg_bytes + (size_t)2 * 1024 * 128));
    if (slice < 1) slice = 1;
    if (slice > batch) slice = batch;
    if (slice > 65535) slice = 65535;  // gridDim.z

    DigitOffset off;
    for (int k = 0; k < 8; k++) off.l[k] = 0;
    for (int w = 0; w < p.nwin - 1; w++) {
        int pos = p.c * w + (p.c - 1);
        off.l[pos >> 5] |= 1u << (pos & 31);
    }

    void *d_digits = nullptr, *d_buckets = nullptr, *d_winsums = nullptr;
    // more threads per segment when there are few segments (occupancy), fewer when there are many
    const size_t est_wgs = segs_per_vec * slice;
    (void)est_wgs;
    // ~200 VGPRs per lane: 2 waves/SIMD, so 512 threads = one workgroup per CU; 1024 would spill
    static const int acc_threads_env = [] {
        const char* e = getenv("BZH_ACC_THREADS");
        return e ? atoi(e) : 0;
    }();
    // 256 threads per chunk (128-item slices at the full chunk size) measured better than 512 wherever several launches share the
    // GPU and also with one batch in flight (k = 14 default 453 vs 440 proofs/s, 366 vs 356 single stream; k = 11 3 683 vs 3 555):
    // fewer, longer slices mean fewer buckets cut by slice boundaries to stitch, and the LDS-bound two workgroups per CU leave
    // more room for the other streams' kernels
    const int acc_threads = acc_threads_env == 128 || acc_threads_env == 256 || acc_threads_env == 512 ? acc_threads_env : 256;
    // per segment: two stitch buffers (+ the ninth limbs of raw buckets and raw head partials when the accumulator is unsaturated:
    // msm_accumulate_body computes the same stride from its template arguments)
    static const bool acc_sat_env = getenv("BZH_ACC_SATURATED") != nullptr;
    const bool acc_u29 = fe29_supported<typename C::Base>() && !acc_sat_env && !use_gs && bases->d_xy29 != nullptr;
    const size_t part_bytes = (size_t)2 * acc_threads * 128 + (acc_u29 ? ((size_t)M_acc + acc_threads) * 16 : 0);
    int rc;
    if ((rc = ws_ensure(ctx, 0, slice * (size_t)p.nwin * n * sizeof(uint16_t), &d_digits))) return rc;
    if ((rc = ws_ensure(ctx, 1, use_gs ? slice * gs_vec_bytes + 4096 : slice * segs_per_vec * (seg_bytes + part_bytes), &d_buckets)))
        return rc;
    uint4* d_partials = (uint4*)((char*)d_buckets + slice * segs_per_vec * seg_bytes);
    if ((rc = ws_ensure(ctx, 2, slice * segs_per_vec * nclass * 128, &d_winsums))) return rc;
    // global-sort workspace, carved from slot 1: buckets | region partials | region heads | region tails | sorted | hist | ends | ids
    uint4 *gs_partials = nullptr, *gs_head = nullptr, *gs_tail = nullptr;
    uint32_t *gs_sorted = nullptr, *gs_hist = nullptr, *gs_ends = nullptr, *gs_maxpop = nullptr, *gs_ids = nullptr;
    if (use_gs) {
        char* cur = (char*)d_buckets + slice * seg_bytes;
        auto carve = [&](size_t bytes) {
            char* r = cur;
            cur += (bytes + 255) & ~(size_t)255;
            return r;
        };
        gs_partials = (uint4*)carve(slice * gs_nreg * 2 * GT * 128);
        gs_head = (uint4*)carve(slice * gs_nreg * 128);
        gs_tail = (uint4*)carve(slice * gs_nreg * 128);
        gs_sorted = (uint32_t*)carve(slice * n_eff * 4);
        gs_hist = (uint32_t*)carve(slice * p.nchunks * (size_t)(M_acc + 1) * 4);
        gs_ends = (uint32_t*)carve(slice * (size_t)(M_acc + 1) * 4);
        gs_maxpop = (uint32_t*)carve(slice * 4);
        gs_ids = (uint32_t*)carve(slice * gs_nreg * 8);
    }

    const size_t acc_lds = ((size_t)M_acc + 2 + 32 + acc_threads) * 4 + p.chunk * 2 + 16;
    int red_threads = p.M < 256 ? p.M : 256;
    const size_t red_lds = (size_t)red_threads * 128 * 2;
    // per context (= per device; the ctx mutex is held): the dynamic-LDS limit is a per-device function attribute, so a
    // process that opens contexts on several GPUs has to raise it on each of them
    if (!ctx->msm_attr_set[C::id]) {
        BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_accumulate<C, 256, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_accumulate<C, 512, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_accumulate<C, 128, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if constexpr (fe29_supported<typename C::Base>()) {
            BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_accumulate<C, 256, true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_accumulate<C, 512, true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_accumulate<C, 128, true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_reduce<C>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gs_hist<512>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gs_scatter<512>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gs_scan<C>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        BZH_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gs_accumulate<C, 512>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ctx->msm_attr_set[C::id] = true;
    }

    for (size_t b0 = 0; b0 < batch; b0 += slice) {
        const size_t nb = (batch - b0 < slice) ? batch - b0 : slice;
        const size_t total = nb * n;
        {
            ScopedTimer t(ctx, BZH_T_MSM_DIGITS);
            hipLaunchKernelGGL((k_msm_digits<SF>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                               d_scalars + b0 * n * 8, n, total, form, p.c, p.nwin, off, (uint16_t*)d_digits, pair);
        }
        if (use_gs) {
            const size_t hist_lds = ((size_t)M_acc + 1) * 4;
            {
                ScopedTimer t(ctx, BZH_T_MSM_DIGITS);  // the sort belongs with the digit stage
                hipLaunchKernelGGL((k_gs_hist<512>), dim3((unsigned)p.nchunks, (unsigned)nb), dim3(512), hist_lds, ctx->stream,
                                   (const uint16_t*)d_digits, n_eff, p.chunk, M_acc, gs_hist);
                hipLaunchKernelGGL((k_gs_scan<C>), dim3((unsigned)nb), dim3(1024), ((size_t)M_acc + 2 + 32) * 4, ctx->stream, gs_hist,
                                   p.nchunks, M_acc, gs_ends, gs_maxpop, (uint4*)d_buckets);
                hipLaunchKernelGGL((k_gs_scatter<512>), dim3((unsigned)p.nchunks, (unsigned)nb), dim3(512), hist_lds, ctx->stream,
                                   (const uint16_t*)d_digits, n_eff, p.chunk, M_acc, (const uint32_t*)gs_hist, gs_sorted);
            }
            {
                ScopedTimer t(ctx, BZH_T_MSM_ACCUMULATE);
                if (ctx->profiling) ctx->alg_bytes[BZH_T_MSM_ACCUMULATE] += (double)nb * (double)n * 32.0 + (double)n * 64.0;
                const size_t region_lds = ((n_eff + gs_nreg - 1) / gs_nreg + 8) * 4;
                hipLaunchKernelGGL((k_gs_accumulate<C, GT>), dim3((unsigned)gs_nreg, (unsigned)nb), dim3(GT), region_lds, ctx->stream, bases->d_xy,
                                   (const uint32_t*)gs_sorted, (const uint32_t*)gs_ends, (const uint32_t*)gs_maxpop, n_eff, M_acc,
                                   (uint4*)d_buckets, gs_partials, gs_head, gs_tail, gs_ids, row_len, row_stride,
                                   pair_in ? n - 2 : (size_t)0);
                hipLaunchKernelGGL((k_gs_stitch<C>), dim3((unsigned)nb), dim3(64), 0, ctx->stream, (const uint4*)gs_head,
                                   (const uint4*)gs_tail, (const uint32_t*)gs_ids, gs_nreg, gs_nreg * nb, M_acc, (uint4*)d_buckets);
            }
        } else
        {
            ScopedTimer t(ctx, BZH_T_MSM_ACCUMULATE);
            if (ctx->profiling) ctx->alg_bytes[BZH_T_MSM_ACCUMULATE] += (double)nb * (double)n * 32.0 + (double)n * 64.0;
            static const bool xcd_map = getenv("BZH_ACC_NO_XCD_MAP") == nullptr;
            const bool xmap = xcd_map && acc_nwin == 1 && nb >= 8 && !bases->vec_col_stride;
            const dim3 grid = xmap ? dim3((unsigned)(((nb * p.nchunks + 7) / 8) * 8)) : dim3((unsigned)nb, (unsigned)acc_nwin, (unsigned)p.nchunks);
            const uint32_t xv = xmap ? (uint32_t)nb : 0u, xc = xmap ? (uint32_t)p.nchunks : 0u;
            // unsaturated-limb accumulation on the Pasta curves (BZH_ACC_SATURATED=1: the 8 x 32 variant, parity-tested against it)
            const bool acc_sat = !acc_u29;
            constexpr bool can29 = fe29_supported<typename C::Base>();
#define BZH_LAUNCH_ACC_U(TT, UU)                                                                                         \
    hipLaunchKernelGGL((k_msm_accumulate<C, TT, UU>), grid, dim3(TT), acc_lds, ctx->stream, bases->d_xy, bases->d_xy29,   \
                       (const uint16_t*)d_digits, n_eff, acc_nwin, M_acc, p.chunk, (uint4*)d_buckets, d_partials, row_len, \
                       row_stride, pair_in ? n - 2 : (size_t)0, ctx->profiling ? ctx->d_add_counter : nullptr,            \
                       bases->vec_col_stride, b0, xv, xc)
#define BZH_LAUNCH_ACC(TT)                                  \
    do {                                                    \
        if constexpr (can29) {                              \
            if (!acc_sat) BZH_LAUNCH_ACC_U(TT, true);       \
            else BZH_LAUNCH_ACC_U(TT, false);               \
        } else {                                            \
            BZH_LAUNCH_ACC_U(TT, false);                    \
        }                                                   \
    } while (0)
            if (acc_threads == 128) BZH_LAUNCH_ACC(128);
            else if (acc_threads == 512) BZH_LAUNCH_ACC(512);
            else BZH_LAUNCH_ACC(256);
#undef BZH_LAUNCH_ACC_U
#undef BZH_LAUNCH_ACC
        }
        const size_t nseg = nb * segs_per_vec;
        bool presum = false, fused_out = false, latency = false;
        {
            ScopedTimer t(ctx, BZH_T_MSM_REDUCE);
            // many segments (throughput regime): sum the chunks of every vector bucket-wise first, then run the
            // running-sum reduction once per vector instead of once per chunk
            // (long chunk lists only when there are enough vectors to hide the serial sum over the chunks)
            presum = !use_gs && acc_nwin == 1 && p.nchunks >= 2 && (p.nchunks <= 32 || nb * nclass >= 8) && nseg * nclass >= 256;
            size_t rseg = nseg, rspv = segs_per_vec, mult = 1;
            if (use_gs) {
                rseg = nb;
                rspv = 1;
            }
            if (presum) {
                if (fe29_supported<typename C::Base>() && !acc_sat_env)
                    hipLaunchKernelGGL((k_msm_chunksum<C, fe29_supported<typename C::Base>()>), dim3((unsigned)((M_acc + 63) / 64), (unsigned)nb), dim3(256), 0,
                                       ctx->stream, (uint4*)d_buckets, M_acc, p.nchunks);
                else
                    hipLaunchKernelGGL((k_msm_chunksum<C, false>), dim3((unsigned)((M_acc + 63) / 64), (unsigned)nb), dim3(256), 0, ctx->stream,
                                       (uint4*)d_buckets, M_acc, p.nchunks);
                rseg = nb;
                rspv = 1;
                mult = p.nchunks;
            }
            // one segment per result (window-table MSM after the chunk pre-sum): the reduction writes the Jacobian result
            // itself, no k_msm_finalize launch
            fused_out = acc_nwin == 1 && rspv == 1;
            uint32_t* fout = fused_out ? d_out + b0 * nclass * 24 : (uint32_t*)nullptr;
            // latency mode: fewer reduction waves than SIMDs -- the dependent chain, not the work, is the cost: four lanes per addition
            static const bool no_quad = getenv("BZH_MSM_NO_QUAD") != nullptr;
            latency = !no_quad && p.M >= 16 && rseg * nclass <= 1024;
            // (the quad reductions in unsaturated limbs on the Pasta curves; BZH_ACC_SATURATED=1: saturated like everything else)
            constexpr bool red29 = fe29_supported<typename C::Base>();
            const bool use_red29 = red29 && !acc_sat_env;
            // the workgroup flavour while its four-wave workgroups fit one or two per CU (a second wave per SIMD interleaves with the
            // first one's dependent chain almost for free); beyond that one wave per segment
            static const size_t wg_max = [] {
                const char* e = getenv("BZH_RED_WG_MAX");   // tuning knob
                const long v = e ? atol(e) : 512;
                return (size_t)(v >= 0 ? v : 512);
            }();
            if (latency && p.M >= 64 && rseg * nclass <= wg_max) {
                if constexpr (red29) {
                    if (use_red29)
                        hipLaunchKernelGGL((k_msm_reduce_quad_wg<C, true>), dim3((unsigned)(rseg * nclass)), dim3(256), (size_t)2 * 64 * 144, ctx->stream,
                                           (const uint4*)d_buckets, p.M, nclass, rspv, mult, (uint4*)d_winsums, form, fout);
                }
                if (!use_red29)
                    hipLaunchKernelGGL((k_msm_reduce_quad_wg<C, false>), dim3((unsigned)(rseg * nclass)), dim3(256), (size_t)2 * 64 * 128, ctx->stream,
                                       (const uint4*)d_buckets, p.M, nclass, rspv, mult, (uint4*)d_winsums, form, fout);
            } else if (latency) {
                if constexpr (red29) {
                    if (use_red29)
                        hipLaunchKernelGGL((k_msm_reduce_quad<C, true>), dim3((unsigned)(rseg * nclass)), dim3(64), 0, ctx->stream,
                                           (const uint4*)d_buckets, p.M, nclass, rspv, mult, (uint4*)d_winsums, form, fout);
                }
                if (!use_red29)
                    hipLaunchKernelGGL((k_msm_reduce_quad<C, false>), dim3((unsigned)(rseg * nclass)), dim3(64), 0, ctx->stream,
                                       (const uint4*)d_buckets, p.M, nclass, rspv, mult, (uint4*)d_winsums, form, fout);
            } else if (p.M >= 64 && rseg * nclass >= 256) {
                hipLaunchKernelGGL((k_msm_reduce_wave<C>), dim3((unsigned)(rseg * nclass)), dim3(64), 0, ctx->stream,
                                   (const uint4*)d_buckets, p.M, nclass, rspv, mult, (uint4*)d_winsums, form, fout);
            } else {
                hipLaunchKernelGGL((k_msm_reduce<C>), dim3((unsigned)(rseg * nclass)), dim3(red_threads), red_lds, ctx->stream,
                                   (const uint4*)d_buckets, p.M, nclass, rspv, mult, (uint4*)d_winsums, form, fout);
            }
        }
        if (!fused_out) {
            ScopedTimer t(ctx, BZH_T_MSM_FINALIZE);
            const size_t fchunks = (presum || use_gs) ? 1 : p.nchunks, fseg = (presum || use_gs) ? nb : nseg;
            if (latency && acc_nwin <= 64)
                hipLaunchKernelGGL((k_msm_finalize_quad<C>), dim3((unsigned)(nb * nclass)), dim3(256), 0, ctx->stream,
                                   (const uint4*)d_winsums, fseg * nclass, acc_nwin, fchunks, pre ? 0 : p.c, form,
                                   d_out + b0 * nclass * 24);
            else
                hipLaunchKernelGGL((k_msm_finalize<C>), dim3((unsigned)(nb * nclass)), dim3(64), 0, ctx->stream,
                                   (const uint4*)d_winsums, fseg * nclass, acc_nwin, fchunks, pre ? 0 : p.c, form,
                                   d_out + b0 * nclass * 24);
        }
        BZH_HIP_TRY(ctx, hipGetLastError());
    }
    return BZH_OK;
}

int msm_run(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_scalars, size_t n, size_t batch, int form,
            uint32_t* d_out_xyz) {
    switch (bases->curve) {
        case BZH_CURVE_VESTA:
            return msm_run_t<VestaCurve, FpParams>(ctx, bases, d_scalars, n, batch, form, d_out_xyz, nullptr);
        case BZH_CURVE_PALLAS:
            return msm_run_t<PallasCurve, FqParams>(ctx, bases, d_scalars, n, batch, form, d_out_xyz, nullptr);
        case BZH_CURVE_BN254:
            return msm_run_t<Bn254Curve, BnFrParams>(ctx, bases, d_scalars, n, batch, form, d_out_xyz, nullptr);
    }
    return BZH_E_ARG;
}

// Paired MSM against a window table of n_pair + 2 points: each of the `batch` dense vectors of n_pair + 4 scalars
//   [ v_0 .. v_(n_pair-1) | a_0 a_1 | b_0 b_1 ]
// yields TWO sums: class 0 takes v_i with bit (log_m - 1) of i clear plus a_0, a_1 on the last two table points,
// class 1 the v_i with that bit set plus b_0, b_1 on the same two points (the L_j / R_j of one IPA round, whose
// scalar vectors have disjoint supports).  d_out_xyz: 2 * batch points, [class 0, class 1] per vector.
int msm_run_paired(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_scalars, size_t n_pair, unsigned log_m, size_t batch,
                   int form, uint32_t* d_out_xyz) {
    if (!bases->pre_c || bases->n != n_pair + 2 || log_m < 1) return BZH_E_ARG;
    const MsmPair pair{log_m, n_pair, 0};
    switch (bases->curve) {
        case BZH_CURVE_VESTA:
            return msm_run_t<VestaCurve, FpParams>(ctx, bases, d_scalars, n_pair + 4, batch, form, d_out_xyz, &pair);
        case BZH_CURVE_PALLAS:
            return msm_run_t<PallasCurve, FqParams>(ctx, bases, d_scalars, n_pair + 4, batch, form, d_out_xyz, &pair);
        case BZH_CURVE_BN254:
            return msm_run_t<Bn254Curve, BnFrParams>(ctx, bases, d_scalars, n_pair + 4, batch, form, d_out_xyz, &pair);
    }
    return BZH_E_ARG;
}

// the fe29 copy of a table (bzh_bases::d_xy29): 20 words per point
template <class P>
__global__ void __launch_bounds__(256) k_points_to_fe29(const uint32_t* __restrict__ xy, uint32_t* __restrict__ xy29, size_t count) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= count) return;
    const Affine<P> q = affine_load<P>(xy + i * 16);
    Fe29<P> x = fe29_zero<P>(), y = x;
    if (!aff_is_id(q)) {
        x = fe29_from_sat_reduced(q.x);
        y = fe29_from_sat_reduced(q.y);
    }
    uint4* o = reinterpret_cast<uint4*>(xy29 + i * 20);
    o[0] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    o[1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    o[2] = make_uint4(x.l[8], y.l[0], y.l[1], y.l[2]);
    o[3] = make_uint4(y.l[3], y.l[4], y.l[5], y.l[6]);
    o[4] = make_uint4(y.l[7], y.l[8], 0u, 0u);
}
template <class C>
static int table_to_fe29(bzh_ctx* ctx, const uint32_t* d_xy, uint32_t* d_xy29, size_t count) {
    if constexpr (fe29_supported<typename C::Base>()) {
        hipLaunchKernelGGL((k_points_to_fe29<typename C::Base>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, d_xy, d_xy29, count);
        BZH_HIP_TRY(ctx, hipGetLastError());
    }
    return BZH_OK;
}

template <class C>
static int bases_precompute_t(bzh_ctx* ctx, bzh_bases* b, int c) {
    const int nwin = (256 + c - 1) / c;
    uint32_t* table = nullptr;
    BZH_HIP_TRY(ctx, hipMalloc((void**)&table, (size_t)nwin * b->n * 64));
    BZH_HIP_TRY(ctx, hipMemcpyAsync(table, b->d_xy, b->n * 64, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL((k_expand_bases<C>), dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, ctx->stream, table, b->n, c,
                       nwin);
    BZH_HIP_TRY(ctx, hipGetLastError());
    BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    BZH_HIP_TRY(ctx, hipFree(b->d_xy));
    b->d_xy = table;
    b->pre_c = c;
    b->pre_nwin = nwin;
    if constexpr (fe29_supported<typename C::Base>()) {
        static const bool no29 = getenv("BZH_ACC_SATURATED") != nullptr;
        if (!no29) {   // the unsaturated-limb copy for the accumulation loop (+25 % of the table's HBM)
            BZH_HIP_TRY(ctx, hipMalloc((void**)&b->d_xy29, (size_t)nwin * b->n * 80));
            const int rc = table_to_fe29<C>(ctx, b->d_xy, b->d_xy29, (size_t)nwin * b->n);
            if (rc) return rc;
            BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return BZH_OK;
}

int bases_precompute(bzh_ctx* ctx, bzh_bases* b, int window_bits) {
    if (b->pre_c != 0 || b->n == 0) return BZH_OK;
    int c = window_bits ? window_bits : plan_precompute_c(b->n);
    if (c < 4 || c > 15) return BZH_E_RANGE;
    switch (b->curve) {
        case BZH_CURVE_VESTA: return bases_precompute_t<VestaCurve>(ctx, b, c);
        case BZH_CURVE_PALLAS: return bases_precompute_t<PallasCurve>(ctx, b, c);
        case BZH_CURVE_BN254: return bases_precompute_t<Bn254Curve>(ctx, b, c);
    }
    return BZH_E_ARG;
}

int bases_to_montgomery(bzh_ctx* ctx, int curve, uint32_t* d_xy, size_t n) {
    const size_t count = n * 2;
    if (count == 0) return BZH_OK;
    dim3 grid((unsigned)((count + 255) / 256)), block(256);
    switch (curve) {
        case BZH_CURVE_VESTA:
            hipLaunchKernelGGL((k_to_montgomery<FqParams>), grid, block, 0, ctx->stream, d_xy, count);
            break;
        case BZH_CURVE_PALLAS:
            hipLaunchKernelGGL((k_to_montgomery<FpParams>), grid, block, 0, ctx->stream, d_xy, count);
            break;
        case BZH_CURVE_BN254:
            hipLaunchKernelGGL((k_to_montgomery<BnFqParams>), grid, block, 0, ctx->stream, d_xy, count);
            break;
        default:
            return BZH_E_ARG;
    }
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}


// ---------------------------------------------------------------------------
// The IPA's generator collapse (ipa.hip): after j rounds the folded generators are
//     G'[i] = sum_{t < cnt} s[t] * G[i + t*m],   cnt = 2^j, m = n / cnt,
// the SAME cnt scalars for every i.  With the SRS window table (row w = 2^(c w) G) and s[t] = sum_w D[t][w] 2^(c w),
//     G'[i] = sum_{(t, w)} D[t][w] * Row_w[i + t*m]:  cnt * nwin items per output, and the item list (t, w, D) is shared by all i.
// |D| = a * 2^h + r with h = ceil((c - 1) / 2): G' = 2^h * sum sgn a Row + sum sgn r Row, each sum by the bucket method with the
// bucket LEVEL as the outer loop: a lane owns one output i, walks the levels from the top and keeps
//     acc (the bucket of this level), running (the sum of the buckets above and at it), total (the sum of the running sums),
// i.e. (items + 2 * levels) additions per output instead of a 255-bit double-and-add per term.  The loop bounds depend on s
// only, so they are wave-uniform, and a wave's table reads are 64 consecutive points.
// ---------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(256) k_collapse_generators(const uint32_t* __restrict__ table, const uint32_t* __restrict__ table29,
                                                               size_t row_stride, int c, int nwin,
                                                               const uint16_t* __restrict__ digits, size_t cnt, size_t m, size_t n_srs,
                                                               uint32_t* __restrict__ out, size_t out_cols,
                                                               unsigned long long* __restrict__ add_counter) {
    using P = typename C::Base;
    extern __shared__ __align__(16) uint32_t lds[];
    const int h = c / 2;                         // |D| <= 2^(c-1) = a * 2^h + r, a <= 2^(c-1-h), r < 2^h
    const int levels_a = 1 << (c - 1 - h), levels_r = (1 << h) - 1, LV = (levels_a > levels_r ? levels_a : levels_r) + 2;
    const int nitems = (int)cnt * nwin;
    uint32_t* startA = lds;                      // [LV]: first list position of level v (exclusive scan), pass A
    uint32_t* startR = lds + LV;
    uint32_t* fillA = lds + 2 * LV;              // running fill cursors of the scatter
    uint32_t* fillR = lds + 3 * LV;
    uint32_t* listA = lds + 4 * LV;              // [nitems]: (t * nwin + w) | sign << 31, sorted by level
    uint32_t* listR = listA + nitems;
    const int tid = threadIdx.x;
    const size_t b = blockIdx.y;
    const uint16_t* dg = digits + b * (size_t)nwin * cnt;   // [w][t]
    for (int i = tid; i < 4 * LV; i += 256) lds[i] = 0;
    __syncthreads();
    for (int e = tid; e < nitems; e += 256) {
        const uint32_t mag = dg[e] & 0x7fffu;
        if (mag >> h) atomicAdd(&startA[mag >> h], 1u);
        if (mag & ((1u << h) - 1u)) atomicAdd(&startR[mag & ((1u << h) - 1u)], 1u);
    }
    __syncthreads();
    if (tid < 2) {   // exclusive scans (a few dozen levels)
        uint32_t* st = tid ? startR : startA;
        uint32_t* fl = tid ? fillR : fillA;
        uint32_t run = 0;
        for (int v = 0; v < LV; v++) {
            const uint32_t k = st[v];
            st[v] = run;
            fl[v] = run;
            run += k;
        }
    }
    __syncthreads();
    for (int e = tid; e < nitems; e += 256) {
        const uint32_t d = dg[e], mag = d & 0x7fffu;
        const uint32_t w = (uint32_t)e / (uint32_t)cnt, t = (uint32_t)e - w * (uint32_t)cnt;
        const uint32_t ent = (t * (uint32_t)nwin + w) | ((d & 0x8000u) << 16);
        if (mag >> h) listA[atomicAdd(&fillA[mag >> h], 1u)] = ent;
        if (mag & ((1u << h) - 1u)) listR[atomicAdd(&fillR[mag & ((1u << h) - 1u)], 1u)] = ent;
    }
    __syncthreads();
    // the scatter order inside a level depends on the atomics: additions commute, the SUM does not depend on it
    const size_t i = blockIdx.x * (size_t)256 + tid;
    if (add_counter && tid == 0) {   // profiling: mixed additions of this workgroup's outputs (both passes)
        const size_t live = min((size_t)256, m - min(m, blockIdx.x * (size_t)256));
        atomicAdd(add_counter, (unsigned long long)(fillA[LV - 1] + fillR[LV - 1]) * live);
    }
    if (i < m) {
        Xyzz<P> sums[2];
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            const uint32_t* st = pass ? startR : startA;
            const uint32_t* ls = pass ? listR : listA;
            const int top = pass ? levels_r : levels_a;
            Xyzz<P> running = xyzz_identity<P>(), total = xyzz_identity<P>();
            bool any = false;
#pragma unroll 1
            for (int v = top; v >= 1; v--) {
                const uint32_t lo = st[v], hi = st[v + 1];
                if (lo < hi) {
                    Xyzz<P> acc;
                    // the level's bucket in unsaturated limbs on the Pasta curves (csrc/curve29.cuh: the mixed addition at 1.28 x),
                    // handed to the saturated running sums through the product-free conversion -- ~12 points per level
                    if constexpr (fe29_supported<P>()) {
                        const Fe29Consts<P> k29 = fe29_consts<P>();
                        Xyzz29<P> acc29 = xyzz29_identity<P>();
                        if (table29) {
                            // operands from the SRS table's fe29 copy (20 words per point, as k_msm_accumulate reads them), the
                            // next item's gather in flight while this one is added
                            auto load29 = [&](uint32_t ent, Fe29<P>& x, Fe29<P>& y) {
                                const uint32_t tw = ent & 0x7fffffffu, t = tw / (uint32_t)nwin, w = tw - t * (uint32_t)nwin;
                                const uint4* q = reinterpret_cast<const uint4*>(table29 + ((size_t)w * row_stride + (size_t)t * m + i) * 20);
                                const uint4 a = q[0], bb = q[1], cc = q[2], d = q[3], f = q[4];
                                x.l[0] = a.x, x.l[1] = a.y, x.l[2] = a.z, x.l[3] = a.w, x.l[4] = bb.x, x.l[5] = bb.y, x.l[6] = bb.z, x.l[7] = bb.w, x.l[8] = cc.x;
                                y.l[0] = cc.y, y.l[1] = cc.z, y.l[2] = cc.w, y.l[3] = d.x, y.l[4] = d.y, y.l[5] = d.z, y.l[6] = d.w, y.l[7] = f.x, y.l[8] = f.y;
                            };
                            uint32_t ent_next = ls[lo];
                            Fe29<P> x_next, y_next;
                            load29(ent_next, x_next, y_next);
                            for (uint32_t e = lo; e < hi; e++) {
                                const uint32_t ent = ent_next;
                                const Fe29<P> qx = x_next;
                                Fe29<P> qy = y_next;
                                if (e + 1 < hi) {
                                    ent_next = ls[e + 1];
                                    load29(ent_next, x_next, y_next);
                                }
                                uint32_t any = 0;   // the identity is stored as zeros
#pragma unroll
                                for (int j = 0; j < 9; j++) any |= qx.l[j] | qy.l[j];
                                if (any) {
                                    if (ent >> 31) qy = fe29_sub<P, 4>(fe29_zero<P>(), qy);
                                    xyzz29_madd_q29(acc29, qx, qy, k29);
                                }
                            }
                        } else {
                            for (uint32_t e = lo; e < hi; e++) {
                                const uint32_t ent = ls[e], tw = ent & 0x7fffffffu;
                                const uint32_t t = tw / (uint32_t)nwin, w = tw - t * (uint32_t)nwin;
                                Affine<P> q = affine_load<P>(table + ((size_t)w * row_stride + (size_t)t * m + i) * 16);
                                if (!aff_is_id(q)) {
                                    if (ent >> 31) q.y = fe_neg(q.y);
                                    xyzz29_madd(acc29, q, k29);
                                }
                            }
                        }
                        acc = xyzz29_to_sat_fast(acc29);
                    } else {
                        acc = xyzz_identity<P>();
                        for (uint32_t e = lo; e < hi; e++) {
                            const uint32_t ent = ls[e], tw = ent & 0x7fffffffu;
                            const uint32_t t = tw / (uint32_t)nwin, w = tw - t * (uint32_t)nwin;
                            Affine<P> q = affine_load<P>(table + ((size_t)w * row_stride + (size_t)t * m + i) * 16);
                            if (!aff_is_id(q)) {
                                if (ent >> 31) q.y = fe_neg(q.y);
                                xyzz_madd(acc, q);
                            }
                        }
                    }
                    xyzz_add(running, acc);
                    any = true;
                }
                if (any) xyzz_add(total, running);
            }
            sums[pass] = total;
        }
        Xyzz<P> r = sums[0];
        for (int k = 0; k < h; k++) r = xyzz_dbl(r);
        xyzz_add(r, sums[1]);
        const Affine<P> a = xyzz_to_affine(r);
        uint32_t* o = out + (b * out_cols + i) * 16;
        fe_store(o, a.x);
        fe_store(o + 8, a.y);
    }
    if (blockIdx.x == 0 && tid < 32) {   // U, W ride along as the last two columns: 2 x 16 words
        const int pt = tid >> 4, wd = tid & 15;
        out[(b * out_cols + m + pt) * 16 + wd] = table[(n_srs + pt) * 16 + wd];
    }
}

// Window-table rows 1 .. nwin-1 for npts points whose row 0 is in place (row w = 2^(c w) * row 0, rows npts points apart):
// one thread per point walks the rows in XYZZ (c doublings each, parked in `scratch` with the running product of the zzz
// coordinates), inverts that product ONCE and walks back handing every row its own inverse (Montgomery's trick along the
// rows of a point: no cross-lane traffic), where 1 / zz = (1 / zzz)^2 * zz^2.
template <class C, bool U>
__global__ void __launch_bounds__(256) k_expand_rows_shared_inverse(uint32_t* __restrict__ table, size_t npts, int c, int nwin,
                                                                      uint4* __restrict__ scratch) {
    using P = typename C::Base;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const Affine<P> p0 = affine_load<P>(table + i * 16);
    if (aff_is_id(p0)) {
        for (int w = 1; w < nwin; w++) {
            uint32_t* o = table + ((size_t)w * npts + i) * 16;
            fe_store(o, p0.x);
            fe_store(o + 8, p0.y);
        }
        return;
    }
    // scratch planes: 10 x uint4 per (row, point): x, y, zz, zzz, prefix
    const size_t stride = npts * (size_t)(nwin - 1);
    auto put = [&](int plane, size_t slot, const Fe<P>& v) {
        scratch[(size_t)(2 * plane) * stride + slot] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
        scratch[(size_t)(2 * plane + 1) * stride + slot] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    };
    auto get = [&](int plane, size_t slot) {
        const uint4 a = scratch[(size_t)(2 * plane) * stride + slot], b = scratch[(size_t)(2 * plane + 1) * stride + slot];
        Fe<P> v;
        v.l[0] = a.x, v.l[1] = a.y, v.l[2] = a.z, v.l[3] = a.w, v.l[4] = b.x, v.l[5] = b.y, v.l[6] = b.z, v.l[7] = b.w;
        return v;
    };
    Fe<P> prefix = fe_one<P>();
    if constexpr (U) {
        // the chain of c (nwin - 1) doublings in unsaturated limbs (xyzz29_dbl: products of 188 instructions, Y3 one fused reduction);
        // a row's coordinates go back to the saturated form as they are parked (the normalisation below multiplies saturated)
        Xyzz29<P> q29;
        {
            const Xyzz<P> q1 = xyzz_dbl_affine(p0);
            q29.x = fe29_from_sat_reduced(q1.x), q29.y = fe29_from_sat_reduced(q1.y);
            q29.zz = fe29_from_sat_reduced(q1.zz), q29.zzz = fe29_from_sat_reduced(q1.zzz);
            q29.id = false;
        }
        for (int w = 1; w < nwin; w++) {
            if (w > 1) q29 = xyzz29_dbl(q29);
            for (int k = 1; k < c; k++) q29 = xyzz29_dbl(q29);
            const Xyzz<P> q = xyzz29_to_sat_fast(q29);
            const size_t slot = (size_t)(w - 1) * npts + i;
            put(4, slot, prefix);   // product of the zzz of the rows before this one
            prefix = fe_mul(prefix, q.zzz);
            put(0, slot, q.x);
            put(1, slot, q.y);
            put(2, slot, q.zz);
            put(3, slot, q.zzz);
        }
    } else {
        Xyzz<P> q = xyzz_dbl_affine(p0);
        for (int w = 1; w < nwin; w++) {
            if (w > 1) q = xyzz_dbl(q);
            for (int k = 1; k < c; k++) q = xyzz_dbl(q);
            const size_t slot = (size_t)(w - 1) * npts + i;
            put(4, slot, prefix);   // product of the zzz of the rows before this one
            prefix = fe_mul(prefix, q.zzz);
            put(0, slot, q.x);
            put(1, slot, q.y);
            put(2, slot, q.zz);
            put(3, slot, q.zzz);
        }
    }
    Fe<P> inv = fe_inv(prefix);
    for (int w = nwin - 1; w >= 1; w--) {
        const size_t slot = (size_t)(w - 1) * npts + i;
        const Fe<P> zzz = get(3, slot), zz = get(2, slot);
        const Fe<P> izzz = fe_mul(inv, get(4, slot));
        inv = fe_mul(inv, zzz);
        const Fe<P> t = fe_mul(izzz, zz);            // 1 / zz = (zz / zzz)^2
        const Fe<P> izz = fe_sqr(t);
        uint32_t* o = table + ((size_t)w * npts + i) * 16;
        fe_store(o, fe_mul(get(0, slot), izz));
        fe_store(o + 8, fe_mul(get(1, slot), izzz));
    }
}

size_t msm_collapse_scratch_bytes(const bzh_bases* srs, size_t cnt, size_t batch, int c_tail) {
    const size_t n = srs->n - 2, m = n / cnt, npts = batch * (m + 2);
    const int nwin_t = (256 + c_tail - 1) / c_tail;
    return npts * (size_t)(nwin_t - 1) * 160 + batch * cnt * (size_t)srs->pre_nwin * 2 + 4096;
}

template <class C, class SF>
static int msm_collapse_table_t(bzh_ctx* ctx, const bzh_bases* srs, const uint32_t* d_s, size_t cnt, size_t batch, int c_tail,
                                uint32_t* d_table29, uint32_t* d_table, void* d_scratch, bzh_bases* out) {
    const size_t n = srs->n - 2, m = n / cnt, cols = m + 2, npts = batch * cols;
    const int c = srs->pre_c, nwin = srs->pre_nwin, nwin_t = (256 + c_tail - 1) / c_tail;
    if (!c || cnt < 2 || m * cnt != n || m < 2 || c_tail < 4 || c_tail > 13 || batch > 65535) return BZH_E_ARG;
    uint16_t* d_digits = (uint16_t*)((char*)d_scratch + npts * (size_t)(nwin_t - 1) * 160);
    DigitOffset off;
    for (int k = 0; k < 8; k++) off.l[k] = 0;
    for (int w = 0; w < nwin - 1; w++) {
        const int pos = c * w + (c - 1);
        off.l[pos >> 5] |= 1u << (pos & 31);
    }
    const size_t total = batch * cnt;
    const MsmPair nopair{0, 0, 0};
    hipLaunchKernelGGL((k_msm_digits<SF>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_s, cnt, total,
                       BZH_FORM_MONTGOMERY, c, nwin, off, d_digits, nopair);
    {
        ScopedTimer t(ctx, BZH_T_MSM_ACCUMULATE);
        const int hh = c / 2, la = 1 << (c - 1 - hh), lr = (1 << hh) - 1, LV = (la > lr ? la : lr) + 2;
        const size_t lds = ((size_t)4 * LV + 2 * cnt * (size_t)nwin) * 4;
        if (lds > 60 * 1024) return BZH_E_RANGE;
        hipLaunchKernelGGL((k_collapse_generators<C>), dim3((unsigned)((m + 255) / 256), (unsigned)batch), dim3(256), lds, ctx->stream,
                           srs->d_xy, (!srs->row_stride || srs->row_stride == srs->n) ? srs->d_xy29 : (const uint32_t*)nullptr,   // (the copy has rows of n points)
                           srs->row_stride ? srs->row_stride : srs->n, c, nwin, (const uint16_t*)d_digits, cnt, m, n, d_table,
                           cols, ctx->profiling ? ctx->d_add_counter : nullptr);
        static const bool rows_sat = getenv("BZH_ACC_SATURATED") != nullptr;
        if (fe29_supported<typename C::Base>() && !rows_sat)
            hipLaunchKernelGGL((k_expand_rows_shared_inverse<C, fe29_supported<typename C::Base>()>), dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, ctx->stream, d_table, npts,
                           c_tail, nwin_t, (uint4*)d_scratch);
        else
            hipLaunchKernelGGL((k_expand_rows_shared_inverse<C, false>), dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, ctx->stream, d_table, npts,
                           c_tail, nwin_t, (uint4*)d_scratch);
    }
    BZH_HIP_TRY(ctx, hipGetLastError());
    *out = bzh_bases();
    out->curve = srs->curve;
    out->n = cols;
    out->d_xy = d_table;
    out->device = srs->device;
    out->pre_c = c_tail;
    out->pre_nwin = nwin_t;
    out->row_stride = npts;
    out->vec_col_stride = cols;
    if (d_table29 && fe29_supported<typename C::Base>()) {
        const int rc = table_to_fe29<C>(ctx, d_table, d_table29, npts * (size_t)nwin_t);
        if (rc) return rc;
        out->d_xy29 = d_table29;
    }
    return BZH_OK;
}

int msm_collapse_table(bzh_ctx* ctx, const bzh_bases* srs, const uint32_t* d_s, size_t cnt, size_t batch, int c_tail, uint32_t* d_table29,
                       uint32_t* d_table, void* d_scratch, bzh_bases* out) {
    switch (srs->curve) {
        case BZH_CURVE_VESTA: return msm_collapse_table_t<VestaCurve, FpParams>(ctx, srs, d_s, cnt, batch, c_tail, d_table29, d_table, d_scratch, out);
        case BZH_CURVE_PALLAS: return msm_collapse_table_t<PallasCurve, FqParams>(ctx, srs, d_s, cnt, batch, c_tail, d_table29, d_table, d_scratch, out);
        case BZH_CURVE_BN254: return msm_collapse_table_t<Bn254Curve, BnFrParams>(ctx, srs, d_s, cnt, batch, c_tail, nullptr, d_table, d_scratch, out);
    }
    return BZH_E_ARG;
}

}  // namespace bzh
