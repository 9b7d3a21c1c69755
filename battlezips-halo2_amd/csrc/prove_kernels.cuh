// Part of the whole-proof translation unit (csrc/prove.hip includes the parts in order; they share one anonymous namespace):
// small device kernels of the prover and the host-side field helpers (portable Fe<P> arithmetic).
#pragma once
namespace bzh {

namespace {

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
// v[b][i] *= s[b * s_stride]   (chain the permutation sets: start from the previous set's hand-over value)
template <class P>
__global__ void __launch_bounds__(256) k_scale_rows(uint32_t* __restrict__ v, size_t n, const uint32_t* __restrict__ s,
                                                      size_t s_stride) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= n) return;
    uint32_t* e = v + (b * n + i) * 8;
    fe_store(e, fe_mul(fe_load<P>(e), fe_load<P>(s + b * s_stride * 8)));
}

// commitment scalars of `count` Lagrange-basis columns with the constant c_v = column_v[ref_row] taken out:
//   sc[v] = [ column_v - c_v (n entries) | 0 | blind_v | c_v ]      against the table (g_lagrange | u | w | g_0)
template <class P>
__global__ void __launch_bounds__(256) k_commit_shift(const uint32_t* __restrict__ polys, size_t pitch, size_t n, size_t ref_row,
                                                        const uint32_t* __restrict__ blinds, uint32_t* __restrict__ sc) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, v = blockIdx.y;
    if (i >= n + 3) return;
    const Fe<P> c = fe_load<P>(polys + (v * pitch + ref_row) * 8);
    Fe<P> o;
    if (i < n) o = fe_sub(fe_load<P>(polys + (v * pitch + i) * 8), c);
    else if (i == n) o = fe_zero<P>();
    else if (i == n + 1) o = fe_load<P>(blinds + v * 8);
    else o = c;
    fe_store(sc + (v * (n + 3) + i) * 8, o);
}

// `ncols` saturated columns of `size` elements (2^256-Montgomery) -> the quotient evaluator's unsaturated planes (fe29.cuh:
// value x 2^261, below 2 p, limbs carried; 9 size words per column)
template <class P>
__global__ void __launch_bounds__(256) k_sat_to_fe29_planes(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t size) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x, c = blockIdx.y;
    if (i >= size) return;
    fe29_store_planes<P>(dst + c * 9 * size, i, size, fe29_from_sat_reduced(fe_load<P>(src + (c * size + i) * 8)));
}

// flag |= any word of rows[b][0 .. words) non-zero
__global__ void __launch_bounds__(256) k_any_nonzero(const uint32_t* __restrict__ p, size_t words, size_t row_stride_words,
                                                       uint32_t* __restrict__ flag) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i < words && p[b * row_stride_words + i]) atomicOr(flag, 1u);
}

// dst[b][j][0..n) = srcs[j] + b * strides[j]   (gather of (polynomial, proof) rows for the batched evaluations)
__global__ void __launch_bounds__(256) k_gather_rows(uint4* __restrict__ dst, const uint4* const* __restrict__ srcs,
                                                       const size_t* __restrict__ strides, size_t n, size_t J) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, j = blockIdx.y, b = blockIdx.z;
    if (i >= 2 * n) return;
    dst[((b * J + j) * n) * 2 + i] = srcs[j][b * strides[j] * 2 + i];
}

// rows of 64-byte draws for every proof of a batch: raw[(b * count + i) * 16 ..] = ChaCha20(key_b, counter0 + i)
__global__ void __launch_bounds__(256) k_chacha20_rows(const uint32_t* __restrict__ keys, uint64_t counter0, size_t count,
                                                        uint32_t* __restrict__ raw) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= count) return;
    uint32_t key[8], out[16];
    for (int k = 0; k < 8; k++) key[k] = keys[b * 8 + k];
    chacha20_block(key, counter0 + i, out);
    uint4* o = reinterpret_cast<uint4*>(raw + (b * count + i) * 16);
    for (int k = 0; k < 4; k++) o[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
}

// ---------------------------------------------------------------------------
// host field helpers (portable Fe<P> arithmetic, Montgomery form unless noted)
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
template <class P>
static Fe<P> h_from_bytes(const uint8_t* b) {  // canonical little-endian -> Montgomery
    uint64_t l[4];
    memcpy(l, b, 32);
    return fe_to_mont(h_load<P>(l));
}
template <class P>
static Fe<P> h_pow_u64(Fe<P> base, uint64_t e) {
    Fe<P> acc = fe_one<P>();
    for (; e; e >>= 1) {
        if (e & 1) acc = fe_mul(acc, base);
        base = fe_sqr(base);
    }
    return acc;
}
// Field::random: 64 bytes little-endian mod p (Montgomery out)
template <class P>
static Fe<P> h_from_u512(const uint8_t* b) {
    uint64_t lo[4], hi[4];
    memcpy(lo, b, 32);
    memcpy(hi, b + 32, 32);
    const Fe<P> r2 = fe_r2<P>();
    return fe_add(fe_mul(h_load<P>(lo), r2), fe_mul(fe_mul(h_load<P>(hi), r2), r2));
}

template <class P>
struct FieldMeta;
template <>
struct FieldMeta<FpParams> {
    static constexpr unsigned S = 32;
    static constexpr uint32_t gen = 5;
    static constexpr int id = BZH_FIELD_FP;
};
template <>
struct FieldMeta<FqParams> {
    static constexpr unsigned S = 32;
    static constexpr uint32_t gen = 5;
    static constexpr int id = BZH_FIELD_FQ;
};
