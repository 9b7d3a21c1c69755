// Shared by csrc/ntt.hip (LDS-tile kernel, planning) and csrc/ntt_wave_*.hip (the in-wave kernel, one translation unit per
// field so that the build compiles them in parallel): pass arguments and the small device helpers.
#pragma once
#include "ctx.hpp"
#include "field.cuh"
#include "fe29.cuh"

namespace bzh {

static constexpr int kTileElems = 2048;  // 64 KiB of LDS per workgroup
static constexpr int kNttThreads = 256;  // two 4-wave workgroups per CU (LDS-bound); 172 VGPRs

struct NttPassArgs {
    const uint32_t* src;  // pass input  (same index map as dst)
    uint32_t* dst;        // pass output: passes before the last may run in place (each tile rewrites
                          // exactly what it read); the last pass transposes and must not.
    unsigned log_n;
    int r;      // log2 R
    int logA;   // log2 A
    int logB;   // log2 B
    int logW;   // log2 W
    int last;
    int nprev;
    int prev_bits[4];
    const uint32_t* sub_tw;  // omega_R^j, j < R/2
    const uint32_t* tw_lo;   // omega_N^e, e < 2^h
    const uint32_t* tw_hi;   // omega_N^(e << h)
    const uint32_t* tw_direct;  // this pass's inter-pass twiddles as an R x B table (row-major, read like the data); or null
    int h;
    const uint32_t* pre_lo;  // first pass: element n *= pre(n)   (coset shift^n)
    const uint32_t* pre_hi;
    const uint32_t* post_lo;  // last pass: output k *= post(k)    (n^-1 * shift^-k)
    const uint32_t* post_hi;
    // Cheap scalings: halo2's extended coset uses shift = ZETA with ZETA^3 = 1, so shift^n takes only
    // three values; a plain inverse needs the single constant n^-1.  cube[i] multiplies index = i mod 3.
    int cube_pre;   // first pass: element n *= cube[n % 3] (cube[0] == 1 is skipped)
    int cube_post;  // last pass:  output k *= cube[k % 3]
    int tw_always;  // this pass's twiddle table carries a folded n^-1: apply it even when the exponent is 0
    int nz;         // first pass of a zero-padded transform: the source holds N >> nz coefficients per vector (rows
                    // >= R >> nz are zero and are not read); the first nz stages then only replicate values
    uint32_t cube[3][8];
    uint32_t* dst29;  // last pass, non-null: vector b's output goes to the unsaturated planes at dst29 + b * 9 * N (fe29.cuh: what the
                      // quotient evaluator reads) INSTEAD of dst -- coeff_to_extended straight into the evaluator's format
};
// the last pass's store: saturated element k of the vector, or its fe29 planes
template <class P>
__device__ __forceinline__ void ntt_store_out(const NttPassArgs& g, uint32_t* vec, size_t vec_index, size_t N, size_t k, const Fe<P>& v) {
    if constexpr (fe29_supported<P>()) {
        if (g.dst29) {
            fe29_store_planes<P>(g.dst29 + vec_index * 9 * N, k, N, fe29_from_sat_reduced(v));
            return;
        }
    }
    fe_store(vec + k * 8, v);
}

template <class P>
__device__ __forceinline__ Fe<P> tile_get(const uint4* t, int idx) {
    uint4 a = t[idx], b = t[kTileElems + idx];
    Fe<P> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
template <class P>
__device__ __forceinline__ void tile_put(uint4* t, int idx, const Fe<P>& v) {
    t[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    t[kTileElems + idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
template <class P>
__device__ __forceinline__ Fe<P> pow_table(const uint32_t* lo, const uint32_t* hi, int h, size_t e) {
    Fe<P> a = fe_load<P>(lo + (e & (((size_t)1 << h) - 1)) * 8);
    Fe<P> b = fe_load<P>(hi + (e >> h) * 8);
    return fe_mul(a, b);
}
template <class P>
__device__ __forceinline__ Fe<P> cube_const(const NttPassArgs& g, unsigned i) {
    Fe<P> r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.l[k] = i == 0 ? g.cube[0][k] : (i == 1 ? g.cube[1][k] : g.cube[2][k]);
    return r;
}
__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }


}  // namespace bzh
