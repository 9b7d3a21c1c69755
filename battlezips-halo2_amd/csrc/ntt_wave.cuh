// The in-wave NTT pass kernel and its launcher template (instantiated per field in csrc/ntt_wave_{fp,fq,bnfr}.hip).
#pragma once
#include "ntt_common.cuh"

namespace bzh {

// ---------------------------------------------------------------------------
// k_ntt_pass_wave: the same pass with the butterflies exchanged INSIDE the wavefront -- no LDS tile, no barrier.
//
// A tile is still R x W = 2048 elements and a workgroup four waves, but every wave owns W/4 whole columns: 512 elements,
// eight per lane in registers.  With t the (bit-reversed-input) row index inside the tile, a radix-2 stage s pairs t and
// t ^ 2^s; a stage runs when bit s of t is a REGISTER bit (both elements of a pair in one lane).  Three row bits live in the
// register index, R - 3 in the lane id above the column bits:
//   phase 0  registers hold t bits 0..2                        -> stages 0..2  (compile-time twiddles; skipped when zero-padded)
//   phase 1  swap the three register bits with lane bits        -> stages 3..5
//   phase 2  swap R - 6 register bits with the next lane bits   -> stages 6..R-1
// A swap of register bit b with lane bit L exchanges, between lanes i and i ^ 2^L, the register with b set in the lower lane
// against the register with b clear in the upper lane: v_permlane32_swap / v_permlane16_swap (gfx950) for distances 32 / 16 --
// one instruction per 32-bit register pair, no select -- and two bank-masked DPP row shifts for distances 8 / 4
// (row_shl / row_shr: the masked lanes keep `old`), quad_perm + v_cndmask for 2 / 1.  Per lane and swap: 4 pairs x 8 limbs
// x (1..2) instructions, against 4 butterflies of ~290 VALU slots per stage.
// Global accesses: a lane reads / writes its own 32-byte elements; lanes that differ in the column bits touch adjacent
// elements (W/4 x 32 B contiguous runs), the rest stride like the tile rows.
// ---------------------------------------------------------------------------
template <int LB>
__device__ __forceinline__ void lane_swap_u32(uint32_t& lo, uint32_t& hi, bool lane_bit) {
    if constexpr (LB == 5) {
        auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0], hi = r[1];
    } else if constexpr (LB == 4) {
        auto r = __builtin_amdgcn_permlane16_swap(lo, hi, false, false);
        lo = r[0], hi = r[1];
    } else if constexpr (LB == 3) {   // distance 8: lanes 0-7 of a row (banks 0, 1) have the bit clear
        const uint32_t nh = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)lo, 0x108 /* row_shl:8 */, 0xf, 0x3, false);
        const uint32_t nl = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)hi, 0x118 /* row_shr:8 */, 0xf, 0xc, false);
        lo = nl, hi = nh;
    } else if constexpr (LB == 2) {   // distance 4: banks 0, 2 have the bit clear
        const uint32_t nh = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)lo, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
        const uint32_t nl = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)hi, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
        lo = nl, hi = nh;
    } else {                          // distance 2 / 1: inside a quad, no bank mask to tell the lanes apart
        const uint32_t send = lane_bit ? lo : hi;
        const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, LB == 1 ? 0x4e /* quad_perm:[2,3,0,1] */ : 0xb1 /* [1,0,3,2] */,
                                                                 0xf, 0xf, false);
        lo = lane_bit ? recv : lo;
        hi = lane_bit ? hi : recv;
    }
}
// swap register bit RB of the eight elements with lane bit LB
template <class P, int RB, int LB>
__device__ __forceinline__ void lane_swap(Fe<P> (&x)[8], int lane) {
    const bool lb = (lane >> LB) & 1;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        if (m & (1 << RB)) continue;
#pragma unroll
        for (int q = 0; q < 8; q++) lane_swap_u32<LB>(x[m].l[q], x[m | (1 << RB)].l[q], lb);
    }
}
// one radix-2 DIT stage on register bit RB; tw(m) = twiddle of the pair whose lower register index is m (null: 1)
template <class P, int RB, class Tw>
__device__ __forceinline__ void reg_stage(Fe<P> (&x)[8], Tw tw) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
        if (m & (1 << RB)) continue;
        Fe<P> v = x[m | (1 << RB)];
        const uint32_t* t = tw(m);
        if (t) v = fe_mul(v, fe_load<P>(t));
        x[m | (1 << RB)] = fe_sub(x[m], v);
        x[m] = fe_add(x[m], v);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// LAST is a template parameter: with both address maps in one function the eight elements stay in scratch (the optimiser
// gives up promoting the array), 272 B per lane
template <class P, int R, bool LAST>
__global__ void __launch_bounds__(kNttThreads) k_ntt_pass_wave(NttPassArgs g) {
    static_assert(R >= 6 && R <= 9, "three register bits + three to six lane bits");
    constexpr int RL = R - 3, CL = 6 - RL, N2 = RL - 3;      // lane row bits, lane column bits, swaps of phase 2
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int logW = g.logW;                                  // == CL + 2 (host)
    const int col = (wave << CL) | (lane & ((1 << CL) - 1));
    const int lr = lane >> CL;                                // phase-0 t bits 3..R-1
    const size_t N = (size_t)1 << g.log_n;
    const uint32_t* vin = g.src + (size_t)blockIdx.y * N * 8;
    uint32_t* vec = g.dst + (size_t)blockIdx.y * N * 8;
    const size_t tile_id = blockIdx.x;
    const size_t B = (size_t)1 << g.logB;
    const size_t tiles_per_a = LAST ? 1 : (B >> logW);
    const size_t a_idx = LAST ? 0 : tile_id / tiles_per_a, j0 = LAST ? 0 : ((tile_id - a_idx * tiles_per_a) << logW);

    Fe<P> x[8];
    // ---- load: element (t, col) of the tile, t = (lr << 3) | m ----
    if (!LAST) {
        const uint32_t* base = g.nz ? g.src + (size_t)blockIdx.y * (N >> g.nz) * 8 + j0 * 8 : vin + ((a_idx << (R + g.logB)) + j0) * 8;
        const int rep = g.nz ? (1 << g.nz) : 1;               // zero-padded source: the low nz bits of t only replicate
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (m & (rep - 1)) continue;
            const uint32_t t = ((uint32_t)lr << 3) | (uint32_t)m;
            const size_t row = bitrev(t, R);
            Fe<P> v = fe_load<P>(base + (row * B + col) * 8);
            if (g.pre_lo) v = fe_mul(v, pow_table<P>(g.pre_lo, g.pre_hi, g.h, row * B + j0 + col));
            if (g.cube_pre) {
                const unsigned c3 = (unsigned)((row * B + j0 + col) % 3);
                if (c3) v = fe_mul(v, cube_const<P>(g, c3));
            }
            x[m] = v;
        }
        // replicate with compile-time register indices (a run-time index would put x[] in scratch)
        if (g.nz >= 1) x[1] = x[0], x[3] = x[2], x[5] = x[4], x[7] = x[6];
        if (g.nz >= 2) x[2] = x[0], x[3] = x[0], x[6] = x[4], x[7] = x[4];
        if (g.nz >= 3) x[4] = x[0], x[5] = x[0], x[6] = x[0], x[7] = x[0];
    } else {
        size_t d = (tile_id << logW) + col, a = 0;
        for (int q = 0; q < g.nprev; q++) {
            a = (a << g.prev_bits[q]) | (d & (((size_t)1 << g.prev_bits[q]) - 1));
            d >>= g.prev_bits[q];
        }
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const uint32_t t = ((uint32_t)lr << 3) | (uint32_t)m;
            const size_t rr = bitrev(t, R);
            Fe<P> v = fe_load<P>(vin + ((a << R) + rr) * 8);
            if (g.pre_lo) v = fe_mul(v, pow_table<P>(g.pre_lo, g.pre_hi, g.h, rr));
            if (g.cube_pre && (rr % 3)) v = fe_mul(v, cube_const<P>(g, (unsigned)(rr % 3)));
            x[m] = v;
        }
    }
    const uint32_t* sub = g.sub_tw;
    // ---- phase 0: stages 0..2 on t bits held by the register index (twiddle index j = low s bits of m) ----
    if (g.nz < 1) reg_stage<P, 0>(x, [&](int) -> const uint32_t* { return nullptr; });
    if (g.nz < 2) reg_stage<P, 1>(x, [&](int m) -> const uint32_t* { return (m & 1) ? sub + ((size_t)(m & 1) << (R - 2)) * 8 : nullptr; });
    if (g.nz < 3) reg_stage<P, 2>(x, [&](int m) -> const uint32_t* { return (m & 3) ? sub + ((size_t)(m & 3) << (R - 3)) * 8 : nullptr; });
    // ---- phase 1: t bits 3..5 into the registers ----
    lane_swap<P, 0, CL + 0>(x, lane);
    lane_swap<P, 1, CL + 1>(x, lane);
    lane_swap<P, 2, CL + 2>(x, lane);
    const uint32_t lo3 = (uint32_t)(lane >> CL) & 7u;        // t bits 0..2 now sit in the lane id
    reg_stage<P, 0>(x, [&](int) -> const uint32_t* { return sub + ((size_t)lo3 << (R - 4)) * 8; });
    reg_stage<P, 1>(x, [&](int m) -> const uint32_t* { return sub + ((size_t)(lo3 | ((uint32_t)(m & 1) << 3)) << (R - 5)) * 8; });
    reg_stage<P, 2>(x, [&](int m) -> const uint32_t* { return sub + ((size_t)(lo3 | ((uint32_t)(m & 3) << 3)) << (R - 6)) * 8; });
    // ---- phase 2: t bits 6..R-1 into the registers (N2 of them); t bits 3..5: bit q in lane bit CL+3+q if q < N2, else register bit q
    if constexpr (N2 >= 1) lane_swap<P, 0, (N2 >= 1 ? CL + 3 : 0)>(x, lane);
    if constexpr (N2 >= 2) lane_swap<P, 1, (N2 >= 2 ? CL + 4 : 0)>(x, lane);
    if constexpr (N2 >= 3) lane_swap<P, 2, (N2 >= 3 ? CL + 5 : 0)>(x, lane);
    const uint32_t mid_lane = N2 ? (((uint32_t)lane >> (CL + 3)) & ((1u << N2) - 1u)) : 0u;   // t bits 3..3+N2-1
    auto mid3 = [&](int m) -> uint32_t { return mid_lane | ((uint32_t)m & (7u & ~((1u << N2) - 1u))); };   // t bits 3..5 of register m
    if constexpr (N2 >= 1) reg_stage<P, 0>(x, [&](int m) -> const uint32_t* { return sub + ((size_t)(lo3 | (mid3(m) << 3)) << (R - 7)) * 8; });
    if constexpr (N2 >= 2)
        reg_stage<P, 1>(x, [&](int m) -> const uint32_t* { return sub + ((size_t)(lo3 | (mid3(m) << 3) | ((uint32_t)(m & 1) << 6)) << (R - 8)) * 8; });
    if constexpr (N2 >= 3)
        reg_stage<P, 2>(x, [&](int m) -> const uint32_t* { return sub + ((size_t)(lo3 | (mid3(m) << 3) | ((uint32_t)(m & 3) << 6)) << (R - 9)) * 8; });
    // ---- store: register m of this lane holds tile row t(m) ----
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const uint32_t t = lo3 | (mid3(m) << 3) | (((uint32_t)m & ((1u << N2) - 1u)) << 6);
        Fe<P> v = x[m];
        if (!LAST) {
            uint32_t* base = vec + ((a_idx << (R + g.logB)) + j0) * 8;
            if (g.tw_direct) {
                v = fe_mul(v, fe_load<P>(g.tw_direct + ((size_t)t * B + j0 + col) * 8));
            } else {
                const size_t e = ((size_t)t * (j0 + col)) << g.logA;
                if (e || g.tw_always) v = fe_mul(v, pow_table<P>(g.tw_lo, g.tw_hi, g.h, e));
            }
            fe_store(base + ((size_t)t * B + col) * 8, v);
        } else {
            const size_t k = ((size_t)t << g.logA) + (tile_id << logW) + col;
            if (g.post_lo) v = fe_mul(v, pow_table<P>(g.post_lo, g.post_hi, g.h, k));
            if (g.cube_post) v = fe_mul(v, cube_const<P>(g, (unsigned)(k % 3)));
            ntt_store_out<P>(g, vec, (size_t)blockIdx.y, N, k, v);
        }
    }
}


// launch one pass of r = 6..9 radix bits on a full 2048-element tile
template <class P>
void launch_ntt_pass_wave(const NttPassArgs& aa, unsigned tiles, unsigned nb, hipStream_t stream) {
    const dim3 grid(tiles, nb), blk(kNttThreads);
#define BZH_WAVE_PASS(RB)                                                                          \
    if (aa.last) hipLaunchKernelGGL((k_ntt_pass_wave<P, RB, true>), grid, blk, 0, stream, aa);      \
    else hipLaunchKernelGGL((k_ntt_pass_wave<P, RB, false>), grid, blk, 0, stream, aa)
    switch (aa.r) {
        case 6: BZH_WAVE_PASS(6); break;
        case 7: BZH_WAVE_PASS(7); break;
        case 8: BZH_WAVE_PASS(8); break;
        default: BZH_WAVE_PASS(9); break;
    }
#undef BZH_WAVE_PASS
}

}  // namespace bzh
