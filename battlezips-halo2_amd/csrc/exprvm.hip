// Gate-expression evaluator over the extended coset.
//
// Device counterpart of halo2_proofs 0.2.0 `poly::Evaluator` as driven by
// `vanishing::Argument::construct` in plonk::create_proof step 6 (UPSTREAM, un-vendored:
// Cargo.lock:382-385; entered from benches/shot.rs:68, src/circuits/board.rs:913-920): every
// custom gate, the permutation and the lookup identities are evaluated at each of the 2^(k+3)
// points of the extended domain and folded with powers of y.  The gates are DATA here
// (the 24 Shot / 57 Board gates, 19 of them from halo2_gadgets, live in crates this repository
// cannot read): a straight-line program over
//     COLUMN(c, rot)  value of column c at row (r + rot) mod size   (rot in extended-domain steps)
//     CONST(i)        i-th constant (challenges, y, selectors folded by the host, ...)
//     SLOT(s)         an earlier intermediate
// with ops ADD / SUB / MUL / NEG / COPY, compiled on the host from halo2-style expression trees
// (battlezips-halo2_amd/bzh2/expr.py).  One thread per row; intermediates live in a small private
// slot file.  Modular-integer VALU work, no MFMA.
#include "ctx.hpp"
#include "field.cuh"

namespace bzh {

static constexpr int kExprSlots = 24;

struct ExprOp {  // mirrors bzh_expr_op
    uint8_t op, dst, a_kind, b_kind;
    int32_t a_idx, b_idx, a_rot, b_rot;
};

template <class P>
__device__ __forceinline__ Fe<P> expr_operand(int kind, int idx, int rot, const Fe<P>* slots, const uint32_t* const* cols,
                                               const size_t* strides, const uint32_t* consts, size_t r, size_t mask, size_t v) {
    if (kind == BZH_EXPR_SLOT) return slots[idx];
    if (kind == BZH_EXPR_CONST) return fe_load<P>(consts + (size_t)idx * 8);
    const size_t row = (r + (size_t)(int64_t)rot) & mask;  // two's complement wrap, size is a power of two
    return fe_load<P>(cols[idx] + (v * strides[idx] + row) * 8);
}

// grid.y = vector (proof) index v: column c of vector v starts strides[c] elements after vector v-1's (0 = shared by
// all vectors), its constants const_stride elements after (0 = shared), its output `size` elements after.
template <class P>
__global__ void __launch_bounds__(256) k_expr_eval(const ExprOp* __restrict__ prog, int nops, const uint32_t* const* __restrict__ cols,
                                                     const size_t* __restrict__ strides, const uint32_t* __restrict__ consts,
                                                     size_t const_stride, size_t size, int result_slot, uint32_t* __restrict__ out) {
    const size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x, v = blockIdx.y;
    if (r >= size) return;
    const size_t mask = size - 1;
    const uint32_t* cv = consts + v * const_stride * 8;
    Fe<P> slots[kExprSlots];
    for (int i = 0; i < nops; i++) {
        const ExprOp op = prog[i];  // wave-uniform: scalar loads
        const Fe<P> a = expr_operand<P>(op.a_kind, op.a_idx, op.a_rot, slots, cols, strides, cv, r, mask, v);
        Fe<P> val;
        if (op.op == BZH_EXPR_NEG) {
            val = fe_neg(a);
        } else if (op.op == BZH_EXPR_COPY) {
            val = a;
        } else {
            const Fe<P> b = expr_operand<P>(op.b_kind, op.b_idx, op.b_rot, slots, cols, strides, cv, r, mask, v);
            val = op.op == BZH_EXPR_ADD ? fe_add(a, b) : (op.op == BZH_EXPR_SUB ? fe_sub(a, b) : fe_mul(a, b));
        }
        slots[op.dst] = val;
    }
    fe_store(out + (v * size + r) * 8, slots[result_slot]);
}

// Register-file variant: programs that keep at most four intermediates live (Sethi-Ullman ordering needs 4 for the
// whole quotient of a Board/Shot-shaped circuit) hold them in VGPRs.  The generic kernel's dynamically indexed slot
// file lives in scratch memory: 32 B written per op and row, 23 GB per quotient launch of a 16-proof batch at k = 14
// (PMC WRITE_SIZE), which made that kernel HBM-bound instead of multiplier-bound.
// Four named values and uniform selects (the op fields are wave-uniform): nothing the optimiser could turn back into
// an indexed private array.
template <class P>
__device__ __forceinline__ Fe<P> fe_select(bool c, const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.l[k] = c ? a.l[k] : b.l[k];
    return r;
}

template <class P>
__global__ void __launch_bounds__(256) k_expr_eval_regs4(const ExprOp* __restrict__ prog, int nops,
                                                           const uint32_t* const* __restrict__ cols,
                                                           const size_t* __restrict__ strides, const uint32_t* __restrict__ consts,
                                                           size_t const_stride, size_t size, int result_slot,
                                                           uint32_t* __restrict__ out) {
    const size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x, v = blockIdx.y;
    if (r >= size) return;
    const size_t mask = size - 1;
    const uint32_t* cv = consts + v * const_stride * 8;
    Fe<P> s0 = fe_zero<P>(), s1 = s0, s2 = s0, s3 = s0;
#define BZH_EXPR_SLOTGET(i) fe_select((i) < 2, fe_select((i) == 0, s0, s1), fe_select((i) == 2, s2, s3))
#define BZH_EXPR_FETCH(dst, kind, idx, rot)                                          \
    if ((kind) == BZH_EXPR_SLOT) {                                                   \
        dst = BZH_EXPR_SLOTGET(idx);                                                 \
    } else if ((kind) == BZH_EXPR_CONST) {                                           \
        dst = fe_load<P>(cv + (size_t)(idx) * 8);                                    \
    } else {                                                                         \
        const size_t row__ = (r + (size_t)(int64_t)(rot)) & mask;                    \
        dst = fe_load<P>(cols[idx] + (v * strides[idx] + row__) * 8);                \
    }
    for (int i = 0; i < nops; i++) {
        const ExprOp op = prog[i];  // wave-uniform: scalar loads
        Fe<P> a, val;
        BZH_EXPR_FETCH(a, op.a_kind, op.a_idx, op.a_rot)
        if (op.op == BZH_EXPR_NEG) {
            val = fe_neg(a);
        } else if (op.op == BZH_EXPR_COPY) {
            val = a;
        } else {
            Fe<P> b;
            BZH_EXPR_FETCH(b, op.b_kind, op.b_idx, op.b_rot)
            val = op.op == BZH_EXPR_ADD ? fe_add(a, b) : (op.op == BZH_EXPR_SUB ? fe_sub(a, b) : fe_mul(a, b));
        }
        const int d = op.dst;
        s0 = fe_select(d == 0, val, s0);
        s1 = fe_select(d == 1, val, s1);
        s2 = fe_select(d == 2, val, s2);
        s3 = fe_select(d == 3, val, s3);
    }
    const Fe<P> res = BZH_EXPR_SLOTGET(result_slot);
    fe_store(out + (v * size + r) * 8, res);
#undef BZH_EXPR_FETCH
#undef BZH_EXPR_SLOTGET
}

template <class P>
static int expr_eval_t(bzh_ctx* ctx, const ExprOp* d_prog, int nops, const uint32_t* const* d_cols, const size_t* d_strides,
                       const uint32_t* d_consts, size_t const_stride, size_t size, int result_slot, size_t batch, int nslots,
                       uint32_t* d_out) {
    ScopedTimer t(ctx, BZH_T_POLY);
    const dim3 grid((unsigned)((size + 255) / 256), (unsigned)batch), block(256);
    if (nslots > 0 && nslots <= 4)
        hipLaunchKernelGGL((k_expr_eval_regs4<P>), grid, block, 0, ctx->stream, d_prog, nops, d_cols, d_strides, d_consts, const_stride,
                           size, result_slot, d_out);
    else
        hipLaunchKernelGGL((k_expr_eval<P>), grid, block, 0, ctx->stream, d_prog, nops, d_cols, d_strides, d_consts, const_stride, size,
                           result_slot, d_out);
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

// nslots: number of distinct intermediate slots the program touches (1 + highest slot index), 0 = unknown
int expr_eval(bzh_ctx* ctx, int field, const void* d_prog, int nops, const uint32_t* const* d_cols, const size_t* d_strides,
              const uint32_t* d_consts, size_t const_stride, size_t size, int result_slot, size_t batch, int nslots, uint32_t* d_out) {
    const ExprOp* p = (const ExprOp*)d_prog;
    switch (field) {
        case BZH_FIELD_FP: return expr_eval_t<FpParams>(ctx, p, nops, d_cols, d_strides, d_consts, const_stride, size, result_slot, batch, nslots, d_out);
        case BZH_FIELD_FQ: return expr_eval_t<FqParams>(ctx, p, nops, d_cols, d_strides, d_consts, const_stride, size, result_slot, batch, nslots, d_out);
        case BZH_FIELD_BN254_FR: return expr_eval_t<BnFrParams>(ctx, p, nops, d_cols, d_strides, d_consts, const_stride, size, result_slot, batch, nslots, d_out);
        case BZH_FIELD_BN254_FQ: return expr_eval_t<BnFqParams>(ctx, p, nops, d_cols, d_strides, d_consts, const_stride, size, result_slot, batch, nslots, d_out);
    }
    return BZH_E_ARG;
}

int expr_slots() { return kExprSlots; }

}  // namespace bzh
