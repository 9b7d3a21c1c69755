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
// (csrc/prove.hip: Compiler; tests/helpers/expr.py for the public bzh_expr_eval).  One thread per row; intermediates live in a small private
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


// ---------------------------------------------------------------------------------------------------------------
// VM v2: the quotient evaluator of the whole-proof driver (csrc/prove.hip, Compiler2).
//
// The real Board / Shot constraint systems are 140 / 81 constraint polynomials with ~1100 / ~700 multiplications
// as trees; after sharing common subexpressions and factoring each gate's selector out of its constraints about half
// remain, but sharing needs more live values than four, and a register file cannot be indexed dynamically (it turns
// into scratch memory: see k_expr_eval_regs4).  So:
//   * four register values r0..r3 used as an evaluation STACK whose depth the compiler tracks: every instruction names
//     its result register statically, and the interpreter dispatches (a wave-uniform switch, scalar branch) to a body
//     specialised for that register -- no per-limb selects;
//   * long-lived values (the folded accumulators, shared subexpressions) live in an LDS slot file
//     [slot][half][thread] of uint4, conflict-free ds_read/write_b128;
//   * leaves (column at a rotation, constant, LDS slot) are operands of the instruction that consumes them.
// Instruction forms: SS r[p] = r[p] op r[p+1];  SL r[p] = r[p] op leaf;  LL r[p] = leafA op leafB;  unary NEG / LOAD /
// STORE.  op in {ADD, SUB, MUL, RSUB (b - a)}.  Modular-integer VALU work, no MFMA.
// ---------------------------------------------------------------------------------------------------------------
struct ExprOp2 {
    uint8_t code, a_kind, b_kind, pad;
    int32_t a_idx, b_idx;
    int16_t a_rot, b_rot;
};
static_assert(sizeof(ExprOp2) == 16, "ExprOp2 layout");

static constexpr int kVm2Threads = 128;

template <class P>
__global__ void __launch_bounds__(kVm2Threads) k_expr_vm2(const ExprOp2* __restrict__ prog, int nops, const uint32_t* const* __restrict__ cols,
                                                           const size_t* __restrict__ strides, const uint32_t* __restrict__ consts,
                                                           size_t const_stride, size_t size, uint32_t* __restrict__ out) {
    extern __shared__ uint4 vm2_lds[];
    const size_t r = blockIdx.x * (size_t)kVm2Threads + threadIdx.x, v = blockIdx.y;
    if (r >= size) return;   // size is a multiple of the block size for every domain the prover uses; no barriers below
    const size_t mask = size - 1;
    const uint32_t* cv = consts + v * const_stride * 8;
    const unsigned t = threadIdx.x;
    Fe<P> r0 = fe_zero<P>(), r1 = r0, r2 = r0, r3 = r0;
    auto leaf = [&](int kind, int idx, int rot) -> Fe<P> {
        if (kind == BZH_EXPR_COLUMN) {
            const size_t row = (r + (size_t)(int64_t)rot) & mask;
            return fe_load<P>(cols[idx] + (v * strides[idx] + row) * 8);
        }
        if (kind == BZH_EXPR_CONST) return fe_load<P>(cv + (size_t)idx * 8);
        Fe<P> x;   // LDS slot
        const uint4 lo = vm2_lds[((size_t)idx * 2) * kVm2Threads + t], hi = vm2_lds[((size_t)idx * 2 + 1) * kVm2Threads + t];
        x.l[0] = lo.x, x.l[1] = lo.y, x.l[2] = lo.z, x.l[3] = lo.w, x.l[4] = hi.x, x.l[5] = hi.y, x.l[6] = hi.z, x.l[7] = hi.w;
        return x;
    };
    auto store = [&](int idx, const Fe<P>& x) {
        vm2_lds[((size_t)idx * 2) * kVm2Threads + t] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
        vm2_lds[((size_t)idx * 2 + 1) * kVm2Threads + t] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    };
#define VM2_ARITH(OP, A, B) ((OP) == 0 ? fe_add(A, B) : ((OP) == 1 ? fe_sub(A, B) : ((OP) == 2 ? fe_mul(A, B) : fe_sub(B, A))))
#define VM2_SS(OP, P0, RA, RB) case (0 << 4) | ((OP) << 2) | (P0): RA = VM2_ARITH(OP, RA, RB); break;
#define VM2_SL(OP, P0, RA)     case (1 << 4) | ((OP) << 2) | (P0): { const Fe<P> b_ = leaf(op.b_kind, op.b_idx, op.b_rot); RA = VM2_ARITH(OP, RA, b_); } break;
#define VM2_LL(OP, P0, RA)     case (2 << 4) | ((OP) << 2) | (P0): { const Fe<P> a_ = leaf(op.a_kind, op.a_idx, op.a_rot); const Fe<P> b_ = leaf(op.b_kind, op.b_idx, op.b_rot); RA = VM2_ARITH(OP, a_, b_); } break;
#define VM2_ALLOPS_SS(P0, RA, RB) VM2_SS(0, P0, RA, RB) VM2_SS(1, P0, RA, RB) VM2_SS(2, P0, RA, RB) VM2_SS(3, P0, RA, RB)
#define VM2_ALLOPS_SL(P0, RA) VM2_SL(0, P0, RA) VM2_SL(1, P0, RA) VM2_SL(2, P0, RA) VM2_SL(3, P0, RA)
#define VM2_ALLOPS_LL(P0, RA) VM2_LL(0, P0, RA) VM2_LL(1, P0, RA) VM2_LL(2, P0, RA)
#define VM2_UN(P0, RA)                                                                                   \
    case (3 << 4) | (0 << 2) | (P0): RA = fe_neg(RA); break;                                              \
    case (3 << 4) | (1 << 2) | (P0): RA = leaf(op.a_kind, op.a_idx, op.a_rot); break;                     \
    case (3 << 4) | (2 << 2) | (P0): store(op.a_idx, RA); break;
    ExprOp2 nxt = prog[0];
    for (int i = 0; i < nops; i++) {
        const ExprOp2 op = nxt;       // wave-uniform: scalar loads, scalar branch
        nxt = prog[i + 1 < nops ? i + 1 : i];   // the next instruction is fetched while this one executes
        switch (op.code) {
            VM2_ALLOPS_SS(0, r0, r1) VM2_ALLOPS_SS(1, r1, r2) VM2_ALLOPS_SS(2, r2, r3)
            VM2_ALLOPS_SL(0, r0) VM2_ALLOPS_SL(1, r1) VM2_ALLOPS_SL(2, r2) VM2_ALLOPS_SL(3, r3)
            VM2_ALLOPS_LL(0, r0) VM2_ALLOPS_LL(1, r1) VM2_ALLOPS_LL(2, r2) VM2_ALLOPS_LL(3, r3)
            VM2_UN(0, r0) VM2_UN(1, r1) VM2_UN(2, r2) VM2_UN(3, r3)
            default: break;
        }
    }
#undef VM2_UN
#undef VM2_ALLOPS_LL
#undef VM2_ALLOPS_SL
#undef VM2_ALLOPS_SS
#undef VM2_LL
#undef VM2_SL
#undef VM2_SS
#undef VM2_ARITH
    fe_store(out + (v * size + r) * 8, r0);
}

// program: ExprOp2[nops] on the device; result in r0; nlds LDS slots (32 bytes per thread each)
int expr_eval2(bzh_ctx* ctx, int field, const void* d_prog, int nops, const uint32_t* const* d_cols, const size_t* d_strides,
               const uint32_t* d_consts, size_t const_stride, size_t size, size_t batch, int nlds, uint32_t* d_out) {
    if (size % kVm2Threads) return BZH_E_ARG;
    const size_t lds = (size_t)std::max(nlds, 1) * 2 * kVm2Threads * sizeof(uint4);
    if (lds > 64 * 1024) return BZH_E_RANGE;
    ScopedTimer t(ctx, BZH_T_QUOTIENT);
    const dim3 grid((unsigned)(size / kVm2Threads), (unsigned)batch), block(kVm2Threads);
    const ExprOp2* p = (const ExprOp2*)d_prog;
    switch (field) {
        case BZH_FIELD_FP:
            hipLaunchKernelGGL((k_expr_vm2<FpParams>), grid, block, lds, ctx->stream, p, nops, d_cols, d_strides, d_consts, const_stride, size, d_out);
            break;
        case BZH_FIELD_FQ:
            hipLaunchKernelGGL((k_expr_vm2<FqParams>), grid, block, lds, ctx->stream, p, nops, d_cols, d_strides, d_consts, const_stride, size, d_out);
            break;
        default: return BZH_E_ARG;
    }
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}

int expr_slots() { return kExprSlots; }

}  // namespace bzh
