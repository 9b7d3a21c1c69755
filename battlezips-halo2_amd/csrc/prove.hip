// Whole-proof entry points: bzh_pk_create / bzh_prove_batch (boundary row b:
// "a coarser seam (whole create_proof) is what batching needs").
//
// Native counterpart of halo2_proofs 0.2.0 `plonk::{keygen_pk, create_proof}` (UPSTREAM, un-vendored:
// Cargo.lock:382-385) as called by the reference at benches/shot.rs:58-71, benches/board.rs:51-71,
// src/circuits/shot.rs:915-930, src/circuits/board.rs:907-922: one call proves `batch` independent
// witnesses of one circuit in lockstep -- every MSM, NTT, gate evaluation, scan and IPA round is ONE launch
// carrying all of them -- with the protocol, message order and randomness draw order per proof of
// create_proof, so each proof is byte-identical to proving its witness alone (the staged test drivers tests/helpers/prover_dev.py,
// oracle/halo2_oracle.py) under the same randomness stream.
//
// The circuit arrives as DATA (serialised constraint system + fixed assignment, format below): the reference's
// own constraint systems include 19 gates of the halo2_gadgets crate that is not on disk.  Host work left:
// transcripts (Blake2b), the lookup sort, challenges and blinds; everything else runs on the device out of
// one grow-only arena owned by the proving key.
//
// Circuit blob, little-endian:
//   u32 magic "BZC1" or "BZC2" | u32 k | u32 num_advice | u32 num_fixed | u32 num_instance | u32 min_degree | u8[32] vk_repr
//   u32 ngates, ngates x expr                                                   (every constraint polynomial of every gate, flattened)
//   u32 nperm, nperm x (u8 kind {0 advice, 1 fixed, 2 instance}, u32 index)
//   u32 nlookups, per lookup: u32 m, m x expr (inputs), m x expr (table)
//   u32 ncopies, ncopies x (u32 col_a, u32 row_a, u32 col_b, u32 row_b)        (indices into the permutation columns)
//   num_fixed x (u32 len, len x u8[32] canonical values)                        (rows past len are zero)
//   "BZC2" only: advice, fixed, instance query lists, each u32 count, count x (u32 column, i32 rotation), in upstream's
//   query REGISTRATION order (= the order of the evaluations in the proof); written by csrc/circuits.hip
//   expr := u8 tag, then  0 const: u8[32] | 1 advice / 2 fixed / 3 instance: u32 column, i32 rotation
//                       | 4 neg: expr | 5 add: expr expr | 6 mul: expr expr | 7 scale: expr, u8[32]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "chacha.hpp"
#include "ctx.hpp"
#include "curve.cuh"

// the table of quotient kernels generated at build time (quotient_builtin.hip); absent (null) in the generator's own link
extern "C" const bzh_builtin_quotient* bzh_builtin_quotients(size_t* count) __attribute__((weak));

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

// ---------------------------------------------------------------------------
// circuit expressions (as serialised) and evaluator expressions (over a column registry)
// ---------------------------------------------------------------------------
enum { CX_CONST = 0, CX_ADVICE = 1, CX_FIXED = 2, CX_INSTANCE = 3, CX_NEG = 4, CX_ADD = 5, CX_MUL = 6, CX_SCALE = 7 };
struct CNode {
    uint8_t tag;
    uint32_t col = 0;
    int32_t rot = 0;
    uint32_t val[8] = {0};  // Montgomery
    int a = -1, b = -1;
};

enum { EX_CONST, EX_SYMBOL, EX_QUERY, EX_NEG, EX_ADD, EX_MUL, EX_SCALE };
struct ENode {
    uint8_t tag;
    int32_t col = 0, rot = 0;  // EX_QUERY: registry index, rotation; EX_SYMBOL: col = symbol id
    uint32_t val[8] = {0};
    int a = -1, b = -1;
};
// challenge symbols bound per proof
enum { SY_THETA, SY_BETA, SY_GAMMA, SY_Y, SY_XN, SY_X1, SY_X2, SY_X4, SY_BD0 /* + permutation column index */ };

struct ConstEnt {
    int sym = -1;  // >= 0: symbol id, else literal
    uint32_t val[8] = {0};
};
struct Program {
    std::vector<bzh_expr_op> ops;
    std::vector<ConstEnt> consts;
    int result_slot = 0;
};

struct EPool {
    std::vector<ENode> n;
    // hash-consing: structurally equal nodes are one node, so that shared subexpressions of the constraint polynomials
    // (a gate's selector product, x_q - x_p of the addition gates, ...) show up as shared nodes of a DAG
    struct NodeKey {
        uint8_t tag;
        int32_t col, rot, a, b;
        uint32_t val[8];
        bool operator<(const NodeKey& o) const { return memcmp(this, &o, sizeof(NodeKey)) < 0; }
    };
    std::map<NodeKey, int> interned;
    int push(const ENode& e) {
        NodeKey k;
        memset(&k, 0, sizeof(k));
        k.tag = e.tag;
        k.col = e.col, k.rot = e.rot, k.a = e.a, k.b = e.b;
        memcpy(k.val, e.val, 32);
        auto it = interned.find(k);
        if (it != interned.end()) return it->second;
        n.push_back(e);
        interned[k] = (int)n.size() - 1;
        return (int)n.size() - 1;
    }
    template <class F>
    int cnst(const F& v) {
        ENode e;
        e.tag = EX_CONST;
        memcpy(e.val, v.l, 32);
        return push(e);
    }
    int sym(int id) {
        ENode e;
        e.tag = EX_SYMBOL;
        e.col = id;
        return push(e);
    }
    int query(int col, int rot = 0) {
        ENode e;
        e.tag = EX_QUERY;
        e.col = col;
        e.rot = rot;
        return push(e);
    }
    int un(uint8_t tag, int a) {
        ENode e;
        e.tag = tag;
        e.a = a;
        return push(e);
    }
    int bin(uint8_t tag, int a, int b) {
        ENode e;
        e.tag = tag;
        e.a = a;
        e.b = b;
        return push(e);
    }
    int neg(int a) { return un(EX_NEG, a); }
    int add(int a, int b) { return bin(EX_ADD, a, b); }
    int sub(int a, int b) { return add(a, neg(b)); }
    int mul(int a, int b) { return bin(EX_MUL, a, b); }
    int horner(const std::vector<int>& terms, int ch) {  // ((t0 * ch + t1) * ch + t2) ...
        int acc = terms[0];
        for (size_t i = 1; i < terms.size(); i++) acc = add(mul(acc, ch), terms[i]);
        return acc;
    }
};

// Sethi-Ullman ordered emission into at most BZH_EXPR_MAX_SLOTS live intermediates (VM v1; tests/helpers/expr.py
// compiles the same format for the public bzh_expr_eval entry point); leaves are free operands
struct Compiler {
    const EPool& pool;
    Program prog;
    std::vector<int> free_slots, depth;
    bool overflow = false;
    explicit Compiler(const EPool& p) : pool(p), depth(p.n.size(), -1) {
        for (int s = BZH_EXPR_MAX_SLOTS - 1; s >= 0; s--) free_slots.push_back(s);
    }
    int depth_of(int i) {
        if (depth[i] >= 0) return depth[i];
        const ENode& e = pool.n[i];
        int d;
        if (e.tag == EX_CONST || e.tag == EX_SYMBOL || e.tag == EX_QUERY) d = 0;
        else if (e.tag == EX_NEG || e.tag == EX_SCALE) d = std::max(1, depth_of(e.a));
        else {
            const int da = depth_of(e.a), db = depth_of(e.b);
            d = da != db ? std::max(da, db) : da + 1;
        }
        return depth[i] = d;
    }
    int alloc() {
        if (free_slots.empty()) {
            overflow = true;
            return 0;
        }
        const int s = free_slots.back();
        free_slots.pop_back();
        return s;
    }
    int const_index(int sym, const uint32_t* val) {
        for (size_t i = 0; i < prog.consts.size(); i++) {
            const ConstEnt& c = prog.consts[i];
            if (sym >= 0 ? c.sym == sym : (c.sym < 0 && !memcmp(c.val, val, 32))) return (int)i;
        }
        ConstEnt c;
        c.sym = sym;
        if (sym < 0) memcpy(c.val, val, 32);
        prog.consts.push_back(c);
        return (int)prog.consts.size() - 1;
    }
    struct Opnd {
        int kind, idx, rot, release;
    };
    Opnd operand(int i) {
        const ENode& e = pool.n[i];
        if (e.tag == EX_CONST) return {BZH_EXPR_CONST, const_index(-1, e.val), 0, -1};
        if (e.tag == EX_SYMBOL) return {BZH_EXPR_CONST, const_index(e.col, nullptr), 0, -1};
        if (e.tag == EX_QUERY) return {BZH_EXPR_COLUMN, e.col, e.rot, -1};
        const int s = emit(i);
        return {BZH_EXPR_SLOT, s, 0, s};
    }
    void push(int op, int dst, const Opnd& a, const Opnd& b) {
        bzh_expr_op o;
        o.op = (uint8_t)op;
        o.dst = (uint8_t)dst;
        o.a_kind = (uint8_t)a.kind;
        o.b_kind = (uint8_t)b.kind;
        o.a_idx = a.idx;
        o.b_idx = b.idx;
        o.a_rot = a.rot;
        o.b_rot = b.rot;
        prog.ops.push_back(o);
    }
    int emit(int i) {
        const ENode& e = pool.n[i];
        const Opnd none{BZH_EXPR_SLOT, 0, 0, -1};
        if (e.tag == EX_CONST || e.tag == EX_SYMBOL || e.tag == EX_QUERY) {
            const Opnd a = operand(i);
            const int d = alloc();
            push(BZH_EXPR_COPY, d, a, none);
            return d;
        }
        if (e.tag == EX_NEG) {
            const Opnd a = operand(e.a);
            const int d = a.release >= 0 ? a.release : alloc();
            push(BZH_EXPR_NEG, d, a, none);
            return d;
        }
        if (e.tag == EX_SCALE) {
            const Opnd a = operand(e.a);
            const int d = a.release >= 0 ? a.release : alloc();
            push(BZH_EXPR_MUL, d, a, Opnd{BZH_EXPR_CONST, const_index(-1, e.val), 0, -1});
            return d;
        }
        // the deeper child first, so that the shallower one never needs more slots than are left
        Opnd a, b;
        if (depth_of(e.b) > depth_of(e.a)) {
            b = operand(e.b);
            a = operand(e.a);
        } else {
            a = operand(e.a);
            b = operand(e.b);
        }
        const int d = a.release >= 0 ? a.release : (b.release >= 0 ? b.release : alloc());
        push(e.tag == EX_ADD ? BZH_EXPR_ADD : BZH_EXPR_MUL, d, a, b);
        if (a.release >= 0 && a.release != d) free_slots.push_back(a.release);
        if (b.release >= 0 && b.release != d) free_slots.push_back(b.release);
        return d;
    }
};


// ---------------------------------------------------------------------------------------------------------------
// Compiler2: the quotient's program for VM v2 (csrc/exprvm.hip: four stack registers + an LDS slot file).
//   value = (sum_j term_j y^(N-1-j)) * t_inv, terms in protocol order.  Consecutive terms of the form S * C_j with the
//   same S (a gate's constraints under its -- compressed -- selector) are folded as
//       ACC <- ACC y^m + S (C_0 y^(m-1) + ... + C_(m-1))
//   so the selector product is evaluated and multiplied in once per gate; inside a gate, subexpressions used more than
//   once are computed once and parked in LDS slots.  The arithmetic is exact field arithmetic: the value, hence every
//   proof byte, is the same as the plain Horner fold's.
// ---------------------------------------------------------------------------------------------------------------
struct ExprOp2 {  // mirrors csrc/exprvm.hip
    uint8_t code, a_kind, b_kind, pad;
    int32_t a_idx, b_idx;
    int16_t a_rot, b_rot;
};
enum { BZH_EXPR_LDS = 3 };
enum { V2_ADD = 0, V2_SUB = 1, V2_MUL = 2, V2_RSUB = 3 };
enum { V2_SS = 0, V2_SL = 1, V2_LL = 2, V2_UN = 3, V2_NEG = 0, V2_LOAD = 1, V2_STORE = 2 };
// LDS slots: 0 ACC, 1 IN, then the shared-subexpression slots, spill slots last (allocated only if a program uses them)
static constexpr int kV2Regs = 4, kV2LdsAcc = 0, kV2LdsInner = 1, kV2LdsCse0 = 2, kV2LdsCseMax = 8, kV2LdsSpills = 2;
enum { SY_YPOW0 = 4096 /* + m: y^m */ };

struct Program2 {
    std::vector<ExprOp2> ops;
    std::vector<ConstEnt> consts;
    bool ok = true;
    int nlds = 2;
};

// FNV-1a over the instruction words: ties a compiled quotient module to the program it was generated from
static uint64_t program2_hash(const Program2& pg, int field) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) {
        for (size_t i = 0; i < n; i++) h = (h ^ ((const uint8_t*)p)[i]) * 1099511628211ull;
    };
    mix(&field, sizeof(field));
    mix(&pg.nlds, sizeof(pg.nlds));
    for (const ExprOp2& o : pg.ops) {
        const int32_t w[7] = {o.code, o.a_kind, o.a_idx, o.a_rot, o.b_kind, o.b_idx, o.b_rot};
        mix(w, sizeof(w));
    }
    return h;
}

// The VM v2 program as straight-line HIP source (compiled by the caller with hipcc / hiprtc against csrc/field.cuh and handed
// back through bzh_pk_set_quotient_module).  The evaluation stack r0..r3 and the slot file become local values; every memory
// operand is loaded one instruction ahead of its use and a scheduling barrier follows every instruction -- without it the
// compiler hoists all ~750 leaf loads to the top (255 VGPRs and scratch); with it 106 VGPRs, four waves per SIMD.  Measured
// on the BoardCircuit program (1 361 instructions, 16 x 2^17 rows): 9.0 ms against the interpreter's 12.5 ms, same bits.
// builtin != 0: the flavour linked into libbzh2.so at build time (csrc/gen_quotient.cpp -> quotient_builtin.hip): kernel named
// after the program hash inside its own namespace, a host launcher, no module-level hash symbol.
static std::string program2_source(const Program2& pg, int field, bool builtin = false) {
    std::string src;
    char buf[512];
    auto add = [&](const char* fmt, auto... a) {
        snprintf(buf, sizeof(buf), fmt, a...);
        src += buf;
    };
    const unsigned long long hash = (unsigned long long)program2_hash(pg, field);
    char kname[64];
    if (builtin) snprintf(kname, sizeof(kname), "bzh_quotient_%016llx", hash);
    else snprintf(kname, sizeof(kname), "jit_quotient");
    add("// generated by libbzh2 (%s): quotient evaluator, %zu instructions\n", builtin ? "bzh_quotient_source_for_circuit" : "bzh_pk_quotient_source",
        pg.ops.size());
    if (builtin) add("namespace bzh_q_%016llx {\n", hash);
    else src += "#include \"field.cuh\"\n";
    src += "using namespace bzh;\n";
    add("typedef %s P;\n", field == BZH_FIELD_FQ ? "FqParams" : "FpParams");
    if (!builtin) add("extern \"C\" __device__ __attribute__((used)) unsigned long long jit_program_hash = 0x%llxull;\n", hash);
    // measured alternatives, all slower: the multiplication inlined (492 vs 514 proofs/s), barriers after multiplications only
    // (498), 0 / 4 / 6 shared-subexpression slots instead of 2 (478 / 503 / 505)
    src += "__device__ __noinline__ Fe<P> mulx(const Fe<P> a, const Fe<P> b) { return fe_mul(a, b); }\n";
    add("extern \"C\" __global__ void __launch_bounds__(128) %s(const uint32_t* const* __restrict__ cols, ", kname);
    src += "const size_t* __restrict__ strides, const uint32_t* __restrict__ consts, size_t const_stride, size_t size, "
           "uint32_t* __restrict__ out) {\n"
           "    const size_t r = blockIdx.x * (size_t)128 + threadIdx.x, v = blockIdx.y;\n"
           "    if (r >= size) return;\n"
           "    const size_t mask = size - 1;\n"
           "    const uint32_t* cv = consts + v * const_stride * 8;\n"
           "    Fe<P> r0 = fe_zero<P>(), r1 = r0, r2 = r0, r3 = r0;\n";
    for (int i = 0; i < std::max(pg.nlds, 1); i++) add("    Fe<P> s%d = r0;\n", i);
    const size_t nops = pg.ops.size();
    auto is_mem = [](int kind) { return kind == BZH_EXPR_COLUMN || kind == BZH_EXPR_CONST; };
    auto emit_load = [&](const char* name, size_t i, int kind, int idx, int rot) {
        if (kind == BZH_EXPR_COLUMN)
            add("    const Fe<P> %s%zu = fe_load<P>(cols[%d] + (v * strides[%d] + ((r + (size_t)(long)(%d)) & mask)) * 8);\n", name, i, idx, idx, rot);
        else if (kind == BZH_EXPR_CONST)
            add("    const Fe<P> %s%zu = fe_load<P>(cv + %d * 8);\n", name, i, idx);
    };
    auto emit_loads = [&](size_t i) {
        if (i >= nops) return;
        const ExprOp2& o = pg.ops[i];
        const int form = o.code >> 4, op = (o.code >> 2) & 3;
        if (form == V2_LL || (form == V2_UN && op == V2_LOAD)) emit_load("la", i, o.a_kind, o.a_idx, o.a_rot);
        if (form == V2_SL || form == V2_LL) emit_load("lb", i, o.b_kind, o.b_idx, o.b_rot);
    };
    auto operand = [&](const char* name, size_t i, int kind, int idx) -> std::string {
        char t[32];
        if (is_mem(kind)) snprintf(t, sizeof(t), "%s%zu", name, i);
        else snprintf(t, sizeof(t), "s%d", idx);
        return t;
    };
    auto arith = [&](int op, const std::string& a, const std::string& b) -> std::string {
        switch (op) {
            case V2_ADD: return "fe_add(" + a + ", " + b + ")";
            case V2_SUB: return "fe_sub(" + a + ", " + b + ")";
            case V2_MUL: return "mulx(" + a + ", " + b + ")";
            default: return "fe_sub(" + b + ", " + a + ")";   // RSUB: b - a
        }
    };
    static const char* const regs[4] = {"r0", "r1", "r2", "r3"};
    emit_loads(0);
    for (size_t i = 0; i < nops; i++) {
        const ExprOp2& o = pg.ops[i];
        const int form = o.code >> 4, op = (o.code >> 2) & 3, pos = o.code & 3;
        emit_loads(i + 1);
        const std::string ra = regs[pos];
        if (form == V2_SS) {
            src += "    " + ra + " = " + arith(op, ra, regs[(pos + 1) & 3]) + ";\n";
        } else if (form == V2_SL) {
            src += "    " + ra + " = " + arith(op, ra, operand("lb", i, o.b_kind, o.b_idx)) + ";\n";
        } else if (form == V2_LL) {
            src += "    " + ra + " = " + arith(op, operand("la", i, o.a_kind, o.a_idx), operand("lb", i, o.b_kind, o.b_idx)) + ";\n";
        } else if (op == V2_NEG) {
            src += "    " + ra + " = fe_neg(" + ra + ");\n";
        } else if (op == V2_LOAD) {
            src += "    " + ra + " = " + operand("la", i, o.a_kind, o.a_idx) + ";\n";
        } else {
            add("    s%d = %s;\n", o.a_idx, ra.c_str());
        }
        src += "    __builtin_amdgcn_sched_barrier(0);\n";
    }
    src += "    fe_store(out + (v * size + r) * 8, r0);\n}\n";
    if (builtin) {
        add("static void launch(unsigned gx, unsigned gy, void* st, const uint32_t* const* cols, const size_t* strides, const uint32_t* consts, "
            "size_t nc, size_t size, uint32_t* out) {\n    hipLaunchKernelGGL(%s, dim3(gx, gy), dim3(128), 0, (hipStream_t)st, cols, strides, consts, nc, size, out);\n}\n", kname);
        add("}  // namespace bzh_q_%016llx\n", hash);
    }
    return src;
}

struct Compiler2 {
    const EPool& pool;
    Program2 prog;
    int depth = 0;                       // registers r0..r(depth-1) hold the evaluation stack
    std::map<int, int> cse;              // node -> LDS slot holding its value (current scope)
    std::map<int, int> hoisted;          // node -> registry column holding its precomputed values (proof-independent)
    std::vector<int> label;              // Sethi-Ullman numbers (leaves 0), memoised per scope
    int spill_used = 0, cse_slots = 2, max_lds = kV2LdsInner;   // measured (k = 14, batch 16): 2 shared-subexpression slots beat 0, 4, 6, 8 -- occupancy matters more than the last 40 multiplications
    explicit Compiler2(const EPool& p) : pool(p), label(p.n.size(), -1) {
        if (const char* e = getenv("BZH_VM2_CSE")) cse_slots = std::max(0, std::min(kV2LdsCseMax, atoi(e)));
    }
    int nlds() const { return max_lds + 1; }

    struct Leaf {
        int kind, idx, rot;
    };
    int const_index(int sym, const uint32_t* val) {
        for (size_t i = 0; i < prog.consts.size(); i++) {
            const ConstEnt& c = prog.consts[i];
            if (sym >= 0 ? c.sym == sym : (c.sym < 0 && !memcmp(c.val, val, 32))) return (int)i;
        }
        ConstEnt c;
        c.sym = sym;
        if (sym < 0) memcpy(c.val, val, 32);
        prog.consts.push_back(c);
        return (int)prog.consts.size() - 1;
    }
    bool is_leaf(int i) const {
        const ENode& e = pool.n[i];
        return e.tag == EX_CONST || e.tag == EX_SYMBOL || e.tag == EX_QUERY || cse.count(i) || hoisted.count(i);
    }
    Leaf leaf_of(int i) {
        auto ih = hoisted.find(i);
        if (ih != hoisted.end()) return {BZH_EXPR_COLUMN, ih->second, 0};
        auto it = cse.find(i);
        if (it != cse.end()) return {BZH_EXPR_LDS, it->second, 0};
        const ENode& e = pool.n[i];
        if (e.tag == EX_CONST) return {BZH_EXPR_CONST, const_index(-1, e.val), 0};
        if (e.tag == EX_SYMBOL) return {BZH_EXPR_CONST, const_index(e.col, nullptr), 0};
        if (e.rot < -32768 || e.rot > 32767) prog.ok = false;
        return {BZH_EXPR_COLUMN, e.col, e.rot};
    }
    int label_of(int i) {
        if (is_leaf(i)) return 0;
        if (label[i] >= 0) return label[i];
        const ENode& e = pool.n[i];
        int d;
        if (e.tag == EX_NEG || e.tag == EX_SCALE) d = std::max(1, label_of(e.a));
        else {
            const int la = label_of(e.a), lb = label_of(e.b);
            d = (la == 0 && lb == 0) ? 1 : (la == lb ? la + 1 : std::max(la, lb));
        }
        return label[i] = d;
    }
    void op(int form, int o, int pos, Leaf a = {0, 0, 0}, Leaf b = {0, 0, 0}) {
        if (pos < 0 || pos >= kV2Regs) prog.ok = false;
        ExprOp2 x;
        x.code = (uint8_t)((form << 4) | (o << 2) | (pos & 3));
        x.a_kind = (uint8_t)a.kind, x.b_kind = (uint8_t)b.kind, x.pad = 0;
        x.a_idx = a.idx, x.b_idx = b.idx;
        x.a_rot = (int16_t)a.rot, x.b_rot = (int16_t)b.rot;
        prog.ops.push_back(x);
    }
    // a - b is add(a, neg(b)) in the pool: peel the negation so that it costs no instruction
    bool is_plain_neg(int i) const { return pool.n[i].tag == EX_NEG && !cse.count(i); }

    // emit node i: its value ends up in a new top-of-stack register
    void emit(int i) {
        if (is_leaf(i)) {
            op(V2_UN, V2_LOAD, depth, leaf_of(i));
            depth++;
            return;
        }
        const ENode& e = pool.n[i];
        if (e.tag == EX_NEG) {
            emit(e.a);
            op(V2_UN, V2_NEG, depth - 1);
        } else if (e.tag == EX_SCALE) {
            const Leaf c{BZH_EXPR_CONST, const_index(-1, e.val), 0};
            if (is_leaf(e.a)) {
                op(V2_LL, V2_MUL, depth, leaf_of(e.a), c);
                depth++;
            } else {
                emit(e.a);
                op(V2_SL, V2_MUL, depth - 1, Leaf{0, 0, 0}, c);
            }
        } else {
            int a = e.a, b = e.b, o = e.tag == EX_ADD ? V2_ADD : V2_MUL;
            if (e.tag == EX_ADD) {   // a + (-b') = a - b' ; (-a') + b = b - a'
                if (is_plain_neg(b)) b = pool.n[b].a, o = V2_SUB;
                else if (is_plain_neg(a)) {
                    const int t = pool.n[a].a;
                    a = b, b = t, o = V2_SUB;
                }
            }
            binary(o, a, b);
        }
        park(i);
    }
    void binary(int o, int a, int b) {
        const bool la = is_leaf(a), lb = is_leaf(b);
        const int rev = o == V2_SUB ? V2_RSUB : o;   // operands swapped
        if (la && lb) {
            op(V2_LL, o, depth, leaf_of(a), leaf_of(b));
            depth++;
        } else if (lb) {
            emit(a);
            op(V2_SL, o, depth - 1, Leaf{0, 0, 0}, leaf_of(b));
        } else if (la) {
            emit(b);
            op(V2_SL, rev, depth - 1, Leaf{0, 0, 0}, leaf_of(a));
        } else {
            const int na = label_of(a), nb = label_of(b);
            const bool a_first = na >= nb;
            const int first = a_first ? a : b, second = a_first ? b : a;
            emit(first);
            if (depth + std::max(1, label_of(second)) > kV2Regs) {
                // not enough registers for the other side: park this one in a spill slot and use it as a leaf
                if (spill_used >= kV2LdsSpills) {
                    prog.ok = false;
                    return;
                }
                const int sl = kV2LdsCse0 + cse_slots + spill_used++;
                max_lds = std::max(max_lds, sl);
                op(V2_UN, V2_STORE, depth - 1, Leaf{BZH_EXPR_LDS, sl, 0});
                depth--;
                emit(second);
                op(V2_SL, a_first ? rev : o, depth - 1, Leaf{0, 0, 0}, Leaf{BZH_EXPR_LDS, sl, 0});
                spill_used--;
            } else {
                emit(second);
                op(V2_SS, a_first ? o : rev, depth - 2);
                depth--;
            }
        }
    }
    // shared subexpression bookkeeping for the current scope
    std::map<int, int> want;   // node -> LDS slot it is to be parked in after its first evaluation
    void park(int i) {
        auto it = want.find(i);
        if (it == want.end() || cse.count(i)) return;
        op(V2_UN, V2_STORE, depth - 1, Leaf{BZH_EXPR_LDS, it->second, 0});
        cse[i] = it->second;
        std::fill(label.begin(), label.end(), -1);   // nodes above it are cheaper to reach now
    }
    void count_uses(int i, std::map<int, int>& uses, std::map<int, int>& weight) {
        const ENode& e = pool.n[i];
        if (e.tag == EX_CONST || e.tag == EX_SYMBOL || e.tag == EX_QUERY || hoisted.count(i)) return;
        if (uses[i]++) return;
        int w = 1;
        if (e.a >= 0) {
            count_uses(e.a, uses, weight);
            w += weight.count(e.a) ? weight[e.a] : 0;
        }
        if (e.b >= 0) {
            count_uses(e.b, uses, weight);
            w += weight.count(e.b) ? weight[e.b] : 0;
        }
        weight[i] = w;
    }
    void open_scope(const std::vector<int>& roots) {
        cse.clear();
        want.clear();
        std::fill(label.begin(), label.end(), -1);
        std::map<int, int> uses, weight;
        for (int r : roots) count_uses(r, uses, weight);
        std::vector<std::pair<long, int>> cand;
        for (auto& kv : uses) {
            if (kv.second >= 2) cand.push_back({-(long)(kv.second - 1) * weight[kv.first], kv.first});
        }
        std::sort(cand.begin(), cand.end());
        for (size_t k = 0; k < cand.size() && k < (size_t)cse_slots; k++) {
            want[cand[k].second] = kV2LdsCse0 + (int)k;
            max_lds = std::max(max_lds, kV2LdsCse0 + (int)k);
        }
    }

    // the whole quotient: terms in protocol order, y = symbol SY_Y, result (times t_inv) in r0
    void quotient(const std::vector<int>& terms, int tinv_node) {
        struct Group {
            int s;                  // shared left factor (-1: none)
            std::vector<int> c;     // the other factors, or the whole terms
        };
        std::vector<Group> groups;
        for (int t : terms) {
            const ENode& e = pool.n[t];
            const int s = (e.tag == EX_MUL) ? e.a : -1;
            if (s >= 0 && !groups.empty() && groups.back().s == s) groups.back().c.push_back(e.b);
            else groups.push_back(Group{s, {s >= 0 ? e.b : t}});
        }
        const Leaf y{BZH_EXPR_CONST, const_index(SY_Y, nullptr), 0};
        const Leaf acc{BZH_EXPR_LDS, kV2LdsAcc, 0}, inner{BZH_EXPR_LDS, kV2LdsInner, 0};
        bool first_group = true;
        for (auto& g : groups) {
            std::vector<int> roots = g.c;
            if (g.s >= 0) roots.push_back(g.s);
            open_scope(roots);
            const size_t m = g.c.size();
            if (m > 64) prog.ok = false;   // y^m symbols are provided up to 64
            for (size_t j = 0; j < m; j++) {
                depth = 0;
                if (j == 0) {
                    emit(g.c[0]);
                } else if (label_of(g.c[j]) < kV2Regs) {
                    op(V2_LL, V2_MUL, 0, inner, y);          // r0 = IN y
                    depth = 1;
                    emit(g.c[j]);                            // r1 = C_j
                    op(V2_SS, V2_ADD, 0);
                    depth = 1;
                } else {
                    emit(g.c[j]);                            // r0 = C_j (needs every register)
                    op(V2_LL, V2_MUL, 1, inner, y);          // r1 = IN y
                    op(V2_SS, V2_ADD, 0);
                }
                if (j + 1 < m) op(V2_UN, V2_STORE, 0, inner);
            }
            // r0 = sum_j C_j y^(m-1-j); times the shared factor
            if (g.s >= 0) {
                if (is_leaf(g.s)) {
                    op(V2_SL, V2_MUL, 0, Leaf{0, 0, 0}, leaf_of(g.s));
                } else if (label_of(g.s) < kV2Regs) {
                    depth = 1;
                    emit(g.s);
                    op(V2_SS, V2_MUL, 0);
                } else {
                    op(V2_UN, V2_STORE, 0, inner);
                    depth = 0;
                    emit(g.s);
                    op(V2_SL, V2_MUL, 0, Leaf{0, 0, 0}, inner);
                }
            }
            if (!first_group) {                              // ACC = ACC y^m + r0
                const Leaf ym{BZH_EXPR_CONST, const_index(m == 1 ? SY_Y : SY_YPOW0 + (int)m, nullptr), 0};
                op(V2_LL, V2_MUL, 1, acc, ym);
                op(V2_SS, V2_ADD, 0);
            }
            op(V2_UN, V2_STORE, 0, acc);
            first_group = false;
        }
        cse.clear();
        want.clear();
        op(V2_SL, V2_MUL, 0, Leaf{0, 0, 0}, leaf_of(tinv_node));
    }
};

// column registry of one batched evaluation: (device pointer, elements between consecutive proofs; 0 = shared)
struct Cols {
    std::vector<const uint32_t*> ptr;
    std::vector<size_t> stride;
    std::map<uint64_t, int> index;
    int add(uint64_t key, const uint32_t* p, size_t s) {
        auto it = index.find(key);
        if (it != index.end()) return it->second;
        const int i = (int)ptr.size();
        index[key] = i;
        ptr.push_back(p);
        stride.push_back(s);
        return i;
    }
    int at(uint64_t key) const { return index.at(key); }
};
// registry keys
enum { K_ADV = 1, K_FIX, K_INST, K_SIGMA, K_IDENT, K_PZ, K_LA, K_LS, K_LZ, K_MISC };
enum { M_L0, M_LLAST, M_LBLIND, M_X, M_TINV, M_AC, M_SC, M_A, M_S, M_ACC, M_Q, M_R, M_F, M_H0 /* + i */ };
static inline uint64_t key(int kind, uint64_t i) { return ((uint64_t)kind << 32) | i; }

// device arena: grow-only blocks, reset at the start of every call
struct Arena {
    struct Block {
        char* p;
        size_t size, used;
    };
    std::vector<Block> blocks;
    int device = 0;
    size_t requested = 0;  // bytes handed out since the last reset
    // A call's allocation sequence is deterministic, so after the first call of a given shape the arena is ONE block
    // that every later call bumps through without touching hipMalloc (overflow blocks are merged at the next reset).
    void reset() {
        if (blocks.size() > 1) {
            const size_t want = requested + (requested >> 4) + ((size_t)1 << 20);
            release();
            Block nb;
            nb.size = want;
            nb.used = 0;
            if (hipMalloc((void**)&nb.p, nb.size) == hipSuccess) blocks.push_back(nb);
        }
        for (auto& b : blocks) b.used = 0;
        requested = 0;
    }
    void release() {
        for (auto& b : blocks) (void)hipFree(b.p);
        blocks.clear();
    }
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        requested += bytes;
        for (auto& b : blocks)
            if (b.size - b.used >= bytes) {
                void* r = b.p + b.used;
                b.used += bytes;
                return r;
            }
        Block nb;
        nb.size = std::max(bytes, (size_t)256 << 20);
        if (hipMalloc((void**)&nb.p, nb.size) != hipSuccess) return nullptr;
        nb.used = bytes;
        blocks.push_back(nb);
        return nb.p;
    }
};

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    uint32_t u32() {
        if (end - p < 4) {
            ok = false;
            return 0;
        }
        uint32_t v;
        memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    uint8_t u8() {
        if (end - p < 1) {
            ok = false;
            return 0;
        }
        return *p++;
    }
    const uint8_t* bytes(size_t n) {
        if ((size_t)(end - p) < n) {
            ok = false;
            return nullptr;
        }
        const uint8_t* r = p;
        p += n;
        return r;
    }
};

}  // namespace

}  // namespace bzh

// ---------------------------------------------------------------------------
// the proving key
// ---------------------------------------------------------------------------
struct bzh_pk {
    int curve = 0, field = 0, device = 0;
    unsigned k = 0, ek = 0;
    size_t n = 0, en = 0, ext = 0;
    int na = 0, nf = 0, ni = 0, degree = 0, bf = 0, chunk_len = 0, nsets = 0, nl = 0, npieces = 0;
    size_t usable = 0;
    uint64_t vk_repr[4] = {0};
    const bzh_bases* srs = nullptr;
    std::vector<uint64_t> srs_g0_u_w;           // G_0, U, W of `srs`, canonical affine, read back once (bzh_verify_batch checks its argument against these)
    const bzh_bases* srs_lagrange = nullptr;   // (g_lagrange | u | w): Params::commit_lagrange for the columns upstream commits in that basis
    std::vector<bzh::CNode> cx;
    std::vector<int> gates;
    std::vector<std::pair<int, int>> perm_columns;  // (kind tag CX_*, index)
    std::vector<std::pair<std::vector<int>, std::vector<int>>> lookups;
    std::vector<std::pair<int, int>> advice_queries, fixed_queries, instance_queries;
    uint64_t omega[4], eomega[4], zeta[4];  // Montgomery limbs for ntt_run
    uint32_t delta[8];                      // Montgomery
    // device, one allocation
    void* dev = nullptr;
    uint32_t *fixed = nullptr, *fixed_polys = nullptr, *fixed_cosets = nullptr, *sigma = nullptr, *ident = nullptr, *sigma_polys = nullptr,
             *sigma_cosets = nullptr, *l0 = nullptr, *l_last = nullptr, *l_blind = nullptr, *x_col = nullptr, *tinv_col = nullptr;
    std::map<uint64_t, bzh::Program> progs;
    // The quotient's VM v2 program: compiled on the host at bzh_pk_create (it depends on the circuit only, not on k or on
    // any witness); q_ok = false when the circuit does not fit VM v2 (the prover then folds through VM v1).
    bzh::Program2 qprog;
    bool q_ok = false;
    uint64_t q_hash = 0;
    // proof-independent subexpressions of the quotient (selector products ...): one VM v1 program each, evaluated once on the
    // extended coset into `hoist` at bzh_pk_create
    std::vector<bzh::Program> hoist_progs;
    uint32_t* hoist = nullptr;
    size_t hoist_cols = 0;
    // The same program as compiled code, launched instead of the interpreter:
    //   q_builtin: a kernel generated at build time and linked into libbzh2.so (the reference's two circuits), found by q_hash;
    //   q_module / q_fn: a code object the caller compiled from bzh_pk_quotient_source (any other circuit).
    // q_select: BZH_QUOTIENT_* -- which of the three runs.
    bzh_quotient_launch_fn q_builtin = nullptr;
    hipModule_t q_module = nullptr;
    hipFunction_t q_fn = nullptr;
    int q_select = BZH_QUOTIENT_INTERPRETER;
    // multiopen structure: rotation sets and the commitments grouped under each
    std::vector<std::vector<int>> rot_sets;
    std::vector<std::vector<uint64_t>> groups;
    // per-call workspaces: one grow-only arena per ctx that has used the key (several worker streams share ONE key)
    std::map<const bzh_ctx*, std::unique_ptr<bzh::Arena>> arenas;
    size_t rng_bytes = 0;
    // verifying key: commitments to the fixed and permutation polynomials (computed at the first verification)
    bool vk_ready = false;
    std::vector<uint64_t> fixed_commitments, sigma_commitments;  // affine canonical x || y
    // The key is immutable after bzh_pk_create except for caches filled on first use (programs, hoisted columns, the vk
    // commitments, G_0/U/W, the quotient module, the arena map): `mu` guards those in short sections.  Calls through different
    // ctxs run concurrently on one key; a call holds its ctx's mutex throughout (lock order: ctx->mu, then pk->mu).
    std::mutex mu;
    bzh::Arena& arena_for(const bzh_ctx* ctx, int dev) {
        std::lock_guard<std::mutex> lk(mu);
        auto& a = arenas[ctx];
        if (!a) {
            a.reset(new bzh::Arena());
            a->device = dev;
        }
        return *a;
    }
};

namespace bzh {
namespace {

static int cx_degree(const bzh_pk& pk, int i) {
    const CNode& e = pk.cx[i];
    switch (e.tag) {
        case CX_CONST: return 0;
        case CX_ADVICE:
        case CX_FIXED:
        case CX_INSTANCE: return 1;
        case CX_NEG:
        case CX_SCALE: return cx_degree(pk, e.a);
        case CX_ADD: return std::max(cx_degree(pk, e.a), cx_degree(pk, e.b));
        default: return cx_degree(pk, e.a) + cx_degree(pk, e.b);
    }
}
struct Query3 {
    int tag, col, rot;
    bool operator==(const Query3& o) const { return tag == o.tag && col == o.col && rot == o.rot; }
};
static void cx_queries(const bzh_pk& pk, int i, std::vector<Query3>& out) {
    const CNode& e = pk.cx[i];
    if (e.tag >= CX_ADVICE && e.tag <= CX_INSTANCE) {
        const Query3 q{e.tag, (int)e.col, e.rot};
        if (std::find(out.begin(), out.end(), q) == out.end()) out.push_back(q);
    } else if (e.tag == CX_NEG || e.tag == CX_SCALE) {
        cx_queries(pk, e.a, out);
    } else if (e.tag == CX_ADD || e.tag == CX_MUL) {
        cx_queries(pk, e.a, out);
        cx_queries(pk, e.b, out);
    }
}

template <class SF>
static int parse_expr(Reader& r, bzh_pk& pk, int depth = 0) {
    if (depth > 4096) {
        r.ok = false;
        return -1;
    }
    CNode nd;
    nd.tag = r.u8();
    if (!r.ok) return -1;
    switch (nd.tag) {
        case CX_CONST: {
            const uint8_t* b = r.bytes(32);
            if (!b) return -1;
            const Fe<SF> v = h_from_bytes<SF>(b);
            memcpy(nd.val, v.l, 32);
            break;
        }
        case CX_ADVICE:
        case CX_FIXED:
        case CX_INSTANCE:
            nd.col = r.u32();
            nd.rot = (int32_t)r.u32();
            if ((nd.tag == CX_ADVICE && nd.col >= (uint32_t)pk.na) || (nd.tag == CX_FIXED && nd.col >= (uint32_t)pk.nf) ||
                (nd.tag == CX_INSTANCE && nd.col >= (uint32_t)pk.ni))
                r.ok = false;
            break;
        case CX_NEG: nd.a = parse_expr<SF>(r, pk, depth + 1); break;
        case CX_ADD:
        case CX_MUL:
            nd.a = parse_expr<SF>(r, pk, depth + 1);
            nd.b = parse_expr<SF>(r, pk, depth + 1);
            break;
        case CX_SCALE: {
            nd.a = parse_expr<SF>(r, pk, depth + 1);
            const uint8_t* b = r.bytes(32);
            if (!b) return -1;
            const Fe<SF> v = h_from_bytes<SF>(b);
            memcpy(nd.val, v.l, 32);
            break;
        }
        default: r.ok = false;
    }
    if (!r.ok) return -1;
    pk.cx.push_back(nd);
    return (int)pk.cx.size() - 1;
}

// circuit expression -> evaluator expression over `reg` (columns looked up by (kind, index))
static int lower(const bzh_pk& pk, int i, EPool& ep, const Cols& reg, int rot_scale) {
    const CNode& e = pk.cx[i];
    switch (e.tag) {
        case CX_CONST: {
            ENode nd;
            nd.tag = EX_CONST;
            memcpy(nd.val, e.val, 32);
            return ep.push(nd);
        }
        case CX_ADVICE: return ep.query(reg.at(key(K_ADV, e.col)), e.rot * rot_scale);
        case CX_FIXED: return ep.query(reg.at(key(K_FIX, e.col)), e.rot * rot_scale);
        case CX_INSTANCE: return ep.query(reg.at(key(K_INST, e.col)), e.rot * rot_scale);
        case CX_NEG: return ep.neg(lower(pk, e.a, ep, reg, rot_scale));
        case CX_SCALE: {
            ENode nd;
            nd.tag = EX_SCALE;
            nd.a = lower(pk, e.a, ep, reg, rot_scale);
            memcpy(nd.val, e.val, 32);
            return ep.push(nd);
        }
        case CX_ADD: {
            const int a = lower(pk, e.a, ep, reg, rot_scale), b = lower(pk, e.b, ep, reg, rot_scale);
            return ep.add(a, b);
        }
        default: {
            const int a = lower(pk, e.a, ep, reg, rot_scale), b = lower(pk, e.b, ep, reg, rot_scale);
            return ep.mul(a, b);
        }
    }
}

#define PV_TRY(expr)           \
    do {                       \
        int rc__ = (expr);     \
        if (rc__) return rc__; \
    } while (0)

template <class C>
struct CurveScalar;
template <>
struct CurveScalar<VestaCurve> {
    using SF = FpParams;
};
template <>
struct CurveScalar<PallasCurve> {
    using SF = FqParams;
};

// ---------------------------------------------------------------------------
// keygen
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// the quotient program of a key (host): column registry, terms in protocol order, compilation, hoisted columns
// ---------------------------------------------------------------------------
// per-proof columns of one prove call; all null when only the program is wanted (compile_quotient, materialize_hoist)
struct QuotientPtrs {
    const uint32_t* adv = nullptr;    // na columns of en elements per proof
    const uint32_t* inst = nullptr;   // ni
    const uint32_t* z = nullptr;      // nsets + nl grand products
    std::vector<const uint32_t*> lk;  // per lookup: A' | S'
};
// The registry fixes the column INDEX every instruction of the compiled program refers to: it must be built by this one
// function, for the compile and for every launch.  stride = elements between consecutive proofs, 0 = shared (key-owned).
static void quotient_registry(const bzh_pk& pk, const QuotientPtrs& q, Cols& reg) {
    const size_t en = pk.en, m = pk.perm_columns.size();
    const int na = pk.na, nf = pk.nf, ni = pk.ni, nsets = pk.nsets, nl = pk.nl, nz = pk.nsets + pk.nl;
    auto at = [](const uint32_t* base, size_t elems) -> const uint32_t* { return base ? base + elems * 8 : nullptr; };
    for (int i = 0; i < na; i++) reg.add(key(K_ADV, i), at(q.adv, (size_t)i * en), (size_t)na * en);
    for (int i = 0; i < nf; i++) reg.add(key(K_FIX, i), at(pk.fixed_cosets, (size_t)i * en), 0);
    for (int i = 0; i < ni; i++) reg.add(key(K_INST, i), at(q.inst, (size_t)i * en), (size_t)ni * en);
    for (size_t j = 0; j < m; j++) reg.add(key(K_SIGMA, j), at(pk.sigma_cosets, j * en), 0);
    for (int i = 0; i < nsets; i++) reg.add(key(K_PZ, i), at(q.z, (size_t)i * en), (size_t)nz * en);
    for (int i = 0; i < nl; i++) {
        const uint32_t* c = (size_t)i < q.lk.size() ? q.lk[i] : nullptr;
        reg.add(key(K_LA, i), c, 2 * en);
        reg.add(key(K_LS, i), at(c, en), 2 * en);
        reg.add(key(K_LZ, i), at(q.z, (size_t)(nsets + i) * en), (size_t)nz * en);
    }
    reg.add(key(K_MISC, M_L0), pk.l0, 0);
    reg.add(key(K_MISC, M_LLAST), pk.l_last, 0);
    reg.add(key(K_MISC, M_LBLIND), pk.l_blind, 0);
    reg.add(key(K_MISC, M_X), pk.x_col, 0);
    reg.add(key(K_MISC, M_TINV), pk.tinv_col, 0);
}

// every term of the quotient's numerator in protocol order (gates, permutation argument, lookups); *tinv = the 1 / (X^n - 1) column
template <class SF>
static std::vector<int> quotient_terms(const bzh_pk& pk, const Cols& reg, EPool& ep, int* tinv) {
    const int e = (int)pk.ext, nsets = pk.nsets, nl = pk.nl, last_rot = -(pk.bf + 1);
    const size_t m = pk.perm_columns.size();
    auto Q = [&](uint64_t kk, int rot = 0) { return ep.query(reg.at(kk), rot); };
    auto col_q = [&](std::pair<int, int> col) {
        return Q(key(col.first == CX_ADVICE ? K_ADV : (col.first == CX_FIXED ? K_FIX : K_INST), col.second));
    };
    const Fe<SF> onef = fe_one<SF>();
    auto one = [&] { return ep.cnst(onef); };
    auto l0 = [&] { return Q(key(K_MISC, M_L0)); };
    auto l_last = [&] { return Q(key(K_MISC, M_LLAST)); };
    auto active = [&] { return ep.sub(one(), ep.add(l_last(), Q(key(K_MISC, M_LBLIND)))); };
    std::vector<int> terms;
    for (int g : pk.gates) terms.push_back(lower(pk, g, ep, reg, e));
    if (nsets) {
        terms.push_back(ep.mul(l0(), ep.sub(one(), Q(key(K_PZ, 0)))));
        const uint64_t zl = key(K_PZ, nsets - 1);
        terms.push_back(ep.mul(l_last(), ep.sub(ep.mul(Q(zl), Q(zl)), Q(zl))));
        for (int i = 1; i < nsets; i++) terms.push_back(ep.mul(l0(), ep.sub(Q(key(K_PZ, i)), Q(key(K_PZ, i - 1), last_rot * e))));
        for (int i = 0; i < nsets; i++) {
            const size_t c0 = (size_t)i * pk.chunk_len, c1 = std::min(m, c0 + pk.chunk_len);
            int left = Q(key(K_PZ, i), e), right = Q(key(K_PZ, i));
            for (size_t gj = c0; gj < c1; gj++) {
                left = ep.mul(left, ep.add(ep.add(col_q(pk.perm_columns[gj]), ep.mul(ep.sym(SY_BETA), Q(key(K_SIGMA, gj)))), ep.sym(SY_GAMMA)));
                const int cur = ep.mul(ep.sym(SY_BD0 + (int)gj), Q(key(K_MISC, M_X)));
                right = ep.mul(right, ep.add(ep.add(col_q(pk.perm_columns[gj]), cur), ep.sym(SY_GAMMA)));
            }
            terms.push_back(ep.mul(active(), ep.sub(left, right)));
        }
    }
    for (int i = 0; i < nl; i++) {
        auto z0 = [&] { return Q(key(K_LZ, i)); };
        auto a_p = [&] { return Q(key(K_LA, i)); };
        auto s_p = [&] { return Q(key(K_LS, i)); };
        auto comp = [&](const std::vector<int>& es) {
            std::vector<int> t;
            for (int x : es) t.push_back(lower(pk, x, ep, reg, e));
            return ep.horner(t, ep.sym(SY_THETA));
        };
        terms.push_back(ep.mul(l0(), ep.sub(one(), z0())));
        terms.push_back(ep.mul(l_last(), ep.sub(ep.mul(z0(), z0()), z0())));
        const int lhs = ep.mul(ep.mul(Q(key(K_LZ, i), e), ep.add(a_p(), ep.sym(SY_BETA))), ep.add(s_p(), ep.sym(SY_GAMMA)));
        const int rhs = ep.mul(ep.mul(z0(), ep.add(comp(pk.lookups[i].first), ep.sym(SY_BETA))),
                               ep.add(comp(pk.lookups[i].second), ep.sym(SY_GAMMA)));
        terms.push_back(ep.mul(active(), ep.sub(lhs, rhs)));
        terms.push_back(ep.mul(l0(), ep.sub(a_p(), s_p())));
        terms.push_back(ep.mul(ep.mul(active(), ep.sub(a_p(), s_p())), ep.sub(a_p(), Q(key(K_LA, i), -e))));
    }
    *tinv = Q(key(K_MISC, M_TINV));
    return terms;
}

// Compile the quotient for VM v2 (host only).  Hoisting: maximal subexpressions over proof-independent columns (stride 0:
// fixed / permutation / Lagrange columns of the key) and literal constants that contain a multiplication -- the
// compressed-selector products q prod (j - q) of every gate -- get one VM v1 program each (pk.hoist_progs) and are referred
// to by the main program as extra registry columns; materialize_hoist evaluates them once on the extended coset.
template <class SF>
static void compile_quotient(bzh_pk& pk) {
    pk.q_ok = false;
    pk.hoist_progs.clear();
    pk.hoist_cols = 0;
    if (pk.en % 128) return;   // VM v2 runs whole 128-row workgroups (tiny test domains take the plain fold)
    Cols reg;
    quotient_registry(pk, QuotientPtrs{}, reg);
    EPool ep;
    int tinv = -1;
    const std::vector<int> terms = quotient_terms<SF>(pk, reg, ep, &tinv);
    Compiler2 cc(ep);
    if (!getenv("BZH_NO_HOIST")) {
        const size_t nn = ep.n.size();
        std::vector<char> indep(nn, 0);
        std::vector<int> muls(nn, 0);
        for (size_t i = 0; i < nn; i++) {   // children precede parents in the pool
            const ENode& e = ep.n[i];
            if (e.tag == EX_CONST) indep[i] = 1;
            else if (e.tag == EX_SYMBOL) indep[i] = 0;
            else if (e.tag == EX_QUERY) indep[i] = reg.stride[e.col] == 0;
            else if (e.tag == EX_NEG) indep[i] = indep[e.a], muls[i] = muls[e.a];
            else if (e.tag == EX_SCALE) indep[i] = indep[e.a], muls[i] = muls[e.a] + 1;
            else indep[i] = indep[e.a] && indep[e.b], muls[i] = muls[e.a] + muls[e.b] + (e.tag == EX_MUL);
        }
        std::vector<int> picked;
        std::vector<char> seen(nn, 0);
        std::vector<int> stack(terms.begin(), terms.end());
        while (!stack.empty()) {
            const int i = stack.back();
            stack.pop_back();
            if (seen[i]) continue;
            seen[i] = 1;
            const ENode& e = ep.n[i];
            if (e.tag == EX_CONST || e.tag == EX_SYMBOL || e.tag == EX_QUERY) continue;
            if (indep[i] && muls[i] >= 1) {
                picked.push_back(i);
                continue;
            }
            if (e.a >= 0) stack.push_back(e.a);
            if (e.b >= 0) stack.push_back(e.b);
        }
        std::sort(picked.begin(), picked.end());
        if (!picked.empty() && picked.size() <= 512) {
            const size_t ncols = reg.ptr.size();
            for (size_t hi = 0; hi < picked.size(); hi++) {
                Compiler c1(ep);
                c1.prog.result_slot = c1.emit(picked[hi]);
                if (c1.overflow) {
                    pk.hoist_progs.clear();
                    return;   // does not fit the evaluators' slot file: the prover folds through VM v1
                }
                pk.hoist_progs.push_back(std::move(c1.prog));
                cc.hoisted[picked[hi]] = (int)(ncols + hi);
            }
        }
    }
    cc.quotient(terms, tinv);
    cc.prog.nlds = cc.nlds();
    if (getenv("BZH_PROVE_TRACE")) {
        size_t muls = 0;
        for (auto& o : cc.prog.ops) muls += ((o.code >> 4) < 3 && ((o.code >> 2) & 3) == V2_MUL);
        fprintf(stderr, "[bzh_pk_create] quotient program (VM v2): %zu terms, %zu ops, %zu multiplications, %d LDS slots, %zu constants, %zu hoisted columns%s\n",
                terms.size(), cc.prog.ops.size(), muls, cc.prog.nlds, cc.prog.consts.size(), pk.hoist_progs.size(), cc.prog.ok ? "" : " -- NOT usable");
        // instruction mix: form (SS/SL/LL/UN) x operation, and the kinds of the memory operands
        size_t hist[4][4] = {{0}}, kinds[4] = {0};
        for (auto& o : cc.prog.ops) {
            const int form = o.code >> 4, oo = (o.code >> 2) & 3;
            hist[form & 3][oo]++;
            if (form == V2_SL || form == V2_LL) kinds[o.b_kind & 3]++;
            if (form == V2_LL || (form == V2_UN && oo != V2_NEG)) kinds[o.a_kind & 3]++;
        }
        fprintf(stderr, "[bzh_pk_create]   mix  SS add/sub/mul/rsub %zu/%zu/%zu/%zu  SL %zu/%zu/%zu/%zu  LL %zu/%zu/%zu/%zu  UN neg/load/store %zu/%zu/%zu ; operands column/const/lds %zu/%zu/%zu\n",
                hist[0][0], hist[0][1], hist[0][2], hist[0][3], hist[1][0], hist[1][1], hist[1][2], hist[1][3], hist[2][0], hist[2][1],
                hist[2][2], hist[2][3], hist[3][0], hist[3][1], hist[3][2], kinds[BZH_EXPR_COLUMN], kinds[BZH_EXPR_CONST], kinds[BZH_EXPR_LDS]);
    }
    pk.qprog = std::move(cc.prog);
    pk.q_ok = pk.qprog.ok;
    pk.q_hash = program2_hash(pk.qprog, pk.field);
    pk.hoist_cols = pk.hoist_progs.size();
    if (!pk.q_ok) {
        pk.hoist_progs.clear();
        pk.hoist_cols = 0;
    }
}

// evaluate the hoisted columns on the extended coset (device; once per key, at bzh_pk_create)
template <class SF>
static int materialize_hoist(bzh_ctx* ctx, bzh_pk& pk) {
    if (!pk.q_ok || pk.hoist_progs.empty()) return BZH_OK;
    const size_t size = pk.en;
    Cols reg;
    quotient_registry(pk, QuotientPtrs{}, reg);   // hoisted programs read key-owned columns only
    const size_t ncols = reg.ptr.size();
    BZH_HIP_TRY(ctx, hipMalloc((void**)&pk.hoist, pk.hoist_cols * size * 32));
    size_t stage_bytes = 0;
    for (const Program& pg : pk.hoist_progs)
        stage_bytes = std::max(stage_bytes, std::max<size_t>(pg.consts.size(), 1) * 32 + pg.ops.size() * sizeof(bzh_expr_op) + ncols * 16 + 1024);
    char* stage_all = nullptr;
    BZH_HIP_TRY(ctx, hipMalloc((void**)&stage_all, stage_bytes * pk.hoist_progs.size()));
    int rc = BZH_OK;
    for (size_t hi = 0; hi < pk.hoist_progs.size() && !rc; hi++) {
        const Program& pg = pk.hoist_progs[hi];
        std::vector<uint32_t> cv(std::max<size_t>(pg.consts.size(), 1) * 8);
        for (size_t i = 0; i < pg.consts.size(); i++) memcpy(&cv[i * 8], pg.consts[i].val, 32);
        char* stage = stage_all + hi * stage_bytes;
        uint32_t* d_consts = (uint32_t*)stage;
        char* d_prog = stage + ((cv.size() * 4 + 255) & ~(size_t)255);
        char* d_ptrs = d_prog + ((pg.ops.size() * sizeof(bzh_expr_op) + 255) & ~(size_t)255);
        char* d_strides = d_ptrs + ((ncols * 8 + 255) & ~(size_t)255);
        if ((rc = h2d_small(ctx, d_consts, cv.data(), cv.size() * 4))) break;
        if ((rc = h2d_small(ctx, d_prog, pg.ops.data(), pg.ops.size() * sizeof(bzh_expr_op)))) break;
        if ((rc = h2d_small(ctx, d_ptrs, reg.ptr.data(), ncols * 8))) break;
        if ((rc = h2d_small(ctx, d_strides, reg.stride.data(), ncols * 8))) break;
        int nslots = pg.result_slot + 1;
        for (auto& o : pg.ops) nslots = std::max(nslots, (int)o.dst + 1);
        rc = expr_eval(ctx, pk.field, d_prog, (int)pg.ops.size(), (const uint32_t* const*)d_ptrs, (const size_t*)d_strides, d_consts, 0, size,
                       pg.result_slot, 1, nslots, pk.hoist + hi * size * 8);
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(stage_all);
    return rc;
}

// what the host half of keygen hands to the device half
template <class SF>
struct ParsedKey {
    std::vector<Fe<SF>> fixed_h;                 // nf x n fixed assignment, Montgomery
    std::vector<uint32_t> map_c, map_r;          // permutation: (column, row) -> (column, row)
    Fe<SF> omega, eomega, delta, zeta;
};

// keygen, host half: parse the circuit blob, derive the constraint-system shape (queries, degree, blinding factors,
// extended domain), the permutation cycles and the multiopen structure, and compile the quotient program.  No device work:
// this is also what the build-time kernel generator runs (bzh_quotient_source_for_circuit).
template <class C>
static int pk_parse_t(const uint8_t* blob, size_t len, bzh_pk& pk, ParsedKey<typename CurveScalar<C>::SF>& po) {
    using SF = typename CurveScalar<C>::SF;
    using FM = FieldMeta<SF>;
    Reader r{blob, blob + len};
    const uint32_t magic = r.u32();
    if (magic != 0x31435A42u && magic != 0x32435A42u) return BZH_E_ARG;  // "BZC1" / "BZC2"
    const bool explicit_queries = magic == 0x32435A42u;
    pk.curve = C::id;
    pk.field = FM::id;
    pk.k = r.u32();
    pk.na = (int)r.u32();
    pk.nf = (int)r.u32();
    pk.ni = (int)r.u32();
    const int min_degree = (int)r.u32();
    const uint8_t* vk = r.bytes(32);
    if (!r.ok || pk.k < 1 || pk.k > 24 || pk.na > 4096 || pk.nf > 4096 || pk.ni > 4096) return BZH_E_ARG;
    memcpy(pk.vk_repr, vk, 32);
    pk.n = (size_t)1 << pk.k;
    const uint32_t ngates = r.u32();
    for (uint32_t g = 0; g < ngates && r.ok; g++) pk.gates.push_back(parse_expr<SF>(r, pk));
    const uint32_t nperm = r.u32();
    for (uint32_t j = 0; j < nperm && r.ok; j++) {
        const int kind = r.u8() + CX_ADVICE;
        const int idx = (int)r.u32();
        if (kind > CX_INSTANCE || idx < 0 || idx >= (kind == CX_ADVICE ? pk.na : (kind == CX_FIXED ? pk.nf : pk.ni))) return BZH_E_ARG;
        pk.perm_columns.push_back({kind, idx});
    }
    const uint32_t nlk = r.u32();
    for (uint32_t l = 0; l < nlk && r.ok; l++) {
        const uint32_t m = r.u32();
        if (!m || m > 64) return BZH_E_ARG;
        std::vector<int> ins, tabs;
        for (uint32_t i = 0; i < m && r.ok; i++) ins.push_back(parse_expr<SF>(r, pk));
        for (uint32_t i = 0; i < m && r.ok; i++) tabs.push_back(parse_expr<SF>(r, pk));
        pk.lookups.push_back({ins, tabs});
    }
    const uint32_t ncopies = r.u32();
    struct Copy {
        uint32_t lc, lr, rc, rr;
    };
    std::vector<Copy> copies;
    for (uint32_t i = 0; i < ncopies && r.ok; i++) {
        Copy c{r.u32(), r.u32(), r.u32(), r.u32()};
        if (c.lc >= nperm || c.rc >= nperm || c.lr >= pk.n || c.rr >= pk.n) return BZH_E_ARG;
        copies.push_back(c);
    }
    if (!r.ok) return BZH_E_ARG;
    const size_t n = pk.n;
    std::vector<Fe<SF>>& fixed_h = po.fixed_h;
    fixed_h.assign((size_t)pk.nf * n, fe_zero<SF>());
    for (int f = 0; f < pk.nf; f++) {
        const uint32_t fl = r.u32();
        if (!r.ok || fl > n) return BZH_E_ARG;
        const uint8_t* b = r.bytes((size_t)fl * 32);
        if (!b) return BZH_E_ARG;
        for (uint32_t i = 0; i < fl; i++) fixed_h[(size_t)f * n + i] = h_from_bytes<SF>(b + 32 * (size_t)i);
    }
    if (!r.ok) return BZH_E_ARG;

    // shape: queries, degree, blinding factors (upstream ConstraintSystem).  "BZC2" carries the query lists in
    // upstream's registration order (a query is registered when it is made: `enable_equality` registers the column's
    // current-row query at once, before any gate of the reference's configure functions -- src/chips/board.rs:199,217
    // before :275); "BZC1" derives them in first-use order: gates, lookups, then the permutation columns.
    std::vector<Query3> used, qs;
    for (int g : pk.gates) cx_queries(pk, g, used);
    for (auto& lk : pk.lookups) {
        for (int e : lk.first) cx_queries(pk, e, used);
        for (int e : lk.second) cx_queries(pk, e, used);
    }
    for (auto& pc : pk.perm_columns) {
        const Query3 q{pc.first, pc.second, 0};
        if (std::find(used.begin(), used.end(), q) == used.end()) used.push_back(q);
    }
    if (explicit_queries) {
        const int tags[3] = {CX_ADVICE, CX_FIXED, CX_INSTANCE};
        const int limits[3] = {pk.na, pk.nf, pk.ni};
        for (int t = 0; t < 3; t++) {
            const uint32_t nq = r.u32();
            if (!r.ok || nq > 65536) return BZH_E_ARG;
            for (uint32_t i = 0; i < nq && r.ok; i++) {
                const Query3 q{tags[t], (int)r.u32(), (int)r.u32()};
                if (q.col < 0 || q.col >= limits[t] || q.rot < -(int)n || q.rot > (int)n) return BZH_E_ARG;
                if (std::find(qs.begin(), qs.end(), q) != qs.end()) return BZH_E_ARG;
                qs.push_back(q);
            }
        }
        if (!r.ok) return BZH_E_ARG;
        for (auto& q : used) {   // every cell the constraint system reads must be in the lists
            if (std::find(qs.begin(), qs.end(), q) == qs.end()) return BZH_E_ARG;
        }
    } else {
        qs = used;
    }
    std::map<int, int> per_col;
    for (auto& q : qs) {
        if (q.tag == CX_ADVICE) {
            pk.advice_queries.push_back({q.col, q.rot});
            per_col[q.col]++;
        } else if (q.tag == CX_FIXED) {
            pk.fixed_queries.push_back({q.col, q.rot});
        } else {
            pk.instance_queries.push_back({q.col, q.rot});
        }
    }
    int deg = 3;
    for (int g : pk.gates) deg = std::max(deg, cx_degree(pk, g));
    for (auto& lk : pk.lookups) {
        int di = 1, dt = 1;
        for (int e : lk.first) di = std::max(di, cx_degree(pk, e));
        for (int e : lk.second) dt = std::max(dt, cx_degree(pk, e));
        deg = std::max(deg, std::max(4, 2 + di + dt));
    }
    pk.degree = std::max(deg, min_degree);
    int maxq = 1;
    for (auto& kv : per_col) maxq = std::max(maxq, kv.second);
    pk.bf = std::max(3, maxq) + 2;
    if ((size_t)pk.bf + 2 > n) return BZH_E_ARG;
    pk.usable = n - (size_t)(pk.bf + 1);
    pk.chunk_len = pk.degree - 2;
    unsigned bl = 0;
    for (int v = pk.degree - 2; v; v >>= 1) bl++;
    pk.ek = pk.k + std::max(1u, bl);
    if (pk.ek > FM::S) return BZH_E_RANGE;
    pk.en = (size_t)1 << pk.ek;
    pk.ext = pk.en / n;
    pk.nl = (int)pk.lookups.size();
    pk.nsets = nperm ? (int)((nperm + pk.chunk_len - 1) / pk.chunk_len) : 0;
    pk.npieces = pk.degree - 1;
    if ((size_t)pk.npieces * n > pk.en) return BZH_E_ARG;

    // domain constants
    uint32_t e[8];
    {  // (p - 1) >> S
        uint32_t pm1[8];
        for (int i = 0; i < 8; i++) pm1[i] = SF::mod(i);
        pm1[0] -= 1;  // p is odd
        for (int i = 0; i < 8; i++) {
            const unsigned s = FM::S, src = i + s / 32;
            const uint64_t lo = src < 8 ? pm1[src] : 0, hi = src + 1 < 8 ? pm1[src + 1] : 0;
            e[i] = (s % 32) ? (uint32_t)(((lo | (hi << 32)) >> (s % 32)) & 0xffffffffu) : (uint32_t)lo;
        }
    }
    const Fe<SF> gen = fe_from_u32<SF>(FM::gen);
    const Fe<SF> root = fe_pow(gen, e);
    auto pow2 = [](Fe<SF> v, unsigned times) {
        for (unsigned i = 0; i < times; i++) v = fe_sqr(v);
        return v;
    };
    const Fe<SF> omega = pow2(root, FM::S - pk.k), eomega = pow2(root, FM::S - pk.ek);
    const Fe<SF> delta = pow2(gen, FM::S);
    Fe<SF> zeta;
    {  // g^((p-1)/3)
        uint32_t q[8];
        uint64_t rem = 0;
        uint32_t pm1[8];
        for (int i = 0; i < 8; i++) pm1[i] = SF::mod(i);
        pm1[0] -= 1;
        for (int i = 7; i >= 0; i--) {
            const uint64_t cur = (rem << 32) | pm1[i];
            q[i] = (uint32_t)(cur / 3);
            rem = cur % 3;
        }
        if (rem) return BZH_E_RANGE;  // no cube root of unity: the coset fast path needs 3 | p - 1
        zeta = fe_pow(gen, q);
    }
    h_store<SF>(pk.omega, omega);
    h_store<SF>(pk.eomega, eomega);
    h_store<SF>(pk.zeta, zeta);
    memcpy(pk.delta, delta.l, 32);
    po.omega = omega, po.eomega = eomega, po.delta = delta, po.zeta = zeta;

    // permutation cycles (upstream permutation::keygen::Assembly::copy)
    const size_t m = nperm;
    std::vector<uint32_t>& map_c = po.map_c;
    std::vector<uint32_t>& map_r = po.map_r;
    map_c.resize(m * n), map_r.resize(m * n);
    std::vector<uint32_t> aux_c(m * n), aux_r(m * n), sizes(m * n, 1);
    for (size_t c = 0; c < m; c++)
        for (size_t rr = 0; rr < n; rr++) {
            map_c[c * n + rr] = aux_c[c * n + rr] = (uint32_t)c;
            map_r[c * n + rr] = aux_r[c * n + rr] = (uint32_t)rr;
        }
    for (auto& cp : copies) {
        size_t li = cp.lc * n + cp.lr, ri = cp.rc * n + cp.rr;
        uint32_t lc = aux_c[li], lr = aux_r[li], rc = aux_c[ri], rr = aux_r[ri];
        if (lc == rc && lr == rr) continue;
        if (sizes[lc * n + lr] < sizes[rc * n + rr]) {
            std::swap(lc, rc);
            std::swap(lr, rr);
        }
        sizes[lc * n + lr] += sizes[rc * n + rr];
        uint32_t ic = rc, ir = rr;
        do {
            const size_t ii = ic * n + ir;
            aux_c[ii] = lc;
            aux_r[ii] = lr;
            const uint32_t nc = map_c[ii], nr = map_r[ii];
            ic = nc;
            ir = nr;
        } while (!(ic == rc && ir == rr));
        std::swap(map_c[li], map_c[ri]);
        std::swap(map_r[li], map_r[ri]);
    }

    // multiopen structure (rotations stand in for the points: distinct rotations <-> distinct points x * omega^r)
    {
        struct Q {
            uint64_t cid;
            int rot;
        };
        std::vector<Q> q;
        const int last_rot = -(pk.bf + 1);
        for (auto& a : pk.instance_queries) q.push_back({key(K_INST, a.first), a.second});
        for (auto& a : pk.advice_queries) q.push_back({key(K_ADV, a.first), a.second});
        for (int i = 0; i < pk.nsets; i++) {
            q.push_back({key(K_PZ, i), 0});
            q.push_back({key(K_PZ, i), 1});
            if (i != pk.nsets - 1) q.push_back({key(K_PZ, i), last_rot});
        }
        for (int i = 0; i < pk.nl; i++) {
            q.push_back({key(K_LZ, i), 0});
            q.push_back({key(K_LA, i), 0});
            q.push_back({key(K_LS, i), 0});
            q.push_back({key(K_LA, i), -1});
            q.push_back({key(K_LZ, i), 1});
        }
        for (auto& a : pk.fixed_queries) q.push_back({key(K_FIX, a.first), a.second});
        for (size_t j = 0; j < m; j++) q.push_back({key(K_SIGMA, j), 0});
        q.push_back({key(K_MISC, M_H0), 0});
        q.push_back({key(K_MISC, M_F), 0});  // the random polynomial
        std::vector<uint64_t> order;
        std::map<uint64_t, std::vector<int>> pts_of;
        for (auto& e2 : q) {
            auto it = pts_of.find(e2.cid);
            if (it == pts_of.end()) {
                order.push_back(e2.cid);
                it = pts_of.insert({e2.cid, {}}).first;
            }
            if (std::find(it->second.begin(), it->second.end(), e2.rot) == it->second.end()) it->second.push_back(e2.rot);
        }
        for (uint64_t cid : order) {
            std::vector<int> ks = pts_of[cid];
            std::sort(ks.begin(), ks.end());
            size_t si = 0;
            for (; si < pk.rot_sets.size(); si++)
                if (pk.rot_sets[si] == ks) break;
            if (si == pk.rot_sets.size()) {
                pk.rot_sets.push_back(ks);
                pk.groups.push_back({});
            }
            pk.groups[si].push_back(cid);
        }
    }
    // randomness per proof: blinding rows and blinds in create_proof's draw order, then the IPA opening
    {
        const size_t bf1 = (size_t)pk.bf + 1;
        size_t draws = (size_t)pk.na * bf1 + pk.na;
        draws += (size_t)pk.nl * (2 * bf1 + 2);
        draws += (size_t)(pk.nsets + pk.nl) * ((size_t)pk.bf + 1);
        draws += n + 1;                 // random polynomial + its blind
        draws += (size_t)pk.npieces;    // h pieces
        draws += 1;                     // f blind
        draws += n + 1 + 2 * (size_t)pk.k;
        pk.rng_bytes = draws * 64;
    }
    compile_quotient<SF>(pk);
    return BZH_OK;
}

// keygen, device half: fixed / permutation / identity polynomials in Lagrange, coefficient and extended-coset form,
// l_0 / l_last / l_blind, X and 1 / (X^n - 1) on the extended coset, the hoisted columns of the quotient program.
template <class C>
static int pk_create_t(bzh_ctx* ctx, const bzh_bases* srs, const uint8_t* blob, size_t len, bzh_pk** out) {
    using SF = typename CurveScalar<C>::SF;
    std::unique_ptr<bzh_pk> pkp(new bzh_pk());
    bzh_pk& pk = *pkp;
    ParsedKey<SF> po;
    PV_TRY(pk_parse_t<C>(blob, len, pk, po));
    pk.device = ctx->device;
    pk.srs = srs;
    if (srs->n != pk.n + 2 || srs->curve != C::id) return BZH_E_ARG;
    const size_t n = pk.n, m = pk.perm_columns.size();
    const std::vector<Fe<SF>>& fixed_h = po.fixed_h;
    const std::vector<uint32_t>&map_c = po.map_c, &map_r = po.map_r;
    const Fe<SF> omega = po.omega, eomega = po.eomega, delta = po.delta, zeta = po.zeta;
    // device allocation: fixed / sigma / ident columns in the three forms, l0 / l_last / l_blind, X and 1/(X^n - 1)
    const size_t en = pk.en, nf = pk.nf;
    const size_t words = (2 * nf * n + nf * en + 3 * m * n + m * en + 3 * en + 2 * en + 3 * n) * 8;
    BZH_HIP_TRY(ctx, hipMalloc(&pk.dev, words * 4 + 256));
    uint32_t* cur = (uint32_t*)pk.dev;
    auto take = [&](size_t elems) {
        uint32_t* p = cur;
        cur += elems * 8;
        return p;
    };
    pk.fixed = take(nf * n);
    pk.fixed_polys = take(nf * n);
    pk.fixed_cosets = take(nf * en);
    pk.sigma = take(m * n);
    pk.ident = take(m * n);
    pk.sigma_polys = take(m * n);
    pk.sigma_cosets = take(m * en);
    pk.l0 = take(en);
    pk.l_last = take(en);
    pk.l_blind = take(en);
    pk.x_col = take(en);
    pk.tinv_col = take(en);
    uint32_t* l_tmp = take(3 * n);
    hipStream_t st = ctx->stream;
    std::vector<Fe<SF>> wp(n), host(std::max(std::max(m * n, en), 3 * n));
    wp[0] = fe_one<SF>();
    for (size_t i = 1; i < n; i++) wp[i] = fe_mul(wp[i - 1], omega);
    std::vector<Fe<SF>> dpow(m ? m : 1);
    dpow[0] = fe_one<SF>();
    for (size_t j = 1; j < m; j++) dpow[j] = fe_mul(dpow[j - 1], delta);
    auto up = [&](uint32_t* dst, const Fe<SF>* src, size_t elems) -> int {
        if (!elems) return BZH_OK;
        BZH_HIP_TRY(ctx, hipMemcpyAsync(dst, src, elems * 32, hipMemcpyHostToDevice, st));
        BZH_HIP_TRY(ctx, hipStreamSynchronize(st));
        return BZH_OK;
    };
    auto to_coeff = [&](uint32_t* dst, const uint32_t* src, size_t count) -> int {
        if (!count) return BZH_OK;
        BZH_HIP_TRY(ctx, hipMemcpyAsync(dst, src, count * n * 32, hipMemcpyDeviceToDevice, st));
        return ntt_run(ctx, pk.field, dst, pk.k, count, pk.omega, nullptr, 1, BZH_FORM_MONTGOMERY);
    };
    auto to_extended = [&](uint32_t* dst, const uint32_t* polys, size_t count) -> int {
        if (!count) return BZH_OK;
        return ntt_run_padded(ctx, pk.field, dst, polys, pk.k, pk.ek, count, pk.eomega, pk.zeta);
    };
    PV_TRY(up(pk.fixed, fixed_h.data(), nf * n));
    PV_TRY(to_coeff(pk.fixed_polys, pk.fixed, nf));
    PV_TRY(to_extended(pk.fixed_cosets, pk.fixed_polys, nf));
    for (size_t j = 0; j < m; j++)
        for (size_t rr = 0; rr < n; rr++) host[j * n + rr] = fe_mul(dpow[j], wp[rr]);
    PV_TRY(up(pk.ident, host.data(), m * n));
    for (size_t j = 0; j < m; j++)
        for (size_t rr = 0; rr < n; rr++) host[j * n + rr] = fe_mul(dpow[map_c[j * n + rr]], wp[map_r[j * n + rr]]);
    PV_TRY(up(pk.sigma, host.data(), m * n));
    PV_TRY(to_coeff(pk.sigma_polys, pk.sigma, m));
    PV_TRY(to_extended(pk.sigma_cosets, pk.sigma_polys, m));
    for (size_t i = 0; i < 3 * n; i++) host[i] = fe_zero<SF>();
    host[0] = fe_one<SF>();
    host[n + pk.usable] = fe_one<SF>();
    for (size_t i = pk.usable + 1; i < n; i++) host[2 * n + i] = fe_one<SF>();
    PV_TRY(up(l_tmp, host.data(), 3 * n));
    PV_TRY(ntt_run(ctx, pk.field, l_tmp, pk.k, 3, pk.omega, nullptr, 1, BZH_FORM_MONTGOMERY));
    PV_TRY(to_extended(pk.l0, l_tmp, 3));  // l0, l_last, l_blind are consecutive
    {
        Fe<SF> x = zeta;
        for (size_t i = 0; i < en; i++) {
            host[i] = x;
            x = fe_mul(x, eomega);
        }
        PV_TRY(up(pk.x_col, host.data(), en));
        std::vector<Fe<SF>> tinv(pk.ext);
        for (size_t i = 0; i < pk.ext; i++) tinv[i] = fe_inv(fe_sub(h_pow_u64(host[i], n), fe_one<SF>()));
        for (size_t i = 0; i < en; i++) host[i] = tinv[i % pk.ext];
        PV_TRY(up(pk.tinv_col, host.data(), en));
    }

    PV_TRY(materialize_hoist<SF>(ctx, pk));
    pk.q_builtin = nullptr;
    if (pk.q_ok) {
        size_t nb = 0;
        const bzh_builtin_quotient* tab = bzh_builtin_quotients ? bzh_builtin_quotients(&nb) : nullptr;
        for (size_t i = 0; i < nb; i++)
            if (tab[i].program_hash == pk.q_hash) pk.q_builtin = tab[i].launch;
    }
    const char* qenv = getenv("BZH_QUOTIENT");
    pk.q_select = (pk.q_builtin && !(qenv && !strcmp(qenv, "interp"))) ? BZH_QUOTIENT_BUILTIN : BZH_QUOTIENT_INTERPRETER;
    BZH_HIP_TRY(ctx, hipStreamSynchronize(st));
    *out = pkp.release();
    return BZH_OK;
}

// ---------------------------------------------------------------------------
// the lockstep prover
// ---------------------------------------------------------------------------
template <class C>
struct Prover {
    using SF = typename CurveScalar<C>::SF;
    using PB = typename C::Base;
    bzh_ctx* ctx;
    bzh_pk& pk;
    const size_t B;
    hipStream_t st;
    const size_t n, en, usable;
    const int field;
    std::vector<bzh_transcript*> T;
    std::vector<const uint8_t*> rng;  // per-proof cursor into the caller's randomness
    // seeded mode (bzh_prove_batch_seeded): the stream of proof b is ChaCha20(seed_b), addressed by 64-byte block; every
    // proof of a batch draws in lockstep, so one counter serves the batch
    bool seeded = false;
    std::vector<uint32_t> seed_keys;   // B x 8 words
    uint32_t* d_seed_keys = nullptr;
    uint64_t seed_ctr = 0;
    std::vector<uint64_t> host_ctr;    // draws taken on the host per proof since the last row draw (must stay in lockstep)
    std::vector<std::map<int, Fe<SF>>> env;
    Arena& arena;   // this ctx's workspace of the (shared) key

    Prover(bzh_ctx* c, bzh_pk& p, size_t batch, Arena& ar)
        : ctx(c), pk(p), B(batch), st(c->stream), n(p.n), en(p.en), usable(p.usable), field(p.field), T(batch, nullptr), rng(batch),
          env(batch), arena(ar) {}
    ~Prover() {
        for (auto t : T)
            if (t) bzh_transcript_free(t);
    }

    // BZH_PROVE_TRACE=1: phase wall times on stderr, with a device sync at every phase boundary
    const bool trace = getenv("BZH_PROVE_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t_last = std::chrono::steady_clock::now();
    void mark(const char* name) {
        if (!trace) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[bzh_prove_batch] %-22s %8.2f ms\n", name, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    }

    uint32_t* dalloc(size_t elems) { return (uint32_t*)arena.alloc(elems * 32); }
    int zero(uint32_t* p, size_t elems) {
        BZH_HIP_TRY(ctx, hipMemsetAsync(p, 0, elems * 32, st));
        return BZH_OK;
    }
    // strided device copy of `rows` rows of `width` elements
    int copy2d(uint32_t* dst, size_t dpitch, const uint32_t* src, size_t spitch, size_t width, size_t rows) {
        if (!rows || !width) return BZH_OK;
        BZH_HIP_TRY(ctx, hipMemcpy2DAsync(dst, dpitch * 32, src, spitch * 32, width * 32, rows, hipMemcpyDeviceToDevice, st));
        return BZH_OK;
    }
    // host Montgomery elements -> device
    int upload(uint32_t* dst, const Fe<SF>* src, size_t elems) { return h2d_small(ctx, dst, src, elems * 32); }

    Fe<SF> draw(size_t b) {
        if (seeded) {
            uint32_t blk[16];
            chacha20_block(&seed_keys[b * 8], seed_ctr + host_ctr[b]++, blk);
            return h_from_u512<SF>(reinterpret_cast<const uint8_t*>(blk));
        }
        const Fe<SF> v = h_from_u512<SF>(rng[b]);
        rng[b] += 64;
        return v;
    }
    // seeded mode: fold the host-side draws into the batch counter (every proof must have taken the same number)
    int seed_sync() {
        for (size_t b = 1; b < B; b++)
            if (host_ctr[b] != host_ctr[0]) return BZH_E_ARG;
        seed_ctr += host_ctr[0];
        std::fill(host_ctr.begin(), host_ctr.end(), 0);
        return BZH_OK;
    }
    // seeded mode: the next `count` 64-byte draws of every proof, generated on the device (B x count x 16 words)
    int seed_rows(size_t count, uint32_t* raw) {
        PV_TRY(seed_sync());
        hipLaunchKernelGGL(k_chacha20_rows, dim3((unsigned)((count + 255) / 256), (unsigned)B), dim3(256), 0, st, d_seed_keys, seed_ctr, count, raw);
        BZH_HIP_TRY(ctx, hipGetLastError());
        seed_ctr += count;
        return BZH_OK;
    }
    // the next `count` draws of every proof, reduced on the device into dst (B x count, proof-major)
    int draw_rows(size_t count, uint32_t* dst) {
        if (!count) return BZH_OK;
        uint32_t* raw = (uint32_t*)arena.alloc(B * count * 64);
        if (!raw) return BZH_E_OOM;
        if (seeded) {
            PV_TRY(seed_rows(count, raw));
            return random_field(ctx, field, raw, B * count, dst);
        }
        char* stage = nullptr;  // one upload for the whole batch, assembled in pinned memory
        PV_TRY(h2d_stage(ctx, B * count * 64, &stage));
        for (size_t b = 0; b < B; b++) {
            memcpy(stage + b * count * 64, rng[b], count * 64);
            rng[b] += count * 64;
        }
        PV_TRY(h2d_commit(ctx, raw, stage, B * count * 64));
        return random_field(ctx, field, raw, B * count, dst);
    }
    Fe<SF> squeeze(size_t b) {
        uint64_t ch[4];
        bzh_transcript_squeeze_challenge(T[b], ch);
        return fe_to_mont(h_load<SF>(ch));
    }
    void write_scalar(size_t b, const Fe<SF>& v) {
        uint64_t s[4];
        h_store<SF>(s, fe_from_mont(v));
        bzh_transcript_write_scalar(T[b], s);
    }

    // ---- transforms ------------------------------------------------------------------------------
    int to_coeff(uint32_t* dst, const uint32_t* src, size_t count) {
        if (!count) return BZH_OK;
        BZH_HIP_TRY(ctx, hipMemcpyAsync(dst, src, count * n * 32, hipMemcpyDeviceToDevice, st));
        return ntt_run(ctx, field, dst, pk.k, count, pk.omega, nullptr, 1, BZH_FORM_MONTGOMERY);
    }
    int to_extended(uint32_t* dst, const uint32_t* polys, size_t count) {
        if (!count) return BZH_OK;
        return ntt_run_padded(ctx, field, dst, polys, pk.k, pk.ek, count, pk.eomega, pk.zeta);
    }
    // Params::commit for `count` polynomials (rows of `pitch` elements): affine canonical points out
    // (lagrange: the rows are evaluations over the domain and the bases g_lagrange -- Params::commit_lagrange; the group
    // element is the same as committing the interpolated coefficients to g, but witness columns are sparse and small in
    // this basis, so most window digits are zero and cost the MSM nothing)
    int commit(const uint32_t* polys, size_t pitch, size_t count, const std::vector<Fe<SF>>& blinds, std::vector<uint64_t>& xy,
               bool lagrange = false) {
        xy.assign(count * 8, 0);
        if (!count) return BZH_OK;
        uint32_t* sc = dalloc(count * (n + 2));
        uint32_t* bl = dalloc(count);
        uint32_t* d_out = dalloc(count * 3);
        if (!sc || !bl || !d_out) return BZH_E_OOM;
        PV_TRY(zero(sc, count * (n + 2)));
        PV_TRY(copy2d(sc, n + 2, polys, pitch, n, count));
        PV_TRY(upload(bl, blinds.data(), count));
        PV_TRY(copy2d(sc + (n + 1) * 8, n + 2, bl, 1, 1, count));
        PV_TRY(msm_run(ctx, lagrange ? pk.srs_lagrange : pk.srs, sc, n + 2, count, BZH_FORM_MONTGOMERY, d_out));
        std::vector<uint64_t> jac(count * 12);
        PV_TRY(d2h_async(ctx, jac.data(), d_out, count * 96));
        PV_TRY(d2h_finish(ctx));
        // Jacobian (Montgomery) -> affine canonical, one inversion
        std::vector<Fe<PB>> pre(count + 1);
        pre[0] = fe_one<PB>();
        for (size_t i = 0; i < count; i++) {
            const Fe<PB> Z = h_load<PB>(&jac[i * 12 + 8]);
            pre[i + 1] = fe_is_zero(Z) ? pre[i] : fe_mul(pre[i], Z);
        }
        Fe<PB> inv = fe_inv(pre[count]);
        for (size_t i = count; i-- > 0;) {
            const Fe<PB> Z = h_load<PB>(&jac[i * 12 + 8]);
            if (fe_is_zero(Z)) continue;
            const Fe<PB> zi = fe_mul(inv, pre[i]);
            inv = fe_mul(inv, Z);
            const Fe<PB> zi2 = fe_sqr(zi), zi3 = fe_mul(zi2, zi);
            h_store<PB>(&xy[i * 8], fe_from_mont(fe_mul(h_load<PB>(&jac[i * 12]), zi2)));
            h_store<PB>(&xy[i * 8 + 4], fe_from_mont(fe_mul(h_load<PB>(&jac[i * 12 + 4]), zi3)));
        }
        return BZH_OK;
    }
    // evaluate `count` polynomials (contiguous, n coefficients each) at one point each
    int evals(const uint32_t* stacked, size_t count, const std::vector<Fe<SF>>& points, std::vector<Fe<SF>>& out) {
        out.resize(count);
        if (!count) return BZH_OK;
        uint32_t* xs = dalloc(count);
        uint32_t* res = dalloc(count);
        if (!xs || !res) return BZH_E_OOM;
        PV_TRY(upload(xs, points.data(), count));
        PV_TRY(poly_eval(ctx, field, stacked, n, count, xs, 1, res));
        PV_TRY(d2h_async(ctx, out.data(), res, count * 32));
        PV_TRY(d2h_finish(ctx));
        return BZH_OK;
    }

    // ---- compiled programs -------------------------------------------------------------------------
    template <class Build>
    int run(uint64_t pkey, Build build, const Cols& reg, size_t size, uint32_t* d_out) {
        const Program* pgp = nullptr;
        {
            std::lock_guard<std::mutex> lk(pk.mu);   // map nodes are stable: the program outlives the lock
            auto it = pk.progs.find(pkey);
            if (it == pk.progs.end()) {
                EPool ep;
                const int root = build(ep);
                Compiler cc(ep);
                cc.prog.result_slot = cc.emit(root);
                if (cc.overflow) return BZH_E_RANGE;
                it = pk.progs.insert({pkey, std::move(cc.prog)}).first;
            }
            pgp = &it->second;
        }
        const Program& pg = *pgp;
        const size_t nc = pg.consts.size(), ncols = reg.ptr.size();
        bool per_proof = false;
        for (auto& c : pg.consts) per_proof |= c.sym >= 0;
        const size_t rows = per_proof ? B : 1;
        std::vector<uint32_t> cv(std::max<size_t>(rows * nc, 1) * 8);
        for (size_t b = 0; b < rows; b++)
            for (size_t i = 0; i < nc; i++) {
                const ConstEnt& c = pg.consts[i];
                if (c.sym >= 0) {
                    auto f = env[b].find(c.sym);
                    if (f == env[b].end()) return BZH_E_ARG;
                    memcpy(&cv[(b * nc + i) * 8], f->second.l, 32);
                } else {
                    memcpy(&cv[(b * nc + i) * 8], c.val, 32);
                }
            }
        char* stage = (char*)arena.alloc(cv.size() * 4 + pg.ops.size() * sizeof(bzh_expr_op) + ncols * 16 + 1024);
        if (!stage) return BZH_E_OOM;
        uint32_t* d_consts = (uint32_t*)stage;
        char* d_prog = stage + ((cv.size() * 4 + 255) & ~(size_t)255);
        char* d_ptrs = d_prog + ((pg.ops.size() * sizeof(bzh_expr_op) + 255) & ~(size_t)255);
        char* d_strides = d_ptrs + ((ncols * 8 + 255) & ~(size_t)255);
        PV_TRY(h2d_small(ctx, d_consts, cv.data(), cv.size() * 4));
        PV_TRY(h2d_small(ctx, d_prog, pg.ops.data(), pg.ops.size() * sizeof(bzh_expr_op)));
        PV_TRY(h2d_small(ctx, d_ptrs, reg.ptr.data(), ncols * 8));
        PV_TRY(h2d_small(ctx, d_strides, reg.stride.data(), ncols * 8));
        int nslots = pg.result_slot + 1;
        for (auto& o : pg.ops) nslots = std::max(nslots, (int)o.dst + 1);  // operands only read slots written before
        return expr_eval(ctx, field, d_prog, (int)pg.ops.size(), (const uint32_t* const*)d_ptrs, (const size_t*)d_strides, d_consts,
                         per_proof ? nc : 0, size, pg.result_slot, B, nslots, d_out);
    }

    // the quotient through VM v2 (the program compiled at bzh_pk_create), as the builtin kernel, the caller's module or the
    // interpreter.  Returns BZH_E_RANGE when the circuit does not fit VM v2 (the caller falls back to the plain fold through `run`).
    int run_quotient(const Cols& reg, size_t size, uint32_t* d_out) {
        if (!pk.q_ok || size % 128 || size != pk.en) return BZH_E_RANGE;
        hipFunction_t q_fn = nullptr;
        bzh_quotient_launch_fn q_builtin = nullptr;
        {
            std::lock_guard<std::mutex> lk(pk.mu);
            if (pk.q_select == BZH_QUOTIENT_MODULE) q_fn = pk.q_fn;
            else if (pk.q_select == BZH_QUOTIENT_BUILTIN) q_builtin = pk.q_builtin;
        }
        const Program2* pgp = &pk.qprog;
        const Program2& pg = *pgp;
        if (!pg.ok) return BZH_E_RANGE;
        const size_t nc = pg.consts.size(), ncols = reg.ptr.size() + pk.hoist_cols;
        std::vector<const uint32_t*> ptrs(reg.ptr);
        std::vector<size_t> strides(reg.stride);
        for (size_t hi = 0; hi < pk.hoist_cols; hi++) {
            ptrs.push_back(pk.hoist + hi * size * 8);
            strides.push_back(0);
        }
        std::vector<uint32_t> cv(std::max<size_t>(B * nc, 1) * 8);
        for (size_t b = 0; b < B; b++)
            for (size_t i = 0; i < nc; i++) {
                const ConstEnt& c = pg.consts[i];
                if (c.sym >= 0) {
                    auto f = env[b].find(c.sym);
                    if (f == env[b].end()) return BZH_E_ARG;
                    memcpy(&cv[(b * nc + i) * 8], f->second.l, 32);
                } else {
                    memcpy(&cv[(b * nc + i) * 8], c.val, 32);
                }
            }
        char* stage = (char*)arena.alloc(cv.size() * 4 + pg.ops.size() * sizeof(ExprOp2) + ncols * 16 + 1024);
        if (!stage) return BZH_E_OOM;
        uint32_t* d_consts = (uint32_t*)stage;
        char* d_prog = stage + ((cv.size() * 4 + 255) & ~(size_t)255);
        char* d_ptrs = d_prog + ((pg.ops.size() * sizeof(ExprOp2) + 255) & ~(size_t)255);
        char* d_strides = d_ptrs + ((ncols * 8 + 255) & ~(size_t)255);
        PV_TRY(h2d_small(ctx, d_consts, cv.data(), cv.size() * 4));
        PV_TRY(h2d_small(ctx, d_prog, pg.ops.data(), pg.ops.size() * sizeof(ExprOp2)));
        PV_TRY(h2d_small(ctx, d_ptrs, ptrs.data(), ncols * 8));
        PV_TRY(h2d_small(ctx, d_strides, strides.data(), ncols * 8));
        if (ctx->profiling) {   // the quotient pass reads every extended column once and writes h: per-proof columns
            double cols_read = 0;   // count per proof, columns of the key once per launch
            for (size_t i = 0; i < reg.stride.size(); i++) cols_read += reg.stride[i] ? (double)B : 1.0;
            ctx->alg_bytes[BZH_T_QUOTIENT] += (cols_read + (double)B) * (double)size * 32.0;
        }
        if (q_builtin) {   // the same program as a kernel generated at build time
            ScopedTimer t(ctx, BZH_T_QUOTIENT);
            q_builtin((unsigned)(size / 128), (unsigned)B, (void*)st, (const uint32_t* const*)d_ptrs, (const size_t*)d_strides, d_consts, nc, size, d_out);
            BZH_HIP_TRY(ctx, hipGetLastError());
            return BZH_OK;
        }
        if (q_fn) {   // the same program as a code object of the caller's (bzh_pk_set_quotient_module)
            ScopedTimer t(ctx, BZH_T_QUOTIENT);
            const uint32_t* const* a_cols = (const uint32_t* const*)d_ptrs;
            const size_t* a_strides = (const size_t*)d_strides;
            const uint32_t* a_consts = d_consts;
            size_t a_nc = nc, a_size = size;
            uint32_t* a_out = d_out;
            void* args[] = {&a_cols, &a_strides, &a_consts, &a_nc, &a_size, &a_out};
            BZH_HIP_TRY(ctx, hipModuleLaunchKernel(q_fn, (unsigned)(size / 128), (unsigned)B, 1, 128, 1, 1, 0, st, args, nullptr));
            return BZH_OK;
        }
        return expr_eval2(ctx, field, d_prog, (int)pg.ops.size(), (const uint32_t* const*)d_ptrs, (const size_t*)d_strides, d_consts, nc, size, B,
                          pg.nlds, d_out);
    }

    int prove(const uint32_t* d_advice_in, const uint64_t* instances, size_t inst_rows, uint8_t* proofs, size_t proof_stride,
              size_t* proof_lens);
};

template <class C>
int Prover<C>::prove(const uint32_t* d_advice_in, const uint64_t* instances, size_t inst_rows, uint8_t* proofs, size_t proof_stride,
                     size_t* proof_lens) {
    const int na = pk.na, nf = pk.nf, ni = pk.ni, bf = pk.bf, nsets = pk.nsets, nl = pk.nl, npieces = pk.npieces;
    const size_t bf1 = (size_t)bf + 1, m = pk.perm_columns.size(), ext = pk.ext;
    const int nz = nsets + nl;
    std::vector<uint64_t> xy;
    std::vector<Fe<SF>> blinds;
    for (size_t b = 0; b < B; b++) {
        PV_TRY(bzh_transcript_new(field, &T[b]));
        bzh_transcript_common_scalar(T[b], pk.vk_repr);
    }

    mark("setup");
    // ---- instance columns ----------------------------------------------------------------------
    uint32_t* inst = dalloc(B * std::max(ni, 1) * n);
    uint32_t* inst_polys = dalloc(B * std::max(ni, 1) * n);
    if (!inst || !inst_polys) return BZH_E_OOM;
    if (ni) {
        PV_TRY(zero(inst, B * ni * n));
        if (inst_rows) {
            std::vector<Fe<SF>> hv(B * ni * inst_rows);
            for (size_t i = 0; i < hv.size(); i++) hv[i] = fe_to_mont(h_load<SF>(instances + 4 * i));
            uint32_t* tmp = dalloc(hv.size());
            if (!tmp) return BZH_E_OOM;
            PV_TRY(upload(tmp, hv.data(), hv.size()));
            PV_TRY(copy2d(inst, n, tmp, inst_rows, inst_rows, B * ni));
        }
        PV_TRY(to_coeff(inst_polys, inst, B * ni));
        blinds.assign(B * ni, fe_one<SF>());
        if (pk.srs_lagrange) PV_TRY(commit(inst, n, B * ni, blinds, xy, true));
        else PV_TRY(commit(inst_polys, n, B * ni, blinds, xy));
        for (size_t b = 0; b < B; b++)
            for (int i = 0; i < ni; i++) bzh_transcript_common_point(T[b], &xy[(b * ni + i) * 8]);
    }

    mark("instance");
    // ---- advice columns ------------------------------------------------------------------------
    uint32_t* adv = dalloc(B * na * n);
    uint32_t* adv_polys = dalloc(B * na * n);
    if (!adv || !adv_polys) return BZH_E_OOM;
    BZH_HIP_TRY(ctx, hipMemcpyAsync(adv, d_advice_in, B * na * n * 32, hipMemcpyDeviceToDevice, st));
    {
        uint32_t* rows = dalloc(B * na * bf1);
        if (!rows) return BZH_E_OOM;
        PV_TRY(draw_rows(na * bf1, rows));
        PV_TRY(copy2d(adv + usable * 8, n, rows, bf1, bf1, B * na));
    }
    std::vector<Fe<SF>> adv_blinds(B * na);
    for (size_t b = 0; b < B; b++)
        for (int i = 0; i < na; i++) adv_blinds[b * na + i] = draw(b);
    PV_TRY(to_coeff(adv_polys, adv, B * na));
    if (pk.srs_lagrange) PV_TRY(commit(adv, n, B * na, adv_blinds, xy, true));
    else PV_TRY(commit(adv_polys, n, B * na, adv_blinds, xy));
    for (size_t b = 0; b < B; b++) {
        for (int i = 0; i < na; i++) bzh_transcript_write_point(T[b], C::id, &xy[(b * na + i) * 8]);
        env[b][SY_THETA] = squeeze(b);
    }
    uint32_t *inst_cosets = nullptr, *adv_cosets = nullptr;
    auto extend_witness = [&]() -> int {  // queued late on purpose: runs on the device while the host sorts the lookups
        if (adv_cosets) return BZH_OK;
        inst_cosets = dalloc(B * std::max(ni, 1) * en);
        adv_cosets = dalloc(B * na * en);
        if (!inst_cosets || !adv_cosets) return BZH_E_OOM;
        PV_TRY(to_extended(inst_cosets, inst_polys, B * ni));
        return to_extended(adv_cosets, adv_polys, B * na);
    };
    auto lag_registry = [&](Cols& reg) {
        for (int i = 0; i < na; i++) reg.add(key(K_ADV, i), adv + (size_t)i * n * 8, (size_t)na * n);
        for (int i = 0; i < nf; i++) reg.add(key(K_FIX, i), pk.fixed + (size_t)i * n * 8, 0);
        for (int i = 0; i < ni; i++) reg.add(key(K_INST, i), inst + (size_t)i * n * 8, (size_t)ni * n);
    };

    mark("advice");
    // ---- lookups: compress, permute (host sort), commit -------------------------------------------
    struct Lk {
        uint32_t *a_c, *s_c, *as, *polys, *cosets;
        std::vector<Fe<SF>> blinds;  // (a, s) per proof
    };
    std::vector<Lk> lk(nl);
    for (int li = 0; li < nl; li++) {
        Lk& d = lk[li];
        d.a_c = dalloc(B * n);
        d.s_c = dalloc(B * n);
        d.as = dalloc(B * 2 * n);
        d.polys = dalloc(B * 2 * n);
        if (!d.a_c || !d.s_c || !d.as || !d.polys) return BZH_E_OOM;
        Cols reg;
        lag_registry(reg);
        for (int side = 0; side < 2; side++) {
            const std::vector<int>& es = side ? pk.lookups[li].second : pk.lookups[li].first;
            PV_TRY(run(key(20 + side, li), [&](EPool& ep) {
                std::vector<int> terms;
                for (int e : es) terms.push_back(lower(pk, e, ep, reg, 1));
                return ep.horner(terms, ep.sym(SY_THETA));
            }, reg, n, side ? d.s_c : d.a_c));
        }
        // compressed columns come back through pinned memory; the permuted pair is assembled in a pinned slot in the
        // device layout (B, 2, n) (rows past `usable` zero until the blinding rows land) and goes up in one piece
        char *ah_c = nullptr, *sh_c = nullptr, *as_c = nullptr;
        PV_TRY(pin_big_reserve(ctx, 4 * B * n * 32 + ((size_t)3 << 20)));
        PV_TRY(pin_big_take(ctx, B * n * 32, &ah_c));
        PV_TRY(pin_big_take(ctx, B * n * 32, &sh_c));
        PV_TRY(pin_big_take(ctx, B * 2 * n * 32, &as_c));
        // the host sorts canonical integers: convert on the device (copies; the Montgomery originals feed the grand product)
        uint32_t* canon = dalloc(2 * B * n);
        if (!canon) return BZH_E_OOM;
        BZH_HIP_TRY(ctx, hipMemcpyAsync(canon, d.a_c, B * n * 32, hipMemcpyDeviceToDevice, st));
        BZH_HIP_TRY(ctx, hipMemcpyAsync(canon + B * n * 8, d.s_c, B * n * 32, hipMemcpyDeviceToDevice, st));
        PV_TRY(field_convert(ctx, field, canon, 2 * B * n, 0));
        PV_TRY(xfer_launch(ctx, ah_c, canon, B * n * 32, hipMemcpyDeviceToHost));
        PV_TRY(xfer_launch(ctx, sh_c, canon + B * n * 8, B * n * 32, hipMemcpyDeviceToHost));
        BZH_HIP_TRY(ctx, hipStreamSynchronize(st));
        const uint64_t* ah = (const uint64_t*)ah_c;
        const uint64_t* sh = (const uint64_t*)sh_c;
        mark(" lk:compress+d2h");
        PV_TRY(extend_witness());
        mark(" lk:extend_witness");
        uint64_t* as = (uint64_t*)as_c;
        for (size_t v = 0; v < 2 * B; v++) memset(as + (v * n + usable) * 4, 0, (n - usable) * 32);
        {  // one sort per proof on host threads
            // short-lived pool, capped: several provers (threads, ranks) run this at once on the same host
            const size_t nthreads = std::min<size_t>({B, (size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)8});
            std::vector<int> rcs(B, BZH_OK);
            std::vector<std::thread> th;
            for (size_t t = 0; t < nthreads; t++)
                th.emplace_back([&, t]() {
                    for (size_t b = t; b < B; b += nthreads)
                        rcs[b] = bzh_permute_expression_pair(field, &ah[b * n * 4], &sh[b * n * 4], usable, BZH_FORM_CANONICAL,
                                                             as + (b * 2) * n * 4, as + (b * 2 + 1) * n * 4);
                });
            for (auto& t : th) t.join();
            for (int rc : rcs)
                if (rc) return rc;
        }
        mark(" lk:sort");
        PV_TRY(h2d_commit(ctx, d.as, as_c, B * 2 * n * 32));
        PV_TRY(field_convert(ctx, field, d.as, B * 2 * n, 1));  // back to Montgomery form (the zero rows stay zero)
        mark(" lk:h2d");
        {
            uint32_t* rows = dalloc(B * 2 * bf1);
            if (!rows) return BZH_E_OOM;
            PV_TRY(draw_rows(2 * bf1, rows));
            PV_TRY(copy2d(d.as + usable * 8, n, rows, bf1, bf1, B * 2));
        }
        d.blinds.resize(B * 2);
        for (size_t b = 0; b < B; b++) {
            d.blinds[2 * b] = draw(b);
            d.blinds[2 * b + 1] = draw(b);
        }
        PV_TRY(to_coeff(d.polys, d.as, B * 2));
        if (pk.srs_lagrange) PV_TRY(commit(d.as, n, B * 2, d.blinds, xy, true));
        else PV_TRY(commit(d.polys, n, B * 2, d.blinds, xy));
        for (size_t b = 0; b < B; b++) {
            bzh_transcript_write_point(T[b], C::id, &xy[(2 * b) * 8]);
            bzh_transcript_write_point(T[b], C::id, &xy[(2 * b + 1) * 8]);
        }
    }
    PV_TRY(extend_witness());
    for (size_t b = 0; b < B; b++) {
        env[b][SY_BETA] = squeeze(b);
        env[b][SY_GAMMA] = squeeze(b);
    }

    mark("lookup");
    // ---- permutation and lookup grand products -----------------------------------------------------
    uint32_t* zs = dalloc(B * std::max(nz, 1) * n);
    uint32_t* z_polys = dalloc(B * std::max(nz, 1) * n);
    uint32_t* z_cosets = dalloc(B * std::max(nz, 1) * en);
    uint32_t* den = dalloc(B * n);
    uint32_t* zt = dalloc(B * n);
    if (!zs || !z_polys || !z_cosets || !den || !zt) return BZH_E_OOM;
    std::vector<Fe<SF>> z_blinds(B * std::max(nz, 1));
    auto finish_product = [&](int slot, int prev_slot) -> int {
        mark("  fp:exprs");
        PV_TRY(poly_batch_invert(ctx, field, den, B * n));
        mark("  fp:invert");
        PV_TRY(poly_vec_mul(ctx, field, zt, den, B * n));
        PV_TRY(poly_prefix_product(ctx, field, zt, n, B));
        mark("  fp:mul+scan");
        if (prev_slot >= 0)
            hipLaunchKernelGGL((k_scale_rows<SF>), dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, st, zt, n,
                               zs + ((size_t)prev_slot * n + usable) * 8, (size_t)nz * n);
        uint32_t* rows = dalloc(B * bf);
        if (!rows) return BZH_E_OOM;
        PV_TRY(draw_rows(bf, rows));
        PV_TRY(copy2d(zt + (n - bf) * 8, n, rows, bf, bf, B));
        for (size_t b = 0; b < B; b++) z_blinds[b * nz + slot] = draw(b);
        return copy2d(zs + (size_t)slot * n * 8, (size_t)nz * n, zt, n, n, B);
    };
    auto lag_col = [&](Cols& reg, std::pair<int, int> col) {
        if (col.first == CX_ADVICE) return reg.add(key(K_ADV, col.second), adv + (size_t)col.second * n * 8, (size_t)na * n);
        if (col.first == CX_FIXED) return reg.add(key(K_FIX, col.second), pk.fixed + (size_t)col.second * n * 8, 0);
        return reg.add(key(K_INST, col.second), inst + (size_t)col.second * n * 8, (size_t)ni * n);
    };
    for (int i = 0; i < nsets; i++) {
        const size_t c0 = (size_t)i * pk.chunk_len, c1 = std::min(m, c0 + pk.chunk_len);
        Cols reg;
        for (size_t gj = c0; gj < c1; gj++) {
            lag_col(reg, pk.perm_columns[gj]);
            reg.add(key(K_SIGMA, gj), pk.sigma + gj * n * 8, 0);
            reg.add(key(K_IDENT, gj), pk.ident + gj * n * 8, 0);
        }
        for (int which = 0; which < 2; which++) {  // 0: denominator, 1: numerator
            PV_TRY(run(key(30 + which, i), [&](EPool& ep) {
                int acc = -1;
                for (size_t gj = c0; gj < c1; gj++) {
                    const int v = ep.query(lag_col(reg, pk.perm_columns[gj]));
                    const int f = which == 0 ? ep.add(ep.add(ep.mul(ep.sym(SY_BETA), ep.query(reg.at(key(K_SIGMA, gj)))), ep.sym(SY_GAMMA)), v)
                                             : ep.add(ep.add(ep.mul(ep.query(reg.at(key(K_IDENT, gj))), ep.sym(SY_BETA)), ep.sym(SY_GAMMA)), v);
                    acc = acc < 0 ? f : ep.mul(acc, f);
                }
                return acc;
            }, reg, n, which == 0 ? den : zt));
        }
        PV_TRY(finish_product(i, i ? i - 1 : -1));
        mark(" gp:perm_set");
    }
    for (int li = 0; li < nl; li++) {
        Cols reg;
        reg.add(key(K_MISC, M_AC), lk[li].a_c, n);
        reg.add(key(K_MISC, M_SC), lk[li].s_c, n);
        reg.add(key(K_MISC, M_A), lk[li].as, 2 * n);
        reg.add(key(K_MISC, M_S), lk[li].as + n * 8, 2 * n);
        PV_TRY(run(key(32, li), [&](EPool& ep) {
            return ep.mul(ep.add(ep.query(0), ep.sym(SY_BETA)), ep.add(ep.query(1), ep.sym(SY_GAMMA)));
        }, reg, n, zt));
        PV_TRY(run(key(33, li), [&](EPool& ep) {
            return ep.mul(ep.add(ep.query(2), ep.sym(SY_BETA)), ep.add(ep.query(3), ep.sym(SY_GAMMA)));
        }, reg, n, den));
        PV_TRY(finish_product(nsets + li, -1));
        mark(" gp:lookup_product");
    }
    if (nz) {
        PV_TRY(to_coeff(z_polys, zs, B * nz));
        if (pk.srs_lagrange) PV_TRY(commit(zs, n, B * nz, z_blinds, xy, true));
        else PV_TRY(commit(z_polys, n, B * nz, z_blinds, xy));
        for (size_t b = 0; b < B; b++)
            for (int i = 0; i < nz; i++) bzh_transcript_write_point(T[b], C::id, &xy[(b * nz + i) * 8]);
        mark(" gp:commit");
        PV_TRY(to_extended(z_cosets, z_polys, B * nz));
        mark(" gp:extend_z");
    }
    for (auto& d : lk) {
        d.cosets = dalloc(B * 2 * en);
        if (!d.cosets) return BZH_E_OOM;
        PV_TRY(to_extended(d.cosets, d.polys, B * 2));
    }

    mark("grand_products");
    // ---- vanishing argument ----------------------------------------------------------------------
    uint32_t* random_poly = dalloc(B * n);
    if (!random_poly) return BZH_E_OOM;
    PV_TRY(draw_rows(n, random_poly));
    std::vector<Fe<SF>> random_blinds(B);
    for (size_t b = 0; b < B; b++) random_blinds[b] = draw(b);
    PV_TRY(commit(random_poly, n, B, random_blinds, xy));
    const Fe<SF> delta = [&] {
        Fe<SF> d;
        memcpy(d.l, pk.delta, 32);
        return d;
    }();
    for (size_t b = 0; b < B; b++) {
        bzh_transcript_write_point(T[b], C::id, &xy[b * 8]);
        env[b][SY_Y] = squeeze(b);
        {
            Fe<SF> yp = env[b][SY_Y];
            for (int mpow = 2; mpow <= 64; mpow++) {   // y^m for the gate-factored fold (m = constraints per gate)
                yp = fe_mul(yp, env[b][SY_Y]);
                env[b][SY_YPOW0 + mpow] = yp;
            }
        }
        Fe<SF> bd = env[b][SY_BETA];
        for (size_t gj = 0; gj < m; gj++) {
            env[b][SY_BD0 + (int)gj] = bd;
            bd = fe_mul(bd, delta);
        }
    }
    mark("vanishing_setup");
    const int last_rot = -(bf + 1);
    uint32_t* h = dalloc(B * en);
    if (!h) return BZH_E_OOM;
    {
        Cols reg;
        QuotientPtrs qp;
        qp.adv = adv_cosets, qp.inst = inst_cosets, qp.z = z_cosets;
        for (int i = 0; i < nl; i++) qp.lk.push_back(lk[i].cosets);
        quotient_registry(pk, qp, reg);
        // VM v2 (gate-factored fold, shared subexpressions in LDS); the plain Horner fold through VM v1 if it does not fit
        int qrc = getenv("BZH_QUOTIENT_V1") ? BZH_E_RANGE : run_quotient(reg, en, h);
        if (qrc == BZH_E_RANGE) {
            qrc = run(key(40, 0), [&](EPool& ep) {
                int tinv = -1;
                const std::vector<int> terms = quotient_terms<SF>(pk, reg, ep, &tinv);
                return ep.mul(ep.horner(terms, ep.sym(SY_Y)), tinv);
            }, reg, en, h);
        }
        PV_TRY(qrc);
    }
    PV_TRY(ntt_run(ctx, field, h, pk.ek, B, pk.eomega, pk.zeta, 1, BZH_FORM_MONTGOMERY));
    uint32_t* d_flag = (uint32_t*)arena.alloc(256);
    if (!d_flag) return BZH_E_OOM;
    uint32_t h_flag = 0;
    if ((size_t)npieces * n < en) {
        BZH_HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, 4, st));
        const size_t words = (en - (size_t)npieces * n) * 8;
        hipLaunchKernelGGL(k_any_nonzero, dim3((unsigned)((words + 255) / 256), (unsigned)B), dim3(256), 0, st,
                           h + (size_t)npieces * n * 8, words, en * 8, d_flag);
        PV_TRY(d2h_async(ctx, &h_flag, d_flag, 4));  // lands at the commit's d2h_finish
    }
    std::vector<Fe<SF>> h_blinds(B * npieces);
    for (size_t b = 0; b < B; b++)
        for (int i = 0; i < npieces; i++) h_blinds[b * npieces + i] = draw(b);
    {
        // pieces of proof b: h[b][i*n .. (i+1)*n) -> (B * npieces) rows; piece rows are n apart inside a proof, proofs en apart
        uint32_t* pieces = dalloc(B * npieces * n);
        if (!pieces) return BZH_E_OOM;
        PV_TRY(copy2d(pieces, (size_t)npieces * n, h, en, (size_t)npieces * n, B));
        PV_TRY(commit(pieces, n, B * npieces, h_blinds, xy));
    }
    if (h_flag) {
        ctx->last_error = "quotient has higher degree than expected: a witness does not satisfy the constraints";
        return BZH_E_RANGE;
    }
    std::vector<Fe<SF>> xs(B);
    Fe<SF> omega_m;
    {
        uint64_t t[4];
        memcpy(t, pk.omega, 32);
        omega_m = h_load<SF>(t);
    }
    const Fe<SF> omega_inv = fe_inv(omega_m);
    for (size_t b = 0; b < B; b++) {
        for (int i = 0; i < npieces; i++) bzh_transcript_write_point(T[b], C::id, &xy[(b * npieces + i) * 8]);
        xs[b] = squeeze(b);
        env[b][SY_XN] = h_pow_u64(xs[b], n);
    }
    std::map<int, Fe<SF>> wp;
    auto rot = [&](size_t b, int r) {
        auto it = wp.find(r);
        if (it == wp.end()) it = wp.insert({r, r >= 0 ? h_pow_u64(omega_m, (uint64_t)r) : h_pow_u64(omega_inv, (uint64_t)(-(int64_t)r))}).first;
        return fe_mul(xs[b], it->second);
    };

    mark("quotient+h_commit");
    // ---- evaluations: one gather of (polynomial, rotation) jobs ----------------------------------------
    // where each committed polynomial lives: (pointer of proof 0, elements between proofs)
    std::map<uint64_t, std::pair<const uint32_t*, size_t>> where;
    for (int i = 0; i < ni; i++) where[key(K_INST, i)] = {inst_polys + (size_t)i * n * 8, (size_t)ni * n};
    for (int i = 0; i < na; i++) where[key(K_ADV, i)] = {adv_polys + (size_t)i * n * 8, (size_t)na * n};
    for (int i = 0; i < nf; i++) where[key(K_FIX, i)] = {pk.fixed_polys + (size_t)i * n * 8, 0};
    for (size_t j = 0; j < m; j++) where[key(K_SIGMA, j)] = {pk.sigma_polys + j * n * 8, 0};
    where[key(K_MISC, M_F)] = {random_poly, n};
    for (int i = 0; i < nsets; i++) where[key(K_PZ, i)] = {z_polys + (size_t)i * n * 8, (size_t)nz * n};
    for (int i = 0; i < nl; i++) {
        where[key(K_LZ, i)] = {z_polys + (size_t)(nsets + i) * n * 8, (size_t)nz * n};
        where[key(K_LA, i)] = {lk[i].polys, 2 * n};
        where[key(K_LS, i)] = {lk[i].polys + n * 8, 2 * n};
    }
    auto gather = [&](const std::vector<std::pair<const uint32_t*, size_t>>& srcs, uint32_t* dst) -> int {
        const size_t J = srcs.size();
        std::vector<const uint32_t*> ps(J);
        std::vector<size_t> ss(J);
        for (size_t j = 0; j < J; j++) {
            ps[j] = srcs[j].first;
            ss[j] = srcs[j].second;
        }
        char* stage = (char*)arena.alloc(J * 16 + 512);
        if (!stage) return BZH_E_OOM;
        char* d_ss = stage + ((J * 8 + 255) & ~(size_t)255);
        PV_TRY(h2d_small(ctx, stage, ps.data(), J * 8));
        PV_TRY(h2d_small(ctx, d_ss, ss.data(), J * 8));
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((2 * n + 255) / 256), (unsigned)J, (unsigned)B), dim3(256), 0, st, (uint4*)dst,
                           (const uint4* const*)stage, (const size_t*)d_ss, n, J);
        BZH_HIP_TRY(ctx, hipGetLastError());
        return BZH_OK;
    };
    {
        std::vector<std::pair<uint64_t, int>> jobs;
        for (auto& a : pk.instance_queries) jobs.push_back({key(K_INST, a.first), a.second});
        for (auto& a : pk.advice_queries) jobs.push_back({key(K_ADV, a.first), a.second});
        for (auto& a : pk.fixed_queries) jobs.push_back({key(K_FIX, a.first), a.second});
        jobs.push_back({key(K_MISC, M_F), 0});
        for (size_t j = 0; j < m; j++) jobs.push_back({key(K_SIGMA, j), 0});
        for (int i = 0; i < nsets; i++) {
            jobs.push_back({key(K_PZ, i), 0});
            jobs.push_back({key(K_PZ, i), 1});
            if (i != nsets - 1) jobs.push_back({key(K_PZ, i), last_rot});
        }
        for (int i = 0; i < nl; i++) {
            jobs.push_back({key(K_LZ, i), 0});
            jobs.push_back({key(K_LZ, i), 1});
            jobs.push_back({key(K_LA, i), 0});
            jobs.push_back({key(K_LA, i), -1});
            jobs.push_back({key(K_LS, i), 0});
        }
        const size_t J = jobs.size();
        std::vector<std::pair<const uint32_t*, size_t>> srcs(J);
        for (size_t j = 0; j < J; j++) srcs[j] = where.at(jobs[j].first);
        uint32_t* gathered = dalloc(B * J * n);
        if (!gathered) return BZH_E_OOM;
        PV_TRY(gather(srcs, gathered));
        std::vector<Fe<SF>> pts(B * J), vals;
        for (size_t b = 0; b < B; b++)
            for (size_t j = 0; j < J; j++) pts[b * J + j] = rot(b, jobs[j].second);
        PV_TRY(evals(gathered, B * J, pts, vals));
        for (size_t b = 0; b < B; b++)
            for (size_t j = 0; j < J; j++) write_scalar(b, vals[b * J + j]);
    }

    mark("evaluations");
    // ---- h(X) = sum_i x^(n i) h_i(X): Horner from the top piece ---------------------------------------
    uint32_t* h_poly = dalloc(B * n);
    if (!h_poly) return BZH_E_OOM;
    {
        Cols reg;
        for (int i = 0; i < npieces; i++) reg.add(key(K_MISC, M_H0 + i), h + (size_t)i * n * 8, en);
        PV_TRY(run(key(41, 0), [&](EPool& ep) {
            std::vector<int> t;
            for (int i = npieces - 1; i >= 0; i--) t.push_back(ep.query(i));
            return ep.horner(t, ep.sym(SY_XN));
        }, reg, n, h_poly));
    }
    std::vector<Fe<SF>> h_blind(B);
    for (size_t b = 0; b < B; b++) {
        Fe<SF> acc = fe_zero<SF>();
        for (int i = npieces - 1; i >= 0; i--) acc = fe_add(fe_mul(acc, env[b][SY_XN]), h_blinds[b * npieces + i]);
        h_blind[b] = acc;
    }
    where[key(K_MISC, M_H0)] = {h_poly, n};

    mark("h_poly");
    // ---- multiopen ------------------------------------------------------------------------------
    auto blind_of = [&](size_t b, uint64_t cid) -> Fe<SF> {
        const int kind = (int)(cid >> 32);
        const size_t i = (size_t)(cid & 0xffffffffu);
        switch (kind) {
            case K_ADV: return adv_blinds[b * na + i];
            case K_PZ: return z_blinds[b * nz + i];
            case K_LZ: return z_blinds[b * nz + nsets + i];
            case K_LA: return lk[i].blinds[2 * b];
            case K_LS: return lk[i].blinds[2 * b + 1];
            case K_MISC: return i == M_H0 ? h_blind[b] : random_blinds[b];
            default: return fe_one<SF>();  // instance, fixed, sigma
        }
    };
    for (size_t b = 0; b < B; b++) {
        env[b][SY_X1] = squeeze(b);
        env[b][SY_X2] = squeeze(b);
    }
    const size_t nq = pk.rot_sets.size();
    uint32_t* q_polys = dalloc(B * nq * n);
    uint32_t* acc_a = dalloc(B * n);
    uint32_t* acc_b = dalloc(B * n);
    if (!q_polys || !acc_a || !acc_b) return BZH_E_OOM;
    std::vector<Fe<SF>> q_blinds(B * nq);
    for (size_t si = 0; si < nq; si++) {
        const std::vector<uint64_t>& cids = pk.groups[si];
        for (size_t b = 0; b < B; b++) {
            Fe<SF> acc = fe_zero<SF>();
            for (uint64_t cid : cids) acc = fe_add(fe_mul(acc, env[b][SY_X1]), blind_of(b, cid));
            q_blinds[b * nq + si] = acc;
        }
        // Horner in x1 over the group's polynomials, in chunks that fit the evaluator's slot file
        uint32_t* prev = nullptr;
        for (size_t s0 = 0; s0 < cids.size(); s0 += 16) {
            const size_t s1 = std::min(cids.size(), s0 + 16);
            Cols reg;
            if (prev) reg.add(key(K_MISC, M_ACC), prev, n);
            for (size_t c = s0; c < s1; c++) {
                const auto& w = where.at(cids[c]);
                reg.add(cids[c], w.first, w.second);
            }
            uint32_t* outp = prev == acc_a ? acc_b : acc_a;
            PV_TRY(run(key(50 + si, s0), [&](EPool& ep) {
                std::vector<int> t;
                for (size_t c = 0; c < reg.ptr.size(); c++) t.push_back(ep.query((int)c));
                return ep.horner(t, ep.sym(SY_X1));
            }, reg, n, outp));
            prev = outp;
        }
        PV_TRY(copy2d(q_polys + si * n * 8, nq * n, prev, n, n, B));
    }
    // evaluations of the q polynomials at their own points, remainders r(X), quotients by prod (X - point)
    {
        std::vector<std::pair<size_t, int>> ev_jobs;
        for (size_t si = 0; si < nq; si++)
            for (int r : pk.rot_sets[si]) ev_jobs.push_back({si, r});
        const size_t J2 = ev_jobs.size();
        std::vector<std::pair<const uint32_t*, size_t>> srcs(J2);
        for (size_t j = 0; j < J2; j++) srcs[j] = {q_polys + ev_jobs[j].first * n * 8, nq * n};
        uint32_t* gathered = dalloc(B * J2 * n);
        if (!gathered) return BZH_E_OOM;
        PV_TRY(gather(srcs, gathered));
        std::vector<Fe<SF>> pts(B * J2), ev;
        for (size_t b = 0; b < B; b++)
            for (size_t j = 0; j < J2; j++) pts[b * J2 + j] = rot(b, ev_jobs[j].second);
        PV_TRY(evals(gathered, B * J2, pts, ev));
        size_t maxpts = 1;
        for (auto& rs : pk.rot_sets) maxpts = std::max(maxpts, rs.size());
        std::vector<Fe<SF>> r_small(B * nq * maxpts, fe_zero<SF>());
        // Lagrange interpolation through (points, evals) per proof and point set: the denominators prod_(m != j) (x_j - x_m) of
        // the whole batch are inverted together (one field inversion per batch instead of one per point: ~12 us each on the host)
        std::vector<Fe<SF>> dinv(B * J2);
        for (size_t b = 0; b < B; b++) {
            size_t o2 = 0;
            for (size_t si = 0; si < nq; si++) {
                const size_t np = pk.rot_sets[si].size();
                for (size_t j = 0; j < np; j++) {
                    Fe<SF> dn = fe_one<SF>();
                    for (size_t mm = 0; mm < np; mm++)
                        if (mm != j) dn = fe_mul(dn, fe_sub(pts[b * J2 + o2 + j], pts[b * J2 + o2 + mm]));
                    dinv[b * J2 + o2 + j] = dn;
                }
                o2 += np;
            }
        }
        {
            std::vector<Fe<SF>> pre(dinv.size() + 1);
            pre[0] = fe_one<SF>();
            for (size_t i = 0; i < dinv.size(); i++) {
                if (fe_is_zero(dinv[i])) return BZH_E_ARG;   // two opening points of one set coincide: not a valid domain
                pre[i + 1] = fe_mul(pre[i], dinv[i]);
            }
            Fe<SF> inv = fe_inv(pre[dinv.size()]);
            for (size_t i = dinv.size(); i-- > 0;) {
                const Fe<SF> d = dinv[i];
                dinv[i] = fe_mul(inv, pre[i]);
                inv = fe_mul(inv, d);
            }
        }
        for (size_t b = 0; b < B; b++) {
            size_t o2 = 0;
            for (size_t si = 0; si < nq; si++) {
                const size_t np = pk.rot_sets[si].size();
                std::vector<Fe<SF>> res(np, fe_zero<SF>());   // coefficient vector of length np
                for (size_t j = 0; j < np; j++) {
                    std::vector<Fe<SF>> num{fe_one<SF>()};
                    for (size_t mm = 0; mm < np; mm++) {
                        if (mm == j) continue;
                        const Fe<SF> xm = pts[b * J2 + o2 + mm];
                        std::vector<Fe<SF>> nx(num.size() + 1);
                        nx[0] = fe_neg(fe_mul(xm, num[0]));
                        for (size_t i = 1; i < num.size(); i++) nx[i] = fe_sub(num[i - 1], fe_mul(xm, num[i]));
                        nx[num.size()] = num.back();
                        num.swap(nx);
                    }
                    const Fe<SF> cf = fe_mul(ev[b * J2 + o2 + j], dinv[b * J2 + o2 + j]);
                    for (size_t i = 0; i < num.size(); i++) res[i] = fe_add(res[i], fe_mul(cf, num[i]));
                }
                for (size_t i = 0; i < np; i++) r_small[(b * nq + si) * maxpts + i] = res[i];
                o2 += np;
            }
        }
        uint32_t* rcols = dalloc(B * nq * n);
        uint32_t* rs_dev = dalloc(B * nq * maxpts);
        uint32_t* f_parts = dalloc(B * nq * n);
        uint32_t* k_a = dalloc(B * n);
        uint32_t* k_b = dalloc(B * n);
        if (!rcols || !rs_dev || !f_parts || !k_a || !k_b) return BZH_E_OOM;
        PV_TRY(zero(rcols, B * nq * n));
        PV_TRY(zero(f_parts, B * nq * n));
        PV_TRY(upload(rs_dev, r_small.data(), r_small.size()));
        PV_TRY(copy2d(rcols, n, rs_dev, maxpts, maxpts, B * nq));
        for (size_t si = 0; si < nq; si++) {
            Cols reg;
            reg.add(key(K_MISC, M_Q), q_polys + si * n * 8, nq * n);
            reg.add(key(K_MISC, M_R), rcols + si * n * 8, nq * n);
            PV_TRY(run(key(42, 0), [&](EPool& ep) { return ep.sub(ep.query(0), ep.query(1)); }, reg, n, k_a));
            uint32_t* cur = k_a;
            uint32_t* nxt = k_b;
            size_t len = n;
            for (int r : pk.rot_sets[si]) {
                // [x, x^-1] per proof, one inversion for the batch
                std::vector<Fe<SF>> xv(2 * B), pre(B + 1);
                pre[0] = fe_one<SF>();
                for (size_t b = 0; b < B; b++) {
                    xv[2 * b] = rot(b, r);
                    pre[b + 1] = fe_is_zero(xv[2 * b]) ? pre[b] : fe_mul(pre[b], xv[2 * b]);
                }
                Fe<SF> inv = fe_inv(pre[B]);
                for (size_t b = B; b-- > 0;) {
                    xv[2 * b + 1] = fe_zero<SF>();
                    if (fe_is_zero(xv[2 * b])) continue;
                    xv[2 * b + 1] = fe_mul(inv, pre[b]);
                    inv = fe_mul(inv, xv[2 * b]);
                }
                uint32_t* d_x = dalloc(2 * B);
                if (!d_x) return BZH_E_OOM;
                PV_TRY(upload(d_x, xv.data(), 2 * B));
                PV_TRY(poly_kate_division(ctx, field, cur, len, B, d_x, nxt));
                std::swap(cur, nxt);
                len--;
            }
            PV_TRY(copy2d(f_parts + si * n * 8, nq * n, cur, len, len, B));
        }
        mark("multiopen_q_kate");
        // f = sum_si x2^(..) f_si (Horner), commit, x3, q evaluations, x4, the opened polynomial
        uint32_t* f_poly = dalloc(B * n);
        uint32_t* p_poly = dalloc(B * n);
        if (!f_poly || !p_poly) return BZH_E_OOM;
        if (nq == 1) {
            PV_TRY(copy2d(f_poly, n, f_parts, n, n, B));
        } else {
            Cols reg;
            for (size_t si = 0; si < nq; si++) reg.add(key(K_MISC, M_H0 + si), f_parts + si * n * 8, nq * n);
            PV_TRY(run(key(43, 0), [&](EPool& ep) {
                std::vector<int> t;
                for (size_t si = 0; si < nq; si++) t.push_back(ep.query((int)si));
                return ep.horner(t, ep.sym(SY_X2));
            }, reg, n, f_poly));
        }
        std::vector<Fe<SF>> f_blinds(B), x3s(B);
        for (size_t b = 0; b < B; b++) f_blinds[b] = draw(b);
        PV_TRY(commit(f_poly, n, B, f_blinds, xy));
        for (size_t b = 0; b < B; b++) {
            bzh_transcript_write_point(T[b], C::id, &xy[b * 8]);
            x3s[b] = squeeze(b);
        }
        std::vector<Fe<SF>> p3(B * nq), v3;
        for (size_t b = 0; b < B; b++)
            for (size_t si = 0; si < nq; si++) p3[b * nq + si] = x3s[b];
        PV_TRY(evals(q_polys, B * nq, p3, v3));
        for (size_t b = 0; b < B; b++) {
            for (size_t si = 0; si < nq; si++) write_scalar(b, v3[b * nq + si]);
            env[b][SY_X4] = squeeze(b);
        }
        {
            Cols reg;
            reg.add(key(K_MISC, M_F), f_poly, n);
            for (size_t si = 0; si < nq; si++) reg.add(key(K_MISC, M_H0 + si), q_polys + si * n * 8, nq * n);
            PV_TRY(run(key(44, 0), [&](EPool& ep) {
                std::vector<int> t;
                for (size_t c = 0; c <= nq; c++) t.push_back(ep.query((int)c));
                return ep.horner(t, ep.sym(SY_X4));
            }, reg, n, p_poly));
        }
        std::vector<uint64_t> p_blinds(B * 4), x3c(B * 4), out_v(B * 4);
        for (size_t b = 0; b < B; b++) {
            Fe<SF> acc = f_blinds[b];
            for (size_t si = 0; si < nq; si++) acc = fe_add(fe_mul(acc, env[b][SY_X4]), q_blinds[b * nq + si]);
            h_store<SF>(&p_blinds[4 * b], fe_from_mont(acc));
            h_store<SF>(&x3c[4 * b], fe_from_mont(x3s[b]));
        }
        mark("multiopen_f_p");
        // the opening draws from each proof's own cursor: a zero stride is not possible, so pass proof 0's cursor and the
        // common distance between the per-proof streams
        const size_t need = 64 * (n + 1 + 2 * (size_t)pk.k);
        if (seeded) {
            uint32_t* raw = (uint32_t*)arena.alloc(B * need);
            if (!raw) return BZH_E_OOM;
            PV_TRY(seed_rows(need / 64, raw));
            PV_TRY(ipa_open(ctx, pk.srs, p_poly, B, p_blinds.data(), x3c.data(), nullptr, need, T.data(), out_v.data(), raw));
        } else {
            std::vector<uint8_t> ipa_rng(B * need);
            for (size_t b = 0; b < B; b++) memcpy(&ipa_rng[b * need], rng[b], need);
            PV_TRY(ipa_open(ctx, pk.srs, p_poly, B, p_blinds.data(), x3c.data(), ipa_rng.data(), need, T.data(), out_v.data()));
        }
    }
    mark("ipa");
    for (size_t b = 0; b < B; b++) {
        const uint8_t* data = nullptr;
        size_t plen = 0;
        PV_TRY(bzh_transcript_proof(T[b], &data, &plen));
        if (plen > proof_stride) return BZH_E_ARG;
        memcpy(proofs + b * proof_stride, data, plen);
        proof_lens[b] = plen;
    }
    return BZH_OK;
}

template <class C>
static int prove_batch_t(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint32_t* d_advice, const uint64_t* instances, size_t inst_rows,
                         const uint8_t* rng, size_t rng_stride, uint8_t* proofs, size_t proof_stride, size_t* proof_lens) {
    Arena& arena = pk->arena_for(ctx, ctx->device);
    arena.reset();
    Prover<C> pv(ctx, *pk, batch, arena);
    if (rng_stride == 0) {  // seeded: rng holds batch x 32 bytes
        pv.seeded = true;
        pv.seed_keys.resize(batch * 8);
        memcpy(pv.seed_keys.data(), rng, batch * 32);
        pv.host_ctr.assign(batch, 0);
        pv.d_seed_keys = (uint32_t*)arena.alloc(batch * 32);
        if (!pv.d_seed_keys) return BZH_E_OOM;
        int rcu = h2d_small(ctx, pv.d_seed_keys, pv.seed_keys.data(), batch * 32);
        if (rcu) return rcu;
    } else {
        for (size_t b = 0; b < batch; b++) pv.rng[b] = rng + b * rng_stride;
    }
    const int rc = pv.prove(d_advice, instances, inst_rows, proofs, proof_stride, proof_lens);
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
}


// ---------------------------------------------------------------------------
// the verifier (halo2_proofs plonk::verify_proof with SingleVerifier, benches/board.rs:80-86): transcript replay,
// expected h(x) from the evaluations, multiopen recombination and the IPA equation.  Host work per proof (threads):
// Blake2b, ~56 point decompressions, a few hundred field operations; device work for the whole batch: the instance
// commitments, one n-term MSM per proof against the SRS table and one small MSM over the proof's own points.
// ---------------------------------------------------------------------------
template <class C>
struct ProofView {
    using SF = typename CurveScalar<C>::SF;
    // outputs of the host pass: left-side linear combination and the right side's (c, u_j)
    std::vector<uint64_t> lc_pts, lc_scal, cu;
    bool ok = false;
};

template <class SF>
static Fe<SF> cx_eval(const bzh_pk& pk, int i, const std::vector<Fe<SF>>& adv, const std::vector<Fe<SF>>& fix,
                      const std::vector<Fe<SF>>& inst) {
    const CNode& e = pk.cx[i];
    auto find = [](const std::vector<std::pair<int, int>>& qs, int col, int rot) -> size_t {
        for (size_t k = 0; k < qs.size(); k++)
            if (qs[k].first == col && qs[k].second == rot) return k;
        return (size_t)-1;
    };
    switch (e.tag) {
        case CX_CONST: {
            Fe<SF> v;
            memcpy(v.l, e.val, 32);
            return v;
        }
        case CX_ADVICE: return adv[find(pk.advice_queries, (int)e.col, e.rot)];
        case CX_FIXED: return fix[find(pk.fixed_queries, (int)e.col, e.rot)];
        case CX_INSTANCE: return inst[find(pk.instance_queries, (int)e.col, e.rot)];
        case CX_NEG: return fe_neg(cx_eval<SF>(pk, e.a, adv, fix, inst));
        case CX_SCALE: {
            Fe<SF> v;
            memcpy(v.l, e.val, 32);
            return fe_mul(cx_eval<SF>(pk, e.a, adv, fix, inst), v);
        }
        case CX_ADD: return fe_add(cx_eval<SF>(pk, e.a, adv, fix, inst), cx_eval<SF>(pk, e.b, adv, fix, inst));
        default: return fe_mul(cx_eval<SF>(pk, e.a, adv, fix, inst), cx_eval<SF>(pk, e.b, adv, fix, inst));
    }
}

// host pass over one proof; inst_xy: this proof's instance commitments.  Returns false on any malformed input.
template <class C>
static bool verify_host(const bzh_pk& pk, const uint64_t* inst_xy, const uint8_t* proof, size_t len, size_t nl_cap,
                        ProofView<C>& out) {
    using SF = typename CurveScalar<C>::SF;
    const int na = pk.na, ni = pk.ni, nsets = pk.nsets, nl = pk.nl, npieces = pk.npieces;
    const size_t n = pk.n, m = pk.perm_columns.size();
    const unsigned k = pk.k;
    bzh_transcript* T = nullptr;
    if (bzh_transcript_new(pk.field, &T)) return false;
    struct Guard {
        bzh_transcript* t;
        ~Guard() { bzh_transcript_free(t); }
    } guard{T};
    size_t off = 0;
    bool bad = false;
    std::vector<uint64_t> pts;     // every point read from the proof, affine canonical
    pts.reserve(((size_t)na + 3 * nl + nsets + npieces + 3 + 2 * (size_t)k + 8) * 8);  // terms keep pointers into it: no regrowth
    auto read_point = [&]() -> size_t {  // index into pts (units of 8 u64)
        const size_t idx = pts.size() / 8;
        pts.resize(pts.size() + 8, 0);
        if (off + 32 > len || !point_decompress(C::id, proof + off, &pts[idx * 8])) {
            bad = true;
            return idx;
        }
        // upstream's Blake2bRead::common_point fails on the identity ("cannot write points at infinity to the
        // transcript"): a proof carrying an identity commitment is rejected, not absorbed as (0, 0)
        {
            uint64_t any = 0;
            for (int i = 0; i < 8; i++) any |= pts[idx * 8 + i];
            if (!any) {
                bad = true;
                return idx;
            }
        }
        off += 32;
        bzh_transcript_common_point(T, &pts[idx * 8]);
        return idx;
    };
    auto read_scalar = [&]() -> Fe<SF> {
        uint64_t l[4] = {0, 0, 0, 0};
        if (off + 32 > len) {
            bad = true;
            return fe_zero<SF>();
        }
        memcpy(l, proof + off, 32);
        off += 32;
        Fe<SF> v = h_load<SF>(l), t = v;
        fe_cond_sub_p(t, 0);
        if (!fe_eq(t, v)) bad = true;  // non-canonical encoding
        bzh_transcript_common_scalar(T, l);
        return fe_to_mont(v);
    };
    auto squeeze = [&]() {
        uint64_t ch[4];
        bzh_transcript_squeeze_challenge(T, ch);
        return fe_to_mont(h_load<SF>(ch));
    };
    bzh_transcript_common_scalar(T, pk.vk_repr);
    for (int i = 0; i < ni; i++) bzh_transcript_common_point(T, inst_xy + 8 * i);
    std::vector<size_t> adv_c(na);
    for (int i = 0; i < na; i++) adv_c[i] = read_point();
    const Fe<SF> theta = squeeze();
    std::vector<size_t> lka(nl), lks(nl), lkz(nl), pz_c(nsets), h_c(npieces);
    for (int i = 0; i < nl; i++) {
        lka[i] = read_point();
        lks[i] = read_point();
    }
    const Fe<SF> beta = squeeze(), gamma = squeeze();
    for (int i = 0; i < nsets; i++) pz_c[i] = read_point();
    for (int i = 0; i < nl; i++) lkz[i] = read_point();
    const size_t rand_c = read_point();
    const Fe<SF> y = squeeze();
    for (int i = 0; i < npieces; i++) h_c[i] = read_point();
    const Fe<SF> x = squeeze();
    if (bad) return false;
    const Fe<SF> one = fe_one<SF>();
    const Fe<SF> xn = h_pow_u64(x, n);
    std::vector<Fe<SF>> inst_ev(pk.instance_queries.size()), adv_ev(pk.advice_queries.size()), fix_ev(pk.fixed_queries.size());
    for (auto& v : inst_ev) v = read_scalar();
    for (auto& v : adv_ev) v = read_scalar();
    for (auto& v : fix_ev) v = read_scalar();
    const Fe<SF> rand_ev = read_scalar();
    std::vector<Fe<SF>> sig_ev(m);
    for (auto& v : sig_ev) v = read_scalar();
    std::vector<Fe<SF>> pz0(nsets), pz1(nsets), pzl(nsets, fe_zero<SF>());
    for (int i = 0; i < nsets; i++) {
        pz0[i] = read_scalar();
        pz1[i] = read_scalar();
        if (i != nsets - 1) pzl[i] = read_scalar();
    }
    std::vector<Fe<SF>> lz0(nl), lz1(nl), la0(nl), lam1(nl), ls0(nl);
    for (int i = 0; i < nl; i++) {
        lz0[i] = read_scalar();
        lz1[i] = read_scalar();
        la0[i] = read_scalar();
        lam1[i] = read_scalar();
        ls0[i] = read_scalar();
    }
    if (bad) return false;
    // Lagrange values at x: l_i(x) = (x^n - 1) w^i / (n (x - w^i))
    Fe<SF> omega;
    {
        uint64_t t[4];
        memcpy(t, pk.omega, 32);
        omega = h_load<SF>(t);
    }
    Fe<SF> nfe = fe_zero<SF>();
    {
        uint64_t t[4] = {(uint64_t)n, 0, 0, 0};
        nfe = fe_to_mont(h_load<SF>(t));
    }
    const Fe<SF> xn1 = fe_sub(xn, one);
    if (fe_is_zero(xn1)) return false;
    auto lag = [&](size_t row) {
        const Fe<SF> wi = h_pow_u64(omega, row);
        return fe_mul(fe_mul(xn1, wi), fe_inv(fe_mul(nfe, fe_sub(x, wi))));
    };
    const Fe<SF> l0 = lag(0), l_last = lag(pk.usable);
    Fe<SF> l_blind = fe_zero<SF>();
    for (size_t r = pk.usable + 1; r < n; r++) l_blind = fe_add(l_blind, lag(r));
    const Fe<SF> active = fe_sub(one, fe_add(l_last, l_blind));
    Fe<SF> delta;
    memcpy(delta.l, pk.delta, 32);
    // the quotient's terms in protocol order, folded with y
    Fe<SF> hacc = fe_zero<SF>();
    auto push = [&](const Fe<SF>& t) { hacc = fe_add(fe_mul(hacc, y), t); };
    for (int g : pk.gates) push(cx_eval<SF>(pk, g, adv_ev, fix_ev, inst_ev));
    auto col_at0 = [&](std::pair<int, int> col) -> Fe<SF> {
        const auto& qs = col.first == CX_ADVICE ? pk.advice_queries : (col.first == CX_FIXED ? pk.fixed_queries : pk.instance_queries);
        const auto& ev = col.first == CX_ADVICE ? adv_ev : (col.first == CX_FIXED ? fix_ev : inst_ev);
        for (size_t q = 0; q < qs.size(); q++)
            if (qs[q].first == col.second && qs[q].second == 0) return ev[q];
        bad = true;
        return fe_zero<SF>();
    };
    if (nsets) {
        push(fe_mul(l0, fe_sub(one, pz0[0])));
        const Fe<SF> zl = pz0[nsets - 1];
        push(fe_mul(l_last, fe_sub(fe_sqr(zl), zl)));
        for (int i = 1; i < nsets; i++) push(fe_mul(l0, fe_sub(pz0[i], pzl[i - 1])));
        Fe<SF> cur_delta = fe_mul(beta, x);
        for (int i = 0; i < nsets; i++) {
            const size_t c0 = (size_t)i * pk.chunk_len, c1 = std::min(m, c0 + pk.chunk_len);
            Fe<SF> left = pz1[i], right = pz0[i];
            for (size_t gj = c0; gj < c1; gj++) {
                const Fe<SF> v = col_at0(pk.perm_columns[gj]);
                left = fe_mul(left, fe_add(fe_add(v, fe_mul(beta, sig_ev[gj])), gamma));
                right = fe_mul(right, fe_add(fe_add(v, cur_delta), gamma));
                cur_delta = fe_mul(cur_delta, delta);
            }
            push(fe_mul(active, fe_sub(left, right)));
        }
    }
    for (int i = 0; i < nl; i++) {
        auto comp = [&](const std::vector<int>& es) {
            Fe<SF> acc = fe_zero<SF>();
            for (int e : es) acc = fe_add(fe_mul(acc, theta), cx_eval<SF>(pk, e, adv_ev, fix_ev, inst_ev));
            return acc;
        };
        push(fe_mul(l0, fe_sub(one, lz0[i])));
        push(fe_mul(l_last, fe_sub(fe_sqr(lz0[i]), lz0[i])));
        const Fe<SF> lhs = fe_mul(fe_mul(lz1[i], fe_add(la0[i], beta)), fe_add(ls0[i], gamma));
        const Fe<SF> rhs = fe_mul(fe_mul(lz0[i], fe_add(comp(pk.lookups[i].first), beta)), fe_add(comp(pk.lookups[i].second), gamma));
        push(fe_mul(active, fe_sub(lhs, rhs)));
        push(fe_mul(l0, fe_sub(la0[i], ls0[i])));
        push(fe_mul(fe_mul(active, fe_sub(la0[i], ls0[i])), fe_sub(la0[i], lam1[i])));
    }
    if (bad) return false;
    const Fe<SF> expected_h = fe_mul(hacc, fe_inv(xn1));

    // multiopen: evaluation of commitment `cid` at rotation r (a permutation product's third rotation is -(blinding + 1)),
    // and its place in the linear combination
    auto eval_of = [&](uint64_t cid, int r) -> Fe<SF> {
        const int kind = (int)(cid >> 32);
        const size_t i = (size_t)(cid & 0xffffffffu);
        auto from = [&](const std::vector<std::pair<int, int>>& qs, const std::vector<Fe<SF>>& ev) {
            for (size_t q = 0; q < qs.size(); q++)
                if (qs[q].first == (int)i && qs[q].second == r) return ev[q];
            bad = true;
            return fe_zero<SF>();
        };
        switch (kind) {
            case K_INST: return from(pk.instance_queries, inst_ev);
            case K_ADV: return from(pk.advice_queries, adv_ev);
            case K_FIX: return from(pk.fixed_queries, fix_ev);
            case K_SIGMA: return sig_ev[i];
            case K_PZ: return r == 0 ? pz0[i] : (r == 1 ? pz1[i] : pzl[i]);
            case K_LZ: return r == 0 ? lz0[i] : lz1[i];
            case K_LA: return r == 0 ? la0[i] : lam1[i];
            case K_LS: return ls0[i];
            default: return i == M_H0 ? expected_h : rand_ev;
        }
    };
    const Fe<SF> x1 = squeeze(), x2 = squeeze();
    const size_t nq = pk.rot_sets.size();
    // left-side linear combination: (point, scalar) pairs; proof / key commitments are weighted later by x4 powers
    struct Term {
        const uint64_t* pt;
        Fe<SF> s;
    };
    std::vector<std::vector<Term>> q_terms(nq);
    std::vector<std::vector<Fe<SF>>> q_evalsets(nq);
    std::vector<Fe<SF>> xn_pows(npieces);
    {
        Fe<SF> pw = one;
        for (int i = 0; i < npieces; i++) {
            xn_pows[i] = pw;
            pw = fe_mul(pw, xn);
        }
    }
    for (size_t si = 0; si < nq; si++) {
        const auto& cids = pk.groups[si];
        const auto& rots = pk.rot_sets[si];
        std::vector<Fe<SF>> evs(rots.size(), fe_zero<SF>());
        for (size_t j = 0; j < cids.size(); j++) {
            for (auto& t : q_terms[si]) t.s = fe_mul(t.s, x1);  // cm = x1 * cm + C
            const uint64_t cid = cids[j];
            const int kind = (int)(cid >> 32);
            const size_t i = (size_t)(cid & 0xffffffffu);
            auto add_term = [&](const uint64_t* pt, const Fe<SF>& s) { q_terms[si].push_back({pt, s}); };
            switch (kind) {
                case K_INST: add_term(inst_xy + 8 * i, one); break;
                case K_ADV: add_term(&pts[adv_c[i] * 8], one); break;
                case K_FIX: add_term(&pk.fixed_commitments[8 * i], one); break;
                case K_SIGMA: add_term(&pk.sigma_commitments[8 * i], one); break;
                case K_PZ: add_term(&pts[pz_c[i] * 8], one); break;
                case K_LZ: add_term(&pts[lkz[i] * 8], one); break;
                case K_LA: add_term(&pts[lka[i] * 8], one); break;
                case K_LS: add_term(&pts[lks[i] * 8], one); break;
                default:
                    if (i == M_H0) {
                        for (int pi = 0; pi < npieces; pi++) add_term(&pts[h_c[pi] * 8], xn_pows[pi]);
                    } else {
                        add_term(&pts[rand_c * 8], one);
                    }
            }
            for (size_t t = 0; t < rots.size(); t++) evs[t] = fe_add(fe_mul(evs[t], x1), eval_of(cid, rots[t]));
        }
        q_evalsets[si] = evs;
    }
    if (bad) return false;
    const size_t f_commit = read_point();
    const Fe<SF> x3 = squeeze();
    std::vector<Fe<SF>> q_evals(nq);
    for (auto& v : q_evals) v = read_scalar();
    if (bad) return false;
    Fe<SF> omega_inv = fe_inv(omega);
    auto rot = [&](int r) { return fe_mul(x, r >= 0 ? h_pow_u64(omega, (uint64_t)r) : h_pow_u64(omega_inv, (uint64_t)(-(int64_t)r))); };
    Fe<SF> f_eval = fe_zero<SF>();
    for (size_t si = 0; si < nq; si++) {
        const auto& rots = pk.rot_sets[si];
        const size_t np = rots.size();
        std::vector<Fe<SF>> ptv(np);
        for (size_t t = 0; t < np; t++) ptv[t] = rot(rots[t]);
        // r(x3) by Lagrange's formula on (points, evals)
        Fe<SF> r_eval = fe_zero<SF>(), den = one;
        for (size_t j = 0; j < np; j++) {
            Fe<SF> num = one, dn = one;
            for (size_t mm = 0; mm < np; mm++) {
                if (mm == j) continue;
                num = fe_mul(num, fe_sub(x3, ptv[mm]));
                dn = fe_mul(dn, fe_sub(ptv[j], ptv[mm]));
            }
            if (fe_is_zero(dn)) return false;
            r_eval = fe_add(r_eval, fe_mul(q_evalsets[si][j], fe_mul(num, fe_inv(dn))));
            den = fe_mul(den, fe_sub(x3, ptv[j]));
        }
        if (fe_is_zero(den)) return false;
        f_eval = fe_add(fe_mul(f_eval, x2), fe_mul(fe_sub(q_evals[si], r_eval), fe_inv(den)));
    }
    const Fe<SF> x4 = squeeze();
    // final commitment = x4^nq f + sum_si x4^(nq-1-si) q_si, final value likewise
    Fe<SF> final_v = f_eval;
    for (size_t si = 0; si < nq; si++) final_v = fe_add(fe_mul(final_v, x4), q_evals[si]);
    std::vector<Fe<SF>> x4p(nq + 1);
    x4p[0] = one;
    for (size_t i = 1; i <= nq; i++) x4p[i] = fe_mul(x4p[i - 1], x4);
    std::vector<Term> lc;
    lc.push_back({&pts[f_commit * 8], x4p[nq]});
    for (size_t si = 0; si < nq; si++)
        for (auto& t : q_terms[si]) lc.push_back({t.pt, fe_mul(t.s, x4p[nq - 1 - si])});
    // the opening argument: S, xi, z, (L_j, R_j, u_j), c, f
    const size_t S = read_point();
    const Fe<SF> xi = squeeze(), z = squeeze();
    std::vector<size_t> Ls(k), Rs(k);
    std::vector<Fe<SF>> us(k);
    for (unsigned j = 0; j < k; j++) {
        Ls[j] = read_point();
        Rs[j] = read_point();
        us[j] = squeeze();
        if (fe_is_zero(us[j])) bad = true;
    }
    if (bad || off + 64 != len) return false;
    uint64_t cl[4], fl[4];
    memcpy(cl, proof + off, 32);
    memcpy(fl, proof + off + 32, 32);
    Fe<SF> cc = h_load<SF>(cl), ff = h_load<SF>(fl);
    {
        Fe<SF> t = cc, t2 = ff;
        fe_cond_sub_p(t, 0);
        fe_cond_sub_p(t2, 0);
        if (!fe_eq(t, cc) || !fe_eq(t2, ff)) return false;
    }
    const Fe<SF> cm = fe_to_mont(cc), fm = fe_to_mont(ff);
    std::vector<Fe<SF>> xp(k ? k : 1);
    if (k) {
        xp[0] = x3;
        for (unsigned i = 1; i < k; i++) xp[i] = fe_sqr(xp[i - 1]);
    }
    Fe<SF> b0 = one;
    for (unsigned j = 0; j < k; j++) b0 = fe_mul(b0, fe_add(one, fe_mul(us[j], xp[k - 1 - j])));
    // batch-invert the u_j
    std::vector<Fe<SF>> pre(k + 1);
    pre[0] = one;
    for (unsigned j = 0; j < k; j++) pre[j + 1] = fe_mul(pre[j], us[j]);
    Fe<SF> inv = fe_inv(pre[k]);
    std::vector<Fe<SF>> uinv(k);
    for (unsigned j = k; j-- > 0;) {
        uinv[j] = fe_mul(inv, pre[j]);
        inv = fe_mul(inv, us[j]);
    }
    for (unsigned j = 0; j < k; j++) {
        lc.push_back({&pts[Ls[j] * 8], uinv[j]});
        lc.push_back({&pts[Rs[j] * 8], us[j]});
    }
    lc.push_back({&pts[S * 8], xi});
    // G_0, U, W are the first and the last two SRS points: supplied by the caller right after this table
    out.lc_pts.assign(nl_cap * 8, 0);
    out.lc_scal.assign(nl_cap * 4, 0);
    if (lc.size() + 3 > nl_cap) return false;
    size_t o = 0;
    for (auto& t : lc) {
        memcpy(&out.lc_pts[o * 8], t.pt, 64);
        h_store<SF>(&out.lc_scal[o * 4], fe_from_mont(t.s));
        o++;
    }
    // scalars of G_0 (-v), U (-c b0 z), W (-f): points filled in by the caller (slots nl_cap-3 .. nl_cap-1)
    h_store<SF>(&out.lc_scal[(nl_cap - 3) * 4], fe_from_mont(fe_neg(final_v)));
    h_store<SF>(&out.lc_scal[(nl_cap - 2) * 4], fe_from_mont(fe_neg(fe_mul(fe_mul(cm, b0), z))));
    h_store<SF>(&out.lc_scal[(nl_cap - 1) * 4], fe_from_mont(fe_neg(fm)));
    out.cu.assign((size_t)(k + 1) * 4, 0);
    h_store<SF>(&out.cu[0], fe_from_mont(cm));
    for (unsigned j = 0; j < k; j++) h_store<SF>(&out.cu[(j + 1) * 4], fe_from_mont(us[j]));
    out.ok = true;
    return true;
}

template <class C>
static int verify_batch_t(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* instances, size_t inst_rows, const uint8_t* proofs,
                          size_t proof_stride, const size_t* proof_lens, const uint64_t* g0_u_w, int* results) {
    using SF = typename CurveScalar<C>::SF;
    Arena& arena = pk->arena_for(ctx, ctx->device);
    arena.reset();
    Prover<C> pv(ctx, *pk, batch, arena);
    const size_t n = pk->n, B = batch;
    const int ni = pk->ni;
    std::vector<uint64_t> xy;
    std::vector<Fe<SF>> blinds;
    {
        std::lock_guard<std::mutex> lkv(pk->mu);
        if (!pk->vk_ready) {  // verifying key: commitments to the fixed and permutation polynomials, blind 1
            const size_t nf = pk->nf, m = pk->perm_columns.size();
            blinds.assign(nf, fe_one<SF>());
            PV_TRY(pv.commit(pk->fixed_polys, n, nf, blinds, pk->fixed_commitments));
            blinds.assign(m, fe_one<SF>());
            PV_TRY(pv.commit(pk->sigma_polys, n, m, blinds, pk->sigma_commitments));
            pk->vk_ready = true;
        }
    }
    // instance commitments of the whole batch (the verifier recomputes them, as upstream does for IPA)
    std::vector<uint64_t> inst_xy(B * std::max(ni, 1) * 8, 0);
    if (ni) {
        uint32_t* inst = pv.dalloc(B * ni * n);
        uint32_t* inst_polys = pv.dalloc(B * ni * n);
        if (!inst || !inst_polys) return BZH_E_OOM;
        PV_TRY(pv.zero(inst, B * ni * n));
        if (inst_rows) {
            std::vector<Fe<SF>> hv(B * ni * inst_rows);
            for (size_t i = 0; i < hv.size(); i++) hv[i] = fe_to_mont(h_load<SF>(instances + 4 * i));
            uint32_t* tmp = pv.dalloc(hv.size());
            if (!tmp) return BZH_E_OOM;
            PV_TRY(pv.upload(tmp, hv.data(), hv.size()));
            PV_TRY(pv.copy2d(inst, n, tmp, inst_rows, inst_rows, B * ni));
        }
        PV_TRY(pv.to_coeff(inst_polys, inst, B * ni));
        blinds.assign(B * ni, fe_one<SF>());
        PV_TRY(pv.commit(inst_polys, n, B * ni, blinds, inst_xy));
    }
    // host pass, one thread per proof
    const size_t ncommit = (size_t)pk->na + 3 * pk->nl + pk->nsets + 1 + pk->npieces + 1 + pk->nf + pk->perm_columns.size() + ni;
    const size_t nl_cap = ncommit + 2 * (size_t)pk->k + 1 + 3 + 4;
    std::vector<ProofView<C>> views(B);
    {
        const size_t nthreads = std::min<size_t>({B, (size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)32});
        std::vector<std::thread> th;
        for (size_t t = 0; t < nthreads; t++)
            th.emplace_back([&, t]() {
                for (size_t b = t; b < B; b += nthreads)
                    verify_host<C>(*pk, &inst_xy[b * std::max(ni, 1) * 8], proofs + b * proof_stride, proof_lens[b], nl_cap, views[b]);
            });
        for (auto& t : th) t.join();
    }
    // device pass over the proofs that parsed; the others are rejected outright
    std::vector<size_t> live;
    for (size_t b = 0; b < B; b++) {
        results[b] = 0;
        if (views[b].ok) live.push_back(b);
    }
    if (live.empty()) return BZH_OK;
    const size_t Bl = live.size(), kk = pk->k;
    std::vector<uint64_t> lc_pts(Bl * nl_cap * 8), lc_scal(Bl * nl_cap * 4), cu(Bl * (kk + 1) * 4);
    for (size_t j = 0; j < Bl; j++) {
        ProofView<C>& v = views[live[j]];
        memcpy(&v.lc_pts[(nl_cap - 3) * 8], g0_u_w, 3 * 64);
        memcpy(&lc_pts[j * nl_cap * 8], v.lc_pts.data(), nl_cap * 64);
        memcpy(&lc_scal[j * nl_cap * 4], v.lc_scal.data(), nl_cap * 32);
        memcpy(&cu[j * (kk + 1) * 4], v.cu.data(), (kk + 1) * 32);
    }
    std::vector<int> ok(Bl, 0);
    PV_TRY(ipa_check_batch(ctx, pk->srs, Bl, nl_cap, lc_pts.data(), lc_scal.data(), cu.data(), ok.data()));
    for (size_t j = 0; j < Bl; j++) results[live[j]] = ok[j];
    return BZH_OK;
}

}  // namespace
}  // namespace bzh

extern "C" {

int bzh_pk_create(bzh_ctx* ctx, const bzh_bases* srs, const uint8_t* circuit, size_t circuit_len, bzh_pk** out) {
    if (!ctx || !srs || !circuit || !out || srs->device != ctx->device) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    switch (srs->curve) {
        case BZH_CURVE_VESTA: return bzh::pk_create_t<bzh::VestaCurve>(ctx, srs, circuit, circuit_len, out);
        case BZH_CURVE_PALLAS: return bzh::pk_create_t<bzh::PallasCurve>(ctx, srs, circuit, circuit_len, out);
    }
    return BZH_E_ARG;  // BN254 has no cube root of unity in Fr's multiplicative generator convention used here
}

int bzh_pk_free(bzh_ctx* ctx, bzh_pk* pk) {
    if (!ctx || !pk) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (pk->dev) (void)hipFree(pk->dev);
    if (pk->hoist) (void)hipFree(pk->hoist);
    if (pk->q_module) (void)hipModuleUnload(pk->q_module);
    for (auto& kv : pk->arenas) kv.second->release();
    delete pk;
    return BZH_OK;
}

int bzh_pk_set_lagrange(bzh_pk* pk, const bzh_bases* g_lagrange) {
    if (!pk) return BZH_E_ARG;
    if (g_lagrange && (g_lagrange->n != pk->n + 2 || g_lagrange->curve != pk->curve || g_lagrange->device != pk->device || !g_lagrange->pre_c))
        return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(pk->mu);
    pk->srs_lagrange = g_lagrange;
    return BZH_OK;
}

int bzh_pk_quotient_stats(bzh_pk* pk, uint32_t* ops, uint32_t* multiplications, uint32_t* lds_slots, uint32_t* hoisted_columns) {
    if (!pk) return BZH_E_ARG;
    uint32_t no = 0, nm = 0, nl = 0;
    if (pk->q_ok) {
        no = (uint32_t)pk->qprog.ops.size();
        nl = (uint32_t)pk->qprog.nlds;
        for (auto& o : pk->qprog.ops) nm += ((o.code >> 4) < 3 && ((o.code >> 2) & 3) == bzh::V2_MUL);
    }
    if (ops) *ops = no;
    if (multiplications) *multiplications = nm;
    if (lds_slots) *lds_slots = nl;
    if (hoisted_columns) *hoisted_columns = (uint32_t)pk->hoist_cols;
    return BZH_OK;
}

static int copy_text(const std::string& src, char* buf, size_t cap, size_t* len) {
    *len = src.size();
    if (buf && cap) {
        const size_t n = std::min(cap - 1, src.size());
        memcpy(buf, src.data(), n);
        buf[n] = 0;
    }
    return BZH_OK;
}

int bzh_pk_quotient_source(bzh_pk* pk, char* buf, size_t cap, size_t* len) {
    if (!pk || !len) return BZH_E_ARG;
    if (!pk->q_ok) return BZH_E_RANGE;   // the circuit does not fit VM v2
    return copy_text(bzh::program2_source(pk->qprog, pk->field), buf, cap, len);
}

int bzh_quotient_source_for_circuit(int curve, const uint8_t* circuit, size_t circuit_len, char* buf, size_t cap, size_t* len,
                                    uint64_t* program_hash) {
    if (!circuit || !len) return BZH_E_ARG;
    bzh_pk pk;
    int rc = BZH_E_ARG;
    if (curve == BZH_CURVE_VESTA) {
        bzh::ParsedKey<bzh::CurveScalar<bzh::VestaCurve>::SF> po;
        rc = bzh::pk_parse_t<bzh::VestaCurve>(circuit, circuit_len, pk, po);
    } else if (curve == BZH_CURVE_PALLAS) {
        bzh::ParsedKey<bzh::CurveScalar<bzh::PallasCurve>::SF> po;
        rc = bzh::pk_parse_t<bzh::PallasCurve>(circuit, circuit_len, pk, po);
    }
    if (rc) return rc;
    if (!pk.q_ok) return BZH_E_RANGE;
    if (program_hash) *program_hash = pk.q_hash;
    return copy_text(bzh::program2_source(pk.qprog, pk.field, true), buf, cap, len);
}

int bzh_pk_quotient_select(bzh_pk* pk, int flavour) {
    if (!pk) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(pk->mu);
    switch (flavour) {
        case BZH_QUOTIENT_INTERPRETER: break;
        case BZH_QUOTIENT_BUILTIN:
            if (!pk->q_builtin) return BZH_E_RANGE;
            break;
        case BZH_QUOTIENT_MODULE:
            if (!pk->q_fn) return BZH_E_RANGE;
            break;
        default: return BZH_E_ARG;
    }
    pk->q_select = flavour;
    return BZH_OK;
}

int bzh_pk_quotient_selected(bzh_pk* pk, int* flavour, int* builtin_available) {
    if (!pk) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(pk->mu);
    if (flavour) *flavour = pk->q_select;
    if (builtin_available) *builtin_available = pk->q_builtin != nullptr;
    return BZH_OK;
}

int bzh_pk_set_quotient_module(bzh_ctx* ctx, bzh_pk* pk, const void* code_object, size_t len) {
    if (!ctx || !pk || pk->device != ctx->device) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::lock_guard<std::mutex> lkp(pk->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (pk->q_module) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipModuleUnload(pk->q_module);
        pk->q_module = nullptr;
        pk->q_fn = nullptr;
    }
    if (pk->q_select == BZH_QUOTIENT_MODULE) pk->q_select = pk->q_builtin ? BZH_QUOTIENT_BUILTIN : BZH_QUOTIENT_INTERPRETER;
    if (!code_object || !len) return BZH_OK;   // back to the key's default
    if (!pk->q_ok) return BZH_E_RANGE;
    hipModule_t mod = nullptr;
    BZH_HIP_TRY(ctx, hipModuleLoadData(&mod, code_object));
    hipFunction_t fn = nullptr;
    hipDeviceptr_t hsym = nullptr;
    size_t hbytes = 0;
    unsigned long long have = 0;
    const bool ok = hipModuleGetFunction(&fn, mod, "jit_quotient") == hipSuccess &&
                    hipModuleGetGlobal(&hsym, &hbytes, mod, "jit_program_hash") == hipSuccess && hbytes == 8 &&
                    hipMemcpy(&have, hsym, 8, hipMemcpyDeviceToHost) == hipSuccess && have == pk->q_hash;
    if (!ok) {
        (void)hipModuleUnload(mod);
        ctx->last_error = "bzh_pk_set_quotient_module: not a module generated from this key's quotient program";
        return BZH_E_ARG;
    }
    pk->q_module = mod;
    pk->q_fn = fn;
    pk->q_select = BZH_QUOTIENT_MODULE;
    return BZH_OK;
}

int bzh_pk_info(const bzh_pk* pk, size_t* rng_bytes_per_proof, size_t* max_proof_bytes, uint32_t* num_advice, uint32_t* n_rows,
                uint32_t* usable_rows) {
    if (!pk) return BZH_E_ARG;
    if (rng_bytes_per_proof) *rng_bytes_per_proof = pk->rng_bytes;
    if (max_proof_bytes) {
        const size_t points = (size_t)pk->na + 2 * pk->nl + pk->nsets + pk->nl + 1 + pk->npieces + 1 + 1 + 2 * (size_t)pk->k;
        const size_t scalars = pk->instance_queries.size() + pk->advice_queries.size() + pk->fixed_queries.size() + 1 +
                               pk->perm_columns.size() + 3 * (size_t)pk->nsets + 5 * (size_t)pk->nl + pk->rot_sets.size() + 2;
        *max_proof_bytes = 32 * (points + scalars);
    }
    if (num_advice) *num_advice = (uint32_t)pk->na;
    if (n_rows) *n_rows = (uint32_t)pk->n;
    if (usable_rows) *usable_rows = (uint32_t)pk->usable;
    return BZH_OK;
}

int bzh_verify_batch(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* instances, size_t instance_rows, const uint8_t* proofs,
                     size_t proof_stride, const size_t* proof_lens, const uint64_t* g0_u_w, int* results) {
    if (!ctx || !pk || !batch || batch > 4096 || !proofs || !proof_lens || !g0_u_w || !results) return BZH_E_ARG;
    if (pk->device != ctx->device || (pk->ni && instance_rows && !instances) || instance_rows > pk->usable) return BZH_E_ARG;
    for (size_t b = 0; b < batch; b++)
        if (proof_lens[b] > proof_stride) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // The commitments of the key and G'_0 are computed against pk->srs: the three points the caller passes must be the
    // same SRS's, or every valid proof would be rejected without an error.  Row 0 of the window table is the raw SRS.
    std::unique_lock<std::mutex> lkp(pk->mu);
    if (pk->srs_g0_u_w.empty()) {
        uint64_t m[3 * 8];
        const size_t idx[3] = {0, pk->n, pk->n + 1};
        for (int i = 0; i < 3; i++)
            BZH_HIP_TRY(ctx, hipMemcpyAsync(&m[i * 8], pk->srs->d_xy + idx[i] * 16, 64, hipMemcpyDeviceToHost, ctx->stream));
        BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint64_t> canon(24);
        for (int i = 0; i < 6; i++) {
            if (pk->curve == BZH_CURVE_VESTA) bzh::h_store<bzh::VestaCurve::Base>(&canon[i * 4], bzh::fe_from_mont(bzh::h_load<bzh::VestaCurve::Base>(&m[i * 4])));
            else bzh::h_store<bzh::PallasCurve::Base>(&canon[i * 4], bzh::fe_from_mont(bzh::h_load<bzh::PallasCurve::Base>(&m[i * 4])));
        }
        pk->srs_g0_u_w = std::move(canon);
    }
    if (memcmp(pk->srs_g0_u_w.data(), g0_u_w, 3 * 64) != 0) {
        ctx->last_error = "bzh_verify_batch: g0_u_w are not G_0, U, W of the SRS this key was built on";
        return BZH_E_ARG;
    }
    lkp.unlock();
    int rc = BZH_E_ARG;
    switch (pk->curve) {
        case BZH_CURVE_VESTA:
            rc = bzh::verify_batch_t<bzh::VestaCurve>(ctx, pk, batch, instances, instance_rows, proofs, proof_stride, proof_lens, g0_u_w, results);
            break;
        case BZH_CURVE_PALLAS:
            rc = bzh::verify_batch_t<bzh::PallasCurve>(ctx, pk, batch, instances, instance_rows, proofs, proof_stride, proof_lens, g0_u_w, results);
            break;
    }
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
}

// rng_stride == 0: `rng` holds batch x 32-byte seeds (bzh_prove_batch_seeded)
static int prove_batch_entry(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* advice, int form, int mem, const uint64_t* instances,
                             size_t instance_rows, const uint8_t* rng, size_t rng_stride, uint8_t* proofs, size_t proof_stride,
                             size_t* proof_lens) {
    if (!ctx || !pk || !batch || batch > 4096 || !advice || !rng || !proofs || !proof_lens) return BZH_E_ARG;
    if ((form != BZH_FORM_CANONICAL && form != BZH_FORM_MONTGOMERY) || (mem != BZH_MEM_HOST && mem != BZH_MEM_DEVICE)) return BZH_E_ARG;
    if (pk->device != ctx->device || (rng_stride != 0 && rng_stride < pk->rng_bytes) || (pk->ni && instance_rows && !instances) ||
        instance_rows > pk->usable)
        return BZH_E_ARG;
    if (mem == BZH_MEM_DEVICE && form != BZH_FORM_MONTGOMERY) return BZH_E_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BZH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t* d_adv = (const uint32_t*)advice;
    void* staged = nullptr;
    const size_t elems = batch * (size_t)pk->na * pk->n;
    if (mem == BZH_MEM_HOST) {
        int rc = bzh::ws_ensure(ctx, 3, elems * 32 + 256, &staged);
        if (rc) return rc;
        BZH_HIP_TRY(ctx, hipMemcpyAsync(staged, advice, elems * 32, hipMemcpyHostToDevice, ctx->stream));
        if (form == BZH_FORM_CANONICAL && (rc = bzh::field_convert(ctx, pk->field, (uint32_t*)staged, elems, 1))) return rc;
        d_adv = (const uint32_t*)staged;
    }
    switch (pk->curve) {
        case BZH_CURVE_VESTA:
            return bzh::prove_batch_t<bzh::VestaCurve>(ctx, pk, batch, d_adv, instances, instance_rows, rng, rng_stride, proofs, proof_stride,
                                                       proof_lens);
        case BZH_CURVE_PALLAS:
            return bzh::prove_batch_t<bzh::PallasCurve>(ctx, pk, batch, d_adv, instances, instance_rows, rng, rng_stride, proofs, proof_stride,
                                                        proof_lens);
    }
    return BZH_E_ARG;
}

int bzh_prove_batch(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* advice, int form, int mem, const uint64_t* instances,
                    size_t instance_rows, const uint8_t* rng, size_t rng_stride, uint8_t* proofs, size_t proof_stride,
                    size_t* proof_lens) {
    if (rng_stride == 0) return BZH_E_ARG;
    return prove_batch_entry(ctx, pk, batch, advice, form, mem, instances, instance_rows, rng, rng_stride, proofs, proof_stride, proof_lens);
}

int bzh_prove_batch_seeded(bzh_ctx* ctx, bzh_pk* pk, size_t batch, const uint64_t* advice, int form, int mem, const uint64_t* instances,
                           size_t instance_rows, const uint8_t* seeds, uint8_t* proofs, size_t proof_stride, size_t* proof_lens) {
    return prove_batch_entry(ctx, pk, batch, advice, form, mem, instances, instance_rows, seeds, 0, proofs, proof_stride, proof_lens);
}

int bzh_rng_expand(const uint8_t* seed, uint64_t first_draw, size_t draws, uint8_t* out) {
    if (!seed || (!out && draws)) return BZH_E_ARG;
    uint32_t key[8], blk[16];
    memcpy(key, seed, 32);
    for (size_t i = 0; i < draws; i++) {
        bzh::chacha20_block(key, first_draw + i, blk);
        memcpy(out + i * 64, blk, 64);
    }
    return BZH_OK;
}

}  // extern "C"
