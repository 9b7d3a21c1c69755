// Part of the whole-proof translation unit (csrc/prove.hip): the COMPILER -- circuit expressions as serialised, the evaluator's
// expression pool (hash-consed DAG), the VM v1 compiler, the VM v2 quotient compiler (y-folding by gate, shared subexpressions,
// hoisting), the program's hash and its straight-line HIP source, the column registry, the device arena and the blob reader.
// Host code only.  Included inside namespace bzh { namespace { (prove_kernels.cuh opens them).
#pragma once
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
static constexpr int kV2Regs = 4, kV2LdsAcc = 0, kV2LdsInner = 1, kV2LdsCse0 = 2, kV2LdsCseMax = 8, kV2LdsSpills = 2, kV2LdsGlobalMax = 6;
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

// The same program over UNSATURATED limbs (csrc/fe29.cuh): values are 9 x 29-bit limbs in R' = 2^261 Montgomery form, not kept
// canonical.  For every value the emitter tracks, while it writes the straight-line code, a bound V in multiples of p and bounds
// L, T on the limbs 0..7 and on limb 8:
//   leaves / constants  carried, V = 2;   a product  V = 2, limbs < 2^29;
//   a + b   limb-wise, NO carry pass (V, L, T add);   a - b   a + K p - b limb-wise, no carry pass either (fe29_sub_lazy: K p written
//   as J copies of (K / J) p so that it dominates b limb by limb; V grows by K, L by J 1.5 2^30);
//   a carry pass (fe29_carry, ~25 instructions) only where a consumer needs it: a product whose operands' limb bounds multiply past
//   1.8e18 (a column of the schoolbook product is 9 such terms + 4 reduction terms of 2^58 in one 64-bit accumulator), a sum or
//   difference that would pass 2^32, a subtrahend above 2 (2^30 - 2), a value parked in a shared-subexpression slot;
//   a fold below 2 p (fe29_fold, ~45 instructions) where bounds would multiply past 100 or add past 100 (the limbs hold 128 p).
// Board: 598 carry passes (one per + and -, the first version) -> 234.  A bound the emitter cannot establish fails the generation
// (empty source), never the arithmetic.  Columns are fe29 planes (fe29_load_planes), constants 12 words apart; the result is
// converted back to the saturated form by fe29_to_sat_div32 as it is stored, so h is what the saturated kernel writes, bit for bit.
static std::string program2_source29_policy(const Program2& pg, int field, bool carry_at_store, int* passes) {
    std::string src;
    int npass = 0;          // carry passes + folds emitted (what the two store policies are compared by)
    bool pg_fail = false;   // a bound the emitter could not establish: no source (the saturated kernel is then the only flavour)
    char buf[640];
    auto add = [&](const char* fmt, auto... a) {
        snprintf(buf, sizeof(buf), fmt, a...);
        src += buf;
    };
    const unsigned long long hash = (unsigned long long)program2_hash(pg, field);
    char kname[64];
    snprintf(kname, sizeof(kname), "bzh_quotient29_%016llx", hash);
    add("// generated by libbzh2 (bzh_quotient_source_for_circuit): quotient evaluator in unsaturated limbs, %zu instructions\n", pg.ops.size());
    add("namespace bzh_q29_%016llx {\n", hash);
    src += "using namespace bzh;\n";
    add("typedef %s P;\n", field == BZH_FIELD_FQ ? "FqParams" : "FpParams");
    // (out of line like the saturated flavour's, for the instruction cache; operands as 9-lane vectors: a 36-byte struct would
    // cross the call through scratch memory)
    src += "__device__ __noinline__ fe29_vec mulv(fe29_vec a, fe29_vec b) { return fe29_pack_vec(fe29_mul(fe29_unpack_vec<P>(a), fe29_unpack_vec<P>(b))); }\n"
           "__device__ __forceinline__ Fe29<P> mulx(const Fe29<P>& a, const Fe29<P>& b) { return fe29_unpack_vec<P>(mulv(fe29_pack_vec(a), fe29_pack_vec(b))); }\n"
           // a b + c d with ONE reduction; d is a per-proof constant the callee fetches itself (four 9-lane operands would not fit the
           // 32 argument registers), first thing, so that the a b multiply-adds cover the load
           "__device__ __noinline__ fe29_vec dot2v(fe29_vec a, fe29_vec b, fe29_vec c, fe29_gbytes d) {\n"
           "    const Fe29<P> dd = fe29_load_const_g<P>(d);\n"
           "    return fe29_pack_vec(fe29_dot2(fe29_unpack_vec<P>(a), fe29_unpack_vec<P>(b), fe29_unpack_vec<P>(c), dd));\n}\n"
           "__device__ __forceinline__ Fe29<P> dot2x(const Fe29<P>& a, const Fe29<P>& b, const Fe29<P>& c, const uint32_t* d) {\n"
           "    return fe29_unpack_vec<P>(dot2v(fe29_pack_vec(a), fe29_pack_vec(b), fe29_pack_vec(c), (fe29_gbytes)d));\n}\n";
    // (132 VGPRs, three waves per SIMD: forcing four with amdgpu_waves_per_eu(4, 4) -- 128 VGPRs, 3 spilled -- measured slower,
    // 34.2 against 33.4 ms per batch of 64)
    // A scheduling barrier after every operation keeps the compiler from hoisting the column loads of the whole program to its top.
    // BZH_QUOTIENT29_NO_BARRIERS=1 (generation time) drops them and caps the kernel at three waves' registers instead: LLVM then
    // merges the program's repeated products (mulv is pure: Board 565 -> 426 calls) but spills ~1 900 words per row to scratch
    // -- measured 30.2 vs 31.1 ms for Board and ShotCircuit's throughput -4 %: not the default.
    static const bool barriers = getenv("BZH_QUOTIENT29_NO_BARRIERS") == nullptr;
    add("extern \"C\" __global__ void __launch_bounds__(128) %s%s(const uint32_t* const* __restrict__ cols, ",
        barriers ? "" : "__attribute__((amdgpu_waves_per_eu(3, 3))) ", kname);
    src += "const size_t* __restrict__ strides, const uint32_t* __restrict__ consts, size_t const_stride, size_t size, "
           "uint32_t* __restrict__ out) {\n"
           "    const size_t r = blockIdx.x * (size_t)128 + threadIdx.x, v = blockIdx.y;\n"
           "    if (r >= size) return;\n"
           "    const size_t mask = size - 1;\n"
           "    const uint32_t* cv = consts + v * const_stride * 12;\n"
           "    Fe29<P> r0 = fe29_zero<P>(), r1 = r0, r2 = r0, r3 = r0;\n";
    const int nslots = std::max(pg.nlds, 1);
    for (int i = 0; i < nslots; i++) add("    Fe29<P> s%d = r0;\n", i);
    // what is known about a value: V (its bound in multiples of p), L (limbs 0..7 at most L), T (limb 8 at most T)
    struct Bnd {
        double V;
        uint64_t L, T;
    };
    static const bool eager = getenv("BZH_QUOTIENT29_EAGER_CARRY") != nullptr;   // (measurement: the carry pass after every + and -)
    constexpr uint64_t kCarried = ((uint64_t)1 << 29) + 8;                          // fe29_carry's output: limbs 0..7 below this
    constexpr uint64_t kBiasLimb = ((uint64_t)1 << 30) + ((uint64_t)1 << 29);       // a limb of a bias K p stays below this
    constexpr double kColumn = 1.8e18;                                              // A * B of a product's limb bounds (header)
    const Bnd zero{0.0, 0, 0}, leaf{2.0, kCarried, ((uint64_t)1 << 23) + 16}, product{2.0, ((uint64_t)1 << 29) - 1, (uint64_t)1 << 23};   // (a product is below 1.8 p: limb 8 < 2^23)
    std::vector<Bnd> bs((size_t)nslots, zero);
    Bnd br[4] = {zero, zero, zero, zero};
    const size_t nops = pg.ops.size();
    auto is_mem = [](int kind) { return kind == BZH_EXPR_COLUMN || kind == BZH_EXPR_CONST; };
    auto emit_load = [&](const char* name, size_t i, int kind, int idx, int rot) {
        if (kind == BZH_EXPR_COLUMN)
            add("    const Fe29<P> %s%zu = fe29_load_planes_g<P>((fe29_gbytes)(cols[%d] + v * strides[%d]), (uint32_t)((r + (size_t)(long)(%d)) & mask), size);\n", name, i, idx, idx, rot);
        else if (kind == BZH_EXPR_CONST)
            add("    const Fe29<P> %s%zu = fe29_load_const<P>(cv + %d * 12);\n", name, i, idx);
    };
    // ---- fused pairs: X = (product) ; Y = slot * constant ; X + Y  ->  one dot2 with one reduction.  Two shapes come out of
    //      Compiler2::quotient: [rP = a * b] [rP+1 = ACC * y^m] [rP += rP+1] (adjacent), and Horner's [rP = IN * y] ... [rP+1 = a * b]
    //      [rP += rP+1] (the first product is deferred to the second: its operands are a slot nothing stores to in between and a constant).
    struct Fuse {
        int slot = -1, cst = -1, dst = -1;   // at the surviving product: the other product's slot and constant, the register that gets the sum
    };
    std::vector<Fuse> fuse(nops);
    std::vector<char> skip(nops, 0);
    static const bool no_dot2 = getenv("BZH_QUOTIENT29_NO_DOT2") != nullptr;
    {
        auto fm = [&](size_t i) { return pg.ops[i].code >> 4; };
        auto opc = [&](size_t i) { return (pg.ops[i].code >> 2) & 3; };
        auto ps = [&](size_t i) { return pg.ops[i].code & 3; };
        auto is_mul = [&](size_t i) { return fm(i) != V2_UN && opc(i) == V2_MUL; };
        auto slot_times_const = [&](size_t i) {
            return is_mul(i) && fm(i) == V2_LL && pg.ops[i].a_kind == BZH_EXPR_LDS && pg.ops[i].b_kind == BZH_EXPR_CONST;
        };
        auto reads = [&](size_t i, int p) {
            if (fm(i) == V2_SS) return ps(i) == p || ps(i) + 1 == p;
            if (fm(i) == V2_SL) return ps(i) == p;
            if (fm(i) == V2_UN) return opc(i) != V2_LOAD && ps(i) == p;
            return false;
        };
        auto writes = [&](size_t i, int p) { return !(fm(i) == V2_UN && opc(i) == V2_STORE) && ps(i) == p; };
        for (size_t k = 2; k < nops && !no_dot2; k++) {
            if (!(fm(k) == V2_SS && opc(k) == V2_ADD) || skip[k]) continue;
            const int P = ps(k);
            if (skip[k - 1] || skip[k - 2] || fuse[k - 1].dst >= 0 || fuse[k - 2].dst >= 0) continue;
            if (slot_times_const(k - 1) && ps(k - 1) == P + 1 && is_mul(k - 2) && ps(k - 2) == P) {   // adjacent
                fuse[k - 2] = Fuse{pg.ops[k - 1].a_idx, pg.ops[k - 1].b_idx, P};
                skip[k - 1] = skip[k] = 1;
                continue;
            }
            if (!(is_mul(k - 1) && ps(k - 1) == P + 1)) continue;
            size_t j = k - 1;
            bool ok = false;
            while (j-- > 0) {   // the last writer of rP before the second product
                if (writes(j, P)) {
                    ok = slot_times_const(j) && !skip[j] && fuse[j].dst < 0;
                    break;
                }
                if (reads(j, P)) break;
            }
            if (!ok) continue;
            for (size_t x = j + 1; x < k && ok; x++)   // its slot operand must still hold the same value
                ok = !(fm(x) == V2_UN && opc(x) == V2_STORE && pg.ops[x].a_idx == pg.ops[j].a_idx);
            if (!ok) continue;
            fuse[k - 1] = Fuse{pg.ops[j].a_idx, pg.ops[j].b_idx, P};
            skip[j] = skip[k] = 1;
        }
    }
    auto emit_loads = [&](size_t i) {
        if (i >= nops || skip[i]) return;   // (a skipped product's operands are a slot and a constant the callee fetches)
        const ExprOp2& o = pg.ops[i];
        const int form = o.code >> 4, op = (o.code >> 2) & 3;
        if (form == V2_LL || (form == V2_UN && op == V2_LOAD)) emit_load("la", i, o.a_kind, o.a_idx, o.a_rot);
        if (form == V2_SL || form == V2_LL) emit_load("lb", i, o.b_kind, o.b_idx, o.b_rot);
    };
    struct Val {
        std::string name;
        Bnd b;
        bool mem;   // a freshly loaded leaf (const Fe29, carried, below 2 p: never needs a carry pass or a fold)
    };
    auto operand = [&](const char* name, size_t i, int kind, int idx) -> Val {
        char t[32];
        if (is_mem(kind)) {
            snprintf(t, sizeof(t), "%s%zu", name, i);
            return Val{t, leaf, true};
        }
        snprintf(t, sizeof(t), "s%d", idx);
        return Val{t, bs[(size_t)idx], false};
    };
    // what the kernel's variable `name` is known to be, after an in-place carry / fold
    auto remember = [&](const Val& x) {
        if (x.mem) return;
        if (x.name[0] == 'r') br[x.name[1] - '0'] = x.b;
        else if (x.name[0] == 's') bs[(size_t)atoi(x.name.c_str() + 1)] = x.b;
    };
    auto carry = [&](Val& x) {
        if (x.b.L <= kCarried) return;
        if (x.mem) {   // (cannot happen: leaves arrive carried)
            pg_fail = true;
            return;
        }
        add("    %s = fe29_carry(%s);\n", x.name.c_str(), x.name.c_str());
        npass++;
        x.b.T += x.b.L >> 29;
        x.b.L = (((uint64_t)1 << 29) - 1) + (x.b.L >> 29);
        remember(x);
    };
    // bring an operand below 2 p (fe29_fold takes a carried value)
    auto fold = [&](Val& x) {
        if (x.b.V <= 2.0) return;
        if (x.mem) {
            pg_fail = true;
            return;
        }
        carry(x);
        add("    %s = fe29_fold(%s);\n", x.name.c_str(), x.name.c_str());
        npass += 2;
        x.b = Bnd{2.0, kCarried, ((uint64_t)1 << 23) + 16};
        remember(x);
    };
    auto pow2_over = [](double b) {
        int k = 4;
        while ((double)k < b + 1.0) k *= 2;
        return k;
    };
    auto limb_max = [](const Bnd& b) { return (double)std::max(b.L, b.T); };
    // dst = a (op) b; returns what dst then is.  When a and b name the same variable (a square) every in-place change of one is
    // a change of the other: `same` keeps the two descriptions equal.
    auto arith = [&](int op, const std::string& dst, Val a, Val b) -> Bnd {
        if (op == V2_RSUB) {   // b - a
            std::swap(a, b);
            op = V2_SUB;
        }
        const bool same = !a.mem && !b.mem && a.name == b.name;
        auto sync = [&](Val& from, Val& to) {
            if (same) to.b = from.b;
        };
        if (op == V2_MUL) {
            for (int pass = 0; pass < 2 && a.b.V * b.b.V > 100.0; pass++) {
                if (a.b.V >= b.b.V) fold(a), sync(a, b);
                else fold(b), sync(b, a);
            }
            for (int pass = 0; pass < 2 && limb_max(a.b) * limb_max(b.b) > kColumn; pass++) {
                if (a.b.L >= b.b.L) carry(a), sync(a, b);
                else carry(b), sync(b, a);
            }
            if (a.b.V * b.b.V > 100.0 || limb_max(a.b) * limb_max(b.b) > kColumn) pg_fail = true;
            add("    %s = mulx(%s, %s);\n", dst.c_str(), a.name.c_str(), b.name.c_str());
            return product;
        }
        if (op == V2_ADD) {
            for (int pass = 0; pass < 2 && a.b.V + b.b.V > 100.0; pass++) {
                if (a.b.V >= b.b.V) fold(a), sync(a, b);
                else fold(b), sync(b, a);
            }
            for (int pass = 0; pass < 2 && a.b.L + b.b.L > 0xffffffffull; pass++) {
                if (a.b.L >= b.b.L) carry(a), sync(a, b);
                else carry(b), sync(b, a);
            }
            if (a.b.V + b.b.V > 100.0 || a.b.L + b.b.L > 0xffffffffull) pg_fail = true;
            const Bnd sum{a.b.V + b.b.V, a.b.L + b.b.L, a.b.T + b.b.T};
            if (eager) {
                add("    %s = fe29_add_c(%s, %s);\n", dst.c_str(), a.name.c_str(), b.name.c_str());
                return Bnd{sum.V, (((uint64_t)1 << 29) - 1) + (sum.L >> 29), sum.T + (sum.L >> 29)};
            }
            add("    %s = fe29_add(%s, %s);\n", dst.c_str(), a.name.c_str(), b.name.c_str());
            return sum;
        }
        // a - b: the bias K p = J copies of (K / J) p has to dominate b limb by limb
        if (b.b.V > 60.0) fold(b), sync(b, a);
        if (b.b.L > 2 * (((uint64_t)1 << 30) - 2) || eager) carry(b), sync(b, a);
        const int J = b.b.L > ((uint64_t)1 << 30) - 2 ? 2 : 1;
        int K = pow2_over(b.b.V);
        while ((uint64_t)K * ((uint64_t)1 << 22) < b.b.T + 2 * (uint64_t)J) K *= 2;
        if (a.b.V + K > 100.0) fold(a), sync(a, b);
        if (a.b.L + (uint64_t)J * kBiasLimb > 0xffffffffull) carry(a), sync(a, b);
        if (same || a.b.V + K > 100.0 || K > 64 || a.b.L + (uint64_t)J * kBiasLimb > 0xffffffffull) pg_fail = true;   // (x - x is never emitted)
        const Bnd diff{a.b.V + K, a.b.L + (uint64_t)J * kBiasLimb, a.b.T + (uint64_t)K * ((uint64_t)1 << 22) + (uint64_t)K};
        if (eager) {
            add("    %s = fe29_sub<P, %d>(%s, %s);\n", dst.c_str(), K, a.name.c_str(), b.name.c_str());
            return Bnd{diff.V, (((uint64_t)1 << 29) - 1) + (diff.L >> 29), diff.T + (diff.L >> 29)};
        }
        add("    %s = fe29_sub_lazy<P, %d, %d>(%s, %s);\n", dst.c_str(), K, J, a.name.c_str(), b.name.c_str());
        return diff;
    };
    // dst = a * b + slot * constant (one reduction)
    auto dot2 = [&](const std::string& dst, Val a, Val b, const Fuse& f) -> Bnd {
        char sn[32];
        snprintf(sn, sizeof(sn), "s%d", f.slot);
        Val c{sn, bs[(size_t)f.slot], false};
        const bool same = !a.mem && !b.mem && a.name == b.name;
        auto sync = [&](Val& from, Val& to) {
            if (same) to.b = from.b;
        };
        // (the constant: carried, below 2 p)
        for (int pass = 0; pass < 4 && a.b.V * b.b.V + c.b.V * 2.0 > 100.0; pass++) {
            if (c.b.V * 2.0 >= a.b.V * b.b.V) fold(c);
            else if (a.b.V >= b.b.V) fold(a), sync(a, b);
            else fold(b), sync(b, a);
        }
        // one column holds 9 terms of each product: A B + C D <= kColumn (the constant's limbs: carried)
        if (limb_max(c.b) * (double)kCarried > 0.5 * kColumn) carry(c);
        if (a.name == c.name) a.b = c.b;
        if (b.name == c.name) b.b = c.b;
        const double room = kColumn - limb_max(c.b) * (double)kCarried;
        for (int pass = 0; pass < 2 && limb_max(a.b) * limb_max(b.b) > room; pass++) {
            if (a.b.L >= b.b.L) carry(a), sync(a, b);
            else carry(b), sync(b, a);
        }
        if (a.b.V * b.b.V + c.b.V * 2.0 > 100.0 || limb_max(a.b) * limb_max(b.b) > room || limb_max(c.b) * (double)kCarried > 0.5 * kColumn)
            pg_fail = true;
        add("    %s = dot2x(%s, %s, %s, cv + %d * 12);\n", dst.c_str(), a.name.c_str(), b.name.c_str(), c.name.c_str(), f.cst);
        return product;
    };
    static const char* const regs[4] = {"r0", "r1", "r2", "r3"};
    emit_loads(0);
    for (size_t i = 0; i < nops; i++) {
        const ExprOp2& o = pg.ops[i];
        const int form = o.code >> 4, op = (o.code >> 2) & 3, pos = o.code & 3;
        emit_loads(i + 1);
        if (skip[i]) continue;
        const std::string ra = regs[pos];
        if (fuse[i].dst >= 0) {   // this product and a skipped slot * constant one, summed
            const std::string rd = regs[fuse[i].dst];
            const Val a = form == V2_LL ? operand("la", i, o.a_kind, o.a_idx) : Val{ra, br[pos], false};
            const Val b = form == V2_SS ? Val{regs[(pos + 1) & 3], br[(pos + 1) & 3], false} : operand("lb", i, o.b_kind, o.b_idx);
            br[fuse[i].dst] = dot2(rd, a, b, fuse[i]);
            if (barriers) src += "    __builtin_amdgcn_sched_barrier(0);\n";
            continue;
        }
        if (form == V2_SS) {
            br[pos] = arith(op, ra, Val{ra, br[pos], false}, Val{regs[(pos + 1) & 3], br[(pos + 1) & 3], false});
        } else if (form == V2_SL) {
            br[pos] = arith(op, ra, Val{ra, br[pos], false}, operand("lb", i, o.b_kind, o.b_idx));
        } else if (form == V2_LL) {
            br[pos] = arith(op, ra, operand("la", i, o.a_kind, o.a_idx), operand("lb", i, o.b_kind, o.b_idx));
        } else if (op == V2_NEG) {
            src += "    { const Fe29<P> z = fe29_zero<P>();\n";
            const Bnd d = arith(V2_SUB, ra, Val{"z", zero, true}, Val{ra, br[pos], false});
            src += "    }\n";
            br[pos] = d;
        } else if (op == V2_LOAD) {
            const Val a = operand("la", i, o.a_kind, o.a_idx);
            src += "    " + ra + " = " + a.name + ";\n";
            br[pos] = a.b;
        } else {
            // a value kept for later uses: carried once here rather than once per copy (policy; the cheaper one is kept)
            Val x{ra, br[pos], false};
            if (carry_at_store) carry(x);
            add("    s%d = %s;\n", o.a_idx, ra.c_str());
            bs[(size_t)o.a_idx] = br[pos];
        }
        if (barriers) src += "    __builtin_amdgcn_sched_barrier(0);\n";
    }
    if (br[0].L > 0xfffffff0ull || br[0].V > 120.0) pg_fail = true;
    if (pg_fail) return std::string();
    *passes = npass;
    src += "    fe_store(out + (v * size + r) * 8, fe29_to_sat_div32(r0));\n}\n";
    add("static void launch(unsigned gx, unsigned gy, void* st, const uint32_t* const* cols, const size_t* strides, const uint32_t* consts, "
        "size_t nc, size_t size, uint32_t* out) {\n    hipLaunchKernelGGL(%s, dim3(gx, gy), dim3(128), 0, (hipStream_t)st, cols, strides, consts, nc, size, out);\n}\n", kname);
    add("}  // namespace bzh_q29_%016llx\n", hash);
    return src;
}
static std::string program2_source29(const Program2& pg, int field) {
    int pa = 0, pb = 0;
    const std::string a = program2_source29_policy(pg, field, true, &pa), b = program2_source29_policy(pg, field, false, &pb);
    if (a.empty() || b.empty()) return a.empty() ? b : a;
    return pb < pa ? b : a;
}

struct Compiler2 {
    const EPool& pool;
    Program2 prog;
    int depth = 0;                       // registers r0..r(depth-1) hold the evaluation stack
    std::map<int, int> cse;              // node -> LDS slot holding its value (current scope)
    std::map<int, int> hoisted;          // node -> registry column holding its precomputed values (proof-independent)
    std::vector<int> label;              // Sethi-Ullman numbers (leaves 0), memoised per scope
    // shared-subexpression slots per gate group.  Round 2 measured 2 against 0, 4, 6, 8 on the INTERPRETER (slots are LDS there:
    // occupancy mattered more than the last 40 multiplications); the builtin kernels hold them in registers and have room
    // (130 -> 138 VGPRs of the 168 that three waves allow): 6 takes 41 more products out of the Board program (581 -> 540)
    int spill_used = 0, cse_slots = 6, max_lds = kV2LdsInner;
    // Values shared ACROSS gate groups (the same sub-polynomial in several gates: Board 94, Shot 71 products that per-group slots
    // recompute): up to `global_slots` more slots, each holding one value from the first to the last group that uses it; values
    // with disjoint ranges share a slot (select_globals).
    struct GlobalEnt {
        int slot, first, last;
    };
    int global_slots = 6;   // Board 541 -> 475 products, Shot 398 -> 356 (2: 495 / 376; perfect sharing: 447 / 327); 138 -> 146 VGPRs
    std::map<int, GlobalEnt> gsel;
    explicit Compiler2(const EPool& p) : pool(p), label(p.n.size(), -1) {
        if (const char* e = getenv("BZH_VM2_CSE")) cse_slots = std::max(0, std::min(kV2LdsCseMax, atoi(e)));
        if (const char* e = getenv("BZH_VM2_GLOBAL")) global_slots = std::max(0, std::min(kV2LdsGlobalMax, atoi(e)));
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
        if (gsel.count(i)) gparked.insert(i);
        std::fill(label.begin(), label.end(), -1);   // nodes above it are cheaper to reach now
    }
    void count_uses(int i, std::map<int, int>& uses, std::map<int, int>& weight) {
        const ENode& e = pool.n[i];
        if (e.tag == EX_CONST || e.tag == EX_SYMBOL || e.tag == EX_QUERY || hoisted.count(i) || cse.count(i)) return;
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
    // distinct operation nodes under i (`stop` and parked / hoisted values are leaves): products and other operations
    void reach(int i, std::set<int>& seen, const std::set<int>& stop, int* products, int* others) const {
        const ENode& e = pool.n[i];
        if (e.tag == EX_CONST || e.tag == EX_SYMBOL || e.tag == EX_QUERY || hoisted.count(i) || stop.count(i)) return;
        if (!seen.insert(i).second) return;
        if (e.tag == EX_MUL || e.tag == EX_SCALE) ++*products;
        else ++*others;
        if (e.a >= 0) reach(e.a, seen, stop, products, others);
        if (e.b >= 0) reach(e.b, seen, stop, products, others);
    }
    // choose the values kept across groups: greedily the best saving per group of residence that still finds a free slot
    void select_globals(const std::vector<std::vector<int>>& group_roots) {
        gsel.clear();
        if (global_slots <= 0) return;
        const int base = kV2LdsCse0 + cse_slots + kV2LdsSpills;
        std::vector<std::vector<std::pair<int, int>>> busy((size_t)global_slots);
        std::set<int> chosen;
        for (int round = 0; round < 256; round++) {
            std::map<int, std::vector<int>> where;
            for (size_t gi = 0; gi < group_roots.size(); gi++) {
                std::set<int> seen;
                int p = 0, o = 0;
                for (int r : group_roots[gi]) reach(r, seen, chosen, &p, &o);
                for (int n : seen) where[n].push_back((int)gi);
            }
            double best_score = 0;
            int best_n = -1, best_slot = -1;
            for (auto& kv : where) {
                if (kv.second.size() < 2) continue;
                std::set<int> seen;
                int p = 0, o = 0;
                reach(kv.first, seen, chosen, &p, &o);
                if (p == 0) continue;   // sums alone are cheaper to redo than to hold
                const int first = kv.second.front(), last = kv.second.back();
                int slot = -1;
                for (int sl = 0; sl < global_slots && slot < 0; sl++) {
                    bool free_ = true;
                    for (auto& iv : busy[(size_t)sl]) free_ &= last < iv.first || first > iv.second;
                    if (free_) slot = sl;
                }
                if (slot < 0) continue;
                const double benefit = (double)(kv.second.size() - 1) * (200.0 * p + 25.0 * o) - 30.0;   // (- the store)
                const double score = benefit / (double)(last - first + 1);
                if (benefit > 0 && score > best_score) best_score = score, best_n = kv.first, best_slot = slot;
            }
            if (best_n < 0) break;
            const std::vector<int>& gs = where[best_n];
            gsel[best_n] = GlobalEnt{base + best_slot, gs.front(), gs.back()};
            busy[(size_t)best_slot].push_back({gs.front(), gs.back()});
            chosen.insert(best_n);
            max_lds = std::max(max_lds, base + best_slot);
        }
    }
    std::set<int> gparked;
    void open_scope(const std::vector<int>& roots, int group = -1) {
        cse.clear();
        want.clear();
        for (auto& kv : gsel) {   // values living across groups: parked ones are leaves here, the others are parked when first computed
            if (group < kv.second.first || group > kv.second.last) continue;
            if (gparked.count(kv.first)) cse[kv.first] = kv.second.slot;
            else want[kv.first] = kv.second.slot;
        }
        std::fill(label.begin(), label.end(), -1);
        std::map<int, int> uses, weight;
        for (int r : roots) count_uses(r, uses, weight);
        std::vector<std::pair<long, int>> cand;
        for (auto& kv : uses) {
            if (kv.second >= 2 && !gsel.count(kv.first)) cand.push_back({-(long)(kv.second - 1) * weight[kv.first], kv.first});
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
        if (getenv("BZH_VM2_DEBUG")) {   // how many products a perfect (unbounded) sharing across all groups would leave
            std::map<int, int> uses, weight;
            size_t horner = 0;
            for (auto& g : groups) {
                for (int r : g.c) count_uses(r, uses, weight);
                if (g.s >= 0) count_uses(g.s, uses, weight);
                horner += g.c.size() - 1 + (g.s >= 0 ? 1 : 0) + 1;
            }
            size_t uniq_mul = 0, shared = 0;
            for (auto& kv : uses) {
                const ENode& e = pool.n[kv.first];
                if (e.tag == EX_MUL || e.tag == EX_SCALE) uniq_mul++;
                if (kv.second >= 2) shared++;
            }
            {   // shared nodes: weight (products under them), the groups that use them
                std::map<int, std::vector<int>> where;
                for (size_t gi = 0; gi < groups.size(); gi++) {
                    std::map<int, int> u2, w2;
                    for (int r : groups[gi].c) count_uses(r, u2, w2);
                    if (groups[gi].s >= 0) count_uses(groups[gi].s, u2, w2);
                    for (auto& kv : u2) where[kv.first].push_back((int)gi);
                }
                for (auto& kv : uses) {
                    if (kv.second < 2 || where[kv.first].size() < 2) continue;
                    const ENode& e = pool.n[kv.first];
                    fprintf(stderr, "[vm2-shared] node %d tag %d weight %d uses %d groups", kv.first, (int)e.tag, weight[kv.first], kv.second);
                    for (int gi : where[kv.first]) fprintf(stderr, " %d", gi);
                    fprintf(stderr, "\n");
                }
            }
            fprintf(stderr, "[vm2] groups %zu, unique product nodes %zu, glue products (y powers, selectors, ACC) ~%zu, nodes used more than once %zu\n",
                    groups.size(), uniq_mul, horner, shared);
        }
        const Leaf y{BZH_EXPR_CONST, const_index(SY_Y, nullptr), 0};
        const Leaf acc{BZH_EXPR_LDS, kV2LdsAcc, 0}, inner{BZH_EXPR_LDS, kV2LdsInner, 0};
        {
            std::vector<std::vector<int>> group_roots;
            for (auto& g : groups) {
                group_roots.push_back(g.c);
                if (g.s >= 0) group_roots.back().push_back(g.s);
            }
            gparked.clear();
            select_globals(group_roots);
        }
        bool first_group = true;
        int group_index = -1;
        for (auto& g : groups) {
            group_index++;
            std::vector<int> roots = g.c;
            if (g.s >= 0) roots.push_back(g.s);
            open_scope(roots, group_index);
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
    size_t live = 0, peak = 0;  // bytes handed out and not released since the last reset, and their high-water mark
    // A call's allocation sequence is deterministic, so after the first call of a given shape the arena is ONE block
    // that every later call bumps through without touching hipMalloc (overflow blocks are merged at the next reset).
    void reset() {
        if (blocks.size() > 1) {
            const size_t want = peak + (peak >> 4) + ((size_t)1 << 20);
            release();
            Block nb;
            nb.size = want;
            nb.used = 0;
            if (hipMalloc((void**)&nb.p, nb.size) == hipSuccess) blocks.push_back(nb);
        }
        for (auto& b : blocks) b.used = 0;
        live = peak = 0;
    }
    void release() {
        for (auto& b : blocks) (void)hipFree(b.p);
        blocks.clear();
    }
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        live += bytes;
        peak = std::max(peak, live);
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
    // Stack discipline for temporaries (a commitment's scalar vectors, gathered rows): everything allocated after mark() is
    // handed back by pop().  Work on the buffers was enqueued on the ctx's one stream, so whatever reuses the memory runs
    // after it.  While the arena is still a list of blocks (a key's first call) only the accounting moves: the merged block
    // of the next call is sized by the high-water mark.
    struct Mark {
        size_t used, live;
        bool single;
    };
    Mark mark() const { return Mark{blocks.size() == 1 ? blocks[0].used : 0, live, blocks.size() == 1}; }
    void pop(const Mark& m) {
        live = m.live;
        if (m.single && blocks.size() == 1) blocks[0].used = m.used;
    }
};
struct ArenaScope {
    Arena& a;
    Arena::Mark m;
    explicit ArenaScope(Arena& ar) : a(ar), m(ar.mark()) {}
    ~ArenaScope() { a.pop(m); }
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
