// Part of the whole-proof translation unit (csrc/prove.hip): the KEY -- bzh_pk, keygen's host half (blob parser, constraint-system
// shape, permutation cycles, multiopen structure, quotient program) and device half (columns, cosets, hoisted columns).
#pragma once
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
    // the builtin kernel's unsaturated-limb flavour (csrc/fe29.cuh) and the key-owned columns it reads as fe29 planes, converted
    // once at bzh_pk_create: fixed cosets | sigma cosets | l0, l_last, l_blind, X, 1/(X^n - 1) | hoisted columns (9 en words each)
    bzh_quotient_launch_fn q_builtin29 = nullptr;
    uint32_t* key29 = nullptr;
    hipModule_t q_module = nullptr;
    hipFunction_t q_fn = nullptr;
    int q_select = BZH_QUOTIENT_INTERPRETER;
    // multiopen structure: rotation sets and the commitments grouped under each
    std::vector<std::vector<int>> rot_sets;
    std::vector<std::vector<uint64_t>> groups;
    // per-call workspaces: one grow-only arena per ctx that has used the key (several worker streams share ONE key)
    std::map<const bzh_ctx*, std::unique_ptr<bzh::Arena>> arenas;
    size_t rng_bytes = 0;
    // bzh_prove_batch / bzh_verify_batch calls running on this key right now, over all ctxs (guarded by `mu`):
    // bzh_pk_set_quotient_module and bzh_pk_free refuse while it is non-zero -- they unload code / free arenas that a
    // call on ANOTHER ctx may be launching from
    int calls_in_flight = 0;
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
    {   // the vk digest is a scalar of the circuit field (upstream: C::Scalar::from_bytes_wide): refuse a non-canonical one
        uint32_t w[8];
        memcpy(w, vk, 32);
        bool lt = false;
        for (int i = 7; i >= 0 && !lt; i--) {
            if (w[i] > SF::mod(i)) return BZH_E_RANGE;
            lt = w[i] < SF::mod(i);
        }
        if (!lt) return BZH_E_RANGE;
    }
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
            if (tab[i].program_hash == pk.q_hash) {
                pk.q_builtin = tab[i].launch;
                pk.q_builtin29 = tab[i].launch29;
            }
    }
    if constexpr (fe29_supported<SF>()) {
        if (pk.q_builtin29 && !getenv("BZH_QUOTIENT_SATURATED")) {
            const size_t cols29 = (size_t)pk.nf + m + 5 + pk.hoist_cols;
            BZH_HIP_TRY(ctx, hipMalloc((void**)&pk.key29, cols29 * 9 * en * 4));
            const dim3 blk(256);
            uint32_t* d = pk.key29;
            auto conv = [&](const uint32_t* src, size_t ncols) {
                if (ncols) hipLaunchKernelGGL((k_sat_to_fe29_planes<SF>), dim3((unsigned)((en + 255) / 256), (unsigned)ncols), blk, 0, st, src, d, en);
                d += ncols * 9 * en;
            };
            conv(pk.fixed_cosets, (size_t)pk.nf);
            conv(pk.sigma_cosets, m);
            conv(pk.l0, 5);       // l0, l_last, l_blind, x_col, tinv_col are consecutive columns of the key's allocation
            conv(pk.hoist, pk.hoist_cols);
            BZH_HIP_TRY(ctx, hipGetLastError());
        } else {
            pk.q_builtin29 = nullptr;
        }
    } else {
        pk.q_builtin29 = nullptr;
    }
    const char* qenv = getenv("BZH_QUOTIENT");
    pk.q_select = (pk.q_builtin && !(qenv && !strcmp(qenv, "interp"))) ? BZH_QUOTIENT_BUILTIN : BZH_QUOTIENT_INTERPRETER;
    BZH_HIP_TRY(ctx, hipStreamSynchronize(st));
    *out = pkp.release();
    return BZH_OK;
}
