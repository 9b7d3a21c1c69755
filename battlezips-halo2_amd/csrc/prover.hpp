// Part of the whole-proof translation unit (csrc/prove.hip): the PROVER -- create_proof for a batch of witnesses in lockstep.
#pragma once
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
    // The quotient's per-proof columns on the extended coset exist for the quotient alone.  When the builtin kernel's
    // unsaturated-limb flavour will run (decided once per call), coeff_to_extended writes them straight as fe29 planes
    // (9 words per element, ntt_store_out) and no saturated copy is ever made.
    const bool q29 = [&] {
        if constexpr (!fe29_supported<SF>()) return false;
        std::lock_guard<std::mutex> lk(pk.mu);
        return pk.q_ok && pk.q_builtin29 && pk.key29 && pk.q_select == BZH_QUOTIENT_BUILTIN && pk.en % 128 == 0 && pk.ek >= 12 &&
               pk.ek <= 27 /* 32-bit byte offsets into a plane (fe29_load_planes_g) */ && !getenv("BZH_QUOTIENT_V1");
    }();
    uint32_t* ext_alloc(size_t cols) { return q29 ? (uint32_t*)arena.alloc(cols * 9 * en * 4) : dalloc(cols * en); }
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
    // an expression program's four host-side pieces -- constants | instructions | column pointers | column strides -- laid out
    // 256-byte aligned in one arena block and sent with ONE staging copy (they were four: ~130 of a proof's ~145 small uploads)
    struct ProgramArgs {
        uint32_t* consts = nullptr;
        char *prog = nullptr, *ptrs = nullptr, *strides = nullptr;
    };
    int upload_program(const void* cv, size_t cv_bytes, const void* ops, size_t op_bytes, const void* ptrs, const void* strides, size_t ncols,
                       ProgramArgs& out) {
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t o_prog = al(cv_bytes), o_ptrs = o_prog + al(op_bytes), o_strides = o_ptrs + al(ncols * 8), total = o_strides + al(ncols * 8);
        char* dev = (char*)arena.alloc(total);
        if (!dev) return BZH_E_OOM;
        char* slot = nullptr;
        PV_TRY(h2d_stage(ctx, total, &slot));
        memcpy(slot, cv, cv_bytes);
        memcpy(slot + o_prog, ops, op_bytes);
        if (ncols) {
            memcpy(slot + o_ptrs, ptrs, ncols * 8);
            memcpy(slot + o_strides, strides, ncols * 8);
        }
        PV_TRY(h2d_commit(ctx, dev, slot, total));
        out.consts = (uint32_t*)dev;
        out.prog = dev + o_prog;
        out.ptrs = dev + o_ptrs;
        out.strides = dev + o_strides;
        return BZH_OK;
    }

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
        if (q29) {
            static const bool unfused = getenv("BZH_QUOTIENT29_UNFUSED") != nullptr;   // experiment: saturated cosets + a conversion pass
            if (!unfused) return ntt_run_padded(ctx, field, nullptr, polys, pk.k, pk.ek, count, pk.eomega, pk.zeta, dst);
            ArenaScope scope(arena);
            uint32_t* sat = dalloc(count * en);
            if (!sat) return BZH_E_OOM;
            PV_TRY(ntt_run_padded(ctx, field, sat, polys, pk.k, pk.ek, count, pk.eomega, pk.zeta));
            hipLaunchKernelGGL((k_sat_to_fe29_planes<SF>), dim3((unsigned)((en + 255) / 256), (unsigned)count), dim3(256), 0, st, sat, dst, en);
            BZH_HIP_TRY(ctx, hipGetLastError());
            return BZH_OK;
        }
        return ntt_run_padded(ctx, field, dst, polys, pk.k, pk.ek, count, pk.eomega, pk.zeta);
    }
    // Params::commit for `count` polynomials (rows of `pitch` elements): affine canonical points out
    // (lagrange: the rows are evaluations over the domain and the bases g_lagrange -- Params::commit_lagrange; the group
    // element is the same as committing the interpolated coefficients to g, but witness columns are sparse and small in
    // this basis, so most window digits are zero and cost the MSM nothing)
    // shift_row >= 0 (Lagrange basis only): the columns are constant over a long stretch that contains that row (the grand
    // products: from the last copy constraint to the blinding rows); the constant is taken out and committed on g_0, the rest
    // of the stretch becomes zero digits that the MSM's sort skips -- the same group element, the same proof bytes.
    int commit(const uint32_t* polys, size_t pitch, size_t count, const std::vector<Fe<SF>>& blinds, std::vector<uint64_t>& xy,
               bool lagrange = false, long shift_row = -1) {
        xy.assign(count * 8, 0);
        if (!count) return BZH_OK;
        ArenaScope scope(arena);   // the scalar vectors and the result buffer are dead when this returns (d2h_finish below)
        static const bool no_shift = getenv("BZH_NO_COMMIT_SHIFT") != nullptr;
        const bool wide = lagrange && pk.srs_lagrange->n == n + 3;   // (g_lagrange | u | w | g_0): every vector spans the whole row
        const bool shift = wide && shift_row >= 0 && !no_shift && count <= 65535;
        const size_t cols = wide ? n + 3 : n + 2;
        uint32_t* sc = dalloc(count * cols);
        uint32_t* bl = dalloc(count);
        uint32_t* d_out = dalloc(count * 3);
        if (!sc || !bl || !d_out) return BZH_E_OOM;
        PV_TRY(upload(bl, blinds.data(), count));
        if (shift) {
            hipLaunchKernelGGL((k_commit_shift<SF>), dim3((unsigned)((n + 3 + 255) / 256), (unsigned)count), dim3(256), 0, st, polys, pitch, n,
                               (size_t)shift_row, bl, sc);
            BZH_HIP_TRY(ctx, hipGetLastError());
        } else {
            PV_TRY(zero(sc, count * cols));
            PV_TRY(copy2d(sc, cols, polys, pitch, n, count));
            PV_TRY(copy2d(sc + (n + 1) * 8, cols, bl, 1, 1, count));
        }
        PV_TRY(msm_run(ctx, lagrange ? pk.srs_lagrange : pk.srs, sc, cols, count, BZH_FORM_MONTGOMERY, d_out));
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
        ArenaScope scope(arena);
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
        ProgramArgs pa;
        PV_TRY(upload_program(cv.data(), cv.size() * 4, pg.ops.data(), pg.ops.size() * sizeof(bzh_expr_op), reg.ptr.data(), reg.stride.data(), ncols, pa));
        uint32_t* d_consts = pa.consts;
        char *d_prog = pa.prog, *d_ptrs = pa.ptrs, *d_strides = pa.strides;
        int nslots = pg.result_slot + 1;
        for (auto& o : pg.ops) nslots = std::max(nslots, (int)o.dst + 1);  // operands only read slots written before
        return expr_eval(ctx, field, d_prog, (int)pg.ops.size(), (const uint32_t* const*)d_ptrs, (const size_t*)d_strides, d_consts,
                         per_proof ? nc : 0, size, pg.result_slot, B, nslots, d_out);
    }

    // the quotient through VM v2 (the program compiled at bzh_pk_create), as the builtin kernel, the caller's module or the
    // interpreter.  Returns BZH_E_RANGE when the circuit does not fit VM v2 (the caller falls back to the plain fold through `run`).
    // The builtin kernel in unsaturated limbs.  qp holds fe29 PLANE buffers here (to_extended wrote them: [proof][column][9 size
    // words]); the key's columns were converted at bzh_pk_create, the constants are converted on the host.  The column order is
    // quotient_registry's (the program's column indices refer to it), followed by the hoisted columns.
    int run_quotient29(const QuotientPtrs& qp, size_t size, uint32_t* d_out) {
        const Program2& pg = pk.qprog;
        const size_t nc = pg.consts.size(), m = pk.perm_columns.size(), col = 9 * size;
        const int na = pk.na, nf = pk.nf, ni = pk.ni, nsets = pk.nsets, nl = pk.nl, nz = pk.nsets + pk.nl;
        std::vector<const uint32_t*> ptrs;
        std::vector<size_t> strides;
        auto add = [&](const uint32_t* p, size_t stride_words) {
            ptrs.push_back(p);
            strides.push_back(stride_words);
        };
        const uint32_t *k_fixed = pk.key29, *k_sigma = k_fixed + (size_t)nf * col, *k_misc = k_sigma + m * col, *k_hoist = k_misc + 5 * col;
        for (int i = 0; i < na; i++) add(qp.adv + (size_t)i * col, (size_t)na * col);
        for (int i = 0; i < nf; i++) add(k_fixed + (size_t)i * col, 0);
        for (int i = 0; i < ni; i++) add(qp.inst + (size_t)i * col, (size_t)ni * col);
        for (size_t j = 0; j < m; j++) add(k_sigma + j * col, 0);
        for (int i = 0; i < nsets; i++) add(qp.z + (size_t)i * col, (size_t)nz * col);
        for (int i = 0; i < nl; i++) {
            add(qp.lk[(size_t)i], 2 * col);
            add(qp.lk[(size_t)i] + col, 2 * col);
            add(qp.z + (size_t)(nsets + i) * col, (size_t)nz * col);
        }
        for (int i = 0; i < 5; i++) add(k_misc + (size_t)i * col, 0);     // l0, l_last, l_blind, X, 1 / (X^n - 1)
        {   // the same shape as the saturated registry, or the program's column indices would point elsewhere
            Cols check;
            quotient_registry(pk, QuotientPtrs{}, check);
            if (check.ptr.size() != ptrs.size()) return BZH_E_ARG;
        }
        for (size_t hi = 0; hi < pk.hoist_cols; hi++) add(k_hoist + hi * col, 0);
        std::vector<uint32_t> cv(std::max<size_t>(B * nc, 1) * 12, 0u);
        for (size_t b = 0; b < B; b++)
            for (size_t i = 0; i < nc; i++) {
                const ConstEnt& c = pg.consts[i];
                Fe<SF> v;
                if (c.sym >= 0) {
                    auto f = env[b].find(c.sym);
                    if (f == env[b].end()) return BZH_E_ARG;
                    v = f->second;
                } else {
                    memcpy(v.l, c.val, 32);
                }
                const Fe29<SF> w = fe29_from_sat_reduced(v);
                memcpy(&cv[(b * nc + i) * 12], w.l, 36);
            }
        ProgramArgs pa;
        PV_TRY(upload_program(cv.data(), cv.size() * 4, pg.ops.data(), 0, ptrs.data(), strides.data(), ptrs.size(), pa));
        if (ctx->profiling) {
            double cols_read = 0;
            for (size_t i = 0; i < strides.size() - pk.hoist_cols; i++) cols_read += strides[i] ? (double)B : 1.0;
            ctx->alg_bytes[BZH_T_QUOTIENT] += (cols_read + (double)B) * (double)size * 32.0;
        }
        ScopedTimer t(ctx, BZH_T_QUOTIENT);
        pk.q_builtin29((unsigned)(size / 128), (unsigned)B, (void*)st, (const uint32_t* const*)pa.ptrs, (const size_t*)pa.strides, pa.consts, nc, size, d_out);
        BZH_HIP_TRY(ctx, hipGetLastError());
        return BZH_OK;
    }

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
        ProgramArgs pa;
        PV_TRY(upload_program(cv.data(), cv.size() * 4, pg.ops.data(), pg.ops.size() * sizeof(ExprOp2), ptrs.data(), strides.data(), ncols, pa));
        uint32_t* d_consts = pa.consts;
        char *d_prog = pa.prog, *d_ptrs = pa.ptrs, *d_strides = pa.strides;
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
    const size_t bf1 = (size_t)bf + 1, m = pk.perm_columns.size();
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
        inst_cosets = ext_alloc(B * std::max(ni, 1));
        adv_cosets = ext_alloc(B * na);
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
            const size_t nthreads = std::min<size_t>({B, (size_t)host_thread_budget(), (size_t)8});
            std::vector<int> rcs(B, BZH_OK);
            std::vector<std::thread> th;
            auto work = [&](size_t t) {
                for (size_t b = t; b < B; b += nthreads)
                    rcs[b] = bzh_permute_expression_pair(field, &ah[b * n * 4], &sh[b * n * 4], usable, BZH_FORM_CANONICAL,
                                                         as + (b * 2) * n * 4, as + (b * 2 + 1) * n * 4);
            };
            for (size_t t = 1; t < nthreads; t++) th.emplace_back(work, t);
            work(0);   // the calling thread takes a share (a single proof starts no thread at all)
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
    uint32_t* z_cosets = ext_alloc(B * std::max(nz, 1));
    // numerators and denominators of ALL grand products side by side, [product][proof][row]: one batch inversion, one
    // element-wise product and one scan for the lot (they were per product; the permutation sets are chained only through a
    // scalar carried from one set's last row into the next, applied afterwards)
    uint32_t* den_all = dalloc(B * std::max(nz, 1) * n);
    uint32_t* zt_all = dalloc(B * std::max(nz, 1) * n);
    if (!zs || !z_polys || !z_cosets || !den_all || !zt_all) return BZH_E_OOM;
    uint32_t* den = den_all;
    uint32_t* zt = zt_all;
    auto product_slot = [&](int slot) {
        den = den_all + (size_t)slot * B * n * 8;
        zt = zt_all + (size_t)slot * B * n * 8;
    };
    std::vector<Fe<SF>> z_blinds(B * std::max(nz, 1));
    auto invert_and_scan_all = [&]() -> int {
        mark("  fp:exprs");
        PV_TRY(poly_batch_invert(ctx, field, den_all, (size_t)nz * B * n));
        mark("  fp:invert");
        PV_TRY(poly_vec_mul(ctx, field, zt_all, den_all, (size_t)nz * B * n));
        PV_TRY(poly_prefix_product(ctx, field, zt_all, n, (size_t)nz * B));
        mark("  fp:mul+scan");
        return BZH_OK;
    };
    auto finish_product = [&](int slot, int prev_slot) -> int {
        product_slot(slot);
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
        product_slot(i);
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
        mark(" gp:perm_set");
    }
    for (int li = 0; li < nl; li++) {
        product_slot(nsets + li);
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
        mark(" gp:lookup_product");
    }
    if (nz) {
        PV_TRY(invert_and_scan_all());
        // blinding rows and blinds in the order the products are made upstream: permutation sets, then lookups
        for (int i = 0; i < nsets; i++) PV_TRY(finish_product(i, i ? i - 1 : -1));
        for (int li = 0; li < nl; li++) PV_TRY(finish_product(nsets + li, -1));
        PV_TRY(to_coeff(z_polys, zs, B * nz));
        if (pk.srs_lagrange) PV_TRY(commit(zs, n, B * nz, z_blinds, xy, true, (long)usable - 1));
        else PV_TRY(commit(z_polys, n, B * nz, z_blinds, xy));
        for (size_t b = 0; b < B; b++)
            for (int i = 0; i < nz; i++) bzh_transcript_write_point(T[b], C::id, &xy[(b * nz + i) * 8]);
        mark(" gp:commit");
        PV_TRY(to_extended(z_cosets, z_polys, B * nz));
        mark(" gp:extend_z");
    }
    for (auto& d : lk) {
        d.cosets = ext_alloc(B * 2);
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
        quotient_registry(pk, qp, reg);   // (q29: the per-proof pointers are plane buffers; only run_quotient29 reads them)
        // VM v2 (gate-factored fold, shared subexpressions in LDS); the plain Horner fold through VM v1 if it does not fit
        int qrc = q29 ? run_quotient29(qp, en, h) : (getenv("BZH_QUOTIENT_V1") ? BZH_E_RANGE : run_quotient(reg, en, h));
        if (qrc == BZH_E_RANGE && !q29) {
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
        ArenaScope scope(arena);   // `gathered` is read by evals() (which ends in a stream sync) and by nothing else
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
        std::vector<Fe<SF>> pts(B * J2), ev;
        for (size_t b = 0; b < B; b++)
            for (size_t j = 0; j < J2; j++) pts[b * J2 + j] = rot(b, ev_jobs[j].second);
        {
            ArenaScope scope(arena);   // `gathered` lives until evals() returns (stream sync)
            uint32_t* gathered = dalloc(B * J2 * n);
            if (!gathered) return BZH_E_OOM;
            PV_TRY(gather(srcs, gathered));
            PV_TRY(evals(gathered, B * J2, pts, ev));
        }
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
        uint32_t* k_a = dalloc(B * nq * n);
        uint32_t* k_b = dalloc(B * nq * n);
        if (!rcols || !rs_dev || !f_parts || !k_a || !k_b) return BZH_E_OOM;
        PV_TRY(zero(rcols, B * nq * n));
        PV_TRY(zero(f_parts, B * nq * n));
        PV_TRY(upload(rs_dev, r_small.data(), r_small.size()));
        PV_TRY(copy2d(rcols, n, rs_dev, maxpts, maxpts, B * nq));
        // (q_si - r_si) / prod_(r in set si) (X - x w^r): one division per point, chained within a set, independent between sets.
        // The sets are ordered by the number of their points (most first) and step t divides every set that still has a t-th
        // point in ONE launch: all of them hold n - t coefficients at that step and they are a prefix of the order, so the
        // vectors [position][proof] stay densely packed from step to step.  (Was one launch chain per set: 16 divisions -> 4.)
        std::vector<size_t> order(nq);
        for (size_t si = 0; si < nq; si++) order[si] = si;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return pk.rot_sets[x].size() > pk.rot_sets[y].size(); });
        const size_t steps = nq ? pk.rot_sets[order[0]].size() : 0;
        size_t nkate = 0;
        for (size_t si = 0; si < nq; si++) nkate += pk.rot_sets[si].size();
        uint32_t* d_xall = dalloc(std::max<size_t>(nkate, 1) * B);
        if (!d_xall) return BZH_E_OOM;
        {   // the opening points, [step][position][proof], in one upload
            std::vector<Fe<SF>> xv(nkate * B);
            size_t at = 0;
            for (size_t t = 0; t < steps; t++)
                for (size_t pos = 0; pos < nq && pk.rot_sets[order[pos]].size() > t; pos++)
                    for (size_t b = 0; b < B; b++) xv[at++] = rot(b, pk.rot_sets[order[pos]][t]);
            PV_TRY(upload(d_xall, xv.data(), xv.size()));
        }
        for (size_t pos = 0; pos < nq; pos++) {
            const size_t si = order[pos];
            Cols reg;
            reg.add(key(K_MISC, M_Q), q_polys + si * n * 8, nq * n);
            reg.add(key(K_MISC, M_R), rcols + si * n * 8, nq * n);
            PV_TRY(run(key(42, 0), [&](EPool& ep) { return ep.sub(ep.query(0), ep.query(1)); }, reg, n, k_a + pos * B * n * 8));
        }
        {
            uint32_t* cur = k_a;
            uint32_t* nxt = k_b;
            size_t len = n, x_at = 0;
            for (size_t t = 0; t < steps; t++) {
                size_t active = 0;
                while (active < nq && pk.rot_sets[order[active]].size() > t) active++;
                PV_TRY(poly_kate_division(ctx, field, cur, len, active * B, d_xall + x_at * 8, nxt));
                x_at += active * B;
                std::swap(cur, nxt);
                len--;
                for (size_t pos = 0; pos < active; pos++)   // the sets whose last point this was
                    if (pk.rot_sets[order[pos]].size() == t + 1)
                        PV_TRY(copy2d(f_parts + order[pos] * n * 8, nq * n, cur + pos * B * len * 8, len, len, B));
            }
            for (size_t pos = 0; pos < nq; pos++)   // a set without points (not produced by the key builder): q - r itself
                if (pk.rot_sets[order[pos]].empty()) PV_TRY(copy2d(f_parts + order[pos] * n * 8, nq * n, k_a + pos * B * n * 8, n, n, B));
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
