// Part of the whole-proof translation unit (csrc/prove.hip): the VERIFIER -- verify_proof for a batch of proofs.
#pragma once
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
        const size_t nthreads = std::min<size_t>({B, (size_t)host_thread_budget(), (size_t)32});
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
