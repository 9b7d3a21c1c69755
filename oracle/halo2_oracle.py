"""CPU ORACLE (test infrastructure, NOT product code): big-int restatement of the halo2 PLONKish
prover and verifier for circuits given as data.

Restates, from the published protocol, halo2_proofs 0.2.0 (Cargo.lock:382-385, UN-VENDORED -- nothing of it
is on disk) `plonk::{keygen, create_proof, verify_proof}` with the permutation, lookup and vanishing
arguments, `poly::multiopen` and the IPA opening of oracle/pasta.py.  Entry points in the reference:
create_proof benches/shot.rs:68, benches/board.rs:61-68, src/circuits/shot.rs:921-928; verify_proof
benches/board.rs:80-86, src/circuits/shot.rs:931-940.

PARITY STATUS: the reference draws all prover randomness from OsRng and holds no proof bytes, and
the upstream source is unavailable, so byte parity with upstream is UNPINNED.  The message order, blinding-row
rules, constraint order and multiopen grouping below follow the upstream design as documented in the halo2
book and remembered from the crate; where that memory could be wrong the proof would differ from upstream in
bytes but not in soundness -- the verifier below is an independent check of every prover message.  This module
is what the product prover (csrc/prove.hip, bzh_prove_batch) is compared against, byte for byte, under a shared
RNG stream.
"""
from __future__ import annotations

import pasta as O

# expression AST: ('const', v) ('advice'|'fixed'|'instance', col, rot) ('neg', e) ('add', a, b) ('mul', a, b) ('scale', e, k)


def expr_degree(e) -> int:
    t = e[0]
    if t == 'const':
        return 0
    if t in ('advice', 'fixed', 'instance'):
        return 1
    if t in ('neg', 'scale'):
        return expr_degree(e[1])
    if t == 'add':
        return max(expr_degree(e[1]), expr_degree(e[2]))
    return expr_degree(e[1]) + expr_degree(e[2])


def expr_queries(e, out):
    t = e[0]
    if t in ('advice', 'fixed', 'instance'):
        q = (t, e[1], e[2])
        if q not in out:
            out.append(q)
    elif t in ('neg', 'scale'):
        expr_queries(e[1], out)
    elif t in ('add', 'mul'):
        expr_queries(e[1], out)
        expr_queries(e[2], out)


def expr_eval(e, leaf, p):
    t = e[0]
    if t == 'const':
        return e[1] % p
    if t in ('advice', 'fixed', 'instance'):
        return leaf(t, e[1], e[2])
    if t == 'neg':
        return (-expr_eval(e[1], leaf, p)) % p
    if t == 'scale':
        return expr_eval(e[1], leaf, p) * e[2] % p
    a, b = expr_eval(e[1], leaf, p), expr_eval(e[2], leaf, p)
    return (a + b) % p if t == 'add' else a * b % p


class ConstraintSystem:
    def __init__(self, k, num_advice, num_fixed, num_instance, gates, perm_columns, lookups=(), degree=None, queries=None):
        """queries: optional (advice, fixed, instance) lists of (column, rotation) in upstream's REGISTRATION order --
        a query is registered when it is made (`enable_equality` registers the column's current-row query before any
        gate of the reference's configure functions does: src/chips/board.rs:199,217 before :275).  Without it the
        order is derived: gates, lookups, then the permutation columns."""
        self.k, self.n = k, 1 << k
        self.num_advice, self.num_fixed, self.num_instance = num_advice, num_fixed, num_instance
        self.gates = list(gates)
        self.perm_columns = list(perm_columns)          # [('advice'|'fixed'|'instance', idx)]
        self.lookups = [(list(a), list(t)) for a, t in lookups]
        qs = []
        for g in self.gates:
            expr_queries(g, qs)
        for a, t in self.lookups:
            for e in a + t:
                expr_queries(e, qs)
        for c in self.perm_columns:                      # permutation columns are queried at the current row
            q = (c[0], c[1], 0)
            if q not in qs:
                qs.append(q)
        if queries is not None:
            listed = [('advice', c, r) for c, r in queries[0]] + [('fixed', c, r) for c, r in queries[1]] + \
                     [('instance', c, r) for c, r in queries[2]]
            assert all(q in listed for q in qs), "a queried cell is missing from the explicit query lists"
            qs = listed
        self.advice_queries = [(c, r) for t, c, r in qs if t == 'advice']
        self.fixed_queries = [(c, r) for t, c, r in qs if t == 'fixed']
        self.instance_queries = [(c, r) for t, c, r in qs if t == 'instance']
        deg = 3                                           # permutation argument
        for g in self.gates:
            deg = max(deg, expr_degree(g))
        for a, t in self.lookups:
            da = max([1] + [expr_degree(e) for e in a])
            dt = max([1] + [expr_degree(e) for e in t])
            deg = max(deg, 4, 2 + da + dt)
        self.degree = max(deg, degree or 0)
        per_col = {}
        for c, r in self.advice_queries:
            per_col[c] = per_col.get(c, 0) + 1
        self.blinding_factors = max(3, max(per_col.values()) if per_col else 1) + 2
        self.usable_rows = self.n - (self.blinding_factors + 1)
        self.chunk_len = self.degree - 2
        self.extended_k = k + max(1, (self.degree - 1 - 1).bit_length())
        assert self.usable_rows >= 1


class Domain:
    def __init__(self, cs: ConstraintSystem, F: O.FieldSpec):
        self.F, self.k, self.n = F, cs.k, cs.n
        self.ek, self.en = cs.extended_k, 1 << cs.extended_k
        self.omega, self.eomega = F.omega(cs.k), F.omega(cs.extended_k)
        self.omega_inv = F.inv(self.omega)
        self.zeta = pow(F.g, (F.p - 1) // 3, F.p)        # extended coset generator (cube root of unity)
        self.delta = pow(F.g, 1 << F.S, F.p)             # generator of the odd-order subgroup
        self.ext = self.en // self.n

    def lagrange_to_coeff(self, v):
        return O.intt(v, self.omega, self.F)

    def coeff_to_extended(self, c):
        return O.coset_ntt(list(c) + [0] * (self.en - len(c)), self.eomega, self.zeta, self.F)

    def extended_to_coeff(self, e):
        p = self.F.p
        c = O.intt(e, self.eomega, self.F)
        zi, s, out = self.F.inv(self.zeta), 1, []
        for x in c:
            out.append(x * s % p)
            s = s * zi % p
        return out

    def rotate(self, x, rot):
        return x * pow(self.omega, rot % self.n, self.F.p) % self.F.p

    def lagrange_basis_ext(self, row):
        v = [0] * self.n
        v[row] = 1
        return self.coeff_to_extended(self.lagrange_to_coeff(v))


def build_permutation(cs: ConstraintSystem, copies):
    """Union of cycles as in halo2's permutation keygen Assembly::copy; copies: [((pcol, row), (pcol, row))]."""
    m, n = len(cs.perm_columns), cs.n
    mapping = [[(c, r) for r in range(n)] for c in range(m)]
    aux = [[(c, r) for r in range(n)] for c in range(m)]
    sizes = [[1] * n for _ in range(m)]
    for (lc, lr), (rc, rr) in copies:
        left, right = aux[lc][lr], aux[rc][rr]
        if left == right:
            continue
        if sizes[left[0]][left[1]] < sizes[right[0]][right[1]]:
            left, right = right, left
        sizes[left[0]][left[1]] += sizes[right[0]][right[1]]
        i = right
        while True:
            aux[i[0]][i[1]] = left
            i = mapping[i[0]][i[1]]
            if i == right:
                break
        mapping[lc][lr], mapping[rc][rr] = mapping[rc][rr], mapping[lc][lr]
    return mapping


class Keys:
    """Proving/verifying key material: fixed columns, permutation polynomials, l_0 / l_last / l_blind."""

    def __init__(self, cs, dom, curve, g, w, u, fixed, copies, vk_repr=0x1234, verifier_only=False):
        """verifier_only: keep what verify_proof reads (the vk: commitments of the fixed and permutation polynomials) and
        skip the prover's extended-coset forms -- the big-int NTTs of size 2^(k+3) that make large k impractical."""
        self.cs, self.dom, self.curve, self.g, self.w, self.u = cs, dom, curve, g, w, u
        F, p, n = dom.F, dom.F.p, cs.n
        self.vk_repr = vk_repr % p
        self.fixed = [list(col) + [0] * (n - len(col)) for col in fixed]
        self.fixed_polys = [dom.lagrange_to_coeff(c) for c in self.fixed]
        mapping = build_permutation(cs, copies)
        dpow = [pow(dom.delta, c, p) for c in range(len(cs.perm_columns))]
        wpow = [1] * n
        for r in range(1, n):
            wpow[r] = wpow[r - 1] * dom.omega % p
        self.sigma = [[dpow[mapping[c][r][0]] * wpow[mapping[c][r][1]] % p for r in range(n)]
                      for c in range(len(cs.perm_columns))]
        self.sigma_polys = [dom.lagrange_to_coeff(s) for s in self.sigma]
        if not verifier_only:
            self.fixed_cosets = [dom.coeff_to_extended(c) for c in self.fixed_polys]
            self.sigma_cosets = [dom.coeff_to_extended(s) for s in self.sigma_polys]
            last = cs.usable_rows                              # row n - (blinding_factors + 1)
            self.l0 = dom.lagrange_basis_ext(0)
            self.l_last = dom.lagrange_basis_ext(last)
            lb = [0] * n
            for r in range(last + 1, n):
                lb[r] = 1
            self.l_blind = dom.coeff_to_extended(dom.lagrange_to_coeff(lb))
        self.fixed_commitments = [self.commit(c, 1) for c in self.fixed_polys]
        self.sigma_commitments = [self.commit(c, 1) for c in self.sigma_polys]

    def commit(self, coeffs, blind):
        cv = self.curve
        try:                                   # C oracle Pippenger when built (same definition, much faster)
            import coracle as C
            cid = {'vesta': 0, 'pallas': 1, 'bn254': 2}[cv.name]
            pts = C.points_to_array(self.g[:len(coeffs)] + [self.w])
            return C.array_to_point(C.msm(cid, C.ints_to_array(list(coeffs) + [blind]), pts, 1))
        except (ImportError, OSError, KeyError):
            return cv.add(cv.msm_naive(coeffs, self.g[:len(coeffs)]), cv.mul(blind, self.w))


def query_sets(queries):
    """multiopen::construct_intermediate_sets: group commitments by the SET of points they are opened at.
    queries: list of (commitment_id, point, payload).  Returns (point_sets, groups): groups[i] = list of
    (commitment_id, [payload per point of point_sets[i]]) in first-seen order."""
    order, pts_of = [], {}
    for cid, pt, pay in queries:
        if cid not in pts_of:
            pts_of[cid] = []
            order.append(cid)
        if pt not in [q[0] for q in pts_of[cid]]:
            pts_of[cid].append((pt, pay))
    point_sets, groups = [], []
    for cid in order:
        key = sorted(pt for pt, _ in pts_of[cid])
        if key not in point_sets:
            point_sets.append(key)
            groups.append([])
        i = point_sets.index(key)
        pay_by_pt = dict(pts_of[cid])
        groups[i].append((cid, [pay_by_pt[pt] for pt in key]))
    return point_sets, groups


def lagrange_interpolate(points, evals, F):
    p = F.p
    res = [0] * len(points)
    for j, (xj, yj) in enumerate(zip(points, evals)):
        num, den = [1], 1
        for m, xm in enumerate(points):
            if m == j:
                continue
            num = [(-xm * num[0]) % p] + [(num[i - 1] - xm * num[i]) % p for i in range(1, len(num))] + [num[-1]]
            den = den * (xj - xm) % p
        c = yj * F.inv(den) % p
        for i, v in enumerate(num):
            res[i] = (res[i] + c * v) % p
    return res


def _constraint_expressions(keys, col_at, z_at, lk_at, sig_at, l0, l_last, l_blind, x_pow, beta, gamma, theta, skip_gates=False):
    """Every quotient-numerator term in protocol order, generic over 'where' it is evaluated:
    col_at(type, col, rot) column value; sig_at(j) permutation polynomial j; z_at(i, rot_key) permutation
    product i at rot in {0, 1, 'last'};
    lk_at(i, name, rot) lookup polys; l0/l_last/l_blind Lagrange values; x_pow = delta-free 'X' value."""
    cs, dom = keys.cs, keys.dom
    p = dom.F.p
    out = []
    for g in ([] if skip_gates else cs.gates):       # skip_gates: the caller has folded the gate terms itself (oracle/accel.py)
        out.append(expr_eval(g, col_at, p))
    nsets = (len(cs.perm_columns) + cs.chunk_len - 1) // cs.chunk_len if cs.perm_columns else 0
    active = (1 - (l_last + l_blind)) % p
    if nsets:
        out.append(l0 * (1 - z_at(0, 0)) % p)
        zl = z_at(nsets - 1, 0)
        out.append(l_last * (zl * zl - zl) % p)
        for i in range(1, nsets):
            out.append(l0 * (z_at(i, 0) - z_at(i - 1, 'last')) % p)
        for i in range(nsets):
            cols = cs.perm_columns[i * cs.chunk_len:(i + 1) * cs.chunk_len]
            left, right = z_at(i, 1), z_at(i, 0)
            cur_delta = beta * x_pow % p * pow(dom.delta, i * cs.chunk_len, p) % p
            for j, c in enumerate(cols):
                v = col_at(c[0], c[1], 0)
                left = left * ((v + beta * sig_at(i * cs.chunk_len + j) + gamma) % p) % p
                right = right * ((v + cur_delta + gamma) % p) % p
                cur_delta = cur_delta * dom.delta % p
            out.append(active * (left - right) % p)
    for i, (ins, tabs) in enumerate(cs.lookups):
        z0, z1 = lk_at(i, 'z', 0), lk_at(i, 'z', 1)
        a_p, a_pm1, s_p = lk_at(i, 'a', 0), lk_at(i, 'a', -1), lk_at(i, 's', 0)
        comp = lambda es: _fold([expr_eval(e, col_at, p) for e in es], theta, p)
        out.append(l0 * (1 - z0) % p)
        out.append(l_last * (z0 * z0 - z0) % p)
        out.append(active * (z1 * (a_p + beta) % p * (s_p + gamma) - z0 * (comp(ins) + beta) % p * (comp(tabs) + gamma)) % p)
        out.append(l0 * (a_p - s_p) % p)
        out.append(active * (a_p - s_p) % p * (a_p - a_pm1) % p)
    return out


def _fold(vals, ch, p):
    acc = 0
    for v in vals:
        acc = (acc * ch + v) % p
    return acc


def quotient_evals(keys, cosets, perm, lk, beta, gamma, theta, y):
    """h(X) = (sum of the constraint terms folded by y) / (X^n - 1) at every point of the extended coset.
    cosets: {'advice'|'fixed'|'instance': [column -> extended evaluations]}; perm / lk: the dicts create_proof builds
    ('coset' of every permutation product; 'a_coset' / 's_coset' / 'z_coset' of every lookup)."""
    cs, dom = keys.cs, keys.dom
    F = dom.F
    p, n, ext, en = F.p, cs.n, dom.ext, dom.en
    last_rot = -(cs.blinding_factors + 1)
    h_eval = []
    for r in range(en):
        col_at = lambda t, c, rot: cosets[t][c][(r + rot * ext) % en]
        z_at = lambda i, key: perm[i]['coset'][(r + {0: 0, 1: ext, 'last': last_rot * ext}[key]) % en]
        lk_at = lambda i, nm, rot: lk[i][nm + '_coset'][(r + rot * ext) % en]
        sig_at = lambda j: keys.sigma_cosets[j][r]
        xr = dom.zeta * pow(dom.eomega, r, p) % p
        terms = _constraint_expressions(keys, col_at, z_at, lk_at, sig_at, keys.l0[r], keys.l_last[r], keys.l_blind[r], xr, beta, gamma, theta)
        num = _fold(terms, y, p)
        h_eval.append(num * F.inv((pow(xr, n, p) - 1) % p) % p)
    return h_eval


def create_proof(keys: Keys, advice, instance, rand_scalars, transcript):
    """advice: num_advice columns (usable rows filled; the blinding rows are overwritten here);
    instance: num_instance columns; rand_scalars: iterator of field elements (the shared RNG stream)."""
    cs, dom, cv = keys.cs, keys.dom, keys.curve
    F = dom.F
    p, n, bf, usable = F.p, cs.n, cs.blinding_factors, cs.usable_rows
    rnd = iter(rand_scalars)
    T = transcript
    T.common_scalar(keys.vk_repr)
    # instance columns
    inst = [list(c) + [0] * (n - len(c)) for c in instance]
    inst_polys = [dom.lagrange_to_coeff(c) for c in inst]
    for poly in inst_polys:
        T.common_point(cv, keys.commit(poly, 1))
    inst_cosets = [dom.coeff_to_extended(c) for c in inst_polys]
    # advice columns: blinding rows, then one blind per column, commit, write
    adv = [list(c) + [0] * (n - len(c)) for c in advice]
    for col in adv:
        for r in range(usable, n):
            col[r] = next(rnd)
    adv_blinds = [next(rnd) for _ in adv]
    adv_polys = [dom.lagrange_to_coeff(c) for c in adv]
    for poly, b in zip(adv_polys, adv_blinds):
        T.write_point(cv, keys.commit(poly, b))
    adv_cosets = [dom.coeff_to_extended(c) for c in adv_polys]
    theta = T.squeeze_challenge()
    cols_lagrange = {'advice': adv, 'fixed': keys.fixed, 'instance': inst}
    # lookups: compressed input/table, permuted pair
    lk = []
    for ins, tabs in cs.lookups:
        def comp(es):
            return [_fold([expr_eval(e, lambda t, c, r, row=row: cols_lagrange[t][c][(row + r) % n], p) for e in es], theta, p)
                    for row in range(n)]
        a_c, s_c = comp(ins), comp(tabs)
        a_p, s_p = O.permute_expression_pair(a_c, s_c, usable, F)
        a_p += [next(rnd) for _ in range(bf + 1)]
        s_p += [next(rnd) for _ in range(bf + 1)]
        a_blind, s_blind = next(rnd), next(rnd)
        a_poly, s_poly = dom.lagrange_to_coeff(a_p), dom.lagrange_to_coeff(s_p)
        T.write_point(cv, keys.commit(a_poly, a_blind))
        T.write_point(cv, keys.commit(s_poly, s_blind))
        lk.append({'a_c': a_c, 's_c': s_c, 'a': a_p, 's': s_p, 'a_poly': a_poly, 's_poly': s_poly, 'a_blind': a_blind, 's_blind': s_blind})
    beta = T.squeeze_challenge()
    gamma = T.squeeze_challenge()
    # permutation products
    nsets = (len(cs.perm_columns) + cs.chunk_len - 1) // cs.chunk_len if cs.perm_columns else 0
    perm = []
    last_z = 1
    for i in range(nsets):
        cols = cs.perm_columns[i * cs.chunk_len:(i + 1) * cs.chunk_len]
        den = [1] * n
        num = [1] * n
        for j, c in enumerate(cols):
            vals = cols_lagrange[c[0]][c[1]]
            sg = keys.sigma[i * cs.chunk_len + j]
            dpow = pow(dom.delta, i * cs.chunk_len + j, p)
            w = 1
            for r in range(n):
                den[r] = den[r] * ((beta * sg[r] + gamma + vals[r]) % p) % p
                num[r] = num[r] * ((dpow * w % p * beta + gamma + vals[r]) % p) % p
                w = w * dom.omega % p
        z = [last_z]
        for r in range(n - 1):
            z.append(z[-1] * num[r] % p * F.inv(den[r]) % p)
        for r in range(n - bf, n):
            z[r] = next(rnd)
        last_z = z[usable]
        blind = next(rnd)
        poly = dom.lagrange_to_coeff(z)
        T.write_point(cv, keys.commit(poly, blind))
        perm.append({'z': z, 'poly': poly, 'blind': blind, 'coset': dom.coeff_to_extended(poly)})
    # lookup products
    for d in lk:
        z = [1]
        for r in range(n - 1):
            nu = (d['a_c'][r] + beta) * (d['s_c'][r] + gamma) % p
            de = (d['a'][r] + beta) * (d['s'][r] + gamma) % p
            z.append(z[-1] * nu % p * F.inv(de) % p)
        for r in range(n - bf, n):
            z[r] = next(rnd)
        d['z'], d['z_blind'] = z, next(rnd)
        d['z_poly'] = dom.lagrange_to_coeff(z)
        T.write_point(cv, keys.commit(d['z_poly'], d['z_blind']))
        for nm in ('a', 's', 'z'):
            d[nm + '_coset'] = dom.coeff_to_extended(d[nm + '_poly'])
    # vanishing argument: random polynomial
    random_poly = [next(rnd) for _ in range(n)]
    random_blind = next(rnd)
    T.write_point(cv, keys.commit(random_poly, random_blind))
    y = T.squeeze_challenge()
    # quotient h(X) on the extended coset
    cosets = {'advice': adv_cosets, 'fixed': keys.fixed_cosets, 'instance': inst_cosets}
    h_eval = quotient_evals(keys, cosets, perm, lk, beta, gamma, theta, y)
    h_coeffs = dom.extended_to_coeff(h_eval)
    npieces = cs.degree - 1                                # quotient degree / n
    assert all(c == 0 for c in h_coeffs[npieces * n:]), "quotient has higher degree than expected: constraints not satisfied?"
    h_pieces = [h_coeffs[i * n:(i + 1) * n] for i in range(npieces)]
    h_blinds = [next(rnd) for _ in h_pieces]
    for piece, b in zip(h_pieces, h_blinds):
        T.write_point(cv, keys.commit(piece, b))
    x = T.squeeze_challenge()
    xn = pow(x, n, p)
    last_rot = -(bf + 1)
    # evaluations
    ev = lambda poly, rot: O.eval_polynomial(poly, dom.rotate(x, rot), F)
    for c, r in cs.instance_queries:
        T.write_scalar(ev(inst_polys[c], r))
    for c, r in cs.advice_queries:
        T.write_scalar(ev(adv_polys[c], r))
    for c, r in cs.fixed_queries:
        T.write_scalar(ev(keys.fixed_polys[c], r))
    T.write_scalar(O.eval_polynomial(random_poly, x, F))
    h_poly, h_blind = [0] * n, 0
    for piece, b in zip(reversed(h_pieces), reversed(h_blinds)):
        h_poly = [(a * xn + c) % p for a, c in zip(h_poly, piece)]
        h_blind = (h_blind * xn + b) % p
    for sp in keys.sigma_polys:
        T.write_scalar(O.eval_polynomial(sp, x, F))
    for i, d in enumerate(perm):
        T.write_scalar(ev(d['poly'], 0))
        T.write_scalar(ev(d['poly'], 1))
        if i != nsets - 1:
            T.write_scalar(ev(d['poly'], last_rot))
    for d in lk:
        T.write_scalar(ev(d['z_poly'], 0))
        T.write_scalar(ev(d['z_poly'], 1))
        T.write_scalar(ev(d['a_poly'], 0))
        T.write_scalar(ev(d['a_poly'], -1))
        T.write_scalar(ev(d['s_poly'], 0))
    # multiopen: (id, point, (poly, blind))
    q = []
    pt = lambda rot: dom.rotate(x, rot)
    for c, r in cs.instance_queries:
        q.append((('inst', c), pt(r), (inst_polys[c], 1)))
    for c, r in cs.advice_queries:
        q.append((('adv', c), pt(r), (adv_polys[c], adv_blinds[c])))
    for i, d in enumerate(perm):
        q.append((('pz', i), pt(0), (d['poly'], d['blind'])))
        q.append((('pz', i), pt(1), (d['poly'], d['blind'])))
        if i != nsets - 1:
            q.append((('pz', i), pt(last_rot), (d['poly'], d['blind'])))
    for i, d in enumerate(lk):
        q.append((('lz', i), pt(0), (d['z_poly'], d['z_blind'])))
        q.append((('la', i), pt(0), (d['a_poly'], d['a_blind'])))
        q.append((('ls', i), pt(0), (d['s_poly'], d['s_blind'])))
        q.append((('la', i), pt(-1), (d['a_poly'], d['a_blind'])))
        q.append((('lz', i), pt(1), (d['z_poly'], d['z_blind'])))
    for c, r in cs.fixed_queries:
        q.append((('fix', c), pt(r), (keys.fixed_polys[c], 1)))
    for j, sp in enumerate(keys.sigma_polys):
        q.append((('sig', j), pt(0), (sp, 1)))
    q.append((('h', 0), pt(0), (h_poly, h_blind)))
    q.append((('rand', 0), pt(0), (random_poly, random_blind)))
    x1 = T.squeeze_challenge()
    x2 = T.squeeze_challenge()
    point_sets, groups = query_sets(q)
    q_polys, q_blinds, q_evalsets = [], [], []
    for pts, grp in zip(point_sets, groups):
        poly, blind, evs = [0] * n, 0, [0] * len(pts)
        for cid, pays in grp:
            cp, cb = pays[0]
            poly = [(a * x1 + b) % p for a, b in zip(poly, cp)]
            blind = (blind * x1 + cb) % p
            evs = [(e * x1 + O.eval_polynomial(cp, ptv, F)) % p for e, ptv in zip(evs, pts)]
        q_polys.append(poly)
        q_blinds.append(blind)
        q_evalsets.append(evs)
    f_poly = None
    for pts, evs, poly in zip(point_sets, q_evalsets, q_polys):
        r_poly = lagrange_interpolate(pts, evs, F)
        pl = list(poly)
        for i, rv in enumerate(r_poly):
            pl[i] = (pl[i] - rv) % p
        for ptv in pts:
            pl = O.kate_division(pl, ptv, F)
        pl = pl + [0] * (n - len(pl))
        f_poly = pl if f_poly is None else [(a * x2 + b) % p for a, b in zip(f_poly, pl)]
    f_blind = next(rnd)
    T.write_point(cv, keys.commit(f_poly, f_blind))
    x3 = T.squeeze_challenge()
    for poly in q_polys:
        T.write_scalar(O.eval_polynomial(poly, x3, F))
    x4 = T.squeeze_challenge()
    p_poly, p_blind = f_poly, f_blind
    for poly, blind in zip(q_polys, q_blinds):
        p_poly = [(a * x4 + b) % p for a, b in zip(p_poly, poly)]
        p_blind = (p_blind * x4 + blind) % p
    rest = list(rnd)
    O.ipa_open(cv, keys.g, keys.w, keys.u, p_poly, p_blind, x3, rest, T)
    return bytes(T.proof)


class _Reader:
    def __init__(self, proof, T, curve):
        self.b, self.o, self.T, self.cv = proof, 0, T, curve

    def point(self):
        raw = bytearray(self.b[self.o:self.o + 32])
        self.o += 32
        if len(raw) != 32:
            raise ValueError("short proof")
        ys = raw[31] >> 7
        raw[31] &= 0x7f
        x = int.from_bytes(raw, "little")
        if x == 0 and ys == 0:
            # upstream's Blake2bRead::common_point errors on the identity ("cannot write points at infinity to the transcript")
            raise ValueError("identity point in proof")
        else:
            cv = self.cv
            y = cv.base.sqrt((x * x * x + cv.a * x + cv.b) % cv.p)
            if y is None or x >= cv.p:
                raise ValueError("bad point")
            pt = (x, y if (y & 1) == ys else cv.p - y)
        self.T.common_point(self.cv, pt)
        return pt

    def scalar(self):
        v = int.from_bytes(self.b[self.o:self.o + 32], "little")
        self.o += 32
        if v >= self.cv.scalar.p:
            raise ValueError("bad scalar")
        self.T.common_scalar(v)
        return v


def verify_proof(keys: Keys, instance, proof: bytes, transcript) -> bool:
    cs, dom, cv = keys.cs, keys.dom, keys.curve
    F = dom.F
    p, n, bf = F.p, cs.n, cs.blinding_factors
    T = transcript
    try:
        T.common_scalar(keys.vk_repr)
        inst = [list(c) + [0] * (n - len(c)) for c in instance]
        inst_commits = [keys.commit(dom.lagrange_to_coeff(c), 1) for c in inst]
        for c in inst_commits:
            T.common_point(cv, c)
        R = _Reader(proof, T, cv)
        adv_c = [R.point() for _ in range(cs.num_advice)]
        theta = T.squeeze_challenge()
        lk_c = [{'a': R.point(), 's': R.point()} for _ in cs.lookups]
        beta = T.squeeze_challenge()
        gamma = T.squeeze_challenge()
        nsets = (len(cs.perm_columns) + cs.chunk_len - 1) // cs.chunk_len if cs.perm_columns else 0
        pz_c = [R.point() for _ in range(nsets)]
        for d in lk_c:
            d['z'] = R.point()
        rand_c = R.point()
        y = T.squeeze_challenge()
        h_c = [R.point() for _ in range(cs.degree - 1)]
        x = T.squeeze_challenge()
        xn = pow(x, n, p)
        inst_ev = {q: R.scalar() for q in cs.instance_queries}
        adv_ev = {q: R.scalar() for q in cs.advice_queries}
        fix_ev = {q: R.scalar() for q in cs.fixed_queries}
        rand_ev = R.scalar()
        sig_ev = [R.scalar() for _ in keys.sigma_polys]
        last_rot = -(bf + 1)
        pz_ev = []
        for i in range(nsets):
            d = {0: R.scalar(), 1: R.scalar()}
            if i != nsets - 1:
                d['last'] = R.scalar()
            pz_ev.append(d)
        lk_ev = [{('z', 0): R.scalar(), ('z', 1): R.scalar(), ('a', 0): R.scalar(), ('a', -1): R.scalar(), ('s', 0): R.scalar()}
                 for _ in cs.lookups]
        # expected h(x): Lagrange values at x from the closed form l_i(x) = (x^n - 1) w^i / (n (x - w^i))
        def lag(row):
            wi = pow(dom.omega, row, p)
            return (xn - 1) * wi % p * F.inv(n * (x - wi) % p) % p
        l0, l_last = lag(0), lag(cs.usable_rows)
        l_blind = sum(lag(r) for r in range(cs.usable_rows + 1, n)) % p
        ev_tab = {'advice': adv_ev, 'fixed': fix_ev, 'instance': inst_ev}
        col_at = lambda t, c, rot: ev_tab[t][(c, rot)]
        z_at = lambda i, key: pz_ev[i][key]
        lk_at = lambda i, nm, rot: lk_ev[i][(nm, rot)]
        terms = _constraint_expressions(keys, col_at, z_at, lk_at, lambda j: sig_ev[j], l0, l_last, l_blind, x, beta, gamma, theta)
        expected_h = _fold(terms, y, p) * F.inv((xn - 1) % p) % p
        h_commit = None
        for c in reversed(h_c):
            h_commit = cv.add(cv.mul(xn, h_commit), c)
        # multiopen: (id, point, (commitment, eval))
        pt = lambda rot: dom.rotate(x, rot)
        q = []
        for c, r in cs.instance_queries:
            q.append((('inst', c), pt(r), (inst_commits[c], inst_ev[(c, r)])))
        for c, r in cs.advice_queries:
            q.append((('adv', c), pt(r), (adv_c[c], adv_ev[(c, r)])))
        for i in range(nsets):
            q.append((('pz', i), pt(0), (pz_c[i], pz_ev[i][0])))
            q.append((('pz', i), pt(1), (pz_c[i], pz_ev[i][1])))
            if i != nsets - 1:
                q.append((('pz', i), pt(last_rot), (pz_c[i], pz_ev[i]['last'])))
        for i, d in enumerate(lk_c):
            e = lk_ev[i]
            q.append((('lz', i), pt(0), (d['z'], e[('z', 0)])))
            q.append((('la', i), pt(0), (d['a'], e[('a', 0)])))
            q.append((('ls', i), pt(0), (d['s'], e[('s', 0)])))
            q.append((('la', i), pt(-1), (d['a'], e[('a', -1)])))
            q.append((('lz', i), pt(1), (d['z'], e[('z', 1)])))
        for c, r in cs.fixed_queries:
            q.append((('fix', c), pt(r), (keys.fixed_commitments[c], fix_ev[(c, r)])))
        for j in range(len(keys.sigma_polys)):
            q.append((('sig', j), pt(0), (keys.sigma_commitments[j], sig_ev[j])))
        q.append((('h', 0), pt(0), (h_commit, expected_h)))
        q.append((('rand', 0), pt(0), (rand_c, rand_ev)))
        x1 = T.squeeze_challenge()
        x2 = T.squeeze_challenge()
        point_sets, groups = query_sets(q)
        q_commits, q_evalsets = [], []
        for pts, grp in zip(point_sets, groups):
            cm, evs = None, [0] * len(pts)
            for cid, pays in grp:
                cm = cv.add(cv.mul(x1, cm), pays[0][0])
                evs = [(e * x1 + pay[1]) % p for e, pay in zip(evs, pays)]
            q_commits.append(cm)
            q_evalsets.append(evs)
        f_commit = R.point()
        x3 = T.squeeze_challenge()
        q_evals = [R.scalar() for _ in point_sets]
        f_eval = 0
        for pts, evs, qe in zip(point_sets, q_evalsets, q_evals):
            r_poly = lagrange_interpolate(pts, evs, F)
            r_eval = O.eval_polynomial(r_poly, x3, F)
            den = 1
            for ptv in pts:
                den = den * (x3 - ptv) % p
            f_eval = (f_eval * x2 + (qe - r_eval) * F.inv(den)) % p
        x4 = T.squeeze_challenge()
        final_c, final_v = f_commit, f_eval
        for cm, qe in zip(q_commits, q_evals):
            final_c = cv.add(cv.mul(x4, final_c), cm)
            final_v = (final_v * x4 + qe) % p
        rest = proof[R.o:]
        # the IPA verifier absorbs its own messages
        return O.ipa_verify(cv, keys.g, keys.w, keys.u, final_c, x3, final_v, rest, T)
    except (ValueError, ZeroDivisionError, IndexError):
        return False
