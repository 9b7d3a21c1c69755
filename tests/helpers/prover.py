"""TEST HELPER (not product code: the product prover is csrc/prove.hip behind bzh_prove_batch).
create_proof on the GPU, staged through Python: the driver that strings the bzh2 stages together.

Host-side mirror of `halo2_proofs::plonk::{keygen_pk, create_proof}` (0.2.0, un-vendored) as called by the
reference at benches/shot.rs:58-71, benches/board.rs:51-71, src/circuits/shot.rs:915-930: same stage order
(instance / advice commitments -> theta -> lookups -> beta, gamma -> permutation and lookup products ->
vanishing commitment -> y -> quotient -> x -> evaluations -> multiopen -> IPA), every heavy step a call
into libbzh2.so:

    Params::commit / commit_lagrange     Context.msm on the SRS window table [G.., U, W] (blind rides on W)
    lagrange_to_coeff / coeff_to_extended / extended_to_coeff        Context.ntt
    lookup compression, permutation numerators / denominators, the quotient numerator   Context.expr_eval
    permute_expression_pair              bzh2.permute_expression_pair
    grand products                       Context.batch_invert + vec_mul + prefix_product
    evaluations                          Context.eval_polynomial
    multiopen                            Context.kate_division, Context.msm
    IPA                                  Context.ipa_open

The circuit arrives as DATA (`Circuit`: column counts, gate expressions, permutation columns and copies,
lookups, fixed columns), because the reference's own constraint systems include 19 gates of the
`halo2_gadgets` crate that is not on disk.  Polynomials are staged through host numpy arrays in this first
version (correctness first); only field-element bookkeeping and the transcript run in Python.

Byte parity: identical to oracle/halo2_oracle.py under a shared RNG byte stream (tests/test_gpu_prover.py);
parity with upstream's bytes is unpinned (the reference uses OsRng and holds no proof: SURVEY F5).
"""
from __future__ import annotations

import numpy as np

from bzh2 import (CURVE_SCALAR_FIELD, Context, Transcript, int_to_limbs, jacobian_to_affine, limbs_to_int,
               permute_expression_pair)
from . import expr as X

from bzh2.circuit_data import MODULI, MULT_GEN, TWO_ADICITY, Circuit, serialize_circuit, _degree, _queries  # noqa: F401


def _arr(ints):
    buf = b"".join(int(v).to_bytes(32, "little") for v in ints)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def _ints(a):
    b = np.ascontiguousarray(a, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


class _Cols:
    """Column registry for one bzh_expr_eval call: name -> index, plus the arrays."""

    def __init__(self):
        self.index, self.arrays = {}, []

    def add(self, name, arr):
        if name not in self.index:
            self.index[name] = len(self.arrays)
            self.arrays.append(np.ascontiguousarray(arr))
        return self.index[name]

    def q(self, name, rot=0):
        return X.Query(self.index[name], rot)


def _lower(e, cols: _Cols, rot_scale):
    """oracle-style tuple expression -> bzh2.expr tree over registered columns."""
    t = e[0]
    if t == 'const':
        return X.Constant(e[1])
    if t in ('advice', 'fixed', 'instance'):
        return cols.q((t, e[1]), e[2] * rot_scale)
    if t == 'neg':
        return X.Negated(_lower(e[1], cols, rot_scale))
    if t == 'scale':
        return X.Scaled(_lower(e[1], cols, rot_scale), e[2])
    a, b = _lower(e[1], cols, rot_scale), _lower(e[2], cols, rot_scale)
    return X.Sum(a, b) if t == 'add' else X.Product(a, b)


def _fold(terms, ch):
    acc = terms[0]
    for t in terms[1:]:
        acc = X.Sum(X.Product(acc, ch), t)
    return acc


class ProvingKey:
    """keygen_pk: fixed / permutation polynomials in coefficient and extended-coset form, l_0 / l_last / l_blind,
    the identity columns delta^j * omega^i, the extended 'X' column and the inverse vanishing column."""

    def __init__(self, ctx: Context, circuit: Circuit, curve: int, g, w, u, vk_repr=0x1234):
        self.ctx, self.c, self.curve = ctx, circuit, curve
        self.field = CURVE_SCALAR_FIELD[curve]
        self.p = p = MODULI[self.field]
        c = circuit
        n, en = c.n, 1 << c.extended_k
        self.en, self.ext = en, en // n
        root = pow(MULT_GEN, (p - 1) >> TWO_ADICITY, p)
        self.omega = pow(root, 1 << (TWO_ADICITY - c.k), p)
        self.eomega = pow(root, 1 << (TWO_ADICITY - c.extended_k), p)
        self.zeta = pow(MULT_GEN, (p - 1) // 3, p)
        self.delta = pow(MULT_GEN, 1 << TWO_ADICITY, p)
        self.vk_repr = vk_repr % p
        self.g0, self.u_pt, self.w_pt = g[0], u, w
        tbl = np.stack([np.concatenate([int_to_limbs(pt[0]), int_to_limbs(pt[1])]) for pt in list(g) + [u, w]])
        self.bases = ctx.upload_bases(curve, tbl).precompute()
        # fixed columns
        self.fixed = [_arr(list(col) + [0] * (n - len(col))) for col in c.fixed]
        self.fixed_polys = self.to_coeff(self.fixed)
        self.fixed_cosets = self.to_extended(self.fixed_polys)
        # permutation
        mapping = _build_permutation(c)
        wp = [pow(self.omega, r, p) for r in range(n)]
        self.ident = [_arr([pow(self.delta, j, p) * wp[r] % p for r in range(n)]) for j in range(len(c.perm_columns))]
        self.sigma = [_arr([pow(self.delta, mapping[j][r][0], p) * wp[mapping[j][r][1]] % p for r in range(n)])
                      for j in range(len(c.perm_columns))]
        self.sigma_polys = self.to_coeff(self.sigma)
        self.sigma_cosets = self.to_extended(self.sigma_polys)
        last = c.usable_rows
        unit = lambda rows: _arr([1 if r in rows else 0 for r in range(n)])
        l_polys = self.to_coeff([unit({0}), unit({last}), unit(set(range(last + 1, n)))])
        self.l0, self.l_last, self.l_blind = self.to_extended(l_polys)
        # X on the extended coset and 1 / (X^n - 1) (period en / n)
        xs, cur = [], self.zeta
        for _ in range(en):
            xs.append(cur)
            cur = cur * self.eomega % p
        self.x_col = _arr(xs)
        tinv = [pow((pow(xs[i], n, p) - 1) % p, p - 2, p) for i in range(self.ext)]
        self.tinv_col = _arr([tinv[i % self.ext] for i in range(en)])

    # ---- device helpers -------------------------------------------------------------------------
    def to_coeff(self, cols):
        if not cols:
            return []
        out = self.ctx.ntt(self.field, np.stack(cols), omega=self.omega, inverse=True)
        return [out[i] for i in range(len(cols))]

    def to_extended(self, polys):
        if not polys:
            return []
        n, en = self.c.n, self.en
        pad = np.zeros((len(polys), en, 4), dtype=np.uint64)
        for i, pl in enumerate(polys):
            pad[i, :n] = pl
        out = self.ctx.ntt(self.field, pad, omega=self.eomega, coset_shift=self.zeta)
        return [out[i] for i in range(len(polys))]

    def commit(self, polys, blinds):
        """Params::commit for a batch: [coeffs.., 0 (U), blind (W)] against the window table -> affine int pairs."""
        if not polys:
            return []
        n = self.c.n
        sc = np.zeros((len(polys), n + 2, 4), dtype=np.uint64)
        for i, (pl, b) in enumerate(zip(polys, blinds)):
            sc[i, :n] = pl
            sc[i, n + 1] = int_to_limbs(b)
        aff = jacobian_to_affine(self.curve, self.ctx.msm(self.bases, sc))
        return [None if not a.any() else (limbs_to_int(a[:4]), limbs_to_int(a[4:])) for a in aff]

    def evals(self, polys, points):
        out = self.ctx.eval_polynomial(self.field, np.stack(polys), _arr(points))
        return _ints(out)


def _build_permutation(c: Circuit):
    m, n = len(c.perm_columns), c.n
    mapping = [[(col, r) for r in range(n)] for col in range(m)]
    aux = [[(col, r) for r in range(n)] for col in range(m)]
    sizes = [[1] * n for _ in range(m)]
    for (lc, lr), (rc, rr) in c.copies:
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


def _from_u512(b: bytes, p: int) -> int:
    return int.from_bytes(b, "little") % p


class _Rng:
    """The shared RNG stream: 64 bytes per Field::random draw."""

    def __init__(self, data: bytes, p: int):
        self.data, self.o, self.p = data, 0, p

    def scalar(self) -> int:
        v = _from_u512(self.data[self.o:self.o + 64], self.p)
        self.o += 64
        return v

    def take(self, count: int) -> bytes:
        """raw bytes of the next `count` draws (reduced on the device by bzh_random_field)"""
        b = self.data[self.o:self.o + 64 * count]
        self.o += 64 * count
        return b

    def rest(self) -> bytes:
        return self.data[self.o:]


def _query_sets(queries):
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
        by = dict(pts_of[cid])
        groups[point_sets.index(key)].append((cid, [by[pt] for pt in key]))
    return point_sets, groups


def _lagrange_interpolate(points, evals, p):
    res = [0] * len(points)
    for j, (xj, yj) in enumerate(zip(points, evals)):
        num, den = [1], 1
        for m, xm in enumerate(points):
            if m == j:
                continue
            num = [(-xm * num[0]) % p] + [(num[i - 1] - xm * num[i]) % p for i in range(1, len(num))] + [num[-1]]
            den = den * (xj - xm) % p
        cf = yj * pow(den, -1, p) % p
        for i, v in enumerate(num):
            res[i] = (res[i] + cf * v) % p
    return res


def create_proof(pk: ProvingKey, advice, instance, rng_bytes: bytes, transcript: Transcript) -> bytes:
    """plonk::create_proof for one circuit instance.  advice / instance: lists of int lists (usable rows).
    rng_bytes: 64 bytes per Field::random draw, in draw order."""
    c, ctx, p, fld, cv = pk.c, pk.ctx, pk.p, pk.field, pk.curve
    n, bf, usable, ext, en = c.n, c.blinding_factors, c.usable_rows, pk.ext, pk.en
    rng = _Rng(rng_bytes, p)
    T = transcript
    T.common_scalar(pk.vk_repr)
    # instance
    inst = [_arr(list(col) + [0] * (n - len(col))) for col in instance]
    inst_polys = pk.to_coeff(inst)
    for pt in pk.commit(inst_polys, [1] * len(inst_polys)):
        T.common_point(pt)
    inst_cosets = pk.to_extended(inst_polys)
    # advice
    adv_i = [list(col) + [0] * (n - len(col)) for col in advice]
    for col in adv_i:
        for r in range(usable, n):
            col[r] = rng.scalar()
    adv_blinds = [rng.scalar() for _ in adv_i]
    adv = [_arr(col) for col in adv_i]
    adv_polys = pk.to_coeff(adv)
    for pt in pk.commit(adv_polys, adv_blinds):
        T.write_point(cv, pt)
    adv_cosets = pk.to_extended(adv_polys)
    theta = T.squeeze_challenge()
    # lookups
    lag = _Cols()
    for i, a in enumerate(adv):
        lag.add(('advice', i), a)
    for i, a in enumerate(pk.fixed):
        lag.add(('fixed', i), a)
    for i, a in enumerate(inst):
        lag.add(('instance', i), a)
    lk = []
    for ins, tabs in c.lookups:
        comp = []
        for es in (ins, tabs):
            tree = _fold([_lower(e, lag, 1) for e in es], X.Constant(theta)) if len(es) > 1 else _lower(es[0], lag, 1)
            comp.append(ctx.expr_eval(fld, X.compile_expression(tree, p), lag.arrays))
        a_c, s_c = comp
        a_p, s_p = permute_expression_pair(fld, a_c, s_c, usable)
        a_p = np.concatenate([a_p, _arr([rng.scalar() for _ in range(bf + 1)])])
        s_p = np.concatenate([s_p, _arr([rng.scalar() for _ in range(bf + 1)])])
        d = {'a_c': a_c, 's_c': s_c, 'a': a_p, 's': s_p, 'a_blind': rng.scalar(), 's_blind': rng.scalar()}
        d['a_poly'], d['s_poly'] = pk.to_coeff([a_p, s_p])
        for pt in pk.commit([d['a_poly'], d['s_poly']], [d['a_blind'], d['s_blind']]):
            T.write_point(cv, pt)
        lk.append(d)
    beta = T.squeeze_challenge()
    gamma = T.squeeze_challenge()
    B, G = X.Constant(beta), X.Constant(gamma)
    # permutation products
    nsets = (len(c.perm_columns) + c.chunk_len - 1) // c.chunk_len if c.perm_columns else 0
    perm, last_z = [], 1
    for i in range(nsets):
        cols = c.perm_columns[i * c.chunk_len:(i + 1) * c.chunk_len]
        reg = _Cols()
        num_t, den_t = None, None
        for j, col in enumerate(cols):
            gj = i * c.chunk_len + j
            reg.add(col, lag.arrays[lag.index[col]])
            reg.add(('sigma', gj), pk.sigma[gj])
            reg.add(('ident', gj), pk.ident[gj])
            v = reg.q(col)
            d_f = X.Sum(X.Sum(X.Product(B, reg.q(('sigma', gj))), G), v)
            n_f = X.Sum(X.Sum(X.Product(reg.q(('ident', gj)), B), G), v)
            den_t = d_f if den_t is None else X.Product(den_t, d_f)
            num_t = n_f if num_t is None else X.Product(num_t, n_f)
        den = ctx.expr_eval(fld, X.compile_expression(den_t, p), reg.arrays)
        num = ctx.expr_eval(fld, X.compile_expression(num_t, p), reg.arrays)
        ratio = ctx.vec_mul(fld, num, ctx.batch_invert(fld, den))
        z = ctx.prefix_product(fld, ratio)
        if last_z != 1:
            z = ctx.vec_mul(fld, z, _arr([last_z] * n))
        z[n - bf:] = _arr([rng.scalar() for _ in range(bf)])
        last_z = limbs_to_int(z[usable])
        blind = rng.scalar()
        poly = pk.to_coeff([z])[0]
        T.write_point(cv, pk.commit([poly], [blind])[0])
        perm.append({'z': z, 'poly': poly, 'blind': blind, 'coset': pk.to_extended([poly])[0]})
    # lookup products
    for d in lk:
        reg = _Cols()
        for nm in ('a_c', 's_c', 'a', 's'):
            reg.add(nm, d[nm])
        nu = X.Product(X.Sum(reg.q('a_c'), B), X.Sum(reg.q('s_c'), G))
        de = X.Product(X.Sum(reg.q('a'), B), X.Sum(reg.q('s'), G))
        num = ctx.expr_eval(fld, X.compile_expression(nu, p), reg.arrays)
        den = ctx.expr_eval(fld, X.compile_expression(de, p), reg.arrays)
        z = ctx.prefix_product(fld, ctx.vec_mul(fld, num, ctx.batch_invert(fld, den)))
        z[n - bf:] = _arr([rng.scalar() for _ in range(bf)])
        d['z'], d['z_blind'] = z, rng.scalar()
        d['z_poly'] = pk.to_coeff([z])[0]
        T.write_point(cv, pk.commit([d['z_poly']], [d['z_blind']])[0])
        d['a_coset'], d['s_coset'], d['z_coset'] = pk.to_extended([d['a_poly'], d['s_poly'], d['z_poly']])
    # vanishing argument
    random_poly = _arr([rng.scalar() for _ in range(n)])
    random_blind = rng.scalar()
    T.write_point(cv, pk.commit([random_poly], [random_blind])[0])
    y = T.squeeze_challenge()
    # quotient numerator on the extended coset, one program
    reg = _Cols()
    for i, a in enumerate(adv_cosets):
        reg.add(('advice', i), a)
    for i, a in enumerate(pk.fixed_cosets):
        reg.add(('fixed', i), a)
    for i, a in enumerate(inst_cosets):
        reg.add(('instance', i), a)
    for j, a in enumerate(pk.sigma_cosets):
        reg.add(('sigma', j), a)
    for i, d in enumerate(perm):
        reg.add(('pz', i), d['coset'])
    for i, d in enumerate(lk):
        for nm in ('a', 's', 'z'):
            reg.add(('l' + nm, i), d[nm + '_coset'])
    for nm, a in (('l0', pk.l0), ('l_last', pk.l_last), ('l_blind', pk.l_blind), ('X', pk.x_col), ('tinv', pk.tinv_col)):
        reg.add(nm, a)
    one = X.Constant(1)
    l0, l_last = reg.q('l0'), reg.q('l_last')
    active = X.Sum(one, X.Negated(X.Sum(l_last, reg.q('l_blind'))))
    last_rot = -(bf + 1)
    terms = [_lower(gt, reg, ext) for gt in c.gates]
    if nsets:
        z0 = reg.q(('pz', 0))
        terms.append(X.Product(l0, X.Sum(one, X.Negated(z0))))
        zl = reg.q(('pz', nsets - 1))
        terms.append(X.Product(l_last, X.Sum(X.Product(zl, zl), X.Negated(zl))))
        for i in range(1, nsets):
            terms.append(X.Product(l0, X.Sum(reg.q(('pz', i)), X.Negated(reg.q(('pz', i - 1), last_rot * ext)))))
        for i in range(nsets):
            cols = c.perm_columns[i * c.chunk_len:(i + 1) * c.chunk_len]
            left, right = reg.q(('pz', i), ext), reg.q(('pz', i))
            for j, col in enumerate(cols):
                gj = i * c.chunk_len + j
                v = reg.q(col)
                left = X.Product(left, X.Sum(X.Sum(v, X.Product(B, reg.q(('sigma', gj)))), G))
                cur = X.Product(X.Constant(beta * pow(pk.delta, gj, p) % p), reg.q('X'))
                right = X.Product(right, X.Sum(X.Sum(v, cur), G))
            terms.append(X.Product(active, X.Sum(left, X.Negated(right))))
    th = X.Constant(theta)
    for i, (ins, tabs) in enumerate(c.lookups):
        z0, z1 = reg.q(('lz', i)), reg.q(('lz', i), ext)
        a_p, a_m1, s_p = reg.q(('la', i)), reg.q(('la', i), -ext), reg.q(('ls', i))
        comp = lambda es: _fold([_lower(e, reg, ext) for e in es], th) if len(es) > 1 else _lower(es[0], reg, ext)
        terms.append(X.Product(l0, X.Sum(one, X.Negated(z0))))
        terms.append(X.Product(l_last, X.Sum(X.Product(z0, z0), X.Negated(z0))))
        lhs = X.Product(X.Product(z1, X.Sum(a_p, B)), X.Sum(s_p, G))
        rhs = X.Product(X.Product(z0, X.Sum(comp(ins), B)), X.Sum(comp(tabs), G))
        terms.append(X.Product(active, X.Sum(lhs, X.Negated(rhs))))
        terms.append(X.Product(l0, X.Sum(a_p, X.Negated(s_p))))
        terms.append(X.Product(X.Product(active, X.Sum(a_p, X.Negated(s_p))), X.Sum(a_p, X.Negated(a_m1))))
    h_tree = X.Product(_fold(terms, X.Constant(y)), reg.q('tinv'))
    h_eval = ctx.expr_eval(fld, X.compile_expression(h_tree, p), reg.arrays)
    h_coeffs = ctx.ntt(fld, h_eval, omega=pk.eomega, inverse=True, coset_shift=pk.zeta)
    npieces = c.degree - 1
    if h_coeffs[npieces * n:].any():
        raise ValueError("quotient has higher degree than expected: the witness does not satisfy the constraints")
    h_pieces = [h_coeffs[i * n:(i + 1) * n] for i in range(npieces)]
    h_blinds = [rng.scalar() for _ in h_pieces]
    for pt in pk.commit(h_pieces, h_blinds):
        T.write_point(cv, pt)
    x = T.squeeze_challenge()
    xn = pow(x, n, p)
    rot = lambda r: x * pow(pk.omega, r % n, p) % p
    # evaluations, one batched launch
    jobs = []
    jobs += [(inst_polys[col], rot(r)) for col, r in c.instance_queries]
    jobs += [(adv_polys[col], rot(r)) for col, r in c.advice_queries]
    jobs += [(pk.fixed_polys[col], rot(r)) for col, r in c.fixed_queries]
    jobs += [(random_poly, x)]
    jobs += [(sp, x) for sp in pk.sigma_polys]
    for i, d in enumerate(perm):
        jobs += [(d['poly'], x), (d['poly'], rot(1))] + ([(d['poly'], rot(last_rot))] if i != nsets - 1 else [])
    for d in lk:
        jobs += [(d['z_poly'], x), (d['z_poly'], rot(1)), (d['a_poly'], x), (d['a_poly'], rot(-1)), (d['s_poly'], x)]
    for v in pk.evals([j[0] for j in jobs], [j[1] for j in jobs]):
        T.write_scalar(v)
    # h(X) = sum x^(n i) h_i(X)
    hp = [_ints(pc) for pc in h_pieces]
    h_poly, h_blind = [0] * n, 0
    for piece, b in zip(reversed(hp), reversed(h_blinds)):
        h_poly = [(a * xn + cc) % p for a, cc in zip(h_poly, piece)]
        h_blind = (h_blind * xn + b) % p
    # multiopen
    I = _ints
    q = []
    for col, r in c.instance_queries:
        q.append((('inst', col), rot(r), (I(inst_polys[col]), 1)))
    for col, r in c.advice_queries:
        q.append((('adv', col), rot(r), (I(adv_polys[col]), adv_blinds[col])))
    for i, d in enumerate(perm):
        pl = I(d['poly'])
        q.append((('pz', i), rot(0), (pl, d['blind'])))
        q.append((('pz', i), rot(1), (pl, d['blind'])))
        if i != nsets - 1:
            q.append((('pz', i), rot(last_rot), (pl, d['blind'])))
    for i, d in enumerate(lk):
        zp, ap, sp_ = I(d['z_poly']), I(d['a_poly']), I(d['s_poly'])
        q.append((('lz', i), rot(0), (zp, d['z_blind'])))
        q.append((('la', i), rot(0), (ap, d['a_blind'])))
        q.append((('ls', i), rot(0), (sp_, d['s_blind'])))
        q.append((('la', i), rot(-1), (ap, d['a_blind'])))
        q.append((('lz', i), rot(1), (zp, d['z_blind'])))
    for col, r in c.fixed_queries:
        q.append((('fix', col), rot(r), (I(pk.fixed_polys[col]), 1)))
    for j, sp in enumerate(pk.sigma_polys):
        q.append((('sig', j), rot(0), (I(sp), 1)))
    q.append((('h', 0), rot(0), (h_poly, h_blind)))
    q.append((('rand', 0), rot(0), (I(random_poly), random_blind)))
    x1 = T.squeeze_challenge()
    x2 = T.squeeze_challenge()
    point_sets, groups = _query_sets(q)
    q_polys, q_blinds = [], []
    for pts, grp in zip(point_sets, groups):
        poly, blind = [0] * n, 0
        for _, pays in grp:
            cp, cb = pays[0]
            poly = [(a * x1 + b_) % p for a, b_ in zip(poly, cp)]
            blind = (blind * x1 + cb) % p
        q_polys.append(poly)
        q_blinds.append(blind)
    # evaluations of every q_poly at every point of its set: one batched launch
    ev_jobs = [(qp, ptv) for pts, qp in zip(point_sets, q_polys) for ptv in pts]
    ev = pk.evals([_arr(j[0]) for j in ev_jobs], [j[1] for j in ev_jobs])
    f_poly, o = None, 0
    for pts, poly in zip(point_sets, q_polys):
        evs = ev[o:o + len(pts)]
        o += len(pts)
        r_poly = _lagrange_interpolate(pts, evs, p)
        pl = list(poly)
        for i, rv in enumerate(r_poly):
            pl[i] = (pl[i] - rv) % p
        arr = _arr(pl)
        for ptv in pts:
            arr = ctx.kate_division(fld, arr, ptv)
        pl = _ints(arr) + [0] * (n - arr.shape[0])
        f_poly = pl if f_poly is None else [(a * x2 + b_) % p for a, b_ in zip(f_poly, pl)]
    f_blind = rng.scalar()
    T.write_point(cv, pk.commit([_arr(f_poly)], [f_blind])[0])
    x3 = T.squeeze_challenge()
    for v in pk.evals([_arr(qp) for qp in q_polys], [x3] * len(q_polys)):
        T.write_scalar(v)
    x4 = T.squeeze_challenge()
    p_poly, p_blind = f_poly, f_blind
    for poly, blind in zip(q_polys, q_blinds):
        p_poly = [(a * x4 + b_) % p for a, b_ in zip(p_poly, poly)]
        p_blind = (p_blind * x4 + blind) % p
    ctx.ipa_open(pk.bases, _arr(p_poly), p_blind, x3, rng.rest(), T)
    return T.proof()
