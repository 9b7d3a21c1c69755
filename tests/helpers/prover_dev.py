"""TEST HELPER (not product code; the product path is libbzh2.so behind include/bzh2.h).
Device-resident create_proof: the same protocol and message order as bzh2/prover.py (and therefore the same
proof bytes), with every polynomial kept in HBM as a Montgomery-form tensor.  Only challenges, blinds, a few
single-row reads (the z(X) hand-over values), the lookup sort and the transcript run on the host.

Reference seam: halo2_proofs::plonk::create_proof as called from benches/shot.rs:68, benches/board.rs:61-68,
src/circuits/shot.rs:921-928, src/circuits/board.rs:913-920."""
from __future__ import annotations

import numpy as np
import torch

from bzh2 import CURVE_SCALAR_FIELD, FORM_MONTGOMERY, Transcript, int_to_limbs, permute_expression_pair
from . import expr as X
from .device import DeviceOps
from .prover import MODULI, MULT_GEN, TWO_ADICITY, Circuit, _build_permutation, _lagrange_interpolate, _query_sets, _Rng


class _Reg:
    """column registry: name -> index into a list of device tensors"""

    def __init__(self):
        self.index, self.cols = {}, []

    def add(self, name, t):
        if name not in self.index:
            self.index[name] = len(self.cols)
            self.cols.append(t)
        return self.index[name]

    def q(self, name, rot=0):
        return X.Query(self.index[name], rot)


def _lower(e, reg: _Reg, rot_scale):
    t = e[0]
    if t == 'const':
        return X.Constant(e[1])
    if t in ('advice', 'fixed', 'instance'):
        return reg.q((t, e[1]), e[2] * rot_scale)
    if t == 'neg':
        return X.Negated(_lower(e[1], reg, rot_scale))
    if t == 'scale':
        return X.Scaled(_lower(e[1], reg, rot_scale), e[2])
    a, b = _lower(e[1], reg, rot_scale), _lower(e[2], reg, rot_scale)
    return X.Sum(a, b) if t == 'add' else X.Product(a, b)


def _horner(terms, ch):
    acc = terms[0]
    for t in terms[1:]:
        acc = X.Sum(X.Product(acc, ch), t)
    return acc


class DeviceProvingKey:
    def __init__(self, ctx, circuit: Circuit, curve: int, g, w, u, device: torch.device, vk_repr=0x1234, window_bits: int = 8):
        self.ctx, self.c, self.curve = ctx, circuit, curve
        self.field = CURVE_SCALAR_FIELD[curve]
        self.p = p = MODULI[self.field]
        self.ops = ops = DeviceOps(ctx, self.field, curve, p, device)
        c = circuit
        n, en = c.n, 1 << c.extended_k
        self.en, self.ext = en, en // n
        root = pow(MULT_GEN, (p - 1) >> TWO_ADICITY, p)
        self.omega = pow(root, 1 << (TWO_ADICITY - c.k), p)
        self.eomega = pow(root, 1 << (TWO_ADICITY - c.extended_k), p)
        self.zeta = pow(MULT_GEN, (p - 1) // 3, p)
        self.delta = pow(MULT_GEN, 1 << TWO_ADICITY, p)
        self.vk_repr = vk_repr % p
        tbl = np.stack([np.concatenate([int_to_limbs(pt[0]), int_to_limbs(pt[1])]) for pt in list(g) + [u, w]])
        # 8-bit windows: a single proof issues its ~25 MSMs one at a time (transcript dependencies), so each one is
        # latency-bound by the bucket reduction, which scales with the bucket count (k=11: 22.4 -> 18.0 ms per proof
        # against the throughput-optimal 10/11-bit table)
        self.bases = ctx.upload_bases(curve, tbl).precompute(window_bits)
        up = ops.upload
        self.fixed = [up(list(col) + [0] * (n - len(col))) for col in c.fixed]
        self.fixed_polys = self.to_coeff(self.fixed)
        self.fixed_cosets = self.to_extended(self.fixed_polys)
        mapping = _build_permutation(c)
        wp = [pow(self.omega, r, p) for r in range(n)]
        m = len(c.perm_columns)
        self.ident = [up([pow(self.delta, j, p) * wp[r] % p for r in range(n)]) for j in range(m)]
        self.sigma = [up([pow(self.delta, mapping[j][r][0], p) * wp[mapping[j][r][1]] % p for r in range(n)]) for j in range(m)]
        self.sigma_polys = self.to_coeff(self.sigma)
        self.sigma_cosets = self.to_extended(self.sigma_polys)
        last = c.usable_rows
        unit = lambda rows: up([1 if r in rows else 0 for r in range(n)])
        self.l0, self.l_last, self.l_blind = self.to_extended(self.to_coeff([unit({0}), unit({last}), unit(set(range(last + 1, n)))]))
        xs, cur = [], self.zeta
        for _ in range(en):
            xs.append(cur)
            cur = cur * self.eomega % p
        self.x_col = up(xs)
        tinv = [pow((pow(xs[i], n, p) - 1) % p, p - 2, p) for i in range(self.ext)]
        self.tinv_col = up([tinv[i % self.ext] for i in range(en)])

    def to_coeff(self, cols):
        if not cols:
            return []
        t = torch.stack(cols).contiguous()
        self.ops.ntt_(t, self.c.k, len(cols), self.omega, None, True)
        return [t[i] for i in range(len(cols))]

    def to_extended(self, polys):
        if not polys:
            return []
        n, en = self.c.n, self.en
        t = self.ops.zeros(len(polys), en)
        for i, pl in enumerate(polys):
            t[i, :n] = pl
        self.ops.ntt_(t, self.c.extended_k, len(polys), self.eomega, self.zeta, False)
        return [t[i] for i in range(len(polys))]

    def commit(self, polys, blinds):
        if not polys:
            return []
        n = self.c.n
        sc = self.ops.zeros(len(polys), n + 2)
        for i, pl in enumerate(polys):
            sc[i, :n] = pl
        sc[:, n + 1] = self.ops.upload(blinds)
        return self.ops.msm(self.bases, sc)

    def evals(self, polys, points):
        return self.ops.evals(torch.stack(polys).contiguous(), points)


def create_proof(pk: DeviceProvingKey, advice, instance, rng_bytes: bytes, transcript: Transcript) -> bytes:
    c, ops, p, cv = pk.c, pk.ops, pk.p, pk.curve
    n, bf, usable, ext, en = c.n, c.blinding_factors, c.usable_rows, pk.ext, pk.en
    rng = _Rng(rng_bytes, p)
    T = transcript
    up = ops.upload
    T.common_scalar(pk.vk_repr)
    inst = []
    for col in instance:                                # instance columns hold a handful of public inputs
        t_ = ops.zeros(n)
        if len(col):
            t_[:len(col)] = up(col)
        inst.append(t_)
    inst_polys = pk.to_coeff(inst)
    for pt in pk.commit(inst_polys, [1] * len(inst_polys)):
        T.common_point(pt)
    inst_cosets = pk.to_extended(inst_polys)
    # advice columns: int lists, or tensors already resident in HBM (Montgomery form, n rows); the
    # blinding rows are drawn column by column, then one blind per column (upstream's draw order)
    adv = []
    for col in advice:
        if isinstance(col, torch.Tensor):
            a = col.clone()
        else:
            a = up(list(col) + [0] * (n - len(col)))
        a[usable:] = up([rng.scalar() for _ in range(usable, n)])
        adv.append(a)
    adv_blinds = [rng.scalar() for _ in adv]
    adv_polys = pk.to_coeff(adv)
    for pt in pk.commit(adv_polys, adv_blinds):
        T.write_point(cv, pt)
    adv_cosets = pk.to_extended(adv_polys)
    theta = T.squeeze_challenge()
    lag = _Reg()
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
            tree = _horner([_lower(e, lag, 1) for e in es], X.Constant(theta))
            comp.append(ops.expr(tree, lag.cols, n))
        a_c, s_c = comp
        # the sort runs on the host (one per lookup argument); values cross in Montgomery form
        ah = a_c.cpu().numpy().view(np.uint64)
        sh = s_c.cpu().numpy().view(np.uint64)
        pa, ps = permute_expression_pair(pk.field, ah, sh, usable, FORM_MONTGOMERY)
        a_p, s_p = ops.zeros(n), ops.zeros(n)
        a_p[:usable] = torch.from_numpy(pa.view(np.int64)).to(ops.dev)
        s_p[:usable] = torch.from_numpy(ps.view(np.int64)).to(ops.dev)
        a_p[usable:] = up([rng.scalar() for _ in range(bf + 1)])
        s_p[usable:] = up([rng.scalar() for _ in range(bf + 1)])
        d = {'a_c': a_c, 's_c': s_c, 'a': a_p, 's': s_p, 'a_blind': rng.scalar(), 's_blind': rng.scalar()}
        d['a_poly'], d['s_poly'] = pk.to_coeff([a_p, s_p])
        for pt in pk.commit([d['a_poly'], d['s_poly']], [d['a_blind'], d['s_blind']]):
            T.write_point(cv, pt)
        lk.append(d)
    beta = T.squeeze_challenge()
    gamma = T.squeeze_challenge()
    B, G = X.Constant(beta), X.Constant(gamma)
    nsets = (len(c.perm_columns) + c.chunk_len - 1) // c.chunk_len if c.perm_columns else 0
    perm, last_z = [], 1
    for i in range(nsets):
        cols = c.perm_columns[i * c.chunk_len:(i + 1) * c.chunk_len]
        reg = _Reg()
        num_t = den_t = None
        for j, col in enumerate(cols):
            gj = i * c.chunk_len + j
            reg.add(col, lag.cols[lag.index[col]])
            reg.add(('sigma', gj), pk.sigma[gj])
            reg.add(('ident', gj), pk.ident[gj])
            v = reg.q(col)
            d_f = X.Sum(X.Sum(X.Product(B, reg.q(('sigma', gj))), G), v)
            n_f = X.Sum(X.Sum(X.Product(reg.q(('ident', gj)), B), G), v)
            den_t = d_f if den_t is None else X.Product(den_t, d_f)
            num_t = n_f if num_t is None else X.Product(num_t, n_f)
        den = ops.expr(den_t, reg.cols, n)
        z = ops.expr(num_t, reg.cols, n)
        ops.batch_invert_(den)
        ops.vec_mul_(z, den)
        ops.prefix_product_(z, n)
        if last_z != 1:
            z = ops.expr(X.Product(X.Query(0), X.Constant(last_z)), [z], n)
        z[n - bf:] = up([rng.scalar() for _ in range(bf)])
        last_z = ops.download(z[usable:usable + 1])[0]
        blind = rng.scalar()
        poly = pk.to_coeff([z])[0]
        T.write_point(cv, pk.commit([poly], [blind])[0])
        perm.append({'poly': poly, 'blind': blind, 'coset': pk.to_extended([poly])[0]})
    for d in lk:
        reg = _Reg()
        for nm in ('a_c', 's_c', 'a', 's'):
            reg.add(nm, d[nm])
        nu = X.Product(X.Sum(reg.q('a_c'), B), X.Sum(reg.q('s_c'), G))
        de = X.Product(X.Sum(reg.q('a'), B), X.Sum(reg.q('s'), G))
        z = ops.expr(nu, reg.cols, n)
        den = ops.expr(de, reg.cols, n)
        ops.batch_invert_(den)
        ops.vec_mul_(z, den)
        ops.prefix_product_(z, n)
        z[n - bf:] = up([rng.scalar() for _ in range(bf)])
        d['z_blind'] = rng.scalar()
        d['z_poly'] = pk.to_coeff([z])[0]
        T.write_point(cv, pk.commit([d['z_poly']], [d['z_blind']])[0])
        d['a_coset'], d['s_coset'], d['z_coset'] = pk.to_extended([d['a_poly'], d['s_poly'], d['z_poly']])
    random_poly = ops.random_field(rng.take(n), n)      # n Field::random draws, reduced on the device
    random_blind = rng.scalar()
    T.write_point(cv, pk.commit([random_poly], [random_blind])[0])
    y = T.squeeze_challenge()
    reg = _Reg()
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
        comp = lambda es: _horner([_lower(e, reg, ext) for e in es], th)
        terms.append(X.Product(l0, X.Sum(one, X.Negated(z0))))
        terms.append(X.Product(l_last, X.Sum(X.Product(z0, z0), X.Negated(z0))))
        lhs = X.Product(X.Product(z1, X.Sum(a_p, B)), X.Sum(s_p, G))
        rhs = X.Product(X.Product(z0, X.Sum(comp(ins), B)), X.Sum(comp(tabs), G))
        terms.append(X.Product(active, X.Sum(lhs, X.Negated(rhs))))
        terms.append(X.Product(l0, X.Sum(a_p, X.Negated(s_p))))
        terms.append(X.Product(X.Product(active, X.Sum(a_p, X.Negated(s_p))), X.Sum(a_p, X.Negated(a_m1))))
    h_coeffs = ops.expr(X.Product(_horner(terms, X.Constant(y)), reg.q('tinv')), reg.cols, en)
    ops.ntt_(h_coeffs, c.extended_k, 1, pk.eomega, pk.zeta, True)
    npieces = c.degree - 1
    if npieces * n < en and bool(h_coeffs[npieces * n:].any().item()):
        raise ValueError("quotient has higher degree than expected: the witness does not satisfy the constraints")
    h_pieces = [h_coeffs[i * n:(i + 1) * n] for i in range(npieces)]
    h_blinds = [rng.scalar() for _ in h_pieces]
    for pt in pk.commit(h_pieces, h_blinds):
        T.write_point(cv, pt)
    x = T.squeeze_challenge()
    xn = pow(x, n, p)
    rot = lambda r: x * pow(pk.omega, r % n, p) % p
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
    # h(X) = sum_i x^(n i) h_i(X): Horner from the top piece
    h_poly = ops.expr(_horner([X.Query(i) for i in reversed(range(npieces))], X.Constant(xn)), h_pieces, n)
    h_blind = 0
    for b in reversed(h_blinds):
        h_blind = (h_blind * xn + b) % p
    q = []
    for col, r in c.instance_queries:
        q.append((('inst', col), rot(r), (inst_polys[col], 1)))
    for col, r in c.advice_queries:
        q.append((('adv', col), rot(r), (adv_polys[col], adv_blinds[col])))
    for i, d in enumerate(perm):
        q.append((('pz', i), rot(0), (d['poly'], d['blind'])))
        q.append((('pz', i), rot(1), (d['poly'], d['blind'])))
        if i != nsets - 1:
            q.append((('pz', i), rot(last_rot), (d['poly'], d['blind'])))
    for i, d in enumerate(lk):
        q.append((('lz', i), rot(0), (d['z_poly'], d['z_blind'])))
        q.append((('la', i), rot(0), (d['a_poly'], d['a_blind'])))
        q.append((('ls', i), rot(0), (d['s_poly'], d['s_blind'])))
        q.append((('la', i), rot(-1), (d['a_poly'], d['a_blind'])))
        q.append((('lz', i), rot(1), (d['z_poly'], d['z_blind'])))
    for col, r in c.fixed_queries:
        q.append((('fix', col), rot(r), (pk.fixed_polys[col], 1)))
    for j, sp in enumerate(pk.sigma_polys):
        q.append((('sig', j), rot(0), (sp, 1)))
    q.append((('h', 0), rot(0), (h_poly, h_blind)))
    q.append((('rand', 0), rot(0), (random_poly, random_blind)))
    x1 = T.squeeze_challenge()
    x2 = T.squeeze_challenge()
    point_sets, groups = _query_sets(q)
    q_polys, q_blinds = [], []
    X1 = X.Constant(x1)
    for pts, grp in zip(point_sets, groups):
        polys = [pays[0][0] for _, pays in grp]
        blind = 0
        for _, pays in grp:
            blind = (blind * x1 + pays[0][1]) % p
        # Horner in x1 over the group's polynomials, in chunks that fit the evaluator's slot file
        acc = None
        for s0 in range(0, len(polys), 16):
            part = polys[s0:s0 + 16]
            leaves = [X.Query(i + (1 if acc is not None else 0)) for i in range(len(part))]
            tree = _horner(([X.Query(0)] if acc is not None else []) + leaves, X1)
            acc = ops.expr(tree, ([acc] if acc is not None else []) + part, n)
        q_polys.append(acc)
        q_blinds.append(blind)
    ev_jobs = [(qp, ptv) for pts, qp in zip(point_sets, q_polys) for ptv in pts]
    ev = pk.evals([j[0] for j in ev_jobs], [j[1] for j in ev_jobs])
    f_parts, o = [], 0
    for pts, poly in zip(point_sets, q_polys):
        evs = ev[o:o + len(pts)]
        o += len(pts)
        r_poly = _lagrange_interpolate(pts, evs, p)
        rcol = ops.zeros(n)
        rcol[:len(r_poly)] = up(r_poly)
        arr = ops.expr(X.Sum(X.Query(0), X.Negated(X.Query(1))), [poly, rcol], n)
        for ptv in pts:
            arr = ops.kate(arr, ptv)
        full = ops.zeros(n)
        full[:arr.shape[0]] = arr
        f_parts.append(full)
    f_poly = f_parts[0] if len(f_parts) == 1 else ops.expr(_horner([X.Query(i) for i in range(len(f_parts))], X.Constant(x2)), f_parts, n)
    f_blind = rng.scalar()
    T.write_point(cv, pk.commit([f_poly], [f_blind])[0])
    x3 = T.squeeze_challenge()
    for v in pk.evals(q_polys, [x3] * len(q_polys)):
        T.write_scalar(v)
    x4 = T.squeeze_challenge()
    cols4 = [f_poly] + q_polys
    p_poly = ops.expr(_horner([X.Query(i) for i in range(len(cols4))], X.Constant(x4)), cols4, n)
    p_blind = f_blind
    for blind in q_blinds:
        p_blind = (p_blind * x4 + blind) % p
    ops.ipa_open(pk.bases, p_poly, p_blind, x3, rng.rest(), T)
    return T.proof()
