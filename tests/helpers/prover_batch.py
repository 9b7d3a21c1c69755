"""TEST HELPER (not product code; the product path is libbzh2.so behind include/bzh2.h).
Lockstep batch prover: B independent proofs of the same circuit advance phase by phase, so that every MSM, NTT,
gate-evaluation, scan and IPA round is ONE launch carrying all B proofs (throughput mode: the kernels of a single
proof are latency-bound).  Protocol, message order and randomness draw order per proof are those of
bzh2/prover_dev.create_proof, so proof b is byte-identical to what create_proof emits for witness b alone.

Reference seam: the batch workloads of BASELINE.json configs[2..3] (many Shot / Board proofs per GPU), each proof
being halo2_proofs::plonk::create_proof as called from benches/shot.rs:68 and benches/board.rs:61-68.

Layout: a per-proof column family is one tensor (B, columns, rows, 4) of Montgomery limbs; proving-key columns
(fixed, permutation, l0 / l_last / l_blind ...) are shared.  Gate programs are compiled once per proving key with
the challenges as expr.Symbol leaves and bound per proof (bzh_expr_eval_batch: per-vector column strides and
constant rows)."""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from bzh2 import FORM_MONTGOMERY, permute_expression_pair
from . import expr as X
from .prover import _lagrange_interpolate, _query_sets, _Rng
from .prover_dev import DeviceProvingKey, _horner

_POOL = ThreadPoolExecutor(max_workers=16)


class _Cols:
    """column registry for one batched evaluation: name -> index; entries are (data_ptr, stride_elems, keepalive)"""

    def __init__(self):
        self.index, self.cols = {}, []

    def add_batched(self, name, t: torch.Tensor, i: int):
        """column i of a (B, m, rows, 4) tensor"""
        if name not in self.index:
            m, rows = t.shape[1], t.shape[2]
            self.index[name] = len(self.cols)
            self.cols.append((t.data_ptr() + i * rows * 32, m * rows, t))
        return self.index[name]

    def add_vec(self, name, t: torch.Tensor):
        """a (B, rows, 4) tensor"""
        if name not in self.index:
            self.index[name] = len(self.cols)
            self.cols.append((t.data_ptr(), t.shape[1], t))
        return self.index[name]

    def add_strided(self, name, t: torch.Tensor, offset_rows: int, stride_rows: int):
        if name not in self.index:
            self.index[name] = len(self.cols)
            self.cols.append((t.data_ptr() + offset_rows * 32, stride_rows, t))
        return self.index[name]

    def add_shared(self, name, t: torch.Tensor):
        if name not in self.index:
            self.index[name] = len(self.cols)
            self.cols.append((t.data_ptr(), 0, t))
        return self.index[name]

    def q(self, name, rot=0):
        return X.Query(self.index[name], rot)

    def args(self):
        return [(c[0], c[1]) for c in self.cols]


def _lower(e, reg: _Cols, rot_scale):
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


class BatchProver:
    """Per-proving-key state of the lockstep prover: compiled programs and the multiopen structure."""

    def __init__(self, pk: DeviceProvingKey):
        self.pk = pk
        self._progs = {}
        self._open = None
        self.trace = None        # set to a list to collect (phase, seconds) with a device sync at every phase boundary

    def _mark(self, name):
        if self.trace is not None:
            import time
            torch.cuda.synchronize()
            self.trace.append((name, time.perf_counter()))

    # ---- compiled programs -------------------------------------------------------------------
    def _program(self, key, build):
        """build() -> expression tree (with Symbol leaves); compiled once, the constant table pre-converted"""
        ent = self._progs.get(key)
        if ent is None:
            ops = self.pk.ops
            prog = X.compile_expression(build(), self.pk.p)
            base = np.zeros((len(prog.consts), 4), dtype=np.uint64)
            sym = []
            for i, cst in enumerate(prog.consts):
                if isinstance(cst, X.Symbol):
                    sym.append((i, cst.name))
                else:
                    base[i] = ops.limbs_mont(cst)
            ent = (prog, prog.as_array(), base, sym)
            self._progs[key] = ent
        return ent

    def _run(self, key, build, reg: _Cols, envs, size: int) -> torch.Tensor:
        prog, arr, base, sym = self._program(key, build)
        ops = self.pk.ops
        B = len(envs)
        if sym:
            rows = np.repeat(base[None], B, axis=0)
            for b, env in enumerate(envs):
                for i, name in sym:
                    rows[b, i] = ops.limbs_mont(env[name])
        else:
            rows = base[None]
        return ops.expr_batch(prog, arr, reg.args(), rows, size, B)

    # ---- batched transforms / commitments ----------------------------------------------------
    def to_coeff(self, t: torch.Tensor) -> torch.Tensor:
        """(..., n, 4) Lagrange values -> coefficients (new tensor)"""
        pk = self.pk
        out = t.contiguous().clone()
        pk.ops.ntt_(out, pk.c.k, out.numel() // (pk.c.n * 4), pk.omega, None, True)
        return out

    def to_extended(self, polys: torch.Tensor) -> torch.Tensor:
        """(..., n, 4) coefficients -> (..., 8n-coset values)"""
        pk = self.pk
        n, en = pk.c.n, pk.en
        out = torch.zeros((*polys.shape[:-2], en, 4), dtype=torch.int64, device=polys.device)
        out[..., :n, :] = polys
        pk.ops.ntt_(out, pk.c.extended_k, out.numel() // (en * 4), pk.eomega, pk.zeta, False)
        return out

    def commit(self, polys: torch.Tensor, blinds) -> list:
        """polys (m, n, 4) (any leading shape flattened), blinds: m ints -> m affine points"""
        pk = self.pk
        n = pk.c.n
        flat = polys.reshape(-1, n, 4)
        sc = pk.ops.zeros(flat.shape[0], n + 2)
        sc[:, :n] = flat
        sc[:, n + 1] = pk.ops.upload(blinds)
        return pk.ops.msm(pk.bases, sc)

    def evals(self, stacked: torch.Tensor, points) -> list:
        """stacked (m, n, 4) contiguous, points: m ints"""
        return self.pk.ops.evals(stacked, points)

    # ---- multiopen structure (identical for every proof; only the point VALUES differ) -----------
    def _open_structure(self, nsets, nl):
        if self._open is not None:
            return self._open
        c = self.pk.c
        last_rot = -(c.blinding_factors + 1)
        q = []
        for col, r in c.instance_queries:
            q.append((('inst', col), r, ('inst', col)))
        for col, r in c.advice_queries:
            q.append((('adv', col), r, ('adv', col)))
        for i in range(nsets):
            q.append((('pz', i), 0, ('pz', i)))
            q.append((('pz', i), 1, ('pz', i)))
            if i != nsets - 1:
                q.append((('pz', i), last_rot, ('pz', i)))
        for i in range(nl):
            q.append((('lz', i), 0, ('lz', i)))
            q.append((('la', i), 0, ('la', i)))
            q.append((('ls', i), 0, ('ls', i)))
            q.append((('la', i), -1, ('la', i)))
            q.append((('lz', i), 1, ('lz', i)))
        for col, r in c.fixed_queries:
            q.append((('fix', col), r, ('fix', col)))
        for j in range(len(c.perm_columns)):
            q.append((('sig', j), 0, ('sig', j)))
        q.append((('h', 0), 0, ('h', 0)))
        q.append((('rand', 0), 0, ('rand', 0)))
        # rotations stand in for the points: distinct rotations <-> distinct points x * omega^r
        rot_sets, groups = _query_sets(q)
        self._open = (rot_sets, [[cid for cid, _ in grp] for grp in groups])
        return self._open


def create_proofs(bp: BatchProver, advice, instances, rng_list, transcripts) -> list:
    """advice: (B, n_advice, n, 4) device tensor of Montgomery limbs (usable rows filled; the blinding rows are
    overwritten), or a list of B lists of columns (int lists / tensors).  instances: B lists of instance columns
    (int lists).  rng_list: B byte strings, 64 bytes per Field::random draw in draw order.  Returns B proofs."""
    pk = bp.pk
    c, ops, p, cv = pk.c, pk.ops, pk.p, pk.curve
    n, bf, usable, ext, en = c.n, c.blinding_factors, c.usable_rows, pk.ext, pk.en
    B = len(transcripts)
    Ts = transcripts
    rngs = [_Rng(rb, p) for rb in rng_list]
    up = ops.upload
    dev = ops.dev
    R = range(B)
    for T in Ts:
        T.common_scalar(pk.vk_repr)

    def draw_rows(count):
        """the next `count` Field::random draws of every proof, reduced on the device -> (B * count, 4)"""
        return ops.random_field(b"".join([rngs[b].take(count) for b in R]), B * count)

    bp._mark('start')
    # ---- instance columns ----------------------------------------------------------------------
    ni = len(instances[0])
    inst = ops.zeros(B, ni, n)
    for i in range(ni):
        ln = len(instances[0][i])
        if ln:
            inst[:, i, :ln] = up([v for b in R for v in instances[b][i]]).view(B, ln, 4)
    inst_polys = bp.to_coeff(inst)
    if ni:
        pts = bp.commit(inst_polys, [1] * (B * ni))
        for b in R:
            for i in range(ni):
                Ts[b].common_point(pts[b * ni + i])

    bp._mark('instance')
    # ---- advice columns ------------------------------------------------------------------------
    if isinstance(advice, torch.Tensor):
        adv = advice.clone()
    else:
        adv = torch.stack([torch.stack([col.to(dev) if isinstance(col, torch.Tensor) else up(list(col) + [0] * (n - len(col)))
                                        for col in cols]) for cols in advice])
    na = adv.shape[1]
    adv[:, :, usable:] = draw_rows(na * (n - usable)).view(B, na, n - usable, 4)
    adv_blinds = [[rngs[b].scalar() for _ in range(na)] for b in R]
    adv_polys = bp.to_coeff(adv)
    pts = bp.commit(adv_polys, [v for b in R for v in adv_blinds[b]])
    for b in R:
        for i in range(na):
            Ts[b].write_point(cv, pts[b * na + i])
    env = [{'theta': Ts[b].squeeze_challenge()} for b in R]
    cosets = {}

    def extend_witness():
        # queued late on purpose: with lookups these transforms run on the device while the host sorts
        if not cosets:
            cosets['inst'] = bp.to_extended(inst_polys)
            cosets['adv'] = bp.to_extended(adv_polys)

    def lag_registry():
        reg = _Cols()
        for i in range(na):
            reg.add_batched(('advice', i), adv, i)
        for i, a in enumerate(pk.fixed):
            reg.add_shared(('fixed', i), a)
        for i in range(ni):
            reg.add_batched(('instance', i), inst, i)
        return reg

    bp._mark('advice')
    # ---- lookups: compress, permute (host sort), commit -------------------------------------------
    TH = X.Symbol('theta')
    nl = len(c.lookups)
    lk = []
    for li, (ins, tabs) in enumerate(c.lookups):
        reg = lag_registry()
        a_c = bp._run(('lk_in', li), lambda: _horner([_lower(e, reg, 1) for e in ins], TH), reg, env, n)
        s_c = bp._run(('lk_tab', li), lambda: _horner([_lower(e, reg, 1) for e in tabs], TH), reg, env, n)
        ah = a_c.cpu().numpy().view(np.uint64)
        sh = s_c.cpu().numpy().view(np.uint64)
        extend_witness()
        # the sort runs on the host, one task per proof (the C call releases the GIL)
        res = list(_POOL.map(lambda b: permute_expression_pair(pk.field, ah[b], sh[b], usable, FORM_MONTGOMERY), R))
        a_s = ops.zeros(B, 2, n)
        a_s[:, 0, :usable] = torch.from_numpy(np.stack([r[0] for r in res]).view(np.int64)).to(dev)
        a_s[:, 1, :usable] = torch.from_numpy(np.stack([r[1] for r in res]).view(np.int64)).to(dev)
        a_s[:, :, usable:] = draw_rows(2 * (bf + 1)).view(B, 2, bf + 1, 4)
        blinds = [(rngs[b].scalar(), rngs[b].scalar()) for b in R]
        polys = bp.to_coeff(a_s)
        pts = bp.commit(polys, [v for b in R for v in blinds[b]])
        for b in R:
            Ts[b].write_point(cv, pts[2 * b])
            Ts[b].write_point(cv, pts[2 * b + 1])
        lk.append({'a_c': a_c, 's_c': s_c, 'as': a_s, 'polys': polys, 'blinds': blinds})
    extend_witness()
    inst_cosets, adv_cosets = cosets['inst'], cosets['adv']
    for b in R:
        env[b]['beta'] = Ts[b].squeeze_challenge()
        env[b]['gamma'] = Ts[b].squeeze_challenge()
    BETA, GAMMA = X.Symbol('beta'), X.Symbol('gamma')

    bp._mark('lookup')
    # ---- permutation and lookup grand products -----------------------------------------------------
    nsets = (len(c.perm_columns) + c.chunk_len - 1) // c.chunk_len if c.perm_columns else 0
    nz = nsets + nl
    zs = ops.zeros(B, max(nz, 1), n)
    z_blinds = [[] for _ in R]
    lag = lag_registry()

    def lag_col(reg, col):
        kind, i = col
        if kind == 'advice':
            reg.add_batched(col, adv, i)
        elif kind == 'fixed':
            reg.add_shared(col, pk.fixed[i])
        else:
            reg.add_batched(col, inst, i)
        return reg.q(col)

    def grand_product(num_key, num_build, den_key, den_build, reg, slot, prev_slot):
        den = bp._run(den_key, den_build, reg, env, n)
        z = bp._run(num_key, num_build, reg, env, n)
        ops.batch_invert_(den)
        ops.vec_mul_(z, den)
        ops.prefix_product_(z, n, B)
        if prev_slot is not None:   # chain the sets: start from the previous set's hand-over value z[usable]
            last = zs[:, prev_slot, usable:usable + 1].expand(B, n, 4).contiguous()
            ops.vec_mul_(z, last)
        z[:, n - bf:] = draw_rows(bf).view(B, bf, 4)
        for b in R:
            z_blinds[b].append(rngs[b].scalar())
        zs[:, slot] = z

    for i in range(nsets):
        cols = c.perm_columns[i * c.chunk_len:(i + 1) * c.chunk_len]
        reg = _Cols()

        def build(which, i=i, cols=cols, reg=reg):
            acc = None
            for j, col in enumerate(cols):
                gj = i * c.chunk_len + j
                v = lag_col(reg, col)
                reg.add_shared(('sigma', gj), pk.sigma[gj])
                reg.add_shared(('ident', gj), pk.ident[gj])
                if which == 'den':
                    f = X.Sum(X.Sum(X.Product(BETA, reg.q(('sigma', gj))), GAMMA), v)
                else:
                    f = X.Sum(X.Sum(X.Product(reg.q(('ident', gj)), BETA), GAMMA), v)
                acc = f if acc is None else X.Product(acc, f)
            return acc
        # the registry has to be populated before either program runs (and identically on cache hits)
        build('den')
        grand_product(('pnum', i), lambda: build('num'), ('pden', i), lambda: build('den'), reg, i, i - 1 if i else None)
    for li, d in enumerate(lk):
        reg = _Cols()
        reg.add_vec('a_c', d['a_c'])
        reg.add_vec('s_c', d['s_c'])
        reg.add_batched('a', d['as'], 0)
        reg.add_batched('s', d['as'], 1)
        grand_product(('lnum', li), lambda: X.Product(X.Sum(reg.q('a_c'), BETA), X.Sum(reg.q('s_c'), GAMMA)),
                      ('lden', li), lambda: X.Product(X.Sum(reg.q('a'), BETA), X.Sum(reg.q('s'), GAMMA)), reg, nsets + li, None)
    if nz:
        z_polys = bp.to_coeff(zs)
        pts = bp.commit(z_polys, [v for b in R for v in z_blinds[b]])
        for b in R:
            for i in range(nz):
                Ts[b].write_point(cv, pts[b * nz + i])
        z_cosets = bp.to_extended(z_polys)
    for d in lk:
        d['cosets'] = bp.to_extended(d['polys'])

    bp._mark('grand_products')
    # ---- vanishing argument ----------------------------------------------------------------------
    random_poly = ops.random_field(b"".join(rngs[b].take(n) for b in R), B * n).view(B, n, 4)
    random_blinds = [rngs[b].scalar() for b in R]
    pts = bp.commit(random_poly, random_blinds)
    for b in R:
        Ts[b].write_point(cv, pts[b])
        env[b]['y'] = Ts[b].squeeze_challenge()
        for gj in range(len(c.perm_columns)):
            env[b][('bd', gj)] = env[b]['beta'] * pow(pk.delta, gj, p) % p
    reg = _Cols()
    for i in range(na):
        reg.add_batched(('advice', i), adv_cosets, i)
    for i, a in enumerate(pk.fixed_cosets):
        reg.add_shared(('fixed', i), a)
    for i in range(ni):
        reg.add_batched(('instance', i), inst_cosets, i)
    for j, a in enumerate(pk.sigma_cosets):
        reg.add_shared(('sigma', j), a)
    for i in range(nsets):
        reg.add_batched(('pz', i), z_cosets, i)
    for i, d in enumerate(lk):
        reg.add_batched(('la', i), d['cosets'], 0)
        reg.add_batched(('ls', i), d['cosets'], 1)
        reg.add_batched(('lz', i), z_cosets, nsets + i)
    for nm, a in (('l0', pk.l0), ('l_last', pk.l_last), ('l_blind', pk.l_blind), ('X', pk.x_col), ('tinv', pk.tinv_col)):
        reg.add_shared(nm, a)
    last_rot = -(bf + 1)

    def build_quotient():
        one = X.Constant(1)
        l0, l_last = reg.q('l0'), reg.q('l_last')
        active = X.Sum(one, X.Negated(X.Sum(l_last, reg.q('l_blind'))))
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
                    left = X.Product(left, X.Sum(X.Sum(v, X.Product(BETA, reg.q(('sigma', gj)))), GAMMA))
                    cur = X.Product(X.Symbol(('bd', gj)), reg.q('X'))
                    right = X.Product(right, X.Sum(X.Sum(v, cur), GAMMA))
                terms.append(X.Product(active, X.Sum(left, X.Negated(right))))
        for i, (ins, tabs) in enumerate(c.lookups):
            z0, z1 = reg.q(('lz', i)), reg.q(('lz', i), ext)
            a_p, a_m1, s_p = reg.q(('la', i)), reg.q(('la', i), -ext), reg.q(('ls', i))
            comp = lambda es: _horner([_lower(e, reg, ext) for e in es], TH)
            terms.append(X.Product(l0, X.Sum(one, X.Negated(z0))))
            terms.append(X.Product(l_last, X.Sum(X.Product(z0, z0), X.Negated(z0))))
            lhs = X.Product(X.Product(z1, X.Sum(a_p, BETA)), X.Sum(s_p, GAMMA))
            rhs = X.Product(X.Product(z0, X.Sum(comp(ins), BETA)), X.Sum(comp(tabs), GAMMA))
            terms.append(X.Product(active, X.Sum(lhs, X.Negated(rhs))))
            terms.append(X.Product(l0, X.Sum(a_p, X.Negated(s_p))))
            terms.append(X.Product(X.Product(active, X.Sum(a_p, X.Negated(s_p))), X.Sum(a_p, X.Negated(a_m1))))
        return X.Product(_horner(terms, X.Symbol('y')), reg.q('tinv'))

    bp._mark('vanishing_setup')
    h = bp._run('quotient', build_quotient, reg, env, en)
    ops.ntt_(h, c.extended_k, B, pk.eomega, pk.zeta, True)
    npieces = c.degree - 1
    if npieces * n < en and bool(h[:, npieces * n:].any().item()):
        raise ValueError("quotient has higher degree than expected: a witness does not satisfy the constraints")
    h_blinds = [[rngs[b].scalar() for _ in range(npieces)] for b in R]
    pts = bp.commit(h[:, :npieces * n].reshape(B, npieces, n, 4), [v for b in R for v in h_blinds[b]])
    for b in R:
        for i in range(npieces):
            Ts[b].write_point(cv, pts[b * npieces + i])
    xs = [Ts[b].squeeze_challenge() for b in R]
    for b in R:
        env[b]['xn'] = pow(xs[b], n, p)
    wp = {}

    def rot(b, r):
        if r not in wp:
            wp[r] = pow(pk.omega, r % n, p)
        return xs[b] * wp[r] % p

    bp._mark('quotient+h_commit')
    # ---- evaluations: one gather of (polynomial, rotation) jobs out of the table of all committed polynomials ----
    nf, ns = len(pk.fixed_polys), len(pk.sigma_polys)
    if not hasattr(pk, '_fixed_stack'):
        pk._fixed_stack = torch.stack(list(pk.fixed_polys) + list(pk.sigma_polys)) if nf + ns else ops.zeros(0, n)
    parts, where, o = [], {}, 0

    def place(kind, t):
        nonlocal o
        for i in range(t.shape[1]):
            where[(kind, i)] = o + i
        o += t.shape[1]
        parts.append(t)
    place('inst', inst_polys)
    place('adv', adv_polys)
    shared = pk._fixed_stack.unsqueeze(0).expand(B, nf + ns, n, 4)
    for i in range(nf):
        where[('fix', i)] = o + i
    for j in range(ns):
        where[('sig', j)] = o + nf + j
    o += nf + ns
    parts.append(shared)
    place('rand', random_poly.unsqueeze(1))
    if nz:
        for i in range(nsets):
            where[('pz', i)] = o + i
        for i in range(nl):
            where[('lz', i)] = o + nsets + i
        o += nz
        parts.append(z_polys)
    for i, d in enumerate(lk):
        where[('la', i)] = o
        where[('ls', i)] = o + 1
        o += 2
        parts.append(d['polys'])
    table = torch.cat(parts, dim=1)
    jobs = [(('inst', col), r) for col, r in c.instance_queries]
    jobs += [(('adv', col), r) for col, r in c.advice_queries]
    jobs += [(('fix', col), r) for col, r in c.fixed_queries]
    jobs += [(('rand', 0), 0)]
    jobs += [(('sig', j), 0) for j in range(ns)]
    for i in range(nsets):
        jobs += [(('pz', i), 0), (('pz', i), 1)] + ([(('pz', i), last_rot)] if i != nsets - 1 else [])
    for i in range(nl):
        jobs += [(('lz', i), 0), (('lz', i), 1), (('la', i), 0), (('la', i), -1), (('ls', i), 0)]
    J = len(jobs)
    idx = torch.tensor([where[j[0]] for j in jobs], dtype=torch.int64, device=dev)
    gathered = table.index_select(1, idx)
    vals = bp.evals(gathered.view(B * J, n, 4), [rot(b, r) for b in R for _, r in jobs])
    for b in R:
        for v in vals[b * J:(b + 1) * J]:
            Ts[b].write_scalar(v)

    bp._mark('evaluations')
    # ---- h(X) = sum_i x^(n i) h_i(X) ---------------------------------------------------------------
    regh = _Cols()
    for i in range(npieces):
        regh.add_strided(i, h, i * n, en)
    h_poly = bp._run('h_poly', lambda: _horner([regh.q(i) for i in reversed(range(npieces))], X.Symbol('xn')), regh, env, n)
    h_blind = []
    for b in R:
        acc = 0
        for v in reversed(h_blinds[b]):
            acc = (acc * env[b]['xn'] + v) % p
        h_blind.append(acc)

    bp._mark('h_poly')
    # ---- multiopen ------------------------------------------------------------------------------
    rot_sets, groups = bp._open_structure(nsets, nl)
    blind_of = []
    for b in R:
        d = {('h', 0): h_blind[b], ('rand', 0): random_blinds[b]}
        for i in range(ni):
            d[('inst', i)] = 1
        for i in range(na):
            d[('adv', i)] = adv_blinds[b][i]
        for i in range(nf):
            d[('fix', i)] = 1
        for j in range(ns):
            d[('sig', j)] = 1
        for i in range(nsets):
            d[('pz', i)] = z_blinds[b][i]
        for i in range(nl):
            d[('lz', i)] = z_blinds[b][nsets + i]
            d[('la', i)] = lk[i]['blinds'][b][0]
            d[('ls', i)] = lk[i]['blinds'][b][1]
        blind_of.append(d)
    for b in R:
        env[b]['x1'] = Ts[b].squeeze_challenge()
        env[b]['x2'] = Ts[b].squeeze_challenge()

    def poly_col(reg, cid):
        if cid == ('h', 0):
            reg.add_vec(cid, h_poly)
        else:
            reg.add_strided(cid, table, where[cid] * n, table.shape[1] * n)
        return reg.q(cid)

    X1 = X.Symbol('x1')
    nq = len(rot_sets)
    q_polys = ops.zeros(B, nq, n)
    q_blinds = [[] for _ in R]
    for si, cids in enumerate(groups):
        for b in R:
            acc = 0
            for cid in cids:
                acc = (acc * env[b]['x1'] + blind_of[b][cid]) % p
            q_blinds[b].append(acc)
        # Horner in x1 over the group's polynomials, in chunks that fit the evaluator's slot file
        acc_t = None
        for s0 in range(0, len(cids), 16):
            part = cids[s0:s0 + 16]
            reg = _Cols()
            if acc_t is not None:
                reg.add_vec('acc', acc_t)

            def build(part=part, reg=reg, first=acc_t is None):
                leaves = [poly_col(reg, cid) for cid in part]
                return _horner(([] if first else [reg.q('acc')]) + leaves, X1)
            for cid in part:            # same registry contents whether or not the program is cached
                poly_col(reg, cid)
            acc_t = bp._run(('q', si, s0), build, reg, env, n)
        q_polys[:, si] = acc_t
    # evaluations of the q polynomials at their own points
    ev_jobs = [(si, r) for si, rs in enumerate(rot_sets) for r in rs]
    J2 = len(ev_jobs)
    idx2 = torch.tensor([j[0] for j in ev_jobs], dtype=torch.int64, device=dev)
    ev = bp.evals(q_polys.index_select(1, idx2).view(B * J2, n, 4), [rot(b, r) for b in R for _, r in ev_jobs])
    maxpts = max(len(rs) for rs in rot_sets)
    r_small = []
    for b in R:
        o2 = 0
        for rs in rot_sets:
            r_poly = _lagrange_interpolate([rot(b, r) for r in rs], ev[b * J2 + o2:b * J2 + o2 + len(rs)], p)
            o2 += len(rs)
            r_small += r_poly + [0] * (maxpts - len(r_poly))
    rcols = ops.zeros(B, nq, n)
    rcols[:, :, :maxpts] = up(r_small).view(B, nq, maxpts, 4)
    f_parts = ops.zeros(B, nq, n)
    for si, rs in enumerate(rot_sets):
        reg = _Cols()
        reg.add_batched('q', q_polys, si)
        reg.add_batched('r', rcols, si)
        arr = bp._run('q_minus_r', lambda: X.Sum(reg.q('q'), X.Negated(reg.q('r'))), reg, env, n)
        for r in rs:
            arr = ops.kate_batch(arr, [rot(b, r) for b in R])
        f_parts[:, si, :arr.shape[1]] = arr
    bp._mark('multiopen_q_kate')
    regf = _Cols()
    for si in range(nq):
        regf.add_batched(si, f_parts, si)
    if nq == 1:
        f_poly = f_parts[:, 0].contiguous()
    else:
        f_poly = bp._run('f_poly', lambda: _horner([regf.q(i) for i in range(nq)], X.Symbol('x2')), regf, env, n)
    f_blinds = [rngs[b].scalar() for b in R]
    pts = bp.commit(f_poly, f_blinds)
    x3s = []
    for b in R:
        Ts[b].write_point(cv, pts[b])
        x3s.append(Ts[b].squeeze_challenge())
    vals = bp.evals(q_polys.view(B * nq, n, 4), [x3s[b] for b in R for _ in range(nq)])
    for b in R:
        for v in vals[b * nq:(b + 1) * nq]:
            Ts[b].write_scalar(v)
        env[b]['x4'] = Ts[b].squeeze_challenge()
    regp = _Cols()
    regp.add_vec('f', f_poly)
    for si in range(nq):
        regp.add_batched(si, q_polys, si)
    p_poly = bp._run('p_poly', lambda: _horner([regp.q('f')] + [regp.q(i) for i in range(nq)], X.Symbol('x4')), regp, env, n)
    p_blinds = []
    for b in R:
        acc = f_blinds[b]
        for v in q_blinds[b]:
            acc = (acc * env[b]['x4'] + v) % p
        p_blinds.append(acc)
    bp._mark('multiopen_f_p')
    need = 64 * (n + 1 + 2 * c.k)
    rests = [rngs[b].rest()[:need] for b in R]
    ops.ipa_open_batch(pk.bases, p_poly, p_blinds, x3s, rests, Ts)
    bp._mark('ipa')
    return [T.proof() for T in Ts]
