"""CPU ORACLE (test infrastructure, NOT product code): run oracle/halo2_oracle.py's prover and verifier at the
reference's real circuit sizes (ShotCircuit k = 11, BoardCircuit k = 12: benches/shot.rs:22, benches/board.rs:22) by
handing the bulk arithmetic to the C oracle (oracle/oracle.c) while every protocol decision -- message order, RNG draw
order, blinding rows, query grouping, constraint order -- stays in halo2_oracle.create_proof, unchanged.

`with accelerated(threads):` swaps, for the duration of the block,
  Curve.msm_naive            -> orc_msm  (upstream best_multiexp, thread-chunked Pippenger)
  Domain.lagrange_to_coeff / coeff_to_extended / extended_to_coeff -> orc_ntt (upstream best_fft)
  pasta.eval_polynomial      -> orc_eval_poly (Horner)
  pasta.fold_bases           -> orc_generator_collapse (parallel_generator_collapse)
  halo2_oracle.quotient_evals-> the gate terms through orc_gate_eval (one postfix pass per extended row), the
                                permutation / lookup terms through halo2_oracle._constraint_expressions as before
Each replacement computes the same mathematical function as the big-int code it replaces;
tests/test_oracle_accel_cpu.py asserts that the accelerated prover emits the big-int prover's bytes."""
from __future__ import annotations

import contextlib

import coracle as C
import halo2_oracle as H
import pasta as O

_CID = {'vesta': 0, 'pallas': 1, 'bn254': 2}
_FID = {O.FP.p: 0, O.FQ.p: 1}


def _fid(F):
    return _FID[F.p]


def _quotient_evals(threads):
    def quotient_evals(keys, cosets, perm, lk, beta, gamma, theta, y):
        cs, dom = keys.cs, keys.dom
        F = dom.F
        p, n, ext, en = F.p, cs.n, dom.ext, dom.en
        last_rot = -(cs.blinding_factors + 1)
        if cs.gates:
            prog, consts, colmap = C.compile_gates(cs.gates)
            cols = [None] * len(colmap)
            for (t, c), i in colmap.items():
                cols[i] = C.ints_to_array(cosets[t][c])
            gate_fold = C.array_to_ints(C.gate_eval(_fid(F), prog, consts, cols, y, 0, en, threads=threads, rot_scale=ext))
        else:
            gate_fold = [0] * en
        # x^n - 1 takes only en / n distinct values on the extended coset
        xn_minus_1_inv = [F.inv((pow(dom.zeta * pow(dom.eomega, r, p) % p, n, p) - 1) % p) for r in range(ext)]
        h_eval = []
        xr = dom.zeta
        for r in range(en):
            col_at = lambda t, c, rot: cosets[t][c][(r + rot * ext) % en]
            z_at = lambda i, key: perm[i]['coset'][(r + {0: 0, 1: ext, 'last': last_rot * ext}[key]) % en]
            lk_at = lambda i, nm, rot: lk[i][nm + '_coset'][(r + rot * ext) % en]
            sig_at = lambda j: keys.sigma_cosets[j][r]
            rest = H._constraint_expressions(keys, col_at, z_at, lk_at, sig_at, keys.l0[r], keys.l_last[r], keys.l_blind[r], xr,
                                             beta, gamma, theta, skip_gates=True)
            acc = gate_fold[r]
            for t in rest:
                acc = (acc * y + t) % p
            h_eval.append(acc * xn_minus_1_inv[r % ext] % p)
            xr = xr * dom.eomega % p
        return h_eval
    return quotient_evals


@contextlib.contextmanager
def accelerated(threads: int = 8):
    C.lib()
    saved = (O.Curve.msm_naive, H.Domain.lagrange_to_coeff, H.Domain.coeff_to_extended, H.Domain.extended_to_coeff,
             O.eval_polynomial, O.fold_bases, H.quotient_evals)

    def msm(self, scalars, points):
        p = self.scalar.p
        return C.array_to_point(C.msm(_CID[self.name], C.ints_to_array([int(s) % p for s in scalars]), C.points_to_array(points), threads))

    def lagrange_to_coeff(self, v):
        return C.array_to_ints(C.ntt(_fid(self.F), C.ints_to_array(v), self.omega, inverse=True, threads=threads))

    def coeff_to_extended(self, c):
        a = C.ints_to_array(list(c) + [0] * (self.en - len(c)))
        return C.array_to_ints(C.ntt(_fid(self.F), a, self.eomega, coset_shift=self.zeta, threads=threads))

    def extended_to_coeff(self, e):
        # inverse NTT, then undo the coset: orc_ntt multiplies coefficient i by shift^-i
        return C.array_to_ints(C.ntt(_fid(self.F), C.ints_to_array(e), self.eomega, inverse=True, coset_shift=self.zeta, threads=threads))

    def eval_polynomial(coeffs, x, field):
        if len(coeffs) < 64:
            return saved[4](coeffs, x, field)
        return C.eval_poly(_fid(field), C.ints_to_array(coeffs), x)

    def fold_bases(curve, g, u):
        out = C.generator_collapse(_CID[curve.name], C.points_to_array(g), u, threads)
        return [C.array_to_point(out[i]) for i in range(out.shape[0])]

    O.Curve.msm_naive = msm
    H.Domain.lagrange_to_coeff, H.Domain.coeff_to_extended, H.Domain.extended_to_coeff = lagrange_to_coeff, coeff_to_extended, extended_to_coeff
    O.eval_polynomial, O.fold_bases, H.quotient_evals = eval_polynomial, fold_bases, _quotient_evals(threads)
    try:
        yield
    finally:
        (O.Curve.msm_naive, H.Domain.lagrange_to_coeff, H.Domain.coeff_to_extended, H.Domain.extended_to_coeff,
         O.eval_polynomial, O.fold_bases, H.quotient_evals) = saved
