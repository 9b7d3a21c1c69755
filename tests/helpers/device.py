"""TEST HELPER (not product code; the product path is libbzh2.so behind include/bzh2.h).
Device-resident field-element arrays for the prover pipeline: torch supplies HBM allocations and the
stream, every operation is a libbzh2.so call with BZH_MEM_DEVICE pointers in Montgomery form.  No torch
type crosses the C ABI (only data_ptr() integers)."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from bzh2 import FORM_MONTGOMERY, MEM_DEVICE, Bases, Context, int_to_limbs, jacobian_to_affine, limbs_to_int, load
from . import expr as X

R256 = 1 << 256
_VP = ctypes.c_void_p
_BASE_MODULUS = {0: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,   # Vesta base field = Fq
                 1: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001}   # Pallas base field = Fp


class DeviceOps:
    def __init__(self, ctx: Context, field: int, curve: int, modulus: int, device: torch.device):
        self.ctx, self.field, self.curve, self.p, self.dev = ctx, field, curve, modulus, device
        self.R = R256 % modulus
        self.Rinv = pow(self.R, modulus - 2, modulus)
        L = load()
        L.bzh_field_convert.argtypes = [_VP, ctypes.c_int, _VP, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        L.bzh_batch_invert.argtypes = [_VP, ctypes.c_int, _VP, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        L.bzh_prefix_product.argtypes = [_VP, ctypes.c_int, _VP, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        L.bzh_vec_mul.argtypes = [_VP, ctypes.c_int, _VP, _VP, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        L.bzh_eval_polynomial.argtypes = [_VP, ctypes.c_int, _VP, ctypes.c_size_t, ctypes.c_size_t, _VP, ctypes.c_size_t, ctypes.c_int,
                                          ctypes.c_int, _VP]
        L.bzh_kate_division.argtypes = [_VP, ctypes.c_int, _VP, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int,
                                        ctypes.c_int, _VP]
        L.bzh_expr_eval.argtypes = [_VP, ctypes.c_int, ctypes.POINTER(X.ExprOp), ctypes.c_size_t, ctypes.POINTER(_VP), ctypes.c_size_t,
                                    _VP, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP]
        L.bzh_ipa_open.argtypes = [_VP, _VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64),
                                   ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_size_t, _VP,
                                   ctypes.POINTER(ctypes.c_uint64)]
        self.L = L

    # ---- host <-> device ---------------------------------------------------------------------
    def mont(self, v: int) -> int:
        return v * self.R % self.p

    def upload(self, ints) -> torch.Tensor:
        """canonical ints -> Montgomery tensor (len, 4) int64 on the device"""
        buf = b"".join([int(v).to_bytes(32, "little") for v in ints])
        if not buf:
            return self.zeros(0)
        t = torch.frombuffer(bytearray(buf), dtype=torch.int64).view(-1, 4).to(self.dev)
        if t.shape[0]:      # Montgomery conversion on the device (one multiplication by R^2 per element)
            self._chk(self.L.bzh_field_convert(self.ctx.handle, self.field, _VP(t.data_ptr()), t.shape[0], 1, MEM_DEVICE),
                      "bzh_field_convert")
        return t

    def download(self, t: torch.Tensor):
        b = t.contiguous().cpu().numpy().tobytes()
        return [int.from_bytes(b[i:i + 32], "little") * self.Rinv % self.p for i in range(0, len(b), 32)]

    def random_field(self, raw: bytes, count: int) -> torch.Tensor:
        """`count` Field::random draws (64 RNG bytes each) reduced on the device -> Montgomery tensor (count, 4)."""
        out = self.zeros(count)
        self.L.bzh_random_field.argtypes = [_VP, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _VP]
        self._chk(self.L.bzh_random_field(self.ctx.handle, self.field, raw, count, FORM_MONTGOMERY, MEM_DEVICE, _VP(out.data_ptr())),
                  "bzh_random_field")
        return out

    def zeros(self, *shape) -> torch.Tensor:
        return torch.zeros((*shape, 4), dtype=torch.int64, device=self.dev)

    def limbs_mont(self, v: int) -> np.ndarray:
        return int_to_limbs(self.mont(v))

    def _chk(self, rc, where):
        self.ctx._check(rc, where)

    # ---- transforms (in place over `batch` contiguous vectors) ----------------------------------
    def ntt_(self, t: torch.Tensor, log_n: int, batch: int, omega: int, shift, inverse: bool):
        self.ctx.ntt_device(self.field, t.data_ptr(), log_n, batch, self.limbs_mont(omega),
                            None if shift is None else self.limbs_mont(shift), inverse, FORM_MONTGOMERY)

    # ---- commitments ------------------------------------------------------------------------
    def msm(self, bases: Bases, scalars: torch.Tensor):
        """scalars (B, m, 4) Montgomery, m <= len(bases) -> list of affine canonical int pairs (None = identity)"""
        B, m = scalars.shape[0], scalars.shape[1]
        out = torch.zeros((B, 12), dtype=torch.int64, device=self.dev)
        self.ctx.msm_device(bases, scalars.data_ptr(), m, B, out.data_ptr(), FORM_MONTGOMERY)
        jac = out.cpu().numpy().view(np.uint64)
        aff = jacobian_to_affine(self.curve, jac, FORM_MONTGOMERY)
        bp = _BASE_MODULUS[self.curve]
        rinv = pow(R256 % bp, bp - 2, bp)
        return [None if not a.any() else (limbs_to_int(a[:4]) * rinv % bp, limbs_to_int(a[4:]) * rinv % bp) for a in aff]

    # ---- element-wise / scans (in place) -----------------------------------------------------
    def batch_invert_(self, t):
        self._chk(self.L.bzh_batch_invert(self.ctx.handle, self.field, _VP(t.data_ptr()), t.numel() // 4, FORM_MONTGOMERY, MEM_DEVICE),
                  "bzh_batch_invert")

    def prefix_product_(self, t, n, batch=1):
        self._chk(self.L.bzh_prefix_product(self.ctx.handle, self.field, _VP(t.data_ptr()), n, batch, FORM_MONTGOMERY, MEM_DEVICE),
                  "bzh_prefix_product")

    def vec_mul_(self, a, b):
        self._chk(self.L.bzh_vec_mul(self.ctx.handle, self.field, _VP(a.data_ptr()), _VP(b.data_ptr()), a.numel() // 4,
                                     FORM_MONTGOMERY, MEM_DEVICE), "bzh_vec_mul")

    def evals(self, coeffs: torch.Tensor, points):
        """coeffs (B, n, 4); points: B canonical ints -> B canonical ints"""
        B, n = coeffs.shape[0], coeffs.shape[1]
        xs = self.upload(points)
        out = self.zeros(B)
        self._chk(self.L.bzh_eval_polynomial(self.ctx.handle, self.field, _VP(coeffs.data_ptr()), n, B, _VP(xs.data_ptr()), B,
                                             FORM_MONTGOMERY, MEM_DEVICE, _VP(out.data_ptr())), "bzh_eval_polynomial")
        return self.download(out)

    def kate(self, coeffs: torch.Tensor, x: int) -> torch.Tensor:
        n = coeffs.shape[0]
        out = self.zeros(n - 1)
        xl = self.limbs_mont(x)
        self._chk(self.L.bzh_kate_division(self.ctx.handle, self.field, _VP(coeffs.data_ptr()), n,
                                           xl.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), FORM_MONTGOMERY, MEM_DEVICE,
                                           _VP(out.data_ptr())), "bzh_kate_division")
        return out

    def expr(self, tree, columns, size: int) -> torch.Tensor:
        """Evaluate a bzh2.expr tree over device columns (each (size, 4)); constants are canonical ints."""
        prog = X.compile_expression(tree, self.p)
        ops = prog.as_array()
        ptrs = (_VP * max(len(columns), 1))(*[c.data_ptr() for c in columns])
        consts = np.ascontiguousarray(np.stack([self.limbs_mont(v) for v in prog.consts]) if prog.consts
                                      else np.zeros((1, 4), dtype=np.uint64))
        out = self.zeros(size)
        rc = self.L.bzh_expr_eval(self.ctx.handle, self.field, ops, len(prog.ops), ptrs, len(columns), _VP(consts.ctypes.data),
                                  len(prog.consts), size.bit_length() - 1, prog.result_slot, FORM_MONTGOMERY, MEM_DEVICE,
                                  _VP(out.data_ptr()))
        self._chk(rc, "bzh_expr_eval")
        return out

    # ---- batched (lockstep proofs) ---------------------------------------------------------------
    def expr_batch(self, prog, ops_arr, cols, const_rows, size: int, batch: int) -> torch.Tensor:
        """One launch of a compiled program over `batch` vectors.  cols: list of (data_ptr, stride_elems) with stride 0 for
        columns shared by all vectors; const_rows: (batch, nconsts, 4) uint64 Montgomery limbs (or (1, nconsts, 4) shared)."""
        L = self.L
        if not hasattr(L, "_bzh_expr_batch_bound"):
            L.bzh_expr_eval_batch.argtypes = [_VP, ctypes.c_int, ctypes.POINTER(X.ExprOp), ctypes.c_size_t, ctypes.POINTER(_VP),
                                              ctypes.POINTER(ctypes.c_size_t), ctypes.c_size_t, _VP, ctypes.c_size_t, ctypes.c_size_t,
                                              ctypes.c_uint, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _VP]
            L._bzh_expr_batch_bound = True
        nc = max(len(cols), 1)
        ptrs = (_VP * nc)(*[c[0] for c in cols])
        strides = (ctypes.c_size_t * nc)(*[c[1] for c in cols])
        nconst = const_rows.shape[1]
        cstride = nconst if const_rows.shape[0] > 1 else 0
        consts = np.ascontiguousarray(const_rows if nconst else np.zeros((1, 1, 4), dtype=np.uint64))
        out = torch.empty((batch, size, 4), dtype=torch.int64, device=self.dev)
        rc = L.bzh_expr_eval_batch(self.ctx.handle, self.field, ops_arr, len(prog.ops), ptrs, strides, len(cols), _VP(consts.ctypes.data),
                                   nconst, cstride, size.bit_length() - 1, prog.result_slot, batch, FORM_MONTGOMERY, MEM_DEVICE,
                                   _VP(out.data_ptr()))
        self._chk(rc, "bzh_expr_eval_batch")
        return out

    def kate_batch(self, coeffs: torch.Tensor, xs) -> torch.Tensor:
        """coeffs (B, m, 4) contiguous, xs: B canonical ints -> (B, m-1, 4): vector b divided by (X - xs[b])"""
        L = self.L
        B, m = coeffs.shape[0], coeffs.shape[1]
        L.bzh_kate_division_batch.argtypes = [_VP, ctypes.c_int, _VP, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64),
                                              ctypes.c_int, ctypes.c_int, _VP]
        out = torch.empty((B, m - 1, 4), dtype=torch.int64, device=self.dev)
        xl = np.ascontiguousarray(np.stack([self.limbs_mont(x) for x in xs]))
        self._chk(L.bzh_kate_division_batch(self.ctx.handle, self.field, _VP(coeffs.data_ptr()), m, B,
                                            xl.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), FORM_MONTGOMERY, MEM_DEVICE,
                                            _VP(out.data_ptr())), "bzh_kate_division_batch")
        return out

    def ipa_open_batch(self, bases: Bases, polys: torch.Tensor, blinds, x3s, rng_list, transcripts):
        """polys (B, n, 4) Montgomery on the device; per-proof blinds / points / rng bytes / transcripts"""
        L = self.L
        u64p = ctypes.POINTER(ctypes.c_uint64)
        L.bzh_ipa_open_batch.argtypes = [_VP, _VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, u64p, u64p, ctypes.c_char_p,
                                         ctypes.c_size_t, ctypes.POINTER(_VP), u64p]
        B = len(transcripts)
        stride = len(rng_list[0])
        bl = np.ascontiguousarray(np.stack([int_to_limbs(v) for v in blinds]))
        xs = np.ascontiguousarray(np.stack([int_to_limbs(v) for v in x3s]))
        out = np.zeros((B, 4), dtype=np.uint64)
        trs = (_VP * B)(*[t.h for t in transcripts])
        rc = L.bzh_ipa_open_batch(self.ctx.handle, bases.handle, _VP(polys.data_ptr()), FORM_MONTGOMERY, MEM_DEVICE, B,
                                  bl.ctypes.data_as(u64p), xs.ctypes.data_as(u64p), b"".join(rng_list), stride, trs,
                                  out.ctypes.data_as(u64p))
        self._chk(rc, "bzh_ipa_open_batch")
        return [limbs_to_int(o) for o in out]

    def ipa_open(self, bases: Bases, poly: torch.Tensor, blind: int, x3: int, rng_bytes: bytes, transcript) -> int:
        out = np.zeros(4, dtype=np.uint64)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        rc = self.L.bzh_ipa_open(self.ctx.handle, bases.handle, _VP(poly.data_ptr()), FORM_MONTGOMERY, MEM_DEVICE,
                                 int_to_limbs(blind).ctypes.data_as(u64p), int_to_limbs(x3).ctypes.data_as(u64p), rng_bytes,
                                 len(rng_bytes), transcript.h, out.ctypes.data_as(u64p))
        self._chk(rc, "bzh_ipa_open")
        return limbs_to_int(out)
