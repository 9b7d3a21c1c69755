"""TEST HELPER (not product code; the product path is libbzh2.so behind include/bzh2.h).
Host-side expression trees and their compiler for bzh_expr_eval (row a13).

Mirrors halo2_proofs::plonk::Expression<F> (UPSTREAM 0.2.0; built by every `meta.create_gate`
of the reference: src/chips/bitify.rs:63-88, src/chips/placement.rs:126-265,
src/chips/transpose.rs:60-88, src/chips/shot.rs:228-297, src/chips/board.rs:264-321):
    Constant(F), Column query (Fixed/Advice/Instance at a Rotation), Negated, Sum, Product, Scaled.
Column queries are resolved by the caller to (column index, rotation in extended-domain steps);
challenges (theta, beta, gamma, y) enter as constants.  `compile_expression` emits the straight-line
program of include/bzh2.h (bzh_expr_op) with at most BZH_EXPR_MAX_SLOTS live intermediates.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

ADD, SUB, MUL, NEG, COPY = 0, 1, 2, 3, 4
SLOT, COLUMN, CONST = 0, 1, 2
MAX_SLOTS = 24


class Expr:
    def __add__(self, o):
        return Sum(self, o)

    def __sub__(self, o):
        return Sum(self, Negated(o))

    def __mul__(self, o):
        return Scaled(self, o) if isinstance(o, int) else Product(self, o)

    def __neg__(self):
        return Negated(self)


@dataclass(frozen=True)
class Constant(Expr):
    value: int


@dataclass(frozen=True)
class Symbol(Expr):
    """A constant bound at evaluation time (a challenge): the program is compiled once per proving key and every
    proof supplies its own value through Program.bind."""
    name: object


@dataclass(frozen=True)
class Query(Expr):
    """Fixed / Advice / Instance query: column index into the caller's column list, rotation in rows
    of the evaluation domain (already multiplied by the extension factor)."""
    column: int
    rotation: int = 0


@dataclass(frozen=True)
class Negated(Expr):
    a: Expr


@dataclass(frozen=True)
class Sum(Expr):
    a: Expr
    b: Expr


@dataclass(frozen=True)
class Product(Expr):
    a: Expr
    b: Expr


@dataclass(frozen=True)
class Scaled(Expr):
    a: Expr
    k: int


class ExprOp(ctypes.Structure):
    _fields_ = [("op", ctypes.c_uint8), ("dst", ctypes.c_uint8), ("a_kind", ctypes.c_uint8), ("b_kind", ctypes.c_uint8),
                ("a_idx", ctypes.c_int32), ("b_idx", ctypes.c_int32), ("a_rot", ctypes.c_int32), ("b_rot", ctypes.c_int32)]


class Program:
    def __init__(self):
        self.ops = []          # (op, dst, (kind, idx, rot), (kind, idx, rot))
        self.consts = []       # ints, or Symbol leaves to be bound per evaluation
        self._const_ix = {}
        self.result_slot = 0

    def const(self, v: int) -> int:
        if v not in self._const_ix:
            self._const_ix[v] = len(self.consts)
            self.consts.append(v)
        return self._const_ix[v]

    def bind(self, env: dict) -> list:
        """constant values for one evaluation: Symbols looked up in `env`"""
        return [env[c.name] if isinstance(c, Symbol) else c for c in self.consts]

    def as_array(self):
        arr = (ExprOp * len(self.ops))()
        for i, (op, dst, a, b) in enumerate(self.ops):
            arr[i] = ExprOp(op, dst, a[0], b[0], a[1], b[1], a[2], b[2])
        return arr


def _depth(e: Expr, memo: dict) -> int:
    """Sethi-Ullman number: slots needed to evaluate e when leaves are free operands (memoised by node identity:
    the quotient's Horner chain is hundreds of nodes deep)."""
    k = id(e)
    if k in memo:
        return memo[k]
    if isinstance(e, (Constant, Query, Symbol)):
        d = 0
    elif isinstance(e, (Negated, Scaled)):
        d = max(1, _depth(e.a, memo))
    else:
        da, db = _depth(e.a, memo), _depth(e.b, memo)
        d = max(da, db) if da != db else da + 1
    memo[k] = d
    return d


def compile_expression(e: Expr, modulus: int) -> Program:
    prog = Program()
    free = list(range(MAX_SLOTS - 1, -1, -1))
    memo: dict = {}

    def alloc():
        if not free:
            raise ValueError("expression needs more than %d live intermediates" % MAX_SLOTS)
        return free.pop()

    def operand(x):
        """-> (kind, idx, rot), slot_to_release_or_None"""
        if isinstance(x, Constant):
            return (CONST, prog.const(x.value % modulus), 0), None
        if isinstance(x, Symbol):
            return (CONST, prog.const(x), 0), None
        if isinstance(x, Query):
            return (COLUMN, x.column, x.rotation), None
        s = emit(x)
        return (SLOT, s, 0), s

    def emit(x) -> int:
        if isinstance(x, (Constant, Query, Symbol)):
            a, _ = operand(x)
            d = alloc()
            prog.ops.append((COPY, d, a, (SLOT, 0, 0)))
            return d
        if isinstance(x, Negated):
            a, ra = operand(x.a)
            d = ra if ra is not None else alloc()
            prog.ops.append((NEG, d, a, (SLOT, 0, 0)))
            return d
        if isinstance(x, Scaled):
            a, ra = operand(x.a)
            d = ra if ra is not None else alloc()
            prog.ops.append((MUL, d, a, (CONST, prog.const(x.k % modulus), 0)))
            return d
        # evaluate the deeper child first so the shallower one never needs more slots than are left
        first_b = _depth(x.b, memo) > _depth(x.a, memo)
        if first_b:
            b, rb = operand(x.b)
            a, ra = operand(x.a)
        else:
            a, ra = operand(x.a)
            b, rb = operand(x.b)
        d = ra if ra is not None else (rb if rb is not None else alloc())
        prog.ops.append((ADD if isinstance(x, Sum) else MUL, d, a, b))
        for r in (ra, rb):
            if r is not None and r != d:
                free.append(r)
        return d

    prog.result_slot = emit(e)
    return prog


def evaluate_tree(e: Expr, columns, row: int, size: int, p: int, env: dict | None = None) -> int:
    """Direct (recursive) evaluation of the tree at one row -- the definition."""
    if isinstance(e, Constant):
        return e.value % p
    if isinstance(e, Symbol):
        return env[e.name] % p
    if isinstance(e, Query):
        return columns[e.column][(row + e.rotation) % size]
    if isinstance(e, Negated):
        return (-evaluate_tree(e.a, columns, row, size, p, env)) % p
    if isinstance(e, Scaled):
        return evaluate_tree(e.a, columns, row, size, p, env) * e.k % p
    a, b = evaluate_tree(e.a, columns, row, size, p, env), evaluate_tree(e.b, columns, row, size, p, env)
    return (a + b) % p if isinstance(e, Sum) else a * b % p
