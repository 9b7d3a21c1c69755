"""ctypes loader for oracle/liboracle.so (CPU ORACLE -- test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this.  numpy arrays of dtype uint64 carry field elements as rows of 4 LE limbs
(canonical form); points are rows of 8 limbs (x||y), (0,0) = identity.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        _LIB.orc_msm.argtypes = [ctypes.c_int, u64p, u64p, ctypes.c_size_t, ctypes.c_int, u64p]
        _LIB.orc_msm_naive.argtypes = [ctypes.c_int, u64p, u64p, ctypes.c_size_t, u64p]
        _LIB.orc_point_mul.argtypes = [ctypes.c_int, u64p, u64p, u64p]
        _LIB.orc_point_walk.argtypes = [ctypes.c_int, u64p, ctypes.c_size_t, u64p]
        _LIB.orc_ntt.argtypes = [ctypes.c_int, u64p, ctypes.c_uint, u64p, ctypes.c_int, u64p, ctypes.c_int]
        _LIB.orc_field_mul.argtypes = [ctypes.c_int, u64p, u64p, u64p]
        _LIB.orc_field_inv.argtypes = [ctypes.c_int, u64p, u64p]
        _LIB.orc_field_consts.argtypes = [ctypes.c_int, u64p, u64p, u64p, u64p]
        _LIB.orc_eval_poly.argtypes = [ctypes.c_int, u64p, ctypes.c_size_t, u64p, u64p]
        _LIB.orc_gate_eval.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_size_t, u64p, ctypes.c_size_t,
                                       ctypes.POINTER(u64p), ctypes.c_size_t, ctypes.c_uint, u64p, ctypes.c_size_t, ctypes.c_size_t,
                                       ctypes.c_int, u64p]
        _LIB.orc_generator_collapse.argtypes = [ctypes.c_int, u64p, ctypes.c_size_t, u64p, ctypes.c_int]
    return _LIB


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def int_to_limbs(x: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint64).copy()


def limbs_to_int(a) -> int:
    return int.from_bytes(np.ascontiguousarray(a, dtype=np.uint64).tobytes(), "little")


def ints_to_array(xs) -> np.ndarray:
    """list of ints -> (n,4) uint64 canonical limbs."""
    buf = b"".join(int(x).to_bytes(32, "little") for x in xs)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def array_to_ints(a: np.ndarray):
    b = np.ascontiguousarray(a, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def points_to_array(pts) -> np.ndarray:
    """list of (x,y)|None -> (n,8) uint64."""
    buf = b"".join((bytes(64) if p is None else int(p[0]).to_bytes(32, "little") + int(p[1]).to_bytes(32, "little"))
                   for p in pts)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).copy()


def array_to_point(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(8)
    x, y = limbs_to_int(a[:4]), limbs_to_int(a[4:])
    return None if (x == 0 and y == 0) else (x, y)


def msm(curve_id: int, scalars: np.ndarray, points: np.ndarray, threads: int = 1):
    n = scalars.shape[0]
    assert scalars.shape == (n, 4) and points.shape == (n, 8)
    out = np.zeros(8, dtype=np.uint64)
    rc = lib().orc_msm(curve_id, _p(scalars), _p(points), n, threads, _p(out))
    assert rc == 0
    return out


def msm_naive(curve_id: int, scalars: np.ndarray, points: np.ndarray):
    n = scalars.shape[0]
    out = np.zeros(8, dtype=np.uint64)
    rc = lib().orc_msm_naive(curve_id, _p(scalars), _p(points), n, _p(out))
    assert rc == 0
    return out


def point_walk(curve_id: int, g_xy: np.ndarray, n: int) -> np.ndarray:
    out = np.zeros((n, 8), dtype=np.uint64)
    g = np.ascontiguousarray(g_xy, dtype=np.uint64).reshape(8)
    rc = lib().orc_point_walk(curve_id, _p(g), n, _p(out))
    assert rc == 0
    return out


def ntt(field_id: int, data: np.ndarray, omega: int, inverse: bool = False, coset_shift: int | None = None,
        threads: int = 1) -> np.ndarray:
    n = data.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    a = np.ascontiguousarray(data, dtype=np.uint64).copy()
    w = int_to_limbs(omega)
    cs = int_to_limbs(coset_shift) if coset_shift is not None else None
    rc = lib().orc_ntt(field_id, _p(a), log_n, _p(w), int(inverse), _p(cs) if cs is not None else None, threads)
    assert rc == 0
    return a


def field_mul(field_id: int, a: int, b: int) -> int:
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_field_mul(field_id, _p(int_to_limbs(a)), _p(int_to_limbs(b)), _p(out))
    return limbs_to_int(out)


def field_inv(field_id: int, a: int) -> int:
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_field_inv(field_id, _p(int_to_limbs(a)), _p(out))
    return limbs_to_int(out)


def eval_poly(field_id: int, coeffs: np.ndarray, x: int) -> int:
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_eval_poly(field_id, _p(np.ascontiguousarray(coeffs)), coeffs.shape[0], _p(int_to_limbs(x)), _p(out))
    return limbs_to_int(out)


def compile_gates(gates):
    """halo2-style expression trees (oracle/halo2_oracle.py AST) -> the postfix program of orc_gate_eval:
    (prog (m, 3) int32, consts list of ints).  One 'end of polynomial' op per gate: acc = acc * y + value."""
    prog, consts, cidx = [], [], {}

    def const(v):
        if v not in cidx:
            cidx[v] = len(consts)
            consts.append(v)
        return cidx[v]

    def emit(e, colmap):
        t = e[0]
        if t == 'const':
            prog.append((0, const(e[1]), 0))
        elif t in ('advice', 'fixed', 'instance'):
            prog.append((1, colmap[(t, e[1])], e[2]))
        elif t == 'neg':
            emit(e[1], colmap)
            prog.append((2, 0, 0))
        elif t == 'scale':
            emit(e[1], colmap)
            prog.append((5, const(e[2]), 0))
        else:
            emit(e[1], colmap)
            emit(e[2], colmap)
            prog.append((3 if t == 'add' else 4, 0, 0))
    colmap = {}

    def cols_of(e):
        t = e[0]
        if t in ('advice', 'fixed', 'instance'):
            colmap.setdefault((t, e[1]), len(colmap))
        elif t in ('neg', 'scale'):
            cols_of(e[1])
        elif t in ('add', 'mul'):
            cols_of(e[1])
            cols_of(e[2])
    for g in gates:
        cols_of(g)
    for g in gates:
        emit(g, colmap)
        prog.append((6, 0, 0))
    return np.array(prog, dtype=np.int32).reshape(-1, 3), consts, colmap


def gate_eval(field_id: int, prog: np.ndarray, consts, cols, y: int, row_lo: int, row_hi: int, threads: int = 1, rot_scale: int = 1):
    """cols: list of (size, 4) uint64 canonical arrays (size a power of two); rotations in the program are multiplied by
    rot_scale (extended-domain steps per row rotation).  Returns (row_hi - row_lo, 4) uint64."""
    size = cols[0].shape[0]
    log_size = size.bit_length() - 1
    pg = np.ascontiguousarray(prog, dtype=np.int32).copy()
    if rot_scale != 1:
        m = pg[:, 0] == 1
        pg[m, 2] *= rot_scale
    cs = ints_to_array(consts) if consts else np.zeros((1, 4), dtype=np.uint64)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    arr = (u64p * len(cols))(*[_p(np.ascontiguousarray(c)) for c in cols])
    keep = [np.ascontiguousarray(c) for c in cols]
    arr = (u64p * len(cols))(*[_p(c) for c in keep])
    out = np.zeros((row_hi - row_lo, 4), dtype=np.uint64)
    rc = lib().orc_gate_eval(field_id, pg.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), pg.shape[0], _p(cs), len(consts), arr, len(cols),
                             log_size, _p(int_to_limbs(y)), row_lo, row_hi, threads, _p(out))
    assert rc == 0
    return out


def generator_collapse(curve_id: int, g: np.ndarray, u: int, threads: int = 1) -> np.ndarray:
    """g: (2 * half, 8) affine points; returns the (half, 8) collapsed generators g_lo + [u] g_hi."""
    a = np.ascontiguousarray(g, dtype=np.uint64).copy()
    half = a.shape[0] // 2
    rc = lib().orc_generator_collapse(curve_id, _p(a), half, _p(int_to_limbs(u)), threads)
    assert rc == 0
    return a[:half]
