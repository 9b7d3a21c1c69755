"""CPU-side checks of the product library: it loads, exports every symbol
include/bzh2.h declares, and its host helpers agree with the oracle.  No device
compute here (that is tests/test_gpu_*.py, -m gpu)."""
import os
import random
import re

import numpy as np
import pytest

import coracle as C
import pasta as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bzh2_lib():
    import __graft_entry__ as g
    import bzh2
    if not os.path.exists(bzh2.lib_path()):
        g.build()
    return bzh2


def test_exports_every_declared_symbol(bzh2_lib):
    hdr = open(os.path.join(ROOT, "include", "bzh2.h")).read()
    declared = set(re.findall(r"\b(bzh_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"bzh_ctx", "bzh_bases"}
    assert declared == set(bzh2_lib.EXPORTS)
    L = bzh2_lib.load()
    for name in declared:
        assert hasattr(L, name), name


def test_no_gpu_is_an_error_not_a_fallback(bzh2_lib):
    if bzh2_lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(bzh2_lib.BzhError) as e:
        bzh2_lib.Context(0)
    assert e.value.status == bzh2_lib.E_NOGPU


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "battlezips-halo2_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "coracle" not in src and "liboracle" not in src and "import pasta" not in src, f


def test_field_omega_matches_oracle(bzh2_lib):
    for fid, F in O.FIELD_BY_ID.items():
        for k in (0, 1, 5, 11, 14, 17, 22, F.S):
            if k > F.S:
                continue
            assert bzh2_lib.limbs_to_int(bzh2_lib.field_omega(fid, k)) == F.omega(k)
            m = bzh2_lib.limbs_to_int(bzh2_lib.field_omega(fid, k, bzh2_lib.FORM_MONTGOMERY))
            assert m == F.omega(k) * F.R % F.p
        with pytest.raises(bzh2_lib.BzhError):
            bzh2_lib.field_omega(fid, F.S + 1)


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_normalize_and_compress_match_oracle(bzh2_lib, cid):
    rng = random.Random(40 + cid)
    cv = O.CURVE_BY_ID[cid]
    p = cv.p
    pts = [cv.random_point(rng) for _ in range(9)] + [None]
    jac = []
    for pt in pts:
        if pt is None:
            jac += [0, 0, 0]
        else:
            z = rng.randrange(1, p)
            jac += [pt[0] * z * z % p, pt[1] * z ** 3 % p, z]
    arr = C.ints_to_array(jac).reshape(-1, 12)
    aff = bzh2_lib.jacobian_to_affine(cid, arr)
    assert [C.array_to_point(a) for a in aff] == pts
    assert bzh2_lib.affine_compress(cid, aff) == [cv.compress(pt) for pt in pts]
    # Montgomery-form round trip
    R = cv.base.R
    arr_m = C.ints_to_array([x * R % p for x in jac]).reshape(-1, 12)
    aff_m = bzh2_lib.jacobian_to_affine(cid, arr_m, bzh2_lib.FORM_MONTGOMERY)
    assert bzh2_lib.affine_compress(cid, aff_m, bzh2_lib.FORM_MONTGOMERY) == [cv.compress(pt) for pt in pts]


def test_cpp_example_client_compiles_against_the_header():
    """examples/shot_prover.cpp (the reference's benches/shot.rs against the C ABI) is the compiled-code client of include/bzh2.h (what a Rust shim's call sequence looks like):
    it has to keep compiling with a plain C++ compiler against the header alone (no HIP, no torch)."""
    import shutil
    import subprocess
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([cxx, "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "shot_prover.cpp")])


def test_builtin_quotient_kernels_cover_the_references_circuits(bzh2_lib):
    """The quotient kernels of ShotCircuit and BoardCircuit are generated when the library is built (csrc/gen_quotient.cpp)
    and found by program hash; the program depends on the circuit only, so the hash is the same at every k -- checked here on
    the host (bzh_quotient_source_for_circuit needs no GPU), together with the generated table inside libbzh2.so."""
    import ctypes
    from bzh2 import circuits as Cm
    L = bzh2_lib.load()

    class Entry(ctypes.Structure):
        _fields_ = [("program_hash", ctypes.c_uint64), ("launch", ctypes.c_void_p), ("name", ctypes.c_char_p), ("launch29", ctypes.c_void_p)]
    L.bzh_builtin_quotients.restype = ctypes.POINTER(Entry)
    L.bzh_builtin_quotients.argtypes = [ctypes.POINTER(ctypes.c_size_t)]
    n = ctypes.c_size_t()
    tab = L.bzh_builtin_quotients(ctypes.byref(n))
    table = {tab[i].name.decode(): tab[i].program_hash for i in range(n.value)}
    assert set(table) == {"ShotCircuit", "BoardCircuit"} and all(tab[i].launch and tab[i].launch29 for i in range(n.value))
    L.bzh_quotient_source_for_circuit.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                                  ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint64)]
    for kind, name, ks in ((Cm.SHOT, "ShotCircuit", (11, 13)), (Cm.BOARD, "BoardCircuit", (12, 14))):
        for k in ks:
            lay = Cm.CircuitLayout(kind, k)
            blob = lay.blob()
            lay.close()
            ln, h = ctypes.c_size_t(), ctypes.c_uint64()
            assert L.bzh_quotient_source_for_circuit(0, blob, len(blob), None, 0, ctypes.byref(ln), ctypes.byref(h)) == 0
            assert h.value == table[name], (name, k)
            buf = ctypes.create_string_buffer(ln.value + 1)
            assert L.bzh_quotient_source_for_circuit(0, blob, len(blob), buf, ln.value + 1, ctypes.byref(ln), ctypes.byref(h)) == 0
            text = buf.value.decode()
            assert ("bzh_quotient_%016x" % h.value) in text and ("bzh_quotient29_%016x" % h.value) in text   # both limb flavours
    assert L.bzh_quotient_source_for_circuit(0, b"junk", 4, None, 0, ctypes.byref(ln), ctypes.byref(h)) == bzh2_lib.E_ARG
