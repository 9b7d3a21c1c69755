"""The unsaturated-limb quotient kernels (csrc/quotient_program.hpp, program2_source29) skip the carry pass after an addition
or subtraction wherever the emitter's limb bounds say the consumer does not need it.  A wrong bound there would not show in
any parity test -- the inputs that reach a worst case are astronomically rare -- so this test re-derives every bound
INDEPENDENTLY from the emitted source text: it walks the straight-line code of the Board and Shot kernels statement by
statement with exact integers, one bound PER LIMB (the emitter keeps two numbers per value), the bias limbs computed from the
modulus here, and checks the preconditions of every operation as csrc/fe29.cuh states them:
  mulx        every column of the 9 x 9 schoolbook product, with its reduction terms and carry, stays below 2^64;
              the operands' values multiply to less than 2^515 (the Montgomery result is then below 2 p)
  dot2x       a b + c d into the same columns with one reduction: the same two conditions on the sums
  fe29_add    no limb passes 2^32
  sub_lazy    J copies of the (K / J) p bias dominate the subtrahend limb by limb; no limb of the result passes 2^32
  fe29_carry  (any input); fe29_fold: carried input, value below 128 p
  the store   fe29_to_sat_div32: exact-digit pass without 32-bit overflow, value below 128 p
The reference computes the same polynomial h(X) in halo2_proofs 0.2.0 `plonk::prover::create_proof` ("h_poly", UPSTREAM) with
canonical field elements; SURVEY section 8 a6."""
import ctypes
import os
import re

import pytest


@pytest.fixture(scope="module")
def bzh2_lib():
    import __graft_entry__ as g
    import bzh2
    if not os.path.exists(bzh2.lib_path()):
        g.build()
    return bzh2

P_FP = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001   # Pallas base field = Vesta scalar field
M29 = (1 << 29) - 1


def limbs_of(v):
    return [(v >> (29 * i)) & M29 if i < 8 else v >> (29 * 8) for i in range(9)]


def bias(p, k):
    """fe29_bias<P, K>: K p with limbs 0..7 raised by 2^30 (limb 0) / 2^30 - 2 (the others) and limb 8 lowered by 2"""
    d = limbs_of(k * p)
    return [d[0] + (1 << 30)] + [d[i] + (1 << 30) - 2 for i in range(1, 8)] + [d[8] - 2]


class Val:
    __slots__ = ("L", "V")

    def __init__(self, L, V):
        self.L, self.V = list(L), V      # per-limb upper bounds (integers), value bound in units of p (float)


CARRIED = [(1 << 29) + 8] * 8
LEAF = Val(CARRIED + [(1 << 23) + 16], 2.0)       # fe29_from_sat_reduced: carried, below 2 p
ZERO = Val([0] * 9, 0.0)


def carry(a):
    assert all(x < (1 << 32) for x in a.L)
    L = [M29] + [M29 + (a.L[i - 1] >> 29) for i in range(1, 8)] + [a.L[8] + (a.L[7] >> 29)]
    return Val(L, a.V)


def check_program(text, p):
    plimbs = limbs_of(p)
    assert plimbs[0] == 1 and plimbs[5] == plimbs[6] == plimbs[7] == 0
    env = {}
    stats = {"mul": 0, "add": 0, "sub": 0, "carry": 0, "fold": 0}
    ident = r"[a-z][a-z0-9]*"
    body = text[text.index("__global__"):]
    stored = False
    for ln in body.split("\n"):
        ln = ln.strip()
        if not ln or ln.startswith("//") or "sched_barrier" in ln or ln in ("}", "{"):
            continue
        m = re.match(r"(?:const )?Fe29<P> (%s) = fe29_load_(planes_g|planes|const)<P>\(" % ident, ln)
        if m:
            env[m.group(1)] = LEAF
            continue
        if re.match(r"Fe29<P> r0 = fe29_zero<P>\(\), r1 = r0, r2 = r0, r3 = r0;", ln):
            for r in ("r0", "r1", "r2", "r3"):
                env[r] = ZERO
            continue
        m = re.match(r"Fe29<P> (s\d+) = r0;", ln)
        if m:
            env[m.group(1)] = ZERO
            continue
        if ln.startswith("{ const Fe29<P> z = fe29_zero<P>();"):
            env["z"] = ZERO
            continue
        m = re.match(r"(%s) = mulx\((%s), (%s)\);" % (ident, ident, ident), ln)
        if m:
            a, b = env[m.group(2)], env[m.group(3)]
            assert a.V * b.V <= 128.0, ln                      # a b < 2^515: (a b + m p) / 2^261 < 2 p
            run = 0                                            # carry into column k
            for k in range(17):
                col = M29 if k < 9 else 0
                col += sum(a.L[j] * b.L[k - j] for j in range(max(0, k - 8), min(k, 8) + 1))
                # reduction terms that land here: digit (< 2^29) of column k - l times limb l of p, l = 1..4 and 8
                col += sum(M29 * plimbs[l] for l in (1, 2, 3, 4, 8) if 0 <= k - l < 9)
                col += run
                assert col < (1 << 64), (ln, k)
                run = col >> 29
            env[m.group(1)] = Val([M29] * 8 + [1 << 23], 2.0)
            stats["mul"] += 1
            continue
        m = re.match(r"(%s) = dot2x\((%s), (%s), (%s), cv \+ \d+ \* 12\);" % (ident, ident, ident, ident), ln)
        if m:   # a b + c d with one reduction, d a per-proof constant (carried, below 2 p)
            a, b, c, d = env[m.group(2)], env[m.group(3)], env[m.group(4)], LEAF
            assert a.V * b.V + c.V * d.V <= 128.0, ln
            run = 0
            for k in range(17):
                col = M29 if k < 9 else 0
                col += sum(a.L[j] * b.L[k - j] + c.L[j] * d.L[k - j] for j in range(max(0, k - 8), min(k, 8) + 1))
                col += sum(M29 * plimbs[l] for l in (1, 2, 3, 4, 8) if 0 <= k - l < 9)
                col += run
                assert col < (1 << 64), (ln, k)
                run = col >> 29
            env[m.group(1)] = Val([M29] * 8 + [1 << 23], 2.0)
            stats["mul"] += 2
            stats["dot2"] = stats.get("dot2", 0) + 1
            continue
        m = re.match(r"(%s) = fe29_add\((%s), (%s)\);" % (ident, ident, ident), ln)
        if m:
            a, b = env[m.group(2)], env[m.group(3)]
            L = [x + y for x, y in zip(a.L, b.L)]
            assert all(x < (1 << 32) for x in L) and a.V + b.V <= 128.0, ln
            env[m.group(1)] = Val(L, a.V + b.V)
            stats["add"] += 1
            continue
        m = re.match(r"(%s) = fe29_sub_lazy<P, (\d+), (\d+)>\((%s), (%s)\);" % (ident, ident, ident), ln)
        if m:
            K, J = int(m.group(2)), int(m.group(3))
            a, b = env[m.group(4)], env[m.group(5)]
            assert K % J == 0 and b.V <= K
            bl = [J * x for x in bias(p, K // J)]
            assert all(bl[i] >= b.L[i] for i in range(9)), ln  # nothing goes negative
            L = [a.L[i] + bl[i] for i in range(9)]
            assert all(x < (1 << 32) for x in L) and a.V + K <= 128.0, ln
            env[m.group(1)] = Val(L, a.V + K)
            stats["sub"] += 1
            continue
        m = re.match(r"(%s) = fe29_carry\((%s)\);" % (ident, ident), ln)
        if m:
            assert m.group(1) == m.group(2)
            env[m.group(1)] = carry(env[m.group(2)])
            stats["carry"] += 1
            continue
        m = re.match(r"(%s) = fe29_fold\((%s)\);" % (ident, ident), ln)
        if m:
            a = env[m.group(2)]
            assert all(x <= (1 << 29) + 8 for x in a.L[:8]) and a.L[8] < (1 << 29) and a.V <= 128.0, ln
            env[m.group(1)] = Val(CARRIED + [(1 << 23) + 16], 2.0)
            stats["fold"] += 1
            continue
        m = re.match(r"(%s) = (%s);" % (ident, ident), ln)
        if m:
            env[m.group(1)] = env[m.group(2)]
            continue
        if ln.startswith("fe_store(out + (v * size + r) * 8, fe29_to_sat_div32(r0));"):
            a = env["r0"]
            c = 0
            for i in range(8):
                assert a.L[i] + c < (1 << 32), "exact-digit pass"
                c = (a.L[i] + c) >> 29
            assert a.L[8] + c < (1 << 29) and a.V <= 128.0
            stored = True
            break
        if re.match(r"(const size_t|const uint32_t\*|if \(r >= size\)|extern|const size_t\* __restrict__ strides)", ln) or ln.startswith("const size_t") or "__global__" in ln:
            continue
        raise AssertionError("statement the checker does not know: " + ln)
    assert stored
    return stats


@pytest.mark.parametrize("kind_name", ["ShotCircuit", "BoardCircuit"])
def test_every_operation_of_the_unsaturated_quotient_kernel_is_inside_its_bounds(bzh2_lib, kind_name):
    from bzh2 import circuits as Cm
    L = bzh2_lib.load()
    L.bzh_quotient_source_for_circuit.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                                  ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint64)]
    kind, k = (Cm.SHOT, 11) if kind_name == "ShotCircuit" else (Cm.BOARD, 12)
    lay = Cm.CircuitLayout(kind, k)
    blob = lay.blob()
    lay.close()
    ln, h = ctypes.c_size_t(), ctypes.c_uint64()
    assert L.bzh_quotient_source_for_circuit(0, blob, len(blob), None, 0, ctypes.byref(ln), ctypes.byref(h)) == 0
    buf = ctypes.create_string_buffer(ln.value + 1)
    assert L.bzh_quotient_source_for_circuit(0, blob, len(blob), buf, ln.value + 1, ctypes.byref(ln), ctypes.byref(h)) == 0
    text = buf.value.decode()
    ns = "namespace bzh_q29_%016x {" % h.value
    assert ns in text
    stats = check_program(text[text.index(ns):], P_FP)
    # the circuits' known sizes: 475 / 356 products with the default six per-group and six cross-group shared-subexpression slots; far fewer carry passes than additions + subtractions
    assert stats["mul"] == (475 if kind_name == "BoardCircuit" else 356), stats
    assert stats["carry"] < (stats["add"] + stats["sub"]) // 2 and stats["dot2"] >= 40, stats


def test_the_checker_rejects_a_missing_carry_pass():
    """the checker itself: a subtrahend that is a lazy sum of three carried values is above the bias limbs"""
    src = """extern "C" __global__ void k() {
    Fe29<P> r0 = fe29_zero<P>(), r1 = r0, r2 = r0, r3 = r0;
    const Fe29<P> la0 = fe29_load_planes<P>(x);
    const Fe29<P> lb0 = fe29_load_planes<P>(x);
    r0 = fe29_add(la0, lb0);
    r0 = fe29_add(r0, la0);
    r1 = fe29_sub_lazy<P, 8, 1>(la0, r0);
    """
    with pytest.raises(AssertionError):
        check_program(src, P_FP)
