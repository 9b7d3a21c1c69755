"""A circuit as DATA for bzh_pk_create (include/bzh2.h): constraint-system description + fixed columns, and the
serialiser of the "BZC1" blob (format at the top of csrc/prove.hip).  The reference's own circuits come ready-made from the
C++ front end (bzh2/circuits.py -> "BZC2" blobs); this class serves hand-built circuits (tests, tools).

Expressions are tuples: ('const', v) ('advice'|'fixed'|'instance', col, rot) ('neg', e) ('add', a, b) ('mul', a, b) ('scale', e, k)."""
from __future__ import annotations

MODULI = {0: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
          1: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001}
TWO_ADICITY, MULT_GEN = 32, 5


def _arr(ints):
    buf = b"".join(int(v).to_bytes(32, "little") for v in ints)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def _ints(a):
    b = np.ascontiguousarray(a, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


class Circuit:
    """Constraint system + fixed assignment, as data.  Expressions are tuples:
    ('const', v) ('advice'|'fixed'|'instance', col, rot) ('neg', e) ('add', a, b) ('mul', a, b) ('scale', e, k)."""

    def __init__(self, k, num_advice, num_fixed, num_instance, gates, perm_columns, lookups, fixed, copies, degree=None):
        self.k, self.n = k, 1 << k
        self.num_advice, self.num_fixed, self.num_instance = num_advice, num_fixed, num_instance
        self.gates, self.perm_columns = list(gates), list(perm_columns)
        self.lookups = [(list(a), list(t)) for a, t in lookups]
        self.fixed, self.copies = fixed, list(copies)
        qs = []
        for g in self.gates:
            _queries(g, qs)
        for a, t in self.lookups:
            for e in a + t:
                _queries(e, qs)
        for c in self.perm_columns:
            if (c[0], c[1], 0) not in qs:
                qs.append((c[0], c[1], 0))
        self.advice_queries = [(c, r) for t, c, r in qs if t == 'advice']
        self.fixed_queries = [(c, r) for t, c, r in qs if t == 'fixed']
        self.instance_queries = [(c, r) for t, c, r in qs if t == 'instance']
        deg = 3
        for g in self.gates:
            deg = max(deg, _degree(g))
        for a, t in self.lookups:
            deg = max(deg, 4, 2 + max([1] + [_degree(e) for e in a]) + max([1] + [_degree(e) for e in t]))
        self.degree = max(deg, degree or 0)
        per_col = {}
        for c, _ in self.advice_queries:
            per_col[c] = per_col.get(c, 0) + 1
        self.blinding_factors = max(3, max(per_col.values()) if per_col else 1) + 2
        self.usable_rows = self.n - (self.blinding_factors + 1)
        self.chunk_len = self.degree - 2
        self.extended_k = k + max(1, (self.degree - 2).bit_length())


_KIND_TAG = {'advice': 0, 'fixed': 1, 'instance': 2}


def _ser_expr(e, p, out):
    t = e[0]
    if t == 'const':
        out.append(b"\x00" + (e[1] % p).to_bytes(32, "little"))
    elif t in _KIND_TAG:
        out.append(bytes([1 + _KIND_TAG[t]]) + int(e[1]).to_bytes(4, "little") + int(e[2]).to_bytes(4, "little", signed=True))
    elif t == 'neg':
        out.append(b"\x04")
        _ser_expr(e[1], p, out)
    elif t in ('add', 'mul'):
        out.append(b"\x05" if t == 'add' else b"\x06")
        _ser_expr(e[1], p, out)
        _ser_expr(e[2], p, out)
    elif t == 'scale':
        out.append(b"\x07")
        _ser_expr(e[1], p, out)
        out.append((e[2] % p).to_bytes(32, "little"))
    else:
        raise ValueError("unknown expression tag %r" % (t,))


def serialize_circuit(c: "Circuit", p: int, vk_repr: int = 0x1234, min_degree: int | None = None) -> bytes:
    """The circuit blob of bzh_pk_create (format: csrc/prove.hip)."""
    u32 = lambda v: int(v).to_bytes(4, "little")
    out = [b"BZC1", u32(c.k), u32(c.num_advice), u32(c.num_fixed), u32(c.num_instance),
           u32(c.degree if min_degree is None else min_degree), (vk_repr % p).to_bytes(32, "little")]
    out.append(u32(len(c.gates)))
    for g in c.gates:
        _ser_expr(g, p, out)
    out.append(u32(len(c.perm_columns)))
    for kind, idx in c.perm_columns:
        out.append(bytes([_KIND_TAG[kind]]) + u32(idx))
    out.append(u32(len(c.lookups)))
    for ins, tabs in c.lookups:
        assert len(ins) == len(tabs)
        out.append(u32(len(ins)))
        for e in ins + tabs:
            _ser_expr(e, p, out)
    out.append(u32(len(c.copies)))
    for (lc, lr), (rc, rr) in c.copies:
        out.append(u32(lc) + u32(lr) + u32(rc) + u32(rr))
    for col in c.fixed:
        col = list(col)[:c.n]
        out.append(u32(len(col)))
        out.append(b"".join((int(v) % p).to_bytes(32, "little") for v in col))
    return b"".join(out)


def _degree(e):
    t = e[0]
    if t == 'const':
        return 0
    if t in ('advice', 'fixed', 'instance'):
        return 1
    if t in ('neg', 'scale'):
        return _degree(e[1])
    return max(_degree(e[1]), _degree(e[2])) if t == 'add' else _degree(e[1]) + _degree(e[2])


def _queries(e, out):
    t = e[0]
    if t in ('advice', 'fixed', 'instance'):
        if (t, e[1], e[2]) not in out:
            out.append((t, e[1], e[2]))
    elif t in ('neg', 'scale'):
        _queries(e[1], out)
    elif t in ('add', 'mul'):
        _queries(e[1], out)
        _queries(e[2], out)
