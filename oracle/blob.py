"""CPU ORACLE (test infrastructure, NOT product code): decoder of the circuit blob the product hands to
bzh_pk_create ("BZC1" / "BZC2", format at the top of battlezips-halo2_amd/csrc/prove.hip), into the plain-data
constraint system the oracle prover / verifier / mock prover work on (oracle/halo2_oracle.py).

The blob is the drop-in for what halo2_proofs 0.2.0 holds in `VerifyingKey::cs` + the fixed columns after
keygen (reference call sites keygen_vk / keygen_pk: benches/shot.rs:60-61, benches/board.rs:53-54)."""
from __future__ import annotations

import struct

KINDS = ('advice', 'fixed', 'instance')


class _Reader:
    def __init__(self, b: bytes):
        self.b, self.o = b, 0

    def u8(self):
        v = self.b[self.o]
        self.o += 1
        return v

    def u32(self):
        v = struct.unpack_from("<I", self.b, self.o)[0]
        self.o += 4
        return v

    def i32(self):
        v = struct.unpack_from("<i", self.b, self.o)[0]
        self.o += 4
        return v

    def fe(self):
        v = int.from_bytes(self.b[self.o:self.o + 32], "little")
        self.o += 32
        return v

    def expr(self):
        t = self.u8()
        if t == 0:
            return ('const', self.fe())
        if t in (1, 2, 3):
            col = self.u32()
            return (KINDS[t - 1], col, self.i32())
        if t == 4:
            return ('neg', self.expr())
        if t == 5:
            a = self.expr()
            return ('add', a, self.expr())
        if t == 6:
            a = self.expr()
            return ('mul', a, self.expr())
        if t == 7:
            a = self.expr()
            return ('scale', a, self.fe())
        raise ValueError("bad expression tag %d" % t)


class DecodedCircuit:
    pass


def decode(blob: bytes) -> DecodedCircuit:
    r = _Reader(blob)
    magic = blob[:4]
    assert magic in (b"BZC1", b"BZC2"), magic
    r.o = 4
    c = DecodedCircuit()
    c.k, c.num_advice, c.num_fixed, c.num_instance, c.min_degree = r.u32(), r.u32(), r.u32(), r.u32(), r.u32()
    c.vk_repr = r.fe()
    c.n = 1 << c.k
    c.gates = [r.expr() for _ in range(r.u32())]                       # flattened constraint polynomials
    c.perm_columns = [(KINDS[r.u8()], r.u32()) for _ in range(r.u32())]
    c.lookups = []
    for _ in range(r.u32()):
        m = r.u32()
        ins = [r.expr() for _ in range(m)]
        c.lookups.append((ins, [r.expr() for _ in range(m)]))
    c.copies = [((r.u32(), r.u32()), (r.u32(), r.u32())) for _ in range(r.u32())]
    c.fixed = []
    for _ in range(c.num_fixed):
        ln = r.u32()
        c.fixed.append([r.fe() for _ in range(ln)] + [0] * (c.n - ln))
    c.queries = None
    if magic == b"BZC2":
        qs = []
        for _ in range(3):
            qs.append([(r.u32(), r.i32()) for _ in range(r.u32())])
        c.queries = tuple(qs)
    assert r.o == len(blob), (r.o, len(blob))
    return c
