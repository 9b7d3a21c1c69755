"""Host mirror of the reference's game / witness marshalling (SURVEY section 8 rows a8, a9 and the
trace helpers of a2 / a5).  Same names, argument meaning and error behaviour as the reference;
100-bit fields are Python ints (bit i = board cell i, cell = 10*y + x).

    BinaryValue           src/utils/binary.rs:16-109   (256-bit little-endian bit array)
    Ship / ShipType       src/utils/ship.rs:10-33, 147-212, 220-311
    Deck                  src/utils/deck.rs:53-71
    Board.state / witness src/utils/board.rs:77-120
    serialize             src/utils/shot.rs:12-19
    compute_shot_trace    src/chips/shot.rs:28-51
    compute_placement_trace  src/chips/placement.rs:380-419
    pedersen_commit(_batch)  src/utils/pedersen.rs:17-28  -> GPU 2-term MSMs (bzh_msm, batch = #commitments)
"""
from __future__ import annotations

import numpy as np

BOARD_SIZE = 100  # src/utils/board.rs:12
SHIP_LENGTHS = (5, 4, 3, 3, 2)  # carrier, battleship, cruiser, submarine, destroyer (src/utils/ship.rs:24-33)
FP_MODULUS = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
FQ_MODULUS = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001

# WitnessOption (src/utils/ship.rs:315-331)
DEFAULT, DUAL_PLACEMENT, NONCONSECUTIVE, EXTRA_BIT, OVERSIZED, UNDERSIZED = range(6)


class BinaryValue:
    """256-bit value with bit access (src/utils/binary.rs:16-109)."""

    def __init__(self, value: int = 0):
        if not 0 <= value < (1 << 256):
            raise ValueError("BinaryValue is 256 bits")
        self.value = value

    @classmethod
    def from_repr(cls, b: bytes):
        return cls(int.from_bytes(b, "little"))

    @classmethod
    def from_u8(cls, v: int):
        return cls(v & 0xFF)

    @classmethod
    def empty(cls):
        return cls(0)

    def to_repr(self) -> bytes:
        return self.value.to_bytes(32, "little")

    def to_fp(self) -> int:
        """Fp::from_repr(..).unwrap(): a non-canonical value is an error upstream."""
        if self.value >= FP_MODULUS:
            raise ValueError("not a canonical Fp representation")
        return self.value

    def lower_u128(self) -> int:
        return self.value & ((1 << 128) - 1)

    def bit(self, i: int) -> int:
        return (self.value >> i) & 1

    def bitfield(self, s: int):
        return [(self.value >> i) & 1 for i in range(s)]

    def zip(self, to: "BinaryValue") -> "BinaryValue":
        """OR of the first 100 bits; both set is a panic upstream (src/utils/binary.rs:97-108)."""
        mask = (1 << BOARD_SIZE) - 1
        clash = self.value & to.value & mask
        if clash:
            raise ValueError("Cannot zip together bit #%d" % ((clash & -clash).bit_length() - 1))
        return BinaryValue((self.value | to.value) & mask)

    def __eq__(self, o):
        return isinstance(o, BinaryValue) and o.value == self.value

    def __repr__(self):
        return "BinaryValue(0x%x)" % self.value


class Ship:
    def __init__(self, ship_type: int, x: int, y: int, z: bool):
        self.ship_type, self.x, self.y, self.z = ship_type, x, y, z

    def length(self) -> int:
        return SHIP_LENGTHS[self.ship_type]

    def coordinates(self, transpose: bool):
        """src/utils/ship.rs:147-161: cell 10*y + x; with `transpose` a vertical ship is indexed 10*x + y."""
        out = []
        for i in range(self.length()):
            xi = self.x if self.z else self.x + i
            yi = self.y + i if self.z else self.y
            out.append((xi * 10 + yi) if (transpose and self.z) else (xi + yi * 10))
        return out

    def bits(self, transpose: bool) -> BinaryValue:
        v = 0
        for c in self.coordinates(transpose):
            v |= 1 << c
        return BinaryValue(v)

    def witness(self, utility: int = DEFAULT):
        """[H, V] placement commitments, optionally tampered (src/utils/ship.rs:189-311)."""
        p = self.bits(True).value
        pl = [0, p] if self.z else [p, 0]
        tgt = 1 if self.z else 0
        coords = self.coordinates(True)
        first, last = coords[0], coords[-1]
        if utility == DUAL_PLACEMENT:
            pl[1 - tgt] |= 1 << first
            pl[tgt] &= ~(1 << first)
        elif utility == NONCONSECUTIVE:
            pl[tgt] = (pl[tgt] & ~(1 << last)) | (1 << (last + 1))
        elif utility == EXTRA_BIT:
            pl[tgt] |= 1
        elif utility == OVERSIZED:
            pl[tgt] |= 1 << (last + 1)
        elif utility == UNDERSIZED:
            pl[tgt] &= ~(1 << last)
        elif utility != DEFAULT:
            raise ValueError("unknown WitnessOption")
        return [BinaryValue(pl[0]), BinaryValue(pl[1])]


class Deck:
    """Five optional ships in the order carrier..destroyer (src/utils/deck.rs:53-71)."""

    def __init__(self, ships=(None,) * 5):
        self.ships = [None if s is None else Ship(i, s[0], s[1], bool(s[2])) for i, s in enumerate(ships)]

    @classmethod
    def from_(cls, ships):
        return cls(ships)


class Board:
    def __init__(self, deck: Deck):
        self.ships = deck

    @classmethod
    def from_(cls, deck: Deck):
        return cls(deck)

    def state(self, utilities=(DEFAULT,) * 5) -> BinaryValue:
        """src/utils/board.rs:77-98: OR of every H placement and of every V placement re-indexed
        j -> (j % 10) * 10 + j // 10."""
        st = 0
        for i, ship in enumerate(self.ships.ships):
            if ship is None:
                continue
            h, v = ship.witness(utilities[i])
            st |= h.value & ((1 << BOARD_SIZE) - 1)
            for j in range(BOARD_SIZE):
                if v.bit(j):
                    st |= 1 << (j % 10 * 10 + j // 10)
        return BinaryValue(st)

    def witness(self, utilities=(DEFAULT,) * 5):
        """[H5, V5, H4, V4, H3a, V3a, H3b, V3b, H2, V2] (src/utils/board.rs:107-120)."""
        out = []
        for i, ship in enumerate(self.ships.ships):
            out += [BinaryValue(0), BinaryValue(0)] if ship is None else ship.witness(utilities[i])
        return out


def serialize(xs, ys) -> BinaryValue:
    """src/utils/shot.rs:12-19: bit (10*y + x) per shot."""
    v = 0
    for x, y in zip(xs, ys):
        v |= 1 << (y * 10 + x)
    return BinaryValue(v)


def compute_shot_trace(board: BinaryValue, shot: BinaryValue):
    """[shot_trace, hit_trace] running sums over the 100 cells (src/chips/shot.rs:28-51)."""
    shot_trace, hit_trace, s, h = [], [], 0, 0
    for i in range(BOARD_SIZE):
        s += shot.bit(i)
        h += board.bit(i) & shot.bit(i)
        shot_trace.append(s)
        hit_trace.append(h)
    return [shot_trace, hit_trace]


def compute_placement_trace(ship: BinaryValue, s: int):
    """[bit_sum, full_window_sum] (src/chips/placement.rs:380-419): running bit count, and the running
    count of fully-set S-wide windows that do not wrap a board row (i % 10 + S > 10 repeats the previous)."""
    bits = ship.bitfield(BOARD_SIZE)
    bit_sum, acc = [], 0
    for b in bits:
        acc += b
        bit_sum.append(acc)
    inc = lambda off: 1 if sum(bits[off:off + s]) == s else 0
    win = [inc(0)]
    for i in range(1, BOARD_SIZE):
        win.append(win[-1] if i % 10 + s > 10 else win[-1] + inc(i))
    return [bit_sum, win]


# ---- Pedersen commitment (src/utils/pedersen.rs:17-28) ---------------------------------------------
# V, R = hash_to_curve("battlezips:hash2curve")(b"v" / b"r") on Pallas.  The reference recomputes them on
# every call; their values are the GENERATOR constants it checks in board_commit_{v,r}.rs:5-14, 2941-2948
# (tests/test_oracle_golden.py re-derives them with the oracle's hash_to_curve).
PEDERSEN_V = (0x1e2542d216c42158aa3fc3f9268467a3bd7d655cae1385d70aaf6299a6692ca4,
              0x32c5c94a039386f80a26f1bc8cffd318e705a08374ba472f37a9d7aa880f14b2)
PEDERSEN_R = (0x1c332c6fa1a9c3d7cfb7d2c81d8bc40c1693cb90e5c65c55df6e95bb730e5277,
              0x0f9890742e8ad0e575d6a964af190d95ccc5ff9ca952893f04962380847950b8)


class PedersenCommitter:
    """Batched native Pedersen commitments [m]V + [t]R on the GPU: one 2-point window table on Pallas,
    one bzh_msm call with batch = number of commitments."""

    def __init__(self, ctx):
        from . import CURVE_PALLAS, int_to_limbs
        self.ctx = ctx
        tbl = np.stack([np.concatenate([int_to_limbs(P[0]), int_to_limbs(P[1])]) for P in (PEDERSEN_V, PEDERSEN_R)])
        self.bases = ctx.upload_bases(CURVE_PALLAS, tbl).precompute()

    def commit_batch(self, messages, trapdoors):
        """messages: Fp values re-read as Fq scalars through their canonical repr (from_repr(..).unwrap():
        >= q is an error); trapdoors: Fq.  Returns affine (x, y) int pairs (None = identity)."""
        from . import CURVE_PALLAS, int_to_limbs, jacobian_to_affine, limbs_to_int
        for m in messages:
            if not 0 <= m < FQ_MODULUS:
                raise ValueError("message repr is not a canonical Fq element")
        sc = np.stack([np.stack([int_to_limbs(m), int_to_limbs(t % FQ_MODULUS)]) for m, t in zip(messages, trapdoors)])
        aff = jacobian_to_affine(CURVE_PALLAS, self.ctx.msm(self.bases, sc))
        out = []
        for a in aff:
            x, y = limbs_to_int(a[:4]), limbs_to_int(a[4:])
            out.append(None if x == 0 and y == 0 else (x, y))
        return out

    def close(self):
        self.bases.free()


def pedersen_commit(ctx, message: int, trapdoor: int):
    c = PedersenCommitter(ctx)
    try:
        return c.commit_batch([message], [trapdoor])[0]
    finally:
        c.close()
