"""Host mirror of the reference's circuit layer over the C ABI (include/bzh2.h, "circuits"; csrc/circuits.hip).

    ShotCircuit.new(board, trapdoor, shot, hit)            src/circuits/shot.rs:65-78
    BoardCircuit.new(ship_commitments, board, trapdoor)    src/circuits/board.rs:63-73
    CircuitLayout(kind, k)                                 Circuit::configure + the keygen synthesize
        .blob()        constraint system + fixed columns for bzh_pk_create (NativeProvingKey.from_blob)
        .describe()    gates / regions / queries (what dev::MockProver reports against)
        .synthesize()  Circuit::synthesize for a batch of circuits -> advice tensor + public inputs

Values are Python ints; BinaryValues are bzh2.game.BinaryValue.  Everything here is a thin binding: the chips, the
floor planner and the witness generation are the C++ of csrc/circuit/*.hpp.
"""
from __future__ import annotations

import ctypes
import json

import numpy as np

from . import FORM_CANONICAL, FORM_MONTGOMERY, MEM_DEVICE, MEM_HOST, BzhError, int_to_limbs, limbs_to_int, load
from .game import BinaryValue

SHOT, BOARD, NUM2BITS_TEST, BITS2NUM_TEST = 0, 1, 2, 3
_VP = ctypes.c_void_p
_U64P = ctypes.POINTER(ctypes.c_uint64)


def _bind():
    L = load()
    if getattr(L, "_bzh_circuits_bound", False):
        return L
    L.bzh_circuit_create.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(_VP)]
    L.bzh_circuit_free.argtypes = [_VP]
    L.bzh_circuit_last_error.restype = ctypes.c_char_p
    L.bzh_circuit_blob.argtypes = [_VP, _VP, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.bzh_circuit_describe.argtypes = [_VP, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.bzh_circuit_info.argtypes = [_VP] + [ctypes.POINTER(ctypes.c_uint32)] * 6
    L.bzh_synthesize_shot.argtypes = [_VP, _VP, ctypes.c_size_t, _VP, _VP, _VP, _VP, _VP, ctypes.c_int, ctypes.c_int, _VP, ctypes.c_uint]
    L.bzh_synthesize_board.argtypes = [_VP, _VP, ctypes.c_size_t, _VP, _VP, _VP, _VP, ctypes.c_int, ctypes.c_int, _VP, ctypes.c_uint]
    L.bzh_synthesize_bitify_test.argtypes = [_VP, _VP, _VP, _VP]
    L.bzh_board_witness.argtypes = [_VP, _VP, _VP, _VP]
    L.bzh_shot_serialize.argtypes = [_VP, _VP, ctypes.c_size_t, _VP]
    L.bzh_pedersen_commit_host.argtypes = [_VP, _VP, _VP]
    L.bzh_fixed_base_tables.argtypes = [ctypes.c_int, _VP, _VP, _VP]
    L.bzh_pedersen_commit_batch.argtypes = [_VP, _VP, _VP, ctypes.c_size_t, _VP]
    L.bzh_circuit_set_vk_repr.argtypes = [_VP, ctypes.c_char_p]
    L.bzh_circuit_vk_repr.argtypes = [_VP, _VP, ctypes.POINTER(ctypes.c_int)]
    L.bzh_vk_digest.argtypes = [ctypes.c_char_p, ctypes.c_size_t, _VP]
    L._bzh_circuits_bound = True
    return L


def _check(rc, where):
    if rc != 0:
        raise BzhError(rc, where, (_bind().bzh_circuit_last_error() or b"").decode())


def _limbs(values) -> np.ndarray:
    return np.stack([int_to_limbs(int(v)) for v in values]) if len(values) else np.zeros((0, 4), dtype=np.uint64)


class ShotCircuit:
    def __init__(self, board: BinaryValue, board_commitment_trapdoor: int, shot: BinaryValue, hit: BinaryValue):
        self.board, self.board_commitment_trapdoor, self.shot, self.hit = board, board_commitment_trapdoor, shot, hit

    new = classmethod(lambda cls, *a: cls(*a))


class BoardCircuit:
    def __init__(self, ship_commitments, board: BinaryValue, board_commitment_trapdoor: int):
        assert len(ship_commitments) == 10
        self.ship_commitments, self.board, self.board_commitment_trapdoor = list(ship_commitments), board, board_commitment_trapdoor

    new = classmethod(lambda cls, *a: cls(*a))


class CircuitLayout:
    """configure + keygen synthesis of one circuit kind at 2^k rows."""

    def __init__(self, kind: int, k: int, bits: int = 0):
        L = _bind()
        h = _VP()
        _check(L.bzh_circuit_create(kind, k, bits, ctypes.byref(h)), "bzh_circuit_create")
        self.handle, self.kind, self.k = h, kind, k
        vals = [ctypes.c_uint32() for _ in range(6)]
        _check(L.bzh_circuit_info(h, *[ctypes.byref(v) for v in vals]), "bzh_circuit_info")
        self.num_advice, self.num_instance_rows, self.n, self.rows_used, self.num_gates, self.num_regions = [v.value for v in vals]

    def close(self):
        if self.handle is not None:
            _bind().bzh_circuit_free(self.handle)
            self.handle = None

    def blob(self) -> bytes:
        L = _bind()
        n = ctypes.c_size_t()
        _check(L.bzh_circuit_blob(self.handle, None, 0, ctypes.byref(n)), "bzh_circuit_blob")
        buf = (ctypes.c_uint8 * n.value)()
        _check(L.bzh_circuit_blob(self.handle, buf, n.value, ctypes.byref(n)), "bzh_circuit_blob")
        return bytes(buf)

    def set_vk_repr(self, vk_repr: int | bytes):
        """install the verifying-key digest upstream's create_proof / verify_proof absorb first (vk.hash_into): an input of the
        boundary (include/bzh2.h "THE VERIFYING-KEY DIGEST").  Call before blob() / NativeProvingKey."""
        raw = vk_repr if isinstance(vk_repr, (bytes, bytearray)) else int(vk_repr).to_bytes(32, "little")
        assert len(raw) == 32
        _check(_bind().bzh_circuit_set_vk_repr(self.handle, bytes(raw)), "bzh_circuit_set_vk_repr")

    def vk_repr(self):
        """(digest as an int, is_placeholder)"""
        out, ph = (ctypes.c_uint8 * 32)(), ctypes.c_int()
        _check(_bind().bzh_circuit_vk_repr(self.handle, out, ctypes.byref(ph)), "bzh_circuit_vk_repr")
        return int.from_bytes(bytes(out), "little"), bool(ph.value)

    def quotient_degree_histogram(self, curve: int = 0):
        """{degree: (terms, tree multiplications)} of the quotient numerator of this circuit (bzh_quotient_degree_histogram)"""
        L = _bind()
        L.bzh_quotient_degree_histogram.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, _VP, _VP]
        blob = self.blob()
        polys, muls = np.zeros(16, dtype=np.uint32), np.zeros(16, dtype=np.uint32)
        _check(L.bzh_quotient_degree_histogram(curve, blob, len(blob), _VP(polys.ctypes.data), _VP(muls.ctypes.data)), "bzh_quotient_degree_histogram")
        return {d: (int(polys[d]), int(muls[d])) for d in range(16) if polys[d]}

    def describe(self) -> dict:
        L = _bind()
        n = ctypes.c_size_t()
        _check(L.bzh_circuit_describe(self.handle, None, 0, ctypes.byref(n)), "bzh_circuit_describe")
        buf = ctypes.create_string_buffer(n.value)
        _check(L.bzh_circuit_describe(self.handle, buf, n.value, ctypes.byref(n)), "bzh_circuit_describe")
        return json.loads(buf.value.decode())

    def synthesize(self, circuits, ctx=None, device_ptr: int | None = None, form: int = FORM_CANONICAL, threads: int = 0):
        """Circuit::synthesize for every circuit of the list.  Returns (advice, instances): advice is a
        (batch, num_advice, n, 4) uint64 host array in `form`, or None when the tensor was written to `device_ptr`
        (Montgomery, on ctx's stream); instances is a list of per-proof public-input columns [[...]] (ints)."""
        L = _bind()
        B = len(circuits)
        inst = np.zeros((B, max(self.num_instance_rows, 1), 4), dtype=np.uint64)
        if device_ptr is not None:
            adv, adv_p, form, mem = None, _VP(device_ptr), FORM_MONTGOMERY, MEM_DEVICE
        else:
            adv = np.zeros((B, self.num_advice, self.n, 4), dtype=np.uint64)
            adv_p, mem = _VP(adv.ctypes.data), MEM_HOST
        ctxh = ctx.handle if ctx is not None else None
        if self.kind == SHOT:
            boards = _limbs([c.board.value for c in circuits])
            traps = _limbs([c.board_commitment_trapdoor for c in circuits])
            shots = _limbs([c.shot.value for c in circuits])
            hits = _limbs([c.hit.value for c in circuits])
            rc = L.bzh_synthesize_shot(ctxh, self.handle, B, _VP(boards.ctypes.data), _VP(traps.ctypes.data), _VP(shots.ctypes.data),
                                       _VP(hits.ctypes.data), adv_p, form, mem, _VP(inst.ctypes.data), threads)
        elif self.kind == BOARD:
            ships = _limbs([s.value for c in circuits for s in c.ship_commitments])
            boards = _limbs([c.board.value for c in circuits])
            traps = _limbs([c.board_commitment_trapdoor for c in circuits])
            rc = L.bzh_synthesize_board(ctxh, self.handle, B, _VP(ships.ctypes.data), _VP(boards.ctypes.data), _VP(traps.ctypes.data),
                                        adv_p, form, mem, _VP(inst.ctypes.data), threads)
        else:
            raise ValueError("use synthesize_bitify_test")
        _check(rc, "bzh_synthesize")
        instances = [[[limbs_to_int(inst[b, r]) for r in range(self.num_instance_rows)]] for b in range(B)]
        return adv, instances

    def synthesize_bitify_test(self, value: int, binary: BinaryValue) -> np.ndarray:
        adv = np.zeros((1, self.num_advice, self.n, 4), dtype=np.uint64)
        v, b = int_to_limbs(value), int_to_limbs(binary.value)
        _check(_bind().bzh_synthesize_bitify_test(self.handle, _VP(v.ctypes.data), _VP(b.ctypes.data), _VP(adv.ctypes.data)), "bzh_synthesize_bitify_test")
        return adv


def vk_digest(pinned_debug: str | bytes) -> int:
    """VerifyingKey::hash_into's scalar from the text of format!("{:?}", vk.pinned()) (bzh_vk_digest)"""
    raw = pinned_debug.encode() if isinstance(pinned_debug, str) else bytes(pinned_debug)
    out = (ctypes.c_uint8 * 32)()
    _check(_bind().bzh_vk_digest(raw, len(raw), out), "bzh_vk_digest")
    return int.from_bytes(bytes(out), "little")


def board_witness(ships, options=None):
    """Board::from(&Deck::from(ships)).{witness, state}(options) through the C ABI (src/utils/board.rs:77-120).
    ships: 5 x (x, y, z) or None; options: 5 WitnessOption ints.  Returns ([BinaryValue] * 10, BinaryValue)."""
    s = np.array([[-1, 0, 0] if sh is None else [sh[0], sh[1], 1 if sh[2] else 0] for sh in ships], dtype=np.int8)
    o = np.array(list(options) if options is not None else [0] * 5, dtype=np.int32)
    out, st = np.zeros((10, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    _check(_bind().bzh_board_witness(_VP(s.ctypes.data), _VP(o.ctypes.data), _VP(out.ctypes.data), _VP(st.ctypes.data)), "bzh_board_witness")
    return [BinaryValue(limbs_to_int(out[i])) for i in range(10)], BinaryValue(limbs_to_int(st))


def shot_serialize(xs, ys) -> BinaryValue:
    x, y = np.array(list(xs), dtype=np.uint8), np.array(list(ys), dtype=np.uint8)
    out = np.zeros(4, dtype=np.uint64)
    _check(_bind().bzh_shot_serialize(_VP(x.ctypes.data), _VP(y.ctypes.data), len(x), _VP(out.ctypes.data)), "bzh_shot_serialize")
    return BinaryValue(limbs_to_int(out))


def pedersen_commit_host(message: int, trapdoor: int):
    m, t, out = int_to_limbs(message), int_to_limbs(trapdoor), np.zeros(8, dtype=np.uint64)
    _check(_bind().bzh_pedersen_commit_host(_VP(m.ctypes.data), _VP(t.ctypes.data), _VP(out.ctypes.data)), "bzh_pedersen_commit_host")
    return limbs_to_int(out[:4]), limbs_to_int(out[4:])


def pedersen_commit_batch(ctx, messages, trapdoors):
    """pedersen_commit (src/utils/pedersen.rs:17-28) for every (message, trapdoor) pair in one device launch
    (bzh_pedersen_commit_batch).  Returns affine (x, y) int pairs, None for the identity."""
    n = len(messages)
    assert n == len(trapdoors) and n > 0
    m, t, out = _limbs(messages), _limbs(trapdoors), np.zeros((n, 8), dtype=np.uint64)
    rc = _bind().bzh_pedersen_commit_batch(ctx.handle, _VP(m.ctypes.data), _VP(t.ctypes.data), n, _VP(out.ctypes.data))
    if rc:
        raise BzhError(rc, "bzh_pedersen_commit_batch", ctx.last_error() if hasattr(ctx, "last_error") else "")
    res = []
    for a in out:
        x, y = limbs_to_int(a[:4]), limbs_to_int(a[4:])
        res.append(None if x == 0 and y == 0 else (x, y))
    return res


def fixed_base_tables(base: int):
    z, u, lg = np.zeros(85, dtype=np.uint64), np.zeros((85, 8, 4), dtype=np.uint64), np.zeros((85, 8, 4), dtype=np.uint64)
    _check(_bind().bzh_fixed_base_tables(base, _VP(z.ctypes.data), _VP(u.ctypes.data), _VP(lg.ctypes.data)), "bzh_fixed_base_tables")
    return ([int(v) for v in z], [[limbs_to_int(u[w, k]) for k in range(8)] for w in range(85)],
            [[limbs_to_int(lg[w, k]) for k in range(8)] for w in range(85)])
