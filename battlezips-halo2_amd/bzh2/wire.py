"""Proof / IO record of the reference's wasm frontend -- binding of bzh_record_* (include/bzh2.h,
csrc/record.hip).

`src/wasm/circuit_wasm.rs:27-31` defines what crosses to JavaScript:
    struct BattleZipsWASM { commitment: Vec<[u8; 32]>, proof: Vec<u8> }
with `commitment` = the public inputs as `BinaryValue::from_fp(fp).to_repr()` (`:75-83`) and `proof` =
`Blake2bWrite::finalize()`; `verify_board` / `verify_shot` (`:86-116`) read the same shape back
(`BinaryValue::from_repr(bin).to_fp()` -- a non-canonical element is an error).  The library keeps serde's JSON shape and
a fixed-stride binary form, which is what the multi-GPU gather carries (bzh2/shard.py, bench.py)."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

from . import BzhError, E_ARG, E_RANGE, int_to_limbs, limbs_to_int, load

_VP = ctypes.c_void_p
KIND_BOARD, KIND_SHOT = 0, 1
HEADER_BYTES = 144


def _bind():
    L = load()
    if getattr(L, "_bzh_record_bound", False):
        return L
    u8p, szp, u32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint32)
    L.bzh_record_stride.argtypes = [ctypes.c_size_t]
    L.bzh_record_stride.restype = ctypes.c_size_t
    L.bzh_record_encode.argtypes = [_VP, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, _VP, ctypes.c_size_t]
    L.bzh_record_decode.argtypes = [_VP, ctypes.c_size_t, _VP, szp, ctypes.POINTER(_VP), szp, u32p, u32p]
    L.bzh_record_to_json.argtypes = [_VP, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, szp]
    L.bzh_record_from_json.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, _VP, ctypes.c_size_t]
    L._bzh_record_bound = True
    return L


def record_stride(proof_stride: int) -> int:
    return _bind().bzh_record_stride(proof_stride)


def _raise(rc, where):
    if rc in (E_RANGE, E_ARG):
        raise ValueError("%s: %s" % (where, "not a canonical / well-sized BattleZipsWASM record" if rc == E_RANGE else "malformed record"))
    if rc:
        raise BzhError(rc, where)


@dataclass
class BattleZipsRecord:
    commitment: list          # public inputs, canonical Fp integers
    proof: bytes
    kind: int = KIND_BOARD    # caller's tags of the fixed-stride form (not part of the JSON shape)
    index: int = 0

    def to_fixed(self, proof_stride: int) -> bytes:
        """fixed-stride binary record (bzh_record_encode); ValueError if the proof exceeds the stride or an input is not canonical"""
        stride = record_stride(proof_stride)
        out = np.zeros(stride, dtype=np.uint8)
        if any(not (0 <= int(v) < (1 << 256)) for v in self.commitment):
            raise ValueError("public inputs are 256-bit values")
        ins = np.stack([int_to_limbs(int(v)) for v in self.commitment]) if self.commitment else np.zeros((1, 4), dtype=np.uint64)
        rc = _bind().bzh_record_encode(_VP(ins.ctypes.data), len(self.commitment), self.proof, len(self.proof), self.kind, self.index,
                                       _VP(out.ctypes.data), stride)
        _raise(rc, "bzh_record_encode")
        return out.tobytes()

    @classmethod
    def from_fixed(cls, record: bytes) -> "BattleZipsRecord":
        rec = np.frombuffer(bytes(record), dtype=np.uint8).copy()
        ins = np.zeros((4, 4), dtype=np.uint64)
        n, plen, pp = ctypes.c_size_t(), ctypes.c_size_t(), _VP()
        kind, index = ctypes.c_uint32(), ctypes.c_uint32()
        rc = _bind().bzh_record_decode(_VP(rec.ctypes.data), rec.size, _VP(ins.ctypes.data), ctypes.byref(n), ctypes.byref(pp), ctypes.byref(plen),
                                       ctypes.byref(kind), ctypes.byref(index))
        _raise(rc, "bzh_record_decode")
        off = pp.value - rec.ctypes.data
        return cls([limbs_to_int(ins[i]) for i in range(n.value)], rec[off:off + plen.value].tobytes(), kind.value, index.value)

    @staticmethod
    def proof_from_fixed(record: bytes) -> bytes:
        return BattleZipsRecord.from_fixed(record).proof

    def to_json(self) -> str:
        rec = np.frombuffer(self.to_fixed(len(self.proof)), dtype=np.uint8).copy()
        n = ctypes.c_size_t()
        L = _bind()
        _raise(L.bzh_record_to_json(_VP(rec.ctypes.data), rec.size, None, 0, ctypes.byref(n)), "bzh_record_to_json")
        buf = ctypes.create_string_buffer(n.value + 1)
        _raise(L.bzh_record_to_json(_VP(rec.ctypes.data), rec.size, buf, n.value + 1, ctypes.byref(n)), "bzh_record_to_json")
        return buf.value.decode()

    @classmethod
    def from_json(cls, text: str, kind: int = KIND_BOARD, index: int = 0) -> "BattleZipsRecord":
        raw = text.encode()
        rec = np.zeros(record_stride(len(raw)), dtype=np.uint8)     # a proof byte takes at least two characters of JSON
        _raise(_bind().bzh_record_from_json(raw, len(raw), kind, index, _VP(rec.ctypes.data), rec.size), "bzh_record_from_json")
        return cls.from_fixed(rec.tobytes())
