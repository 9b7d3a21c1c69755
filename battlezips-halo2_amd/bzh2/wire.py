"""Proof / IO record of the reference's wasm frontend.

`src/wasm/circuit_wasm.rs:27-31` defines what crosses to JavaScript:
    struct BattleZipsWASM { commitment: Vec<[u8; 32]>, proof: Vec<u8> }
with `commitment` = the public inputs as `BinaryValue::from_fp(fp).to_repr()` (32 little-endian bytes each,
`:75-83`) and `proof` = `Blake2bWrite::finalize()`; `verify_board` / `verify_shot` (`:86-116`) read the same
shape back (`BinaryValue::from_repr(bin).to_fp()` -- a non-canonical element is an error).  serde serialises
both fields as arrays of numbers; this module keeps that JSON shape so records made here are accepted by the
wasm frontend's verifier entry points and vice versa.  The fixed-stride form `{u32 length, bytes}` is what the
multi-GPU gather carries (bzh2/shard.py).
"""
from __future__ import annotations

import json
from dataclasses import dataclass

from .game import BinaryValue


@dataclass
class BattleZipsRecord:
    commitment: list  # public inputs, canonical Fp integers
    proof: bytes

    def to_json(self) -> str:
        return json.dumps({"commitment": [list(BinaryValue(v).to_repr()) for v in self.commitment], "proof": list(self.proof)},
                          separators=(",", ":"))

    @classmethod
    def from_json(cls, text: str) -> "BattleZipsRecord":
        d = json.loads(text)
        if set(d) != {"commitment", "proof"}:
            raise ValueError("not a BattleZipsWASM record")
        commitment = []
        for c in d["commitment"]:
            if len(c) != 32 or any(not (isinstance(b, int) and 0 <= b <= 255) for b in c):
                raise ValueError("commitment entries are 32 bytes")
            commitment.append(BinaryValue.from_repr(bytes(c)).to_fp())   # rejects non-canonical field elements
        if any(not (isinstance(b, int) and 0 <= b <= 255) for b in d["proof"]):
            raise ValueError("proof is a byte array")
        return cls(commitment, bytes(d["proof"]))

    # ---- fixed-stride form for the RCCL gather -------------------------------------------------
    def to_fixed(self, proof_stride: int) -> bytes:
        if len(self.proof) > proof_stride:
            raise ValueError("proof longer than the record stride")
        return len(self.proof).to_bytes(4, "little") + self.proof + bytes(proof_stride - len(self.proof))

    @staticmethod
    def proof_from_fixed(record: bytes) -> bytes:
        n = int.from_bytes(record[:4], "little")
        if n > len(record) - 4:
            raise ValueError("corrupt record length")
        return bytes(record[4:4 + n])
