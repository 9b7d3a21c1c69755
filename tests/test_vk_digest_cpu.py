"""The verifying-key digest as an input of the boundary (include/bzh2.h "THE VERIFYING-KEY DIGEST"): upstream's create_proof
and verify_proof absorb pk.get_vk().hash_into(transcript) first -- the keys of benches/shot.rs:59-61, benches/board.rs:52-54,
src/circuits/shot.rs:915-918, src/circuits/board.rs:907-910.  Host-only checks: the digest recipe against hashlib, the setter
/ getter on a circuit, the blob offset the header publishes, the refusal of a non-canonical value."""
import hashlib
import re
import os

import pytest

import blob as B
import pasta as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(text: bytes) -> int:
    h = hashlib.blake2b(len(text).to_bytes(8, "little") + text, digest_size=64, person=b"Halo2-Verify-Key").digest()
    return int.from_bytes(h, "little") % O.FP.p          # Fp::from_bytes_wide


@pytest.mark.parametrize("text", [b"", b"x", b"PinnedVerificationKey { base_modulus: \"0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001\" }",
                                  bytes(range(256)) * 9])
def test_vk_digest_is_blake2b_verify_key_of_the_length_prefixed_text(text):
    from bzh2 import circuits as Cm
    assert Cm.vk_digest(text) == _digest(text)


def test_a_circuit_carries_the_placeholder_until_the_digest_is_set():
    from bzh2 import circuits as Cm
    hdr = open(os.path.join(ROOT, "include", "bzh2.h")).read()
    off = int(re.search(r"#define BZH_CIRCUIT_BLOB_VK_REPR_OFFSET (\d+)", hdr).group(1))
    placeholder = int(re.search(r"#define BZH_VK_REPR_PLACEHOLDER (0x[0-9a-fA-F]+)", hdr).group(1), 16)
    for kind, k in ((Cm.SHOT, 11), (Cm.BOARD, 12)):
        lay = Cm.CircuitLayout(kind, k)
        try:
            assert lay.vk_repr() == (placeholder, True)
            before = lay.blob()
            assert B.decode(before).vk_repr == placeholder
            d = _digest(b"a verifying key's Debug text, kind %d" % kind)
            lay.set_vk_repr(d)
            assert lay.vk_repr() == (d, False)
            after = lay.blob()
            assert after[off:off + 32] == d.to_bytes(32, "little")
            assert before[:off] == after[:off] and before[off + 32:] == after[off + 32:], "only the digest field may change"
            assert B.decode(after).vk_repr == d
            from bzh2 import BzhError
            for bad in (O.FP.p, (1 << 256) - 1):
                with pytest.raises(BzhError) as e:
                    lay.set_vk_repr(bad)
                assert e.value.status == -4
            assert lay.vk_repr() == (d, False)
        finally:
            lay.close()
