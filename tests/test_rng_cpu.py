"""bzh_rng_expand (the ChaCha20 stream bzh_prove_batch_seeded draws from) against a plain-Python ChaCha20 that is itself
checked on the RFC 8439 section 2.3.2 block vector.  Host code only: no device compute."""
import os
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def chacha20_block(key: bytes, counter: int, nonce64: int = 0) -> bytes:
    """ChaCha20 block, 64-bit counter in words 12-13 and 64-bit nonce in words 14-15 (the original layout; RFC 8439's
    32-bit counter + 96-bit nonce is the same state with the fields read differently)."""
    def rotl(v, c):
        return ((v << c) & 0xffffffff) | (v >> (32 - c))
    s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(struct.unpack("<8I", key)) + \
        [counter & 0xffffffff, (counter >> 32) & 0xffffffff, nonce64 & 0xffffffff, (nonce64 >> 32) & 0xffffffff]
    x = s[:]

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xffffffff; x[d] ^= x[a]; x[d] = rotl(x[d], 16)
        x[c] = (x[c] + x[d]) & 0xffffffff; x[b] ^= x[c]; x[b] = rotl(x[b], 12)
        x[a] = (x[a] + x[b]) & 0xffffffff; x[d] ^= x[a]; x[d] = rotl(x[d], 8)
        x[c] = (x[c] + x[d]) & 0xffffffff; x[b] ^= x[c]; x[b] = rotl(x[b], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(x[i] + s[i]) & 0xffffffff for i in range(16)])


def expand(seed: bytes, first: int, draws: int) -> bytes:
    return b"".join(chacha20_block(seed, first + i) for i in range(draws))


def test_python_chacha20_matches_rfc8439_block_vector():
    key = bytes(range(32))
    # RFC 8439 2.3.2: counter = 1, nonce = 00 00 00 09 00 00 00 4a 00 00 00 00 -> state words 12..15 = 1, 0x09000000, 0x4a000000, 0
    blk = chacha20_block(key, 1 | (0x09000000 << 32), 0x4a000000)
    assert blk.hex() == ("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                         "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


def test_rng_expand_matches_python_chacha20():
    import __graft_entry__ as g
    import bzh2
    from bzh2 import native as N
    if not os.path.exists(bzh2.lib_path()):
        g.build()
    for seed in (bytes(32), bytes(range(32)), bytes((7 * i + 3) & 0xff for i in range(32))):
        assert N.rng_expand(seed, 0, 4) == expand(seed, 0, 4)
        assert N.rng_expand(seed, (1 << 32) - 2, 4) == expand(seed, (1 << 32) - 2, 4)   # counter carries into word 13
    assert N.rng_expand(bytes(32), 9, 0) == b""
