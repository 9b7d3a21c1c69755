"""Shared test helper: field elements / scalars drawn UNIFORMLY below the modulus (rejection sampling on
bitlen(modulus)-bit draws), as (n, 4) uint64 little-endian limbs.  The top window digit of a 255-bit scalar
is then exercised by ordinary random inputs, not only by hand-placed r-1 cases."""
import numpy as np

MODULI = {
    "fp": 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
    "fq": 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
    "bn254_fr": 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
}
# scalar field of curve id (0 vesta, 1 pallas, 2 bn254 G1) and modulus of field id (0 Fp, 1 Fq, 2 BN254 Fr)
SCALAR_MODULUS = {0: MODULI["fp"], 1: MODULI["fq"], 2: MODULI["bn254_fr"]}
FIELD_MODULUS = {0: MODULI["fp"], 1: MODULI["fq"], 2: MODULI["bn254_fr"]}


def _less_than(a: np.ndarray, m: int) -> np.ndarray:
    ml = [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    lt = a[:, 0] < np.uint64(ml[0])
    for i in (1, 2, 3):
        lt = (a[:, i] < np.uint64(ml[i])) | ((a[:, i] == np.uint64(ml[i])) & lt)
    return lt


def uniform_below(rng: np.random.Generator, n: int, modulus: int) -> np.ndarray:
    bits = modulus.bit_length()
    top_mask = np.uint64((1 << (bits - 192)) - 1)
    out = np.zeros((n, 4), dtype=np.uint64)
    have = 0
    while have < n:
        m = max(64, int((n - have) * 2.2))
        a = np.frombuffer(rng.bytes(m * 32), dtype=np.uint64).reshape(m, 4).copy()
        a[:, 3] &= top_mask
        a = a[_less_than(a, modulus)]
        take = min(n - have, a.shape[0])
        out[have:have + take] = a[:take]
        have += take
    return out
