#!/usr/bin/env python3
"""Prove the case of examples/export_case.py through the ctypes binding and print the same FNV-1a checksum the C++ client
prints: both must agree (same library, same inputs, same bytes).  Usage: check_case.py case.bin"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "battlezips-halo2_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402

import bzh2  # noqa: E402
from bzh2 import native as N, synth  # noqa: E402
from export_case import fnv1a  # noqa: E402


def main():
    raw = open(sys.argv[1], "rb").read()
    assert raw[:4] == b"BZX1"
    k, na, ni, rows, batch = struct.unpack_from("<5I", raw, 4)
    (rng_stride,) = struct.unpack_from("<Q", raw, 24)
    lens = struct.unpack_from("<5Q", raw, 32)
    o = 72
    parts = []
    for ln in lens:
        parts.append(raw[o:o + ln])
        o += ln
    n = 1 << k
    srs = np.frombuffer(parts[0], dtype=np.uint64).reshape(n + 2, 8)
    pt = lambda a: (bzh2.limbs_to_int(a[:4]), bzh2.limbs_to_int(a[4:]))
    circ, _, inst = synth.battlezips_shaped(k, 7)
    advice = np.frombuffer(parts[2], dtype=np.uint64).reshape(batch, na, n, 4)
    with bzh2.Context(0) as ctx:
        pk = N.NativeProvingKey(ctx, circ, bzh2.CURVE_VESTA, [pt(a) for a in srs[:n]], pt(srs[n + 1]), pt(srs[n]))
        rbs = [parts[4][b * rng_stride:(b + 1) * rng_stride] for b in range(batch)]
        proofs = pk.prove_batch(advice, [inst] * batch, rbs)
        ok = pk.verify_batch([inst] * batch, proofs)
        pk.close()
    print('{"fnv1a": "%s", "verified": %d, "proof_bytes": %d}' % (fnv1a(proofs), sum(ok), len(proofs[0])))


if __name__ == "__main__":
    main()
