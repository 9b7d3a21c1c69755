"""Params::new on the host (csrc/params.hip): the product's hash_to_curve against the reference's two known answers
(`generator` tests, src/utils/constants/fixed_bases/board_commit_{v,r}.rs:2941-2948, data in tests/golden/fixed_bases.json)
and against the oracle's independent hash_to_curve (oracle/pasta.py) on Vesta for the SRS messages."""
import json
import os

import pasta as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "fixed_bases.json")))


def test_hash_to_curve_reproduces_the_reference_generators():
    import bzh2
    from bzh2 import params as Pm
    for name in ("v", "r"):
        b = GOLD["bases"][name]
        h = b["hash_to_curve"]
        got = Pm.hash_to_curve(bzh2.CURVE_PALLAS, h["domain"], h["message"].encode())
        assert got == (int(b["generator"][0], 16), int(b["generator"][1], 16))


def test_srs_generators_match_the_oracle():
    import bzh2
    from bzh2 import params as Pm
    k = 5
    g, w, u = Pm.generators(k)
    cv = O.VESTA
    for i in (0, 1, 2, 17, 31):
        want = O.hash_to_curve("vesta", "Halo2-Parameters", bytes([0]) + i.to_bytes(4, "little"))
        assert (bzh2.limbs_to_int(g[i, :4]), bzh2.limbs_to_int(g[i, 4:])) == want
        assert cv.is_on_curve(want)
    assert w == O.hash_to_curve("vesta", "Halo2-Parameters", bytes([1]))
    assert u == O.hash_to_curve("vesta", "Halo2-Parameters", bytes([2]))
    # messages of other lengths / domains
    for dom, msg in (("z.cash:test", b""), ("z.cash:test", b"Trans rights now!"), ("battlezips:hash2curve", b"v" * 200)):
        for cid, name in ((bzh2.CURVE_PALLAS, "pallas"), (bzh2.CURVE_VESTA, "vesta")):
            assert Pm.hash_to_curve(cid, dom, msg) == O.hash_to_curve(name, dom, msg)
