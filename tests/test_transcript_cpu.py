"""Host transcript (bzh_transcript_*, hand-written Blake2b) against the oracle built on
hashlib.blake2b: same challenges and same proof bytes for a mixed absorb/squeeze script.
Reference call sites: Blake2bWrite::init at benches/shot.rs:66-67, src/circuits/board.rs:911-912."""
import hashlib
import random

import pytest

import pasta as O


@pytest.fixture(scope="module")
def bzh2_lib():
    import __graft_entry__ as g
    import bzh2
    import os
    if not os.path.exists(bzh2.lib_path()):
        g.build()
    return bzh2


def test_blake2b_long_stream_matches_hashlib(bzh2_lib):
    """> 128-byte streams cross block boundaries; challenge = digest of a clone after absorbing 0x00."""
    rng = random.Random(1)
    t = bzh2_lib.Transcript(bzh2_lib.FIELD_FP)
    ref = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
    for i in range(40):
        s = rng.randrange(O.P)
        t.common_scalar(s)
        ref.update(b"\x02" + O.to_repr(s))
        if i % 7 == 3:
            ref.update(b"\x00")
            want = int.from_bytes(ref.copy().digest(), "little") % O.P
            assert t.squeeze_challenge() == want
    t.close()


@pytest.mark.parametrize("fid,cid", [(0, 0), (1, 1)])
def test_transcript_script_matches_oracle(bzh2_lib, fid, cid):
    rng = random.Random(5 + fid)
    F, cv = O.FIELD_BY_ID[fid], O.CURVE_BY_ID[cid]
    a = bzh2_lib.Transcript(fid)
    b = O.Blake2bTranscript(F)
    assert a.squeeze_challenge() == b.squeeze_challenge()          # empty transcript
    for step in range(30):
        k = rng.randrange(5)
        if k == 0:
            pt = cv.random_point(rng) if rng.random() < 0.9 else None
            a.write_point(cid, pt)
            b.write_point(cv, pt)
        elif k == 1:
            s = rng.randrange(F.p)
            a.write_scalar(s)
            b.write_scalar(s)
        elif k == 2:
            pt = cv.random_point(rng)
            a.common_point(pt)
            b.common_point(cv, pt)
        elif k == 3:
            s = rng.choice([0, 1, F.p - 1, rng.randrange(F.p)])
            a.common_scalar(s)
            b.common_scalar(s)
        else:
            assert a.squeeze_challenge() == b.squeeze_challenge()
    assert a.squeeze_challenge() == b.squeeze_challenge()
    assert a.proof() == bytes(b.proof)
    a.close()
