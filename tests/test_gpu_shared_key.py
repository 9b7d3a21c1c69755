"""One proving key, several worker streams.  bzh_pk is immutable after bzh_pk_create except for caches filled on first use
(programs, hoisted columns, vk commitments) -- the per-call workspace belongs to the ctx -- so host threads prove
concurrently on ONE key, each through its own ctx, and get the bytes a private key gives (the reference's ProvingKey is
likewise shared by reference across create_proof calls: benches/shot.rs:61-68 builds it once)."""
import threading

import pytest

from helpers import real_parity as R

pytestmark = pytest.mark.gpu


def test_threads_share_one_key_and_get_the_private_keys_proofs(gpu_ctx):
    import bzh2
    from bzh2 import circuits as Cm, native as N, params as Pm
    lay = Cm.CircuitLayout(Cm.SHOT, 11)
    prm = Pm.Params(gpu_ctx, 11)
    workers = 4
    ctxs = [bzh2.Context(0) for _ in range(workers)]
    try:
        jobs = []
        for wi in range(workers):
            circuits = R.shot_circuits(Cm, 300 + wi, 3)
            adv, insts = lay.synthesize(circuits)
            seeds = [R.rng_stream("shared-%d-%d" % (wi, b), 32) for b in range(3)]
            jobs.append((adv, insts, seeds))
        # reference run: a private key, one thread, the creating ctx
        pk0 = N.NativeProvingKey(gpu_ctx, lay.blob(), bzh2.CURVE_VESTA, params=prm)
        want = [pk0.prove_batch(adv, insts, None, seeds=seeds) for adv, insts, seeds in jobs]
        pk0.close()
        # shared key: the FIRST proofs of the key run concurrently (program compilation and hoisted columns race here), twice over
        pk = N.NativeProvingKey(gpu_ctx, lay.blob(), bzh2.CURVE_VESTA, params=prm)
        got, errors = [[None, None] for _ in range(workers)], []

        def work(wi):
            try:
                adv, insts, seeds = jobs[wi]
                for rep in range(2):
                    got[wi][rep] = pk.prove_batch(adv, insts, None, seeds=seeds, ctx=ctxs[wi])
                assert pk.verify_batch(insts, got[wi][1], ctx=ctxs[wi]) == [True] * 3
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
        ths = [threading.Thread(target=work, args=(wi,)) for wi in range(workers)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errors, errors
        for wi in range(workers):
            assert got[wi][0] == want[wi] and got[wi][1] == want[wi], wi
        # the interpreter, selected on the shared key, serves every ctx too
        pk.quotient_select(N.QUOTIENT_INTERPRETER)
        again = pk.prove_batch(jobs[2][0], jobs[2][1], None, seeds=jobs[2][2], ctx=ctxs[1])
        assert again == want[2]
        pk.close()
    finally:
        for c in ctxs:
            c.close()
        prm.close()
        lay.close()
