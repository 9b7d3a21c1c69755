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


def test_module_swap_and_free_are_refused_while_another_ctx_proves_on_the_key(gpu_ctx):
    """bzh_pk_set_quotient_module unloads a code object and bzh_pk_free releases every ctx's workspace: both refuse (BZH_E_ARG,
    key untouched) while a bzh_prove_batch on the key is still running on another ctx -- a launch of the unloaded code or a
    kernel on a freed arena would fault the GPU.  Thread A proves a batch; the main thread keeps asking for both through its
    own ctx until A is done: every answer during the call is a refusal, A's proofs are the expected bytes, and both requests
    succeed afterwards."""
    import time
    import bzh2
    from bzh2 import BzhError, circuits as Cm, native as N, params as Pm
    lay = Cm.CircuitLayout(Cm.BOARD, 12)
    prm = Pm.Params(gpu_ctx, 12)
    other = bzh2.Context(0)
    try:
        pk = N.NativeProvingKey(gpu_ctx, lay.blob(), bzh2.CURVE_VESTA, params=prm)
        circuits = R.board_circuits(Cm, 77, 16)
        adv, insts = lay.synthesize(circuits)
        seeds = [R.rng_stream("busy-%d" % b, 32) for b in range(16)]
        want = pk.prove_batch(adv, insts, None, seeds=seeds, ctx=other)      # warm: caches filled, workspace grown
        state = {"running": False, "done": False, "proofs": None, "error": None}

        def prove():
            try:
                state["running"] = True
                state["proofs"] = pk.prove_batch(adv, insts, None, seeds=seeds, ctx=other)
            except BaseException as e:  # noqa: BLE001
                state["error"] = e
            finally:
                state["done"] = True
        th = threading.Thread(target=prove)
        th.start()
        while not state["running"]:
            time.sleep(0)
        time.sleep(0.002)      # let the call take the key (it runs for tens of milliseconds)
        refused = {"module": 0, "free": 0}
        while not state["done"]:
            for what, call in (("module", lambda: pk.set_quotient_module(None)), ("free", pk.close)):
                try:
                    call()
                    th.join(timeout=1.0)     # allowed only once the call has left the library (its thread ends right after)
                    if th.is_alive():
                        state["error"] = AssertionError("%s went through while a proof was running" % what)
                except BzhError as e:
                    assert e.status == -1
                    refused[what] += 1
            if pk.handle is None:
                break
        th.join()
        assert state["error"] is None, state["error"]
        assert refused["module"] >= 1 and refused["free"] >= 1, refused
        assert state["proofs"] == want
        if pk.handle is not None:
            pk.set_quotient_module(None)      # idle key: allowed (returns to the builtin kernel)
            assert pk.prove_batch(adv[:2], insts[:2], None, seeds=seeds[:2]) == want[:2]
            pk.close()
        assert pk.handle is None
    finally:
        other.close()
        prm.close()
        lay.close()


def test_params_close_in_a_cleanup_path_closes_borrowing_keys_first(gpu_ctx):
    """Params.close() is what `finally:` blocks call: with a key still open on the tables it closes the key first (warning)
    instead of raising over the original failure; strict=True keeps the hard error."""
    import warnings
    import bzh2
    from bzh2 import circuits as Cm, native as N, params as Pm
    lay = Cm.CircuitLayout(Cm.SHOT, 11)
    prm = Pm.Params(gpu_ctx, 11)
    try:
        pk = N.NativeProvingKey(gpu_ctx, lay.blob(), bzh2.CURVE_VESTA, params=prm)
        with pytest.raises(RuntimeError):
            prm.close(strict=True)
        assert prm.handle is not None and pk.handle is not None
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            prm.close()
        assert pk.handle is None and prm.handle is None
        assert any(issubclass(x.category, ResourceWarning) for x in w)
    finally:
        prm.close()
        lay.close()
