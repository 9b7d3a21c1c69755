"""The reference's REAL circuits end to end on the GPU: ShotCircuit (k = 11, benches/shot.rs:22) and BoardCircuit
(k = 12, benches/board.rs:22) built by the C++ front end (bzh_circuit_*), witnesses synthesised per proof
(bzh_synthesize_*), proved by bzh_prove_batch and checked by bzh_verify_batch AND by the oracle's verify_proof on the
constraint system decoded from the same blob (oracle/blob.py) -- the analogue of the reference's `production` tests
(src/circuits/shot.rs:880-941, src/circuits/board.rs:879-933), plus BASELINE.json configs[2] (1024 Shot proofs).
Small circuits with `enable_equality` before their gates are compared byte for byte with the oracle PROVER."""
import os
import random

import numpy as np
import pytest

import blob as B
import coracle as C
import halo2_oracle as H
import pasta as O

pytestmark = pytest.mark.gpu

FP, FQ = O.FP.p, O.FQ.p
THREADS = max(1, min(32, os.cpu_count() or 1))
PATTERN_1 = [(3, 3, True), (5, 4, False), (0, 1, False), (0, 5, True), (6, 1, False)]      # src/circuits/board.rs:101-107
PATTERN_2 = [(3, 4, False), (9, 6, True), (0, 0, False), (0, 6, False), (6, 1, True)]     # src/circuits/board.rs:134-140


def _srs(n, seed):
    cv = O.VESTA
    g0 = cv.random_point(random.Random(seed))
    walk = C.point_walk(0, C.points_to_array([g0])[0], n + 2)
    pts = [C.array_to_point(walk[i]) for i in range(n + 2)]
    return pts[:n], pts[n], pts[n + 1]        # g, u, w


def _oracle_keys(circ, g, w, u, monkeypatch, verifier_only=True):
    cv, F = O.VESTA, O.FP
    fast = lambda self, scalars, points: C.array_to_point(C.msm(0, C.ints_to_array([int(s) % F.p for s in scalars]),
                                                                   C.points_to_array(points), THREADS))
    monkeypatch.setattr(type(cv), "msm_naive", fast)
    if verifier_only:
        monkeypatch.setattr(H.Domain, "lagrange_to_coeff",
                            lambda self, v: C.array_to_ints(C.ntt(0, C.ints_to_array(v), self.omega, inverse=True, threads=THREADS)))
    cs = H.ConstraintSystem(circ.k, circ.num_advice, circ.num_fixed, circ.num_instance, circ.gates, circ.perm_columns, circ.lookups,
                            degree=circ.min_degree, queries=circ.queries)
    return H.Keys(cs, H.Domain(cs, F), cv, g, w, u, circ.fixed, circ.copies, vk_repr=circ.vk_repr, verifier_only=verifier_only)


def _random_deck(rng):
    """a valid random fleet: rejection-sample non-overlapping in-bounds placements"""
    lens = (5, 4, 3, 3, 2)
    while True:
        used, deck = set(), []
        for L in lens:
            for _ in range(200):
                z = rng.random() < 0.5
                x, y = rng.randrange(10 - (0 if z else L - 1)), rng.randrange(10 - (L - 1 if z else 0))
                cells = {(x, y + i) if z else (x + i, y) for i in range(L)}
                if not cells & used:
                    used |= cells
                    deck.append((x, y, z))
                    break
            else:
                break
        if len(deck) == 5:
            return deck, used


def _shot_circuits(Cm, rng, count):
    from bzh2.game import BinaryValue
    out = []
    for _ in range(count):
        deck, used = _random_deck(rng)
        _, state = Cm.board_witness(deck, None)
        x, y = rng.randrange(10), rng.randrange(10)
        hit = 1 if (x, y) in used else 0
        out.append(Cm.ShotCircuit(state, rng.randrange(FQ), Cm.shot_serialize([x], [y]), BinaryValue.from_u8(hit)))
    return out


@pytest.fixture(scope="module")
def shot_setup(gpu_ctx):
    import bzh2
    from bzh2 import circuits as Cm, native as N
    lay = Cm.CircuitLayout(Cm.SHOT, 11)
    blob = lay.blob()
    g, u, w = _srs(1 << 11, 1101)
    pk = N.NativeProvingKey(gpu_ctx, blob, bzh2.CURVE_VESTA, g, w, u)
    yield lay, blob, pk, (g, w, u)
    pk.close()
    lay.close()


@pytest.fixture(scope="module")
def board_setup(gpu_ctx):
    import bzh2
    from bzh2 import circuits as Cm, native as N
    lay = Cm.CircuitLayout(Cm.BOARD, 12)
    blob = lay.blob()
    g, u, w = _srs(1 << 12, 1201)
    pk = N.NativeProvingKey(gpu_ctx, blob, bzh2.CURVE_VESTA, g, w, u)
    yield lay, blob, pk, (g, w, u)
    pk.close()
    lay.close()


def test_shot_circuit_production_roundtrip(gpu_ctx, oracle_c, shot_setup, monkeypatch):
    """create_proof -> verify_proof for distinct ShotCircuit witnesses; the native verifier and the oracle verifier agree
    on valid proofs, on a wrong public input and on a tampered proof."""
    from bzh2 import circuits as Cm
    lay, blob, pk, (g, w, u) = shot_setup
    rng = random.Random(11)
    circuits = _shot_circuits(Cm, rng, 6)
    adv, insts = lay.synthesize(circuits)
    rbs = [bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes)) for _ in circuits]
    proofs = pk.prove_batch(adv, insts, rbs)
    assert len(set(proofs)) == len(proofs)
    assert pk.verify_batch(insts, proofs) == [True] * len(proofs)
    keys = _oracle_keys(B.decode(blob), g, w, u, monkeypatch)
    for b in (0, 3):
        assert H.verify_proof(keys, insts[b], proofs[b], O.Blake2bTranscript(O.FP))
    wrong = [[list(insts[0][0])]]
    wrong[0][0][3] ^= 1                                                  # flip the public hit bit
    bad = proofs[1][:500] + bytes([proofs[1][500] ^ 2]) + proofs[1][501:]
    assert pk.verify_batch([wrong[0], insts[1], insts[1]], [proofs[0], bad, proofs[1]]) == [False, False, True]
    assert not H.verify_proof(keys, wrong[0], proofs[0], O.Blake2bTranscript(O.FP))
    assert not H.verify_proof(keys, insts[1], bad, O.Blake2bTranscript(O.FP))


def test_shot_witness_from_the_device_path_gives_the_same_proofs(gpu_ctx, shot_setup):
    """bzh_synthesize_shot with BZH_MEM_DEVICE (compact pinned staging + expansion kernel on the ctx's stream) feeds
    bzh_prove_batch the same tensor as the host path: identical proof bytes."""
    import torch
    from bzh2 import circuits as Cm
    lay, blob, pk, _ = shot_setup
    rng = random.Random(12)
    circuits = _shot_circuits(Cm, rng, 5)
    adv, insts = lay.synthesize(circuits)
    rbs = [bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes)) for _ in circuits]
    want = pk.prove_batch(adv, insts, rbs)
    dev = torch.empty((len(circuits), lay.num_advice, lay.n, 4), dtype=torch.int64, device="cuda")
    dev.fill_(-1)                                                        # stale contents must be overwritten / zeroed
    torch.cuda.synchronize()
    _, insts2 = lay.synthesize(circuits, ctx=gpu_ctx, device_ptr=dev.data_ptr())
    assert insts2 == insts
    assert pk.prove_batch(None, insts, rbs, device_ptr=dev.data_ptr()) == want


def test_seeded_randomness_gives_the_proofs_of_the_expanded_streams(gpu_ctx, shot_setup, board_setup):
    """bzh_prove_batch_seeded draws each proof's randomness on the device from ChaCha20(seed) (the reference draws from OsRng
    inside create_proof); the proofs are those of bzh_prove_batch fed with the same streams expanded by bzh_rng_expand
    (checked against a Python ChaCha20 in tests/test_rng_cpu.py) -- host single draws, device row draws and the opening's
    rows all address the stream by position."""
    from bzh2 import circuits as Cm, native as N
    rng = random.Random(14)
    lay, blob, pk, _ = shot_setup
    circuits = _shot_circuits(Cm, rng, 3)
    adv, insts = lay.synthesize(circuits)
    seeds = [bytes(rng.getrandbits(8) for _ in range(32)) for _ in circuits]
    assert pk.rng_bytes % 64 == 0
    streams = [N.rng_expand(sd, 0, pk.rng_bytes // 64) for sd in seeds]
    want = pk.prove_batch(adv, insts, streams)
    got = pk.prove_batch(adv, insts, None, seeds=seeds)
    assert got == want and len(set(got)) == 3
    assert pk.verify_batch(insts, got) == [True] * 3
    # BoardCircuit (lookup argument, more host-side draws), one proof
    layb, _, pkb, _ = board_setup
    deck, _ = _random_deck(rng)
    ships, state = Cm.board_witness(deck, None)
    cb = [Cm.BoardCircuit(ships, state, rng.randrange(FQ))]
    advb, instb = layb.synthesize(cb)
    sd = bytes(rng.getrandbits(8) for _ in range(32))
    assert pkb.prove_batch(advb, instb, None, seeds=[sd]) == pkb.prove_batch(advb, instb, [N.rng_expand(sd, 0, pkb.rng_bytes // 64)])


def test_compiled_quotient_flavours_give_the_same_proofs(gpu_ctx, shot_setup, board_setup):
    """The key's evaluator program runs three ways: the interpreter (k_expr_vm2), the kernel generated when libbzh2.so was
    built (found by program hash at bzh_pk_create: the default for the reference's circuits, no compiler at run time) and a
    code object compiled by the caller (bzh_pk_quotient_source -> hipcc -> bzh_pk_set_quotient_module, the path for foreign
    circuits).  Same proofs bit for bit; a module generated from another circuit's program is refused; NULL returns to the
    key's default.  (tests/test_gpu_real_circuit_parity.py compares the first two with the ORACLE prover.)"""
    import shutil
    import bzh2
    from bzh2 import circuits as Cm, native as N
    rng = random.Random(15)
    lay, blob, pk, _ = shot_setup
    circuits = _shot_circuits(Cm, rng, 3)
    adv, insts = lay.synthesize(circuits)
    seeds = [bytes(rng.getrandbits(8) for _ in range(32)) for _ in circuits]
    assert pk.quotient_selected() == (N.QUOTIENT_BUILTIN, True)
    builtin = pk.prove_batch(adv, insts, None, seeds=seeds)
    pk.quotient_select(N.QUOTIENT_INTERPRETER)
    try:
        assert pk.quotient_selected()[0] == N.QUOTIENT_INTERPRETER
        want = pk.prove_batch(adv, insts, None, seeds=seeds)
        assert builtin == want
        with pytest.raises(bzh2.BzhError) as e:
            pk.quotient_select(N.QUOTIENT_MODULE)                          # no module installed
        assert e.value.status == bzh2.E_RANGE
        if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
            return
        src = pk.quotient_source()
        assert "jit_quotient" in src and "jit_program_hash" in src
        shot_code = N.compile_quotient_source(src)
        assert shot_code is not None
        pk.set_quotient_module(shot_code)
        assert pk.quotient_selected()[0] == N.QUOTIENT_MODULE
        assert pk.prove_batch(adv, insts, None, seeds=seeds) == want
        # BoardCircuit: the Shot module is refused, its own flavours agree
        layb, _, pkb, _ = board_setup
        deck, _ = _random_deck(rng)
        ships, state = Cm.board_witness(deck, None)
        cb = [Cm.BoardCircuit(ships, state, rng.randrange(FQ))]
        advb, instb = layb.synthesize(cb)
        sd = [bytes(rng.getrandbits(8) for _ in range(32))]
        wantb = pkb.prove_batch(advb, instb, None, seeds=sd)              # builtin
        with pytest.raises(bzh2.BzhError) as e:
            pkb.set_quotient_module(shot_code)
        assert e.value.status == bzh2.E_ARG
        assert pkb.quotient_selected()[0] == N.QUOTIENT_BUILTIN
        pkb.quotient_select(N.QUOTIENT_INTERPRETER)
        try:
            assert pkb.prove_batch(advb, instb, None, seeds=sd) == wantb
        finally:
            pkb.quotient_select(N.QUOTIENT_BUILTIN)
        pk.set_quotient_module(None)
        assert pk.quotient_selected()[0] == N.QUOTIENT_BUILTIN          # NULL: back to the key's default
    finally:
        pk.set_quotient_module(None)
        pk.quotient_select(N.QUOTIENT_BUILTIN)
    assert pk.prove_batch(adv, insts, None, seeds=seeds) == want


def test_shot_unsatisfied_witnesses_are_refused_or_rejected(gpu_ctx, shot_setup):
    """A wrong hit assertion, a non-boolean hit, two shots, no shot (src/circuits/shot.rs:261-640): bzh_prove_batch
    returns BZH_E_RANGE, or the proof is rejected by bzh_verify_batch."""
    import bzh2
    from bzh2 import circuits as Cm
    from bzh2.game import BinaryValue
    lay, blob, pk, _ = shot_setup
    rng = random.Random(13)
    _, state = Cm.board_witness(PATTERN_1, None)
    cases = [Cm.ShotCircuit(state, 7, Cm.shot_serialize([8], [8]), BinaryValue.from_u8(1)),       # miss asserted as hit
             Cm.ShotCircuit(state, 7, Cm.shot_serialize([3], [5]), BinaryValue.from_u8(2)),       # non-boolean assertion
             Cm.ShotCircuit(state, 7, Cm.shot_serialize([3, 9], [3, 9]), BinaryValue.from_u8(1)),  # two shots
             Cm.ShotCircuit(state, 7, BinaryValue.empty(), BinaryValue.from_u8(0))]               # no shot
    for c in cases:
        adv, insts = lay.synthesize([c])
        rb = [bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes))]
        try:
            proofs = pk.prove_batch(adv, insts, rb)
        except bzh2.BzhError as e:
            assert e.status == bzh2.E_RANGE
        else:
            assert pk.verify_batch(insts, proofs) == [False]


def test_board_circuit_production_roundtrip_and_malicious_witnesses(gpu_ctx, oracle_c, board_setup, monkeypatch):
    """BoardCircuit at the reference's k = 12: valid fleets prove and verify (native + oracle verifier); every
    WitnessOption (src/utils/ship.rs:315-331) and the overflow / collision boards of src/circuits/board.rs:542-828 are
    refused (BZH_E_RANGE) or yield a rejected proof."""
    import bzh2
    from bzh2 import circuits as Cm
    lay, blob, pk, (g, w, u) = board_setup
    rng = random.Random(21)
    circuits = []
    for deck in (PATTERN_1, PATTERN_2, _random_deck(rng)[0]):
        ships, state = Cm.board_witness(deck, None)
        circuits.append(Cm.BoardCircuit(ships, state, rng.randrange(FQ)))
    adv, insts = lay.synthesize(circuits)
    rbs = [bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes)) for _ in circuits]
    proofs = pk.prove_batch(adv, insts, rbs)
    assert pk.verify_batch(insts, proofs) == [True, True, True]
    keys = _oracle_keys(B.decode(blob), g, w, u, monkeypatch)
    assert H.verify_proof(keys, insts[0], proofs[0], O.Blake2bTranscript(O.FP))
    swapped = [insts[1], insts[0], insts[2]]
    assert pk.verify_batch(swapped, proofs) == [False, False, True]
    bad_cases = []
    for opt in (1, 2, 3, 4, 5):                                           # DualPlacement .. Undersized, on one ship each
        opts = [0] * 5
        opts[(opt - 1) % 5] = opt
        bad_cases.append((PATTERN_1, opts))
    bad_cases.append(([(3, 4, False), (9, 6, True), (9, 0, False), (0, 6, False), (6, 1, True)], [0] * 5))   # horizontal overflow
    bad_cases.append(([(3, 6, True), (5, 4, False), (0, 1, False), (0, 5, True), (6, 1, False)], [0] * 5))   # vertical overflow
    bad_cases.append(([(3, 3, True), (5, 4, False), (4, 1, False), (0, 5, True), (6, 1, False)], [0] * 5))   # collision
    bad_cases.append(([None, (5, 4, False), (0, 1, False), (0, 5, True), (6, 1, True)], [0] * 5))           # missing carrier
    for deck, opts in bad_cases:
        ships, state = Cm.board_witness(deck, opts)
        adv1, inst1 = lay.synthesize([Cm.BoardCircuit(ships, state, 99)])
        rb = [bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes))]
        try:
            pr = pk.prove_batch(adv1, inst1, rb)
        except bzh2.BzhError as e:
            assert e.status == bzh2.E_RANGE, (deck, opts)
        else:
            assert pk.verify_batch(inst1, pr) == [False], (deck, opts)


@pytest.mark.parametrize("kind_name,bits,k", [("num2bits", 6, 5), ("bits2num", 9, 5), ("num2bits", 20, 6)])
def test_small_real_circuits_byte_identical_to_the_oracle_prover(gpu_ctx, oracle_c, kind_name, bits, k):
    """The reference's bitify test circuits (src/chips/bitify.rs:255-403: enable_equality on every column BEFORE the
    gate, a constants column, selector compression) at sizes the big-int oracle prover finishes: bzh_prove_batch emits
    the oracle's bytes under the same randomness, with the query order carried by the BZC2 blob."""
    import bzh2
    from bzh2 import circuits as Cm, native as N
    from bzh2.game import BinaryValue
    cv, F = O.VESTA, O.FP
    kind = Cm.NUM2BITS_TEST if kind_name == "num2bits" else Cm.BITS2NUM_TEST
    lay = Cm.CircuitLayout(kind, k, bits)
    try:
        blob = lay.blob()
        circ = B.decode(blob)
        assert circ.queries is not None and circ.queries[0][:3] == [(0, 0), (1, 0), (2, 0)]   # registered by enable_equality
        rng = random.Random(100 * bits + k)
        g = [cv.random_point(rng) for _ in range(1 << k)]
        w, u = cv.random_point(rng), cv.random_point(rng)
        cs = H.ConstraintSystem(circ.k, circ.num_advice, circ.num_fixed, circ.num_instance, circ.gates, circ.perm_columns, circ.lookups,
                                degree=circ.min_degree, queries=circ.queries)
        keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, circ.fixed, circ.copies, vk_repr=circ.vk_repr)
        pk = N.NativeProvingKey(gpu_ctx, blob, bzh2.CURVE_VESTA, g, w, u)
        try:
            value = rng.getrandbits(bits)
            adv = lay.synthesize_bitify_test(value, BinaryValue(value))
            rbytes = bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes))
            rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(pk.rng_bytes // 64)]
            adv_cols = [C.array_to_ints(adv[0, c]) for c in range(adv.shape[1])]
            want = H.create_proof(keys, adv_cols, [], rs, O.Blake2bTranscript(F))
            assert H.verify_proof(keys, [], want, O.Blake2bTranscript(F))
            got = pk.prove_batch(adv, [[]], [rbytes])
            assert got == [want]
            assert pk.verify_batch([[]], got) == [True]
        finally:
            pk.close()
    finally:
        lay.close()


def test_config2_batch_of_1024_shot_proofs(gpu_ctx, oracle_c, shot_setup, monkeypatch):
    """BASELINE.json configs[2]: a batch of 1024 ShotCircuit proofs at k = 11 (here in slices of 128, distinct fleets,
    shots and trapdoors, witnesses synthesised on the host and expanded on the device): every proof through
    bzh_verify_batch, a sample of 8 through the oracle's verify_proof."""
    import torch
    from bzh2 import circuits as Cm
    lay, blob, pk, (g, w, u) = shot_setup
    rng = random.Random(1024)
    total, slice_ = 1024, 128
    dev = torch.empty((slice_, lay.num_advice, lay.n, 4), dtype=torch.int64, device="cuda")
    keep = []
    hits = 0
    for s in range(total // slice_):
        circuits = _shot_circuits(Cm, rng, slice_)
        hits += sum(c.hit.value for c in circuits)
        _, insts = lay.synthesize(circuits, ctx=gpu_ctx, device_ptr=dev.data_ptr())
        rbs = [rng.randbytes(pk.rng_bytes) for _ in circuits]
        proofs = pk.prove_batch(None, insts, rbs, device_ptr=dev.data_ptr())
        assert len(set(proofs)) == slice_
        assert pk.verify_batch(insts, proofs) == [True] * slice_, s
        keep.append((insts[s % slice_], proofs[s % slice_]))
    assert 0 < hits < total                                               # both hit and miss witnesses were proved
    keys = _oracle_keys(B.decode(blob), g, w, u, monkeypatch)
    for inst, proof in keep:
        assert H.verify_proof(keys, inst, proof, O.Blake2bTranscript(O.FP))
