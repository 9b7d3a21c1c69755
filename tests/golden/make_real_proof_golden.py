"""Generate tests/golden/real_proofs.json: proofs of the reference's REAL BoardCircuit made by the ORACLE prover
(oracle/halo2_oracle.py, bulk arithmetic through the C oracle: oracle/accel.py) at sizes where running the oracle
inside the GPU test would take too long (k = 17, ~10 min on 8 cores).  The GPU test
(tests/test_gpu_real_circuit_parity.py) asserts bzh_prove_batch emits exactly these bytes on the same Params::new SRS,
witness and randomness stream.

Inputs are all derived from the tag stored with each entry: SRS = Params::new(k) (hash-to-curve generators; the
host routine bzh_params_generators is checked against the oracle's hash_to_curve in tests/test_params_cpu.py and
tests/test_gpu_params.py), witness = helpers.real_parity.board_circuits(seed), randomness = SHAKE-256(tag).

Run:  python tests/golden/make_real_proof_golden.py [k ...]      (from the repo root; no GPU needed)"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "battlezips-halo2_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import accel as A                                   # noqa: E402
import halo2_oracle as H                            # noqa: E402
import pasta as O                                   # noqa: E402
from bzh2 import circuits as Cm, params as Pm       # noqa: E402
from helpers import real_parity as R                # noqa: E402


def make(kind, k, seed):
    tag = "bzh2-golden-%s-k%d-seed%d" % (kind, k, seed)
    lay = Cm.CircuitLayout(Cm.BOARD if kind == "board" else Cm.SHOT, k)
    blob = lay.blob()
    g_arr, w, u = Pm.generators(k)
    circuits = (R.board_circuits if kind == "board" else R.shot_circuits)(Cm, seed, 1)
    adv, insts = lay.synthesize(circuits)
    t0 = time.time()
    with A.accelerated(R.THREADS):
        keys = R.oracle_keys(blob, R.points_of(g_arr), w, u)
        stream = R.rng_stream(tag, 64 * (2 * lay.n + 4096))
        proof = R.oracle_prove(keys, adv[0], insts[0], stream)
        assert H.verify_proof(keys, insts[0], proof, O.Blake2bTranscript(O.FP))
    print("%s: %d bytes in %.0f s" % (tag, len(proof), time.time() - t0), flush=True)
    lay.close()
    return {"tag": tag, "kind": kind, "k": k, "seed": seed, "proof_hex": proof.hex()}


if __name__ == "__main__":
    ks = [int(a) for a in sys.argv[1:]] or [17]
    path = os.path.join(HERE, "real_proofs.json")
    entries = json.load(open(path)) if os.path.exists(path) else []
    for k in ks:
        e = make("board", k, 1700 + k)
        entries = [x for x in entries if x["tag"] != e["tag"]] + [e]
    json.dump(entries, open(path, "w"), indent=0)
