"""bench.py as the driver runs it: `--gpus N` starts N rank processes (here 2 ranks sharing the one GPU of this box over gloo;
on an 8-GPU node the same launcher binds rank r to device r over RCCL), and BASELINE.json configs[3] -- a FIXED batch of
Board (k = 14) + Shot (k = 11) proofs sharded over the ranks -- yields, after the one gather, exactly the records of the
unsharded run: same public inputs, same proof bytes, every (kind, index) once (records are the
BattleZipsWASM shape of src/wasm/circuit_wasm.rs:27-31 through bzh_record_encode / bzh_record_decode)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_sharded_fixed_batch_gathers_the_unsharded_runs_records(gpu_ctx, tmp_path):
    from bzh2.wire import BattleZipsRecord, KIND_BOARD, KIND_SHOT
    common = ["--workload", "mixed_board_shot", "--mix-divisor", "32", "--steps", "1", "--warmup", "0", "--batch", "8", "--no-cpu-baseline"]
    one = _bench(common + ["--gpus", "1", "--records-out", str(tmp_path / "one.npy")])
    two = _bench(common + ["--gpus", "2", "--dist-backend", "gloo", "--records-out", str(tmp_path / "two.npy")])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["last_batches_verified"] and two["config"]["last_batches_verified"]

    def load(path):
        a = np.load(path)
        recs = [BattleZipsRecord.from_fixed(a[i].tobytes()) for i in range(a.shape[0])]
        return {(r.kind, r.index): r for r in recs}, len(recs)
    r1, n1 = load(tmp_path / "one.npy")
    r2, n2 = load(tmp_path / "two.npy")
    assert n1 == n2 == len(r1) == len(r2) == 256 // 32 + 2560 // 32
    assert sum(1 for k in r1 if k[0] == KIND_BOARD) == 8 and sum(1 for k in r1 if k[0] == KIND_SHOT) == 80
    assert r1.keys() == r2.keys()
    for key in r1:
        assert r1[key].commitment == r2[key].commitment and r1[key].proof == r2[key].proof, key
    assert len({r.proof for r in r1.values()}) == n1


def test_weak_scaling_two_ranks_line(gpu_ctx):
    """proof_k11 on two ranks: the line reports n_gpus = 2 and the whole-job rate"""
    line = _bench(["--workload", "proof_k11", "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "16",
                   "--concurrency", "2", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["last_batches_verified"]
    assert line["value"] > 0 and abs(line["value"] - 2 * 16 * 2 * 2 / (line["ms_per_step"] * 2 / 1e3)) < 1e-6 * line["value"]


def test_the_rccl_path_runs_with_one_rank(gpu_ctx, tmp_path):
    """`--force-dist`: with WORLD_SIZE = 1 bench.py still does init_process_group("nccl", device_id=...) -- RCCL --, the device
    all_gather of the proof records (uint8, fixed stride), the barriers and the MAX all_reduce of the elapsed time: everything the
    8-GPU run executes except a second peer.  The gathered records decode and verify like the ungrouped run's."""
    from bzh2.wire import BattleZipsRecord, KIND_SHOT
    common = ["--workload", "proof_k11", "--steps", "2", "--warmup", "1", "--batch", "8", "--concurrency", "2", "--no-cpu-baseline",
              "--other-workloads", "none"]
    grouped = _bench(common + ["--force-dist", "--records-out", str(tmp_path / "g.npy")])
    plain = _bench(common + ["--records-out", str(tmp_path / "p.npy")])
    assert grouped["n_gpus"] == 1 and grouped["config"]["dist"].startswith("nccl process group, world 1")
    assert plain["config"]["dist"].startswith("none")
    assert grouped["config"]["last_batches_verified"] and grouped["config"]["host_threads_per_rank"]["world"] == 1
    a, b = np.load(tmp_path / "g.npy"), np.load(tmp_path / "p.npy")
    assert a.shape == b.shape and a.shape[0] == 16
    ra = [BattleZipsRecord.from_fixed(a[i].tobytes()) for i in range(a.shape[0])]
    rb = [BattleZipsRecord.from_fixed(b[i].tobytes()) for i in range(b.shape[0])]
    assert all(r.kind == KIND_SHOT for r in ra) and sorted(r.index for r in ra) == sorted(r.index for r in rb)
    assert {r.index: r.proof for r in ra} == {r.index: r.proof for r in rb}      # seeds derive from the proof index: same bytes
