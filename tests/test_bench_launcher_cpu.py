"""`python bench.py --gpus N` must start N rank processes itself (one per GPU, before anything touches the GPU) when it
is not already running under torchrun, and refuse a --gpus that contradicts WORLD_SIZE.  Checked here without a GPU
through --dry-run (gloo rendezvous on 127.0.0.1 + the record gather of bzh2/shard.py); the GPU counterpart is
tests/test_gpu_bench_ranks.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_launches_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--dist-backend", "gloo", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["ranks_gathered"] == [0, 1]


def test_gpus_1_stays_one_process():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_gpus_that_contradicts_world_size_is_refused():
    r = _run(["--gpus", "8", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_force_dist_joins_a_group_with_one_rank():
    """--force-dist: world size 1 still goes through init_process_group, the record gather and the barrier (the GPU counterpart
    runs them over RCCL: tests/test_gpu_bench_ranks.py)"""
    r = _run(["--gpus", "1", "--dry-run", "--force-dist", "--dist-backend", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["ranks_gathered"] == [0] and line["config"]["dist"].startswith("gloo process group")
