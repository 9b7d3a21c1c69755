"""N>1 path on CPU: world_size-2 gloo run of the proof-batch partition and the record gather
(the only collective of the design; on the GPU box the same code runs over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bzh2.shard import gather_records, shard_range


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 256, 2816):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(shard_range(total, r, world))
            assert seen == list(range(total))
            sizes = [len(shard_range(total, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_range(total, rank, world)
        # record i = 12 limbs derived from the global proof index (stands in for a commitment record)
        local = torch.tensor([[i * 1000 + j for j in range(12)] for i in mine], dtype=torch.int64).reshape(len(mine), 12)
        counts = [len(shard_range(total, r, world)) for r in range(world)]
        full = gather_records(local, counts, dist)
        q.put((rank, full.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 8])
def test_gather_records_world2_gloo(total):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [[i * 1000 + j for j in range(12)] for i in range(total)]
    for _, full in got:
        assert full == want
