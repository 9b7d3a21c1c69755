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


# ---- fixed batch of real witnesses sharded over two ranks (BASELINE.json configs[3]) -------------------------------
def _fixed_batch(total):
    """the fixed batch every rank can rebuild: witness i = fleet / shot / trapdoor derived from i alone"""
    import random
    from bzh2 import circuits as Cm
    from bzh2.game import BinaryValue
    decks = [[(3, 3, True), (5, 4, False), (0, 1, False), (0, 5, True), (6, 1, False)],
             [(3, 4, False), (9, 6, True), (0, 0, False), (0, 6, False), (6, 1, True)]]
    out = []
    for i in range(total):
        rng = random.Random(i)
        _, state = Cm.board_witness(decks[i % 2], None)
        x, y = rng.randrange(10), rng.randrange(10)
        hit = (state.value >> (10 * y + x)) & 1
        out.append(Cm.ShotCircuit(state, rng.getrandbits(250), Cm.shot_serialize([x], [y]), BinaryValue.from_u8(hit)))
    return out


def _witness_records(layout, circuits):
    """one fixed-stride record per witness: its 4 public inputs (4 x 4 limbs) + a checksum of the advice tensor"""
    import numpy as np
    adv, insts = layout.synthesize(circuits, form=1)
    recs = []
    for b, inst in enumerate(insts):
        limbs = []
        for v in inst[0]:
            limbs += [int(x) for x in np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.int64)]
        limbs.append(int(np.bitwise_xor.reduce(adv[b].reshape(-1)).astype(np.int64)))
        limbs.append(int(adv[b].sum(dtype=np.uint64).astype(np.int64)))
        recs.append(limbs)
    return torch.tensor(recs, dtype=torch.int64).reshape(len(circuits), 18)


def _synth_worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bzh2 import circuits as Cm
        lay = Cm.CircuitLayout(Cm.SHOT, 11)
        mine = shard_range(total, rank, world)
        batch = _fixed_batch(total)
        local = _witness_records(lay, [batch[i] for i in mine])
        counts = [len(shard_range(total, r, world)) for r in range(world)]
        full = gather_records(local, counts, dist)
        q.put((rank, full.tolist()))
    finally:
        dist.destroy_process_group()


def _run_world2(target, args):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_fixed_batch_of_shot_witnesses_sharded_world2_equals_one_rank():
    """Strong-scaling shard of a FIXED batch (bench.py --workload mixed_board_shot): two gloo ranks synthesise
    shard_range(total, rank, 2) of the same 7 ShotCircuit witnesses with the product's C++ front end and gather fixed-stride
    records; the gathered batch equals the one-rank run record for record."""
    from bzh2 import circuits as Cm
    total = 7
    got = _run_world2(_synth_worker, (total,))
    lay = Cm.CircuitLayout(Cm.SHOT, 11)
    want = _witness_records(lay, _fixed_batch(total)).tolist()
    lay.close()
    for _, full in got:
        assert full == want


# ---- one MSM split over two ranks (BASELINE.json configs[4], multi-GPU) ---------------------------------------------
def _msm_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import random
        import numpy as np
        import coracle as C
        import pasta as O
        from bzh2.shard import combine_msm_partials
        g = O.VESTA.random_point(random.Random(5))
        bases = C.point_walk(0, C.points_to_array([g])[0], n)
        sc = np.frombuffer(np.random.default_rng(6).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
        sc[:, 3] &= (1 << 61) - 1
        mine = shard_range(n, rank, world)
        lo, hi = mine.start, mine.stop
        # the local partial sum: on the GPU box this is bzh_msm over the rank's N / world points; here the C oracle stands in
        part = C.array_to_point(C.msm(0, np.ascontiguousarray(sc[lo:hi]), np.ascontiguousarray(bases[lo:hi]), 1))
        jac = np.concatenate([C.points_to_array([part])[0], np.array([1, 0, 0, 0], dtype=np.uint64)]) if part else np.zeros(12, dtype=np.uint64)
        total = combine_msm_partials(0, jac, dist)
        q.put((rank, [int(v) for v in total]))
    finally:
        dist.destroy_process_group()


def test_msm_split_over_two_ranks_combines_to_the_full_msm():
    """N / 2 points per rank, all_gather of the 96-byte partials, local adds (bzh_jacobian_sum): the result equals the
    single-rank MSM over all N points."""
    import random
    import numpy as np
    import bzh2
    import coracle as C
    import pasta as O
    n = 301
    got = _run_world2(_msm_worker, (n,))
    g = O.VESTA.random_point(random.Random(5))
    bases = C.point_walk(0, C.points_to_array([g])[0], n)
    sc = np.frombuffer(np.random.default_rng(6).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= (1 << 61) - 1
    want = O.VESTA.compress(C.array_to_point(C.msm(0, sc, bases, 2)))
    for _, total in got:
        aff = bzh2.jacobian_to_affine(0, np.array(total, dtype=np.uint64))
        assert bzh2.affine_compress(0, aff) == [want]
