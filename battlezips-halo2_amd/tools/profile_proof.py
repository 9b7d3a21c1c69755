#!/usr/bin/env python3
"""cProfile of the device-resident create_proof (host-side orchestration cost).
Usage: profile_proof.py [k] [batch]   (batch > 1: the lockstep prover, bzh2/prover_batch.py)"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "battlezips-halo2_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

torch.zeros(1, device="cuda")
import bzh2
from bzh2 import prover_dev as D, synth
import bench

k = int(sys.argv[1]) if len(sys.argv) > 1 else 11
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
ctx = bzh2.Context(0, stream=torch.cuda.current_stream().cuda_stream)
circ, adv, inst = synth.battlezips_shaped(k, 1)
n = 1 << k
pts = bench.make_bases(ctx, 0, n + 2, 2)
g = [(bzh2.limbs_to_int(a[:4]), bzh2.limbs_to_int(a[4:])) for a in pts]
pk = D.DeviceProvingKey(ctx, circ, 0, g[:n], g[n + 1], g[n], dev, window_bits=8 if batch == 1 else 0)
adv_dev = [pk.ops.upload(c) for c in adv]
rb = np.random.default_rng(1).bytes(64 * (3 * n + 2048))
if batch > 1:
    from bzh2 import prover_batch as PB
    bp = PB.BatchProver(pk)
    adv_b = torch.stack(adv_dev).unsqueeze(0).repeat(batch, 1, 1, 1).contiguous()
    run = lambda: PB.create_proofs(bp, adv_b, [inst] * batch, [rb] * batch, [bzh2.Transcript(0) for _ in range(batch)])
else:
    run = lambda: D.create_proof(pk, adv_dev, inst, rb, bzh2.Transcript(0))
run()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    run()
torch.cuda.synchronize()
pr.disable()
if batch > 1:
    bp.trace = []
    run()
    tr = bp.trace
    print("phase wall times (ms, device synced at each boundary), batch", batch)
    for (a, ta), (b_, tb) in zip(tr, tr[1:]):
        print("  %-22s %8.2f" % (b_, (tb - ta) * 1e3))
    print("  %-22s %8.2f" % ("total", (tr[-1][1] - tr[0][1]) * 1e3))
    bp.trace = None
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
