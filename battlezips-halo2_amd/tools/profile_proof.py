#!/usr/bin/env python3
"""cProfile of one device-resident create_proof (host-side orchestration cost). Usage: profile_proof.py [k]"""
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
dev = torch.device("cuda", 0)
ctx = bzh2.Context(0, stream=torch.cuda.current_stream().cuda_stream)
circ, adv, inst = synth.battlezips_shaped(k, 1)
n = 1 << k
pts = bench.make_bases(ctx, 0, n + 2, 2)
g = [(bzh2.limbs_to_int(a[:4]), bzh2.limbs_to_int(a[4:])) for a in pts]
pk = D.DeviceProvingKey(ctx, circ, 0, g[:n], g[n + 1], g[n], dev)
adv_dev = [pk.ops.upload(c) for c in adv]
rb = np.random.default_rng(1).bytes(64 * (3 * n + 2048))
D.create_proof(pk, adv_dev, inst, rb, bzh2.Transcript(0))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    D.create_proof(pk, adv_dev, inst, rb, bzh2.Transcript(0))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
