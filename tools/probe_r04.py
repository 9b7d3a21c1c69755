import sys, time, os
sys.path.insert(0, "battlezips-halo2_amd"); sys.path.insert(0, ".")
import torch
torch.zeros(1, device="cuda")
import numpy as np, bzh2
import bench
ctx = bzh2.Context(0)
for curve in (0, 2):
    for lg in (16, 20):
        t = time.time(); b = bench.make_bases(ctx, curve, 1 << lg, 5); print("make_bases curve", curve, "2^%d" % lg, round(time.time() - t, 2), "s", flush=True)
from bzh2 import circuits as Cm
import random
rng = random.Random(1)
Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
for n in (1, 30, 2816, 65536):
    ms, ts = [rng.randrange(1 << 100) for _ in range(n)], [rng.randrange(Q) for _ in range(n)]
    Cm.pedersen_commit_batch(ctx, ms, ts)
    m, t_, out = Cm._limbs(ms), Cm._limbs(ts), np.zeros((n, 8), dtype=np.uint64)
    L = Cm._bind(); VP = Cm._VP
    t = time.time()
    reps = 20 if n < 10000 else 3
    for _ in range(reps):
        L.bzh_pedersen_commit_batch(ctx.handle, VP(m.ctypes.data), VP(t_.ctypes.data), n, VP(out.ctypes.data))
    dt = (time.time() - t) / reps
    t = time.time()
    hn = min(n, 64)
    for i in range(hn):
        Cm.pedersen_commit_host(ms[i], ts[i])
    dh = (time.time() - t) / hn
    print("pedersen batch n=%d: %.3f ms per call, %.2f us per commitment; host %.1f us per commitment" % (n, dt * 1e3, dt / n * 1e6, dh * 1e6), flush=True)
