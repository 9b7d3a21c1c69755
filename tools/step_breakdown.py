import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "battlezips-halo2_amd"))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--no-cpu-baseline", "--concurrency", "1"]
import bench, torch, bzh2
dev = torch.device("cuda:0")
ctx = bzh2.Context(0)
wl = bench.Workload("proof_k14", ctx, dev, 1, concurrency=1, batch=16)
r = wl.runner
for it in range(4):
    t0 = time.perf_counter(); circuits = r._circuits(0, 16)
    t1 = time.perf_counter(); _, insts = r.layout.synthesize(circuits, ctx=r.ctxs[0], device_ptr=r.adv[0].data_ptr(), threads=4); r.ctxs[0].sync()
    t2 = time.perf_counter(); blob = r.np_rng[0].bytes(r.rng_bytes * 16); rbs = [blob[i * r.rng_bytes:(i + 1) * r.rng_bytes] for i in range(16)]
    t3 = time.perf_counter(); proofs = r.pks[0].prove_batch(None, insts, rbs, device_ptr=r.adv[0].data_ptr())
    t4 = time.perf_counter()
    print("circuits %.2f  synth %.2f  rng %.2f (%d B/proof)  prove_batch %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, r.rng_bytes, (t4-t3)*1e3), flush=True)
