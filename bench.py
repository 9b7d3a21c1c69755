#!/usr/bin/env python3
"""bench.py -- halo2 create_proof on MI355X through libbzh2.so.

Default workload `proof_k14` (BASELINE.json configs[1..2]: BoardCircuit-sized proofs, k=14, IPA/Pasta, batched):
one "step" = `--concurrency` host threads each proving `--batch` independent witnesses with ONE bzh_prove_batch call
(csrc/prove.hip: the whole create_proof behind the C ABI, proofs advanced in lockstep so that every MSM, NTT, gate
evaluation, scan and IPA round is one launch for the batch).  The circuit has the reference's Board/Shot shape
(bzh2/synth.py: 11 advice / 8 fixed / 1 instance columns, 24 gates, degree 9, 13 permutation columns, one 10-bit
lookup); a proof covers column commitments, lookup permute + grand product, permutation grand products, the
vanishing argument (quotient over the 8n coset), evaluations, multiopen and the IPA opening, transcript included.
Witness columns are synthetic and resident in HBM before the timed region starts; every proof draws its own blinding
randomness and its bytes come back to the host inside the timed region.  Defaults: --batch 24 --concurrency 4
(96 proofs per step).  `--batch 1 --concurrency 1` is the single-proof latency configuration.
`--driver python` runs the ctypes-level drivers (bzh2/prover_dev.py, bzh2/prover_batch.py) instead.

`board_k14` / `board_k12` / `shot_k11` / `board_k17` / `shot_k11_batch` time only the MSM + NTT schedule of
such a proof (28 MSMs of n, 17 iNTT(n), 18 coset NTT(8n), 1 extended iNTT(8n));
`msm24` / `msm20` / `ntt22` are the config-5 microbenches.

roofline: the dominant kernel is k_msm_accumulate; `achieved` = the library's own count of algorithmic bytes
(32 B per scalar + 64 B per base point per launch, bzh_ctx_work) / its HIP-event time on the launch stream
(bzh_ctx_timings), both taken live over the timed region and summed over the host threads' contexts.

Multi-GPU: one process per GPU (torchrun); proofs are independent, so every rank runs the same per-GPU
workload (weak scaling) and the only collective is one RCCL all_gather of the ranks' outputs at the end of
the timed region.
"""
import argparse
import types
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "battlezips-halo2_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bzh2  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def pin_to_gpu_numa(local_rank):
    """Run this rank's host threads on the CPU socket its GPU hangs off (two-socket hosts: a prover thread on the far
    socket pays for every launch, challenge upload and commitment read-back across the socket link).  Threads created
    afterwards inherit the mask.  Returns the NUMA node, or None when the topology cannot be read."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:
        pass
    return None


def run_threads(fn, count):
    """fn(0) on the calling thread, fn(1..count-1) on helper threads; an exception in any of them is re-raised here
    (a failed batch must fail the run, not silently shorten the step)."""
    import threading
    errors = []

    def guarded(i):
        try:
            fn(i)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
    ths = [threading.Thread(target=guarded, args=(i,)) for i in range(1, count)]
    for t in ths:
        t.start()
    guarded(0)
    for t in ths:
        t.join()
    if errors:
        raise errors[0]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="proof_k14",
                    choices=["board_k14", "board_k12", "shot_k11", "board_k17", "shot_k11_batch", "msm24", "msm20", "ntt22",
                             "proof_k11", "proof_k12", "proof_k14", "proof_k8", "proof_k17", "verify_k11", "verify_k14", "mixed_board_shot"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + several ranks on one GPU rehearses the multi-rank path on a single-GPU box (ranks share device "
                         "local_rank %% device_count); the driver's multi-GPU runs use nccl (RCCL)")
    ap.add_argument("--no-precompute", action="store_true", help="plain bases: no fixed-base window table for the SRS")
    ap.add_argument("--batch", type=int, default=24,
                    help="proof_k* workloads: proofs advanced in lockstep per step (bzh2/prover_batch.py: one launch per kernel "
                         "class per phase for the whole batch); 1 = the single-proof latency path (bzh2/prover_dev.py)")
    ap.add_argument("--driver", default="native", choices=["native", "python"],
                    help="proof_k* workloads: native = bzh_prove_batch (csrc/prove.hip, the C-ABI whole-proof entry point); "
                         "python = the ctypes-level drivers bzh2/prover_dev.py (--batch 1) / bzh2/prover_batch.py")
    ap.add_argument("--window-bits", type=int, default=0, help="SRS window-table width (0: 8 for --batch 1, else the planner's)")
    ap.add_argument("--concurrency", type=int, default=4,
                    help="proof_k* workloads: independent proofs in flight per GPU (host threads, one ctx + stream each)")
    return ap.parse_args()


def rand_field(shape_elems, gen, device):
    """Random 253-bit values as (.., 4) int64 limbs: valid Montgomery representatives for every field here."""
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (*shape_elems, 4), dtype=torch.int64, device=device, generator=gen)
    t[..., 3] &= (1 << 61) - 1
    return t


def make_bases(ctx, curve, n, seed):
    """n random multiples of the curve generator (-1, 2) [(1, 2) on BN254], made with the product's own
    MSM (n single-point MSMs), normalised on the host."""
    F = {0: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
         1: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001}
    gx = 1 if curve == bzh2.CURVE_BN254 else F[curve] - 1
    g = np.concatenate([bzh2.int_to_limbs(gx), bzh2.int_to_limbs(2)]).reshape(1, 8)
    hb = ctx.upload_bases(curve, g)
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 8), dtype=np.uint64)
    step = 1 << 16
    for i0 in range(0, n, step):
        m = min(step, n - i0)
        sc = np.frombuffer(rng.bytes(m * 32), dtype=np.uint64).reshape(m, 1, 4).copy()
        sc[:, :, 3] &= (1 << 61) - 1
        jac = ctx.msm(hb, sc)
        out[i0:i0 + m] = bzh2.jacobian_to_affine(curve, jac)
    hb.free()
    return out


class Workload:
    """Device-resident inputs + the list of library calls that make one step."""

    def __init__(self, name, ctx, device, seed, precompute=True, concurrency=1, batch=1, window_bits=0, driver="native"):
        self.name, self.ctx = name, ctx
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.calls = []
        self.curve, self.field = bzh2.CURVE_VESTA, bzh2.FIELD_FP
        self.alg_bytes_msm_launch = 0
        self.units_per_step = 1
        self.desc = {}
        proof_shapes = {"board_k14": 14, "board_k12": 12, "shot_k11": 11, "board_k17": 17}
        if name in proof_shapes or name == "shot_k11_batch":
            k = 11 if name == "shot_k11_batch" else proof_shapes[name]
            proofs = 64 if name == "shot_k11_batch" else 1
            n, ext = 1 << k, 1 << (k + 3)
            self.k = k
            self.units_per_step = proofs
            self.bases = ctx.upload_bases(self.curve, make_bases(ctx, self.curve, n, seed + 1))
            if precompute:
                self.bases.precompute()  # SRS window table: one-time, outside the timed region
            self.msm_scalars = rand_field((28 * proofs, n), gen, device)
            self.msm_out = torch.zeros((28 * proofs, 12), dtype=torch.int64, device=device)
            self.cols = rand_field((17 * proofs, n), gen, device)
            self.ext = rand_field((18 * proofs, ext), gen, device)
            self.hx = rand_field((proofs, ext), gen, device)
            self.w_n = bzh2.field_omega(self.field, k, bzh2.FORM_MONTGOMERY)
            self.w_ext = bzh2.field_omega(self.field, k + 3, bzh2.FORM_MONTGOMERY)
            # halo2's extended coset generator is ZETA, a primitive cube root of unity: 5^((p-1)/3)
            p = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
            zeta = pow(5, (p - 1) // 3, p)
            self.zeta = bzh2.int_to_limbs(zeta * pow(2, 256, p) % p)
            self.calls = [
                ("msm", lambda: ctx.msm_device(self.bases, self.msm_scalars.data_ptr(), n, 28 * proofs, self.msm_out.data_ptr())),
                ("intt_n", lambda: ctx.ntt_device(self.field, self.cols.data_ptr(), k, 17 * proofs, self.w_n, None, True)),
                ("coset_ntt_8n", lambda: ctx.ntt_device(self.field, self.ext.data_ptr(), k + 3, 18 * proofs, self.w_ext, self.zeta, False)),
                ("ext_intt_8n", lambda: ctx.ntt_device(self.field, self.hx.data_ptr(), k + 3, proofs, self.w_ext, self.zeta, True)),
            ]
            self.alg_bytes_msm_launch = 28 * proofs * n * 32 + n * 64
            self.alg_bytes_step = self.alg_bytes_msm_launch + 64 * (17 * n + 19 * ext) * proofs
            self.desc = {"k": k, "proofs_per_step": proofs, "msm": "%dx2^%d vesta" % (28 * proofs, k),
                         "ntt": "%dx iNTT 2^%d + %dx coset NTT 2^%d + %dx coset iNTT 2^%d (Fp)" % (17 * proofs, k, 18 * proofs, k + 3, proofs, k + 3)}
            self.result = self.msm_out
        elif name == "mixed_board_shot":
            # BASELINE.json configs[3] on one GPU: Board-sized (k = 14) and Shot-sized (k = 11) proofs in the 1 : 10 ratio of
            # "256 Board + 2560 Shot", two host threads per circuit, every thread its own ctx / stream / proving key
            import threading
            from bzh2 import native as N, synth
            from bzh2.device import DeviceOps
            self.k = 14
            b14 = max(batch // 2, 1)
            plan = [(14, b14), (11, 10 * b14), (14, b14), (11, 10 * b14)]
            self.mix = []
            as_pt = lambda a: (bzh2.limbs_to_int(a[:4]), bzh2.limbs_to_int(a[4:]))
            for wi, (k, bsz) in enumerate(plan):
                n = 1 << k
                if wi == 0:
                    wctx = ctx
                else:
                    st = torch.cuda.Stream(device=device)
                    wctx = bzh2.Context(device.index or 0, stream=st.cuda_stream)
                circ, adv, inst = synth.battlezips_shaped(k, seed + k)
                g = [as_pt(a) for a in make_bases(wctx, self.curve, n + 2, seed + 1)]
                npk = N.NativeProvingKey(wctx, circ, self.curve, g[:n], g[n + 1], g[n])
                ops = DeviceOps(wctx, self.field, self.curve, npk.p, device)
                adv_b = torch.stack([ops.upload(list(col) + [0] * (n - len(col))) for col in adv]).unsqueeze(0).repeat(bsz, 1, 1, 1).contiguous()
                rbs = [np.random.default_rng(seed + 100 * wi + i).bytes(npk.rng_bytes) for i in range(bsz)]
                self.mix.append((npk, adv_b, [inst] * bsz, rbs, wctx))
            torch.cuda.synchronize(device)
            self.workers = [(None if wi == 0 else True, types.SimpleNamespace(ctx=m[4])) for wi, m in enumerate(self.mix)]
            self.proofs_made = [0] * len(plan)

            def prove_one(wi):
                npk, adv_b, insts, rbs, _ = self.mix[wi]
                proofs = npk.prove_batch(None, insts, rbs, device_ptr=adv_b.data_ptr())
                self.proofs_made[wi] = len(proofs)
                if wi == 0:
                    self.last_proof = proofs[0]

            def prove():
                run_threads(prove_one, len(plan))
            self.calls = [("bzh_prove_batch x4 (2 Board k=14, 2 Shot k=11)", prove)]
            self.units_per_step = sum(b for _, b in plan)
            self.alg_bytes_msm_launch = 0
            self.alg_bytes_step = 0
            self.last_proof = b""
            self.desc = {"mix": "per step %d Board-sized (k=14) + %d Shot-sized (k=11) proofs" % (2 * b14, 20 * b14)}
            self.result = torch.zeros((1, 12), dtype=torch.int64, device=device)
        elif name.startswith("verify_k"):
            # the reference's second benchmark (benches/board.rs:80-86): verify_proof over a batch of proofs of the
            # BattleZips-shaped circuit, made once (untimed) by bzh_prove_batch; one step = one bzh_verify_batch call
            from bzh2 import native as N, synth
            from bzh2.device import DeviceOps
            k = int(name[len("verify_k"):])
            n = 1 << k
            self.k = k
            circ, adv, inst = synth.battlezips_shaped(k, seed)
            pts = make_bases(ctx, self.curve, n + 2, seed + 1)
            as_pt = lambda a: (bzh2.limbs_to_int(a[:4]), bzh2.limbs_to_int(a[4:]))
            g = [as_pt(a) for a in pts]
            self.npk = N.NativeProvingKey(ctx, circ, self.curve, g[:n], g[n + 1], g[n])
            ops = DeviceOps(ctx, self.field, self.curve, self.npk.p, device)
            adv_b = torch.stack([ops.upload(list(col) + [0] * (n - len(col))) for col in adv]).unsqueeze(0).repeat(batch, 1, 1, 1).contiguous()
            torch.cuda.synchronize(device)
            rbs = [np.random.default_rng(seed + 100 + i).bytes(self.npk.rng_bytes) for i in range(batch)]
            self.insts = [inst] * batch
            self.proofs = self.npk.prove_batch(None, self.insts, rbs, device_ptr=adv_b.data_ptr())
            self.accepted = 0

            def verify():
                res = self.npk.verify_batch(self.insts, self.proofs)
                self.accepted = sum(res)
            self.calls = [("bzh_verify_batch", verify)]
            self.units_per_step = batch
            self.alg_bytes_msm_launch = 0
            self.alg_bytes_step = 0
            self.desc = {"k": k, "batch": batch, "proof_bytes": len(self.proofs[0])}
            self.result = torch.zeros((1, 12), dtype=torch.int64, device=device)
        elif name.startswith("proof_k"):
            # a COMPLETE proof per step: bzh2/prover_dev.create_proof on a circuit with the shape of the reference's
            # Shot / Board circuits (bzh2/synth.py); witness columns are resident in HBM before the timed region
            from bzh2 import prover_dev as D, synth
            k = int(name[len("proof_k"):])
            n = 1 << k
            self.k = k
            circ, adv, inst = synth.battlezips_shaped(k, seed)
            pts = make_bases(ctx, self.curve, n + 2, seed + 1)
            as_pt = lambda a: (bzh2.limbs_to_int(a[:4]), bzh2.limbs_to_int(a[4:]))
            g = [as_pt(a) for a in pts]
            wb = window_bits or (8 if batch == 1 else 0)
            self.window_bits = wb
            self.driver = driver
            if driver == "native":
                self._setup_native(ctx, device, circ, adv, inst, g, n, seed, batch, concurrency, wb)
                return
            self.pk = D.DeviceProvingKey(ctx, circ, self.curve, g[:n], g[n + 1], g[n], device, window_bits=wb)
            self.adv_dev = [self.pk.ops.upload(col) for col in adv]
            self.inst = inst
            self.circ = circ
            ndraws = 3 * n + 2048
            self.rng_pool = [np.random.default_rng(seed + 100 + i).bytes(64 * ndraws) for i in range(4)]
            self.step_no = 0
            self.last_proof = b""

            # `concurrency` independent proofs per step, one host thread each with its own ctx + HIP stream;
            # the proving key (SRS window table, fixed / permutation polynomials) is shared read-only
            import copy
            import threading
            self.units_per_step = concurrency
            self.workers = []
            for wi in range(concurrency):
                if wi == 0:
                    self.workers.append((None, self.pk))
                    continue
                st = torch.cuda.Stream(device=device)
                wctx = bzh2.Context(device.index or 0, stream=st.cuda_stream)
                wpk = copy.copy(self.pk)
                wpk.ctx = wctx
                wpk.ops = type(self.pk.ops)(wctx, self.pk.field, self.pk.curve, self.pk.p, device)
                self.workers.append((st, wpk))

            def prove_one(wi, rb):
                st, wpk = self.workers[wi]
                if st is None:
                    self.last_proof = D.create_proof(wpk, self.adv_dev, self.inst, rb, bzh2.Transcript(bzh2.FIELD_FP))
                else:
                    with torch.cuda.stream(st):
                        D.create_proof(wpk, self.adv_dev, self.inst, rb, bzh2.Transcript(bzh2.FIELD_FP))
                    st.synchronize()

            def prove():
                rbs = [self.rng_pool[(self.step_no + wi) % len(self.rng_pool)] for wi in range(concurrency)]
                self.step_no += 1
                run_threads(lambda wi: prove_one(wi, rbs[wi]), concurrency)
            self.calls = [("create_proof", prove)]
            if batch > 1:
                # lockstep batch: the same witness columns for every proof of the batch (each proof still draws its own
                # blinding randomness, so the proofs differ), stacked once and resident in HBM
                # `concurrency` such batches run side by side (host threads, one ctx + stream each): one batch's host
                # phases (transcripts, lookup sort, challenge uploads) overlap the other's kernels
                from bzh2 import prover_batch as PB
                bps = [PB.BatchProver(wpk) for _, wpk in self.workers]
                adv_b = torch.stack(self.adv_dev).unsqueeze(0).repeat(batch, 1, 1, 1).contiguous()
                self.units_per_step = batch * concurrency
                self.rng_pool = [np.random.default_rng(seed + 100 + i).bytes(64 * ndraws) for i in range(batch * concurrency + 3)]
                self.distinct = 0

                def prove_batch_one(wi, step_no):
                    rbs = [self.rng_pool[(step_no + wi * batch + b) % len(self.rng_pool)] for b in range(batch)]
                    st = self.workers[wi][0]
                    trs = [bzh2.Transcript(bzh2.FIELD_FP) for _ in range(batch)]
                    if st is None:
                        proofs = PB.create_proofs(bps[wi], adv_b, [self.inst] * batch, rbs, trs)
                        self.last_proof = proofs[0]
                        self.distinct = len(set(proofs))
                    else:
                        with torch.cuda.stream(st):
                            PB.create_proofs(bps[wi], adv_b, [self.inst] * batch, rbs, trs)
                        st.synchronize()

                def prove_batch():
                    sn = self.step_no
                    self.step_no += 1
                    run_threads(lambda wi: prove_batch_one(wi, sn), concurrency)
                self.calls = [("create_proofs", prove_batch)]
            self.alg_bytes_msm_launch = 0
            self.alg_bytes_step = 0  # filled from the library's own counters (bzh_ctx_work) after the timed region
            self.desc = {"k": k, "circuit": "synthetic, BattleZips-shaped: 11 advice / 8 fixed / 1 instance, 24 gates, degree 9, "
                                            "13 permutation columns, one 10-bit lookup (bzh2/synth.py)",
                         "proof_bytes": None}
            self.result = torch.zeros((1, 12), dtype=torch.int64, device=device)
        elif name in ("msm24", "msm20"):
            k = 24 if name == "msm24" else 20
            n = 1 << k
            self.k = k
            self.bases = ctx.upload_bases(self.curve, make_bases(ctx, self.curve, n, seed + 1))
            if precompute:
                self.bases.precompute()
            self.msm_scalars = rand_field((1, n), gen, device)
            self.msm_out = torch.zeros((1, 12), dtype=torch.int64, device=device)
            self.calls = [("msm", lambda: ctx.msm_device(self.bases, self.msm_scalars.data_ptr(), n, 1, self.msm_out.data_ptr()))]
            self.alg_bytes_msm_launch = n * 96
            self.alg_bytes_step = n * 96
            self.desc = {"msm": "1x2^%d vesta" % k}
            self.result = self.msm_out
        elif name == "ntt22":
            k = 22
            self.k = k
            self.data = rand_field((1, 1 << k), gen, device)
            self.w = bzh2.field_omega(self.field, k, bzh2.FORM_MONTGOMERY)
            self.calls = [("ntt", lambda: ctx.ntt_device(self.field, self.data.data_ptr(), k, 1, self.w, None, False))]
            self.alg_bytes_step = 64 << k
            self.desc = {"ntt": "1x NTT 2^22 (Fp)"}
            self.result = self.data[:, :4].contiguous().view(1, 16)[:, :12].contiguous()
        else:
            raise ValueError(name)

    def _setup_native(self, ctx, device, circ, adv, inst, g, n, seed, batch, concurrency, wb):
        """proof_k* through bzh_pk_create / bzh_prove_batch: `concurrency` host threads, each with its own ctx + stream +
        proving key, each proving `batch` witnesses per step in lockstep; witness tensor resident in HBM."""
        import threading
        from bzh2 import native as N
        from bzh2.device import DeviceOps
        self.workers = []
        self.npks = []
        for wi in range(concurrency):
            if wi == 0:
                st, wctx = None, ctx
            else:
                st = torch.cuda.Stream(device=device)
                wctx = bzh2.Context(device.index or 0, stream=st.cuda_stream)
            self.npks.append(N.NativeProvingKey(wctx, circ, self.curve, g[:n], g[n + 1], g[n], window_bits=wb))
            self.workers.append((st, self.npks[-1]))
        ops = DeviceOps(ctx, self.field, self.curve, self.npks[0].p, device)
        adv_b = torch.stack([ops.upload(list(col) + [0] * (n - len(col))) for col in adv]).unsqueeze(0).repeat(batch, 1, 1, 1).contiguous()
        torch.cuda.synchronize(device)
        self.adv_b = adv_b
        nbytes = self.npks[0].rng_bytes
        self.rng_pool = [np.random.default_rng(seed + 100 + i).bytes(nbytes) for i in range(batch * concurrency + 3)]
        self.step_no = 0
        self.last_proof = b""
        self.distinct = 0
        self.units_per_step = batch * concurrency
        self.inst = inst

        def prove_one(wi, sn):
            rbs = [self.rng_pool[(sn + wi * batch + b) % len(self.rng_pool)] for b in range(batch)]
            proofs = self.workers[wi][1].prove_batch(None, [self.inst] * batch, rbs, device_ptr=adv_b.data_ptr())
            self.last_batch[wi] = proofs
            if wi == 0:
                self.last_proof = proofs[0]
                self.distinct = len(set(proofs))

        self.last_batch = [[] for _ in range(concurrency)]
        self.proof_stride = self.npks[0].max_proof_bytes

        def proof_records():
            """the last step's proofs of this rank as fixed-stride records {u32 length, bytes}: what the final gather carries"""
            recs = bytearray()
            for proofs in self.last_batch:
                for pr in proofs:
                    recs += len(pr).to_bytes(4, "little") + pr + bytes(self.proof_stride - len(pr))
            a = np.frombuffer(bytes(recs), dtype=np.uint8).reshape(-1, 4 + self.proof_stride)
            return torch.from_numpy(a.copy()).to(device)
        self.records = proof_records

        def prove():
            sn = self.step_no
            self.step_no += 1
            run_threads(lambda wi: prove_one(wi, sn), concurrency)
        self.calls = [("bzh_prove_batch", prove)]
        self.solo_step = lambda: prove_one(0, self.step_no)
        self.alg_bytes_msm_launch = 0
        self.alg_bytes_step = 0
        self.desc = {"k": self.k, "circuit": "synthetic, BattleZips-shaped: 11 advice / 8 fixed / 1 instance, 24 gates, degree 9, "
                                            "13 permutation columns, one 10-bit lookup (bzh2/synth.py)",
                     "proof_bytes": None, "driver": "native (bzh_prove_batch)"}
        self.result = torch.zeros((1, 12), dtype=torch.int64, device=device)

    def step(self):
        for _, fn in self.calls:
            fn()


def cpu_baseline(workload_name, k):
    """Time the C oracle ("port": CPU restatement, not the Rust crate) on one full step's worth of
    MSM+NTT work for the proof workloads, all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle as C
    import pasta as O
    import random
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # the 1-GPU box's CPU share; more threads only shrink the per-thread windows
    n, ext = 1 << k, 1 << (k + 3)
    rng = np.random.default_rng(99)

    def rnd(m):
        a = np.frombuffer(rng.bytes(m * 32), dtype=np.uint64).reshape(m, 4).copy()
        a[:, 3] &= (1 << 61) - 1
        return a

    g = O.VESTA.random_point(random.Random(1))
    bases = C.point_walk(0, C.points_to_array([g])[0], n)
    n_msm, n_intt, n_cntt = 2, 2, 2
    t0 = time.perf_counter()
    for _ in range(n_msm):
        C.msm(0, rnd(n), bases, cores)
    t_msm = (time.perf_counter() - t0) / n_msm
    a = rnd(n)
    t0 = time.perf_counter()
    for _ in range(n_intt):
        C.ntt(0, a, O.FP.omega(k), inverse=True, threads=cores)
    t_intt = (time.perf_counter() - t0) / n_intt
    e = rnd(ext)
    t0 = time.perf_counter()
    for _ in range(n_cntt):
        C.ntt(0, e, O.FP.omega(k + 3), coset_shift=O.FP.g, threads=cores)
    t_cntt = (time.perf_counter() - t0) / n_cntt
    per_step = 28 * t_msm + 17 * t_intt + 19 * t_cntt
    return {"value": 1.0 / per_step, "unit": "proof-workloads/s", "cores": cores, "kind": "port",
            "sample": "C oracle (liboracle.so, halo2-style chunked Pippenger + radix-2 FFT): %d MSM 2^%d, %d iNTT 2^%d, "
                      "%d coset NTT 2^%d timed, scaled to 28/17/19 per step" % (n_msm, k, n_intt, k, n_cntt, k + 3),
            "seconds_per_step": per_step}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.dist_backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    numa_node = pin_to_gpu_numa(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    stream = torch.cuda.current_stream(device)
    ctx = bzh2.Context(local_rank, stream=stream.cuda_stream)
    wl = Workload(args.workload, ctx, device, seed=1234 + rank, precompute=not args.no_precompute, concurrency=args.concurrency, batch=args.batch,
                  window_bits=args.window_bits, driver=args.driver)

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        wl.step()
    barrier()
    all_ctx = [ctx] + [w[1].ctx for w in getattr(wl, "workers", []) if w[0] is not None]
    all_ctx = list({id(c): c for c in all_ctx}.values())
    for c in all_ctx:
        c.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    if dist is not None:  # the one collective: gather every rank's commitments (fixed-stride records)
        from bzh2.shard import gather_records
        local = wl.records() if hasattr(wl, "records") else wl.result  # complete proofs: the proof records themselves
        gathered = gather_records(local, [local.shape[0]] * world, dist)
        assert gathered.shape[0] == local.shape[0] * world
    barrier()
    elapsed = time.perf_counter() - t0
    timings = ctx.timings()
    for c in all_ctx[1:]:  # proofs in flight on their own ctx + stream: sum their kernel classes into the report
        for kname, v in c.timings().items():
            for f in v:
                timings[kname][f] += v[f]
    for c in all_ctx:
        c.profile(False)
    # The same kernels with ONE batch in flight (two extra, untimed steps on the first context): with several batches
    # overlapping, every kernel shares the GPU with the others' and its own duration stretches, which says nothing about
    # the kernel.  Reported beside the timed region's figures as roofline.single_batch_in_flight.
    solo = None
    if rank == 0 and hasattr(wl, "solo_step") and len(all_ctx) > 1:
        wl.solo_step()
        torch.cuda.synchronize(device)
        ctx.profile(True)
        for _ in range(2):
            wl.solo_step()
        torch.cuda.synchronize(device)
        solo = ctx.timings()["msm_accumulate"]
        ctx.profile(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        units = wl.units_per_step * args.steps * world
        is_proof = args.workload.startswith(("board", "shot"))
        is_full = args.workload.startswith("proof_k")
        is_verify = args.workload.startswith("verify_k")
        is_mixed = args.workload == "mixed_board_shot"
        acc = timings["msm_accumulate"]
        nt = timings["ntt"]
        if args.workload == "ntt22":
            dom_ms = nt["ms"] / max(nt["launches"], 1) * (nt["launches"] / max(args.steps, 1))  # all passes of one NTT
            alg = wl.alg_bytes_step
            dom_name = "k_ntt_pass (all passes of one 2^22 NTT)"
        elif is_full or is_verify or is_mixed:
            # MSM launches of a proof differ in size (28 column commits batched by phase, then the halving IPA
            # rounds): average the bytes the library counted per launch (bzh_ctx_work) over the same launches
            dom_ms = acc["ms"] / max(acc["launches"], 1)
            alg = acc["algorithmic_bytes"] / max(acc["launches"], 1)
            dom_name = "k_msm_accumulate (mean over the %d launches of one step)" % (acc["launches"] // max(args.steps, 1))
            wl.alg_bytes_step = (acc["algorithmic_bytes"] + nt["algorithmic_bytes"]) / max(args.steps, 1)
        else:
            dom_ms = acc["ms"] / max(acc["launches"], 1)
            alg = wl.alg_bytes_msm_launch
            dom_name = "k_msm_accumulate"
        achieved = alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM-side traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs of this same command; tools/pmc_traffic.py), if one exists for this workload
        traffic, traffic_src = None, None
        for tag in ("r01_h", "r01_e"):
            tj = os.path.join(ROOT, "profiles", "%s_%s_pmc_traffic.json" % (tag, args.workload))
            if traffic is not None or not os.path.exists(tj) or args.workload == "ntt22":
                continue
            try:
                for e in json.load(open(tj))["kernels"]:
                    if "k_msm_accumulate" in e["kernel"] and ("512" in e["kernel"] or e.get("workgroup", 0) >= 512):
                        traffic = e["read_bytes_raw"] + e["write_bytes"]
                        traffic_src = os.path.relpath(tj, ROOT) + " (raw FETCH_SIZE: 64-B gathers, see its correction note)"
            except Exception:
                traffic = None
        line = {
            "metric": ("%s proof MSM+NTT workloads per second" % args.workload) if is_proof else
                      (("complete proofs per second, BattleZips-shaped circuit, k=%d, IPA/Pasta" % wl.k) if is_full
                       else ("%s runs per second" % args.workload)),
            "value": units / elapsed,
            "unit": "proof-workloads/s" if is_proof else ("proofs/s" if is_full else "runs/s"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x8 (255-bit modular integer, Montgomery)", "data": "synthetic",
            "config": dict({"workload": args.workload, "curve": "vesta/Fp (IPA over Pasta, the reference's locked build)",
                            "stages": "MSM commits + NTT/iNTT/coset-NTT of one proof; NOT included: synthesis, quotient, "
                                      "grand products, multiopen/IPA, transcript",
                            "form": "montgomery", "srs_window_table": not args.no_precompute, "parallelism": "independent proofs per GPU (dp%d)" % world,
                            "algorithmic_GB_per_step": wl.alg_bytes_step / 1e9,
                            "whole_step_GBps": wl.alg_bytes_step * args.steps / elapsed / 1e9}, **wl.desc),
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": dom_ms},
            "kernel_ms": {kname: v for kname, v in timings.items() if v["launches"]},
        }
        line["config"]["host_threads_pinned_to_numa_node"] = numa_node
        if (is_full or is_verify) and acc["ms"] > 0:
            # the bound that actually binds this kernel: bucket additions per second against the rate
            # the same XYZZ mixed addition reaches in isolation (profiles/r01_ubench_field_gfx950.txt), i.e. the integer
            # multiplier peak; kernel time is summed over the host threads' contexts, so overlapping batches understate it
            wb = getattr(wl, "window_bits", 0) or 11
            nwin = (256 + wb - 1) // wb
            npts = (1 << wl.k) + 2
            scalars = (acc["algorithmic_bytes"] - 64.0 * npts * acc["launches"]) / 32.0
            adds = scalars * nwin
            line["roofline"]["alu_equivalent"] = {"unit": "G mixed additions/s", "achieved": adds / (acc["ms"] * 1e-3) / 1e9, "peak": 14.2,
                                                  "frac": adds / (acc["ms"] * 1e-3) / 1e9 / 14.2, "table_rows": nwin}
            if solo is not None and solo["ms"] > 0:
                s_ms = solo["ms"] / max(solo["launches"], 1)
                s_alg = solo["algorithmic_bytes"] / max(solo["launches"], 1)
                s_adds = (solo["algorithmic_bytes"] - 64.0 * npts * solo["launches"]) / 32.0 * nwin
                line["roofline"]["single_batch_in_flight"] = {
                    "avg_launch_ms": s_ms, "achieved": s_alg / (s_ms * 1e-3) / 1e9, "frac": s_alg / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "alu_equivalent": {"achieved": s_adds / (solo["ms"] * 1e-3) / 1e9, "peak": 14.2, "frac": s_adds / (solo["ms"] * 1e-3) / 1e9 / 14.2},
                    "note": "same kernel, one batch in flight (2 untimed steps after the timed region); in the timed region "
                            "%d batches share the GPU and every launch stretches accordingly" % len(all_ctx)}
        if is_mixed:
            line["metric"] = "complete proofs per second, Board-sized (k=14) : Shot-sized (k=11) = 1 : 10, IPA/Pasta"
            line["unit"] = "proofs/s"
            line["config"]["stages"] = "complete create_proof for every proof of the mix (bzh_prove_batch), witnesses resident in HBM"
        if is_verify:
            line["metric"] = "proof verifications per second, BattleZips-shaped circuit, k=%d, IPA/Pasta" % wl.k
            line["unit"] = "verifications/s"
            line["config"]["stages"] = ("complete verify_proof: instance commitments, transcript replay, expected h(x), multiopen "
                                        "recombination, IPA equation (one n-term MSM per proof)")
            line["config"]["accepted_in_last_batch"] = wl.accepted
            assert wl.accepted == wl.units_per_step, "a valid proof was rejected"
        if is_full:
            line["config"]["stages"] = ("complete create_proof: commitments, lookup, permutation, vanishing, quotient, evaluations, "
                                        "multiopen, IPA, transcript; witness synthesis excluded (columns resident in HBM)")
            line["config"]["proof_bytes"] = len(wl.last_proof)
            line["config"]["proofs_in_flight_per_gpu"] = args.concurrency * args.batch
            line["config"]["concurrent_batches"] = args.concurrency
            line["config"]["batch"] = args.batch
            line["config"]["srs_window_bits"] = wl.window_bits or "planner"
            if args.batch > 1 or args.driver == "native":
                line["config"]["distinct_proofs_in_last_batch"] = wl.distinct
        if world == 1 and (is_proof or is_full) and not args.no_cpu_baseline and args.workload != "shot_k11_batch":
            line["cpu_baseline"] = cpu_baseline(args.workload, wl.k)
            if is_full:
                line["cpu_baseline"]["unit"] = "proofs/s (upper bound)"
                line["cpu_baseline"]["sample"] += ("; only the proof's MSM+NTT schedule is timed on the CPU, so a CPU prover's "
                                                   "complete-proof rate is below this figure")
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
