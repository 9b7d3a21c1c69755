#!/usr/bin/env python3
"""bench.py -- halo2 create_proof for the BattleZips circuits on MI355X through libbzh2.so.

Default workload `proof_k14` (BASELINE.json configs[1..2]): complete proofs of the reference's BoardCircuit
(src/circuits/board.rs, 57 gates, built by the C++ front end csrc/circuit/*.hpp) in a 2^14-row table, IPA over Pasta.
One "step" = `--concurrency` host threads, each taking `--batch` DISTINCT witnesses (fleets from a pool, a fresh
trapdoor per proof and step) through
    bzh_synthesize_board  (Circuit::synthesize: host C++ witness generation, compact pinned staging, expansion kernel)
    bzh_prove_batch       (create_proof for the batch in lockstep: every MSM, NTT, gate evaluation, scan and IPA round
                           is one launch for the batch; transcript included; proof bytes back on the host)
both INSIDE the timed region.  `proof_k11` is the ShotCircuit at the reference's own size (benches/shot.rs:22),
`proof_k12` the BoardCircuit at its own size (benches/board.rs:22), `proof_k17` the Board scale-up.  After the timed
region the last batch of every thread is checked with bzh_verify_batch (untimed).
`--batch 1 --concurrency 1` is the single-proof latency configuration.

`mixed_board_shot` is BASELINE.json configs[3]: a FIXED batch of 256 Board (k = 14) + 2560 Shot (k = 11) proofs per step
(divided by --mix-divisor), sharded over the ranks with bzh2.shard.shard_range (strong scaling), gathered as
fixed-stride proof records at the end.  `verify_k*` is the reference's second benchmark (benches/board.rs:80-86).
`board_k14` / `board_k12` / `shot_k11` / `board_k17` / `shot_k11_batch` time only the MSM + NTT schedule of a proof
(SURVEY.md section 3.1); `msm24` / `msm20` / `ntt22` are the config-5 microbenches -- with --gpus N > 1 `msm24` splits the
2^24 points over the ranks (N/8 points each), all_gathers the partial sums (96 B each) and adds them locally.

roofline: the dominant kernel is k_msm_accumulate; `achieved` = the library's own count of algorithmic bytes
(32 B per scalar + 64 B per base point per launch, bzh_ctx_work) / its HIP-event time on the launch stream
(bzh_ctx_timings), both taken live over the timed region and summed over the host threads' contexts.

Multi-GPU: one process per GPU (torchrun); proofs are independent, so every rank runs the same per-GPU
workload (weak scaling; `mixed_board_shot` and multi-rank `msm24`: strong) and the only collective is one RCCL all_gather
of the ranks' proof records at the end of the timed region.
"""
import argparse
import types
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "battlezips-halo2_amd"))


def launch_ranks():
    """`bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment (i.e. not under torchrun): start the N rank
    processes here -- one per GPU, rank r on device r, RCCL rendezvous on 127.0.0.1 -- BEFORE this process imports torch,
    loads libbzh2.so or touches the GPU in any way (children are fresh interpreters, not forks or execs of a process that
    has initialised HIP), wait for them and exit with their status.  Rank 0 prints the one JSON line."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    known, _ = ap.parse_known_args()
    if "WORLD_SIZE" in os.environ:
        if int(os.environ["WORLD_SIZE"]) != known.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (python -m torch.distributed.run "
                             "--nproc-per-node N bench.py --gpus N ..., or plain `python bench.py --gpus N`)" % (known.gpus, os.environ["WORLD_SIZE"]))
        return
    if known.gpus <= 1:
        return
    import socket
    import subprocess
    rc = 0
    for attempt in (0, 1):
        # The port is probed and released before the children bind it (they import torch first): another process can take it
        # in between.  The parent never touches the GPU, so a launch that dies early is simply started again, once, with
        # fresh children and a new port.
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        t_launch = time.time()
        procs = []
        for r in range(known.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(known.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc = 0
        pending = list(procs)
        while pending:
            for pr in list(pending):
                code = pr.poll()
                if code is None:
                    continue
                pending.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in pending:       # a failed rank leaves the others waiting in a collective: stop exactly those children
                        other.terminate()
            time.sleep(0.05)
        if rc == 0 or attempt == 1 or time.time() - t_launch > 120:
            break
        sys.stderr.write("bench.py: a rank exited with %d within %.0f s of the launch (rendezvous on port %d?): launching the ranks once more\n"
                         % (rc, time.time() - t_launch, port))
    sys.exit(128 - rc if rc < 0 else rc)      # a child killed by signal N reports 128 + N, like a shell


if __name__ == "__main__":
    launch_ranks()

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
                             "proof_k11", "proof_k12", "proof_k14", "proof_k17", "verify_k11", "verify_k14", "mixed_board_shot"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true",
                    help="latency runs: the timed region runs without the library's HIP event records around its kernel classes (two "
                         "extra, untimed steps afterwards fill kernel_ms and the roofline instead)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + several ranks on one GPU rehearses the multi-rank path on a single-GPU box (ranks share device "
                         "local_rank %% device_count); the driver's multi-GPU runs use nccl (RCCL)")
    ap.add_argument("--no-precompute", action="store_true", help="plain bases: no fixed-base window table for the SRS")
    ap.add_argument("--batch", type=int, default=64,
                    help="proof_k* workloads: witnesses synthesised and proved in lockstep per bzh_prove_batch call; "
                         "1 = the single-proof latency path")
    ap.add_argument("--circuit", default="auto", choices=["auto", "shot", "board"],
                    help="proof_k* workloads: which of the reference's circuits (auto: ShotCircuit at k = 11, BoardCircuit otherwise)")
    ap.add_argument("--mix-divisor", type=int, default=1, help="mixed_board_shot: divide the fixed batch (256 Board + 2560 Shot) by this")
    ap.add_argument("--window-bits", type=int, default=0, help="SRS window-table width (0: 8 for --batch 1, else the planner's)")
    ap.add_argument("--no-quotient-codegen", action="store_true",
                    help="proof_k* workloads: the interpreted quotient evaluator (default: the circuit's kernel built into libbzh2.so)")
    ap.add_argument("--explicit-rng", action="store_true",
                    help="proof_k* workloads: generate every proof's random stream on the host (numpy) and pass it to bzh_prove_batch "
                         "instead of a 32-byte seed per proof expanded on the device (bzh_prove_batch_seeded)")
    ap.add_argument("--curve", default="vesta", choices=["vesta", "pallas", "bn254"],
                    help="msm24 / msm20 / ntt22 (BASELINE configs[4]): the curve of the MSM, the NTT runs over its scalar field "
                         "(vesta: Fp, the reference's commitment curve; pallas: Fq; bn254: G1 / Fr -- no reference, SURVEY F3)")
    ap.add_argument("--records-out", default=None,
                    help="rank 0 writes the gathered proof records of the last step (uint8 array, numpy .npy) here")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: every rank joins the process group, gathers one fixed-stride record per rank and rank 0 prints "
                         "the line -- checks the --gpus launcher and the rendezvous on a CPU-only machine (tests/test_bench_launcher_cpu.py)")
    ap.add_argument("--force-dist", action="store_true",
                    help="world size 1: still join a process group on --dist-backend (nccl = RCCL) and run the record all_gather, the "
                         "barriers and the MAX all_reduce of the multi-GPU path (tests/test_gpu_bench_ranks.py: the first 8-GPU run must "
                         "not be the first time that code executes)")
    ap.add_argument("--other-workloads", default="auto", choices=["auto", "all", "none"],
                    help="after the main timed region, short regions for the rest of BASELINE.json's metric -- proof_k11 (Shot 128 x 8), "
                         "proof_k12, proof_k17 (8 x 4), verify_k14, msm24 and ntt22 on vesta and bn254 -- reported under "
                         "config.other_workloads, each with its own roofline and cpu_baseline.  auto: when the main workload is the default "
                         "proof_k14 at the default batch / concurrency; with several ranks only the proof sizes run (weak scaling)")
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


PATTERN_1 = [(3, 3, True), (5, 4, False), (0, 1, False), (0, 5, True), (6, 1, False)]      # src/circuits/board.rs:101-107
FQ_MODULUS = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001


def random_fleet(rng):
    """a valid random fleet (rejection sampling of in-bounds, non-overlapping placements) and its occupied cells"""
    while True:
        used, deck = set(), []
        for length in (5, 4, 3, 3, 2):
            for _ in range(200):
                z = rng.random() < 0.5
                x, y = rng.randrange(10 - (0 if z else length - 1)), rng.randrange(10 - (length - 1 if z else 0))
                cells = {(x, y + i) if z else (x + i, y) for i in range(length)}
                if not cells & used:
                    used |= cells
                    deck.append((x, y, z))
                    break
            else:
                break
        if len(deck) == 5:
            return deck, used


class ProofRunner:
    EXPLICIT_RNG = False   # --explicit-rng
    QUOTIENT_CODEGEN = True   # --no-quotient-codegen
    """`workers` host threads, each with its own ctx + stream + proving key + advice tensor in HBM, proving slices of
    `batch` witnesses of one real circuit: bzh_synthesize_{shot,board} (device output) then bzh_prove_batch."""

    POOL = 64   # distinct fleets / shots the witnesses cycle through; every proof also gets its own trapdoor
    SYNTH_THREADS = 4   # host threads per bzh_synthesize_* call (main() lowers it to the rank's share of the cores)

    def __init__(self, kind, k, ctx, device, seed, batch, workers, window_bits=0, first_ctx=None):
        import random
        from bzh2 import circuits as Cm, native as N
        from bzh2.game import BinaryValue
        self.kind, self.k, self.batch, self.device = kind, k, batch, device
        self.explicit_rng = ProofRunner.EXPLICIT_RNG
        self.Cm = Cm
        self.layout = Cm.CircuitLayout(Cm.SHOT if kind == "shot" else Cm.BOARD, k)
        blob = self.layout.blob()
        d = self.layout.describe()
        self.circuit_desc = ("%sCircuit of the reference (src/circuits/%s.rs) via the C++ front end: %d gates / %d constraint polynomials, %d advice + %d fixed "
                             "(incl. %d selector) columns, degree %d, %d permutation columns, 1 lookup, %d regions, %d of 2^%d rows used"
                             % (kind.capitalize(), kind, len(d["gates"]), d["num_polys"], d["num_advice"], d["num_fixed"],
                                d["num_fixed"] - 10, d["degree"], len(d["permutation"]), len(d["regions"]), d["rows_used"], k))
        n = 1 << k
        # the reference's own SRS: Params::<vesta::Affine>::new(k) (benches/shot.rs:58, benches/board.rs:51) -- hash-to-curve
        # generators, g_lagrange by a device group FFT, cached on disk; one pair of window tables shared by every worker
        from bzh2 import params as Pm
        self.params = Pm.Params(first_ctx if first_ctx is not None else ctx, k, window_bits=window_bits)
        rng = random.Random(seed)
        self.pool = []
        for _ in range(self.POOL):
            deck, used = random_fleet(rng)
            ships, state = Cm.board_witness(deck, None)
            x, y = rng.randrange(10), rng.randrange(10)
            self.pool.append((ships, state, Cm.shot_serialize([x], [y]), BinaryValue.from_u8(1 if (x, y) in used else 0)))
        self.rng_seed = seed
        self.ctxs, self.streams, self.pks, self.adv = [], [], [], []
        for wi in range(workers):
            if wi == 0 and first_ctx is not None:
                st, wctx = None, first_ctx
            else:
                st = torch.cuda.Stream(device=device)
                wctx = bzh2.Context(device.index or 0, stream=st.cuda_stream)
            self.streams.append(st)
            self.ctxs.append(wctx)
            if wi == 0:   # ONE proving key (fixed / sigma / hoisted columns) shared by every worker stream; workspaces are per ctx
                self.pk = N.NativeProvingKey(wctx, blob, bzh2.CURVE_VESTA, params=self.params)
                if os.environ.get("BZH_BENCH_COEFF_COMMITS"):   # experiment: coefficient-basis commitments everywhere
                    self.pk.set_lagrange(None)
            self.pks.append(self.pk)
            self.adv.append(torch.zeros((batch, self.layout.num_advice, n, 4), dtype=torch.int64, device=device))
        torch.cuda.synchronize(device)
        self.rng_bytes = self.pks[0].rng_bytes
        self.proof_stride = self.pks[0].max_proof_bytes
        self.last_batch = [[] for _ in range(workers)]
        self.last_insts = [[] for _ in range(workers)]
        self.step_batches = [[] for _ in range(workers)]      # every batch of the current plan (for the record gather)
        self.np_rng = [np.random.default_rng(seed + 1000 + wi) for wi in range(workers)]
        self.kind_id = 1 if kind == "shot" else 0           # bzh2.wire.KIND_SHOT / KIND_BOARD
        # The quotient evaluator: the reference's circuits have their kernel inside libbzh2.so (generated at build time, picked
        # by bzh_pk_create); --no-quotient-codegen selects the interpreter instead.
        from bzh2 import native as Nn
        if not ProofRunner.QUOTIENT_CODEGEN:
            self.pk.quotient_select(Nn.QUOTIENT_INTERPRETER)
        self.quotient_codegen = self.pk.quotient_selected()[0] != Nn.QUOTIENT_INTERPRETER

    def worker_ctxs(self):
        return [(st, types.SimpleNamespace(ctx=c)) for st, c in zip(self.streams, self.ctxs)]

    def close(self):
        """release the key, the SRS tables, the worker contexts and the advice tensors (the next workload gets the HBM)"""
        torch.cuda.synchronize(self.device)
        self.pk.close()
        self.params.close()
        for st, c in zip(self.streams, self.ctxs):
            if st is not None:      # contexts this runner created (worker 0 may run on the caller's)
                c.close()
        self.layout.close()
        self.adv, self.pks, self.ctxs, self.streams = [], [], [], []

    def _circuits(self, lo, count):
        Cm = self.Cm
        out = []
        for i in range(lo, lo + count):
            ships, state, shot, hit = self.pool[i % self.POOL]
            trapdoor = (0x9e3779b97f4a7c15 * (i + 1) * (self.rng_seed + 3) + (i << 130)) % FQ_MODULUS
            out.append(Cm.ShotCircuit(state, trapdoor, shot, hit) if self.kind == "shot" else Cm.BoardCircuit(ships, state, trapdoor))
        return out

    def _prove_slice(self, wi, lo, count):
        circuits = self._circuits(lo, count)
        _, insts = self.layout.synthesize(circuits, ctx=self.ctxs[wi], device_ptr=self.adv[wi].data_ptr(), threads=ProofRunner.SYNTH_THREADS)
        if self.explicit_rng:   # the caller supplies every random byte (2 MB per proof at k = 14): the parity tests' mode
            blob = self.np_rng[wi].bytes(self.rng_bytes * count)
            rbs = [blob[i * self.rng_bytes:(i + 1) * self.rng_bytes] for i in range(count)]
            proofs = self.pk.prove_batch(None, insts, rbs, device_ptr=self.adv[wi].data_ptr(), ctx=self.ctxs[wi])
        else:                   # a 32-byte seed per proof, expanded on the device (bzh_prove_batch_seeded) -- create_proof's OsRng.
            # BENCHMARK seeds: a function of (run seed, circuit, proof index) so that a sharded run makes the proofs of the unsharded
            # one (tests/test_gpu_bench_ranks.py); a real caller draws them from the OS (include/bzh2.h)
            import hashlib
            seeds = [hashlib.blake2b(b"bzh2-bench-seed %d %s %d" % (self.rng_seed, self.kind.encode(), i), digest_size=32).digest()
                     for i in range(lo, lo + count)]
            proofs = self.pk.prove_batch(None, insts, None, device_ptr=self.adv[wi].data_ptr(), seeds=seeds, ctx=self.ctxs[wi])
        self.last_batch[wi], self.last_insts[wi] = proofs, insts
        self.step_batches[wi].append((lo, insts, proofs))

    def plan(self, lo, count):
        """jobs (one per worker) that together prove witnesses [lo, lo + count) in slices of `batch`"""
        slices = [(s, min(self.batch, lo + count - s)) for s in range(lo, lo + count, self.batch)]
        W = len(self.pks)

        def job(wi):
            def run():
                self.step_batches[wi] = []
                for s, c in slices[wi::W]:
                    self._prove_slice(wi, s, c)
            return run
        return [job(wi) for wi in range(W)]

    def has_records(self):
        return any(self.step_batches)

    def proof_records(self):
        """the last step's proofs of this rank as fixed-stride BattleZipsWASM records (bzh_record_encode: public inputs +
        proof bytes, tagged with circuit kind and proof index), in index order: what the final gather carries"""
        from bzh2.wire import BattleZipsRecord
        items = []
        for batches in self.step_batches:
            for lo, insts, proofs in batches:
                for j, pr in enumerate(proofs):
                    items.append((lo + j, BattleZipsRecord([int(v) for v in insts[j][0]], pr, self.kind_id, lo + j).to_fixed(self.proof_stride)))
        items.sort()
        a = np.frombuffer(b"".join(r for _, r in items), dtype=np.uint8).reshape(len(items), -1)
        return torch.from_numpy(a.copy()).to(self.device)

    def verify_last(self):
        ok = True
        for wi in range(len(self.ctxs)):
            if self.last_batch[wi]:
                ok = ok and all(self.pk.verify_batch(self.last_insts[wi], self.last_batch[wi], ctx=self.ctxs[wi]))
        return ok


class Workload:
    """Device-resident inputs + the list of library calls that make one step."""

    def __init__(self, name, ctx, device, seed, precompute=True, concurrency=1, batch=1, window_bits=0, circuit="auto", rank=0, world=1,
                 mix_divisor=1, dist=None, curve="vesta"):
        self.name, self.ctx = name, ctx
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.calls = []
        self.curve = {"vesta": bzh2.CURVE_VESTA, "pallas": bzh2.CURVE_PALLAS, "bn254": bzh2.CURVE_BN254}[curve]
        self.field = bzh2.CURVE_SCALAR_FIELD[self.curve]
        self.curve_desc = {"vesta": "vesta / Fp (IPA over Pasta, the reference's locked build)", "pallas": "pallas / Fq",
                           "bn254": "bn254 G1 / Fr (no reference: SURVEY F3)"}[curve]
        if curve != "vesta" and name not in ("msm24", "msm20", "ntt22"):
            raise SystemExit("--curve applies to the config-5 microbenches (msm24 / msm20 / ntt22); the circuits are Vesta / Fp")
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
            # BASELINE.json configs[3]: a FIXED batch of 256 Board (k = 14) + 2560 Shot (k = 11) proofs per step, sharded over
            # the ranks (strong scaling): rank r proves shard_range(total, r, world) of each kind
            from bzh2.shard import shard_range
            self.k = 14
            nb, ns = max(256 // mix_divisor, 1), max(2560 // mix_divisor, 1)
            mine_b, mine_s = shard_range(nb, rank, world), shard_range(ns, rank, world)
            fixed_seed = seed - rank           # the fixed batch is the same set of proofs whatever the world size
            self.provers = [ProofRunner("board", 14, ctx, device, fixed_seed, batch=max(batch // 2, 1), workers=2, window_bits=window_bits, first_ctx=ctx),
                            ProofRunner("shot", 11, ctx, device, fixed_seed + 7, batch=max(batch // 2, 1) * 8, workers=2, window_bits=window_bits)]
            self.shares = [(mine_b.start, len(mine_b)), (mine_s.start, len(mine_s))]
            self.workers = [w for p in self.provers for w in p.worker_ctxs()]

            def prove():
                jobs = []
                for p, (lo, cnt) in zip(self.provers, self.shares):
                    jobs += p.plan(lo, cnt)
                run_threads(lambda i: jobs[i](), len(jobs))
            self.calls = [("bzh_synthesize_* + bzh_prove_batch, fixed batch shard", prove)]
            self.units_per_step = nb + ns                 # whole-job units: the fixed batch (every rank adds its share)
            self.local_units_per_step = len(mine_b) + len(mine_s)
            self.fixed_total = True
            stride = max(p.proof_stride for p in self.provers)     # one record format for both circuits
            for p in self.provers:
                p.proof_stride = stride
            self.records = lambda: torch.cat([p.proof_records() for p in self.provers if p.has_records()], dim=0)
            self.record_counts = lambda: [len(shard_range(nb, r, world)) + len(shard_range(ns, r, world)) for r in range(world)]
            self.verify_last = lambda: all(p.verify_last() for p in self.provers)
            self.alg_bytes_msm_launch = 0
            self.alg_bytes_step = 0
            self.last_proof = b""
            self.desc = {"mix": "per step %d BoardCircuit (k=14) + %d ShotCircuit (k=11) proofs, fixed batch; this rank: %d + %d"
                                % (nb, ns, len(mine_b), len(mine_s))}
            self.result = torch.zeros((1, 12), dtype=torch.int64, device=device)
        elif name.startswith("verify_k"):
            # the reference's second benchmark (benches/board.rs:80-86): verify_proof over a batch of proofs of the real
            # circuit, made once (untimed) by bzh_prove_batch; one step = one bzh_verify_batch call
            k = int(name[len("verify_k"):])
            self.k = k
            self.runner = ProofRunner("shot" if k == 11 else "board", k, ctx, device, seed, batch=batch, workers=1, window_bits=window_bits, first_ctx=ctx)
            self.runner.plan(0, batch)[0]()
            self.insts, self.proofs = self.runner.last_insts[0], self.runner.last_batch[0]
            self.accepted = 0

            def verify():
                res = self.runner.pks[0].verify_batch(self.insts, self.proofs)
                self.accepted = sum(res)
            self.calls = [("bzh_verify_batch", verify)]
            self.units_per_step = batch
            self.alg_bytes_msm_launch = 0
            self.alg_bytes_step = 0
            self.desc = {"k": k, "batch": batch, "proof_bytes": len(self.proofs[0]), "circuit": self.runner.circuit_desc}
            self.result = torch.zeros((1, 12), dtype=torch.int64, device=device)
        elif name.startswith("proof_k"):
            # COMPLETE proofs of the reference's circuit per step: witness synthesis + create_proof, both timed
            k = int(name[len("proof_k"):])
            self.k = k
            kind = circuit if circuit != "auto" else ("shot" if k == 11 else "board")
            if k >= 16:
                # 2^17-row tables: 8 proofs per batch already fill the GPU (and 21 GB of cosets per worker); BZH_BENCH_BIG_BATCH lifts the cap
                batch = min(batch, int(os.environ.get("BZH_BENCH_BIG_BATCH", "8")))
            wb = window_bits or (8 if batch == 1 else 0)
            self.window_bits = wb
            self.runner = ProofRunner(kind, k, ctx, device, seed, batch=batch, workers=concurrency, window_bits=wb, first_ctx=ctx)
            self.workers = self.runner.worker_ctxs()
            self.step_no = 0

            def prove():
                jobs = self.runner.plan(self.step_no * batch * concurrency, batch * concurrency)
                self.step_no += 1
                run_threads(lambda i: jobs[i](), len(jobs))
            self.calls = [("bzh_synthesize_%s + bzh_prove_batch" % kind, prove)]
            self.solo_step = lambda: self.runner.plan(0, batch)[0]()
            self.units_per_step = batch * concurrency
            self.batch = batch
            self.records = self.runner.proof_records
            self.verify_last = self.runner.verify_last
            self.alg_bytes_msm_launch = 0
            self.alg_bytes_step = 0
            self.desc = {"k": k, "circuit": self.runner.circuit_desc, "proof_bytes": None,
                         "driver": "native (bzh_synthesize_* + %s)" % ("bzh_prove_batch" if ProofRunner.EXPLICIT_RNG else "bzh_prove_batch_seeded"),
                         "quotient_evaluator": "builtin kernel (the circuit's program as straight-line code, generated and linked when libbzh2.so was built)" if self.runner.quotient_codegen else "interpreted (k_expr_vm2)",
                         "randomness": ("every draw generated on the host and passed in (%d bytes per proof)" % self.runner.rng_bytes) if ProofRunner.EXPLICIT_RNG
                         else "a fresh 32-byte seed per proof, expanded on the device (ChaCha20) -- the reference draws OsRng inside create_proof"}
            self.result = torch.zeros((1, 12), dtype=torch.int64, device=device)
        elif name in ("msm24", "msm20"):
            # config 5: one 2^k-point MSM.  With several ranks the POINTS are split (shard_range: N / world per GPU, no
            # exchange of inputs); each step all-gathers the 96-byte partial sums and every rank adds them locally
            from bzh2.shard import combine_msm_partials, shard_range
            k = 24 if name == "msm24" else 20
            n = 1 << k
            self.k = k
            mine = shard_range(n, rank, world)
            nl = len(mine)
            # the rank's points: G_r, 2 G_r, ..., nl G_r for a rank-specific random G_r, made on the device (bzh_bases_walk: the
            # 1 GB of a 2^24-point table never crosses PCIe; SURVEY 8d's "cheap generator walk")
            self.bases = ctx.bases_walk(self.curve, make_bases(ctx, self.curve, 1, 77 + rank)[0], nl)
            if precompute:
                self.bases.precompute()
            self.msm_scalars = rand_field((1, nl), gen, device)
            self.msm_out = torch.zeros((1, 12), dtype=torch.int64, device=device)
            self.calls = [("msm", lambda: ctx.msm_device(self.bases, self.msm_scalars.data_ptr(), nl, 1, self.msm_out.data_ptr()))]
            if world > 1:
                self.fixed_total = True
                self.total_point = None

                def combine():
                    self.total_point = combine_msm_partials(self.curve, self.msm_out[0], dist, bzh2.FORM_MONTGOMERY)
                self.calls.append(("all_gather 96-B partials + local adds", combine))
            self.alg_bytes_msm_launch = nl * 96
            self.alg_bytes_step = n * 96
            self.desc = {"msm": "1x2^%d %s%s" % (k, curve, (", points split over %d ranks (%d each)" % (world, nl)) if world > 1 else ""),
                         "stages": "one 2^%d-point multi-scalar multiplication (digits, bucket accumulation, reduction, final sum)" % k}
            self.result = self.msm_out
        elif name == "ntt22":
            k = 22
            self.k = k
            self.data = rand_field((1, 1 << k), gen, device)
            self.w = bzh2.field_omega(self.field, k, bzh2.FORM_MONTGOMERY)
            self.calls = [("ntt", lambda: ctx.ntt_device(self.field, self.data.data_ptr(), k, 1, self.w, None, False))]
            self.alg_bytes_step = 64 << k
            self.desc = {"ntt": "1x NTT 2^22 over the scalar field of %s" % curve,
                         "stages": "one forward radix-2 NTT of 2^22 elements, in place (all passes)"}
            self.result = self.data[:, :4].contiguous().view(1, 16)[:, :12].contiguous()
        else:
            raise ValueError(name)

    def step(self):
        for _, fn in self.calls:
            fn()

    def close(self):
        for r in [getattr(self, "runner", None)] + list(getattr(self, "provers", [])):
            if r is not None:
                r.close()
        if getattr(self, "bases", None) is not None:
            self.bases.free()
        for a in ("msm_scalars", "msm_out", "cols", "ext", "hx", "data", "result", "runner", "provers", "bases"):
            if hasattr(self, a):
                setattr(self, a, None)
        self.calls = []
        torch.cuda.empty_cache()


def _cpu_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))  # the 1-GPU box's CPU share; more threads only shrink the per-thread windows


def cpu_baseline(workload_name, k):
    """MSM + NTT schedule workloads (board_k14 ...): the C oracle ("port": CPU restatement, not the Rust crate) on one step's
    worth of MSM + NTT work, all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle as C
    import pasta as O
    import random
    cores = _cpu_cores()
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


def cpu_baseline_proof(runner):
    """A COMPLETE create_proof schedule for the runner's real circuit on the host cores with the C oracle (oracle/oracle.c:
    restatements of upstream's best_multiexp, best_fft, the per-row gate evaluation and the IPA generator collapse), every
    stage of SURVEY section 3.1 steps 1-9 timed on a bounded sample of the real data and scaled to one proof:
      commitments (instance / advice / permuted lookup in the Lagrange basis on the REAL witness columns -- sparse, as upstream
      sees them -- and the dense z / random / h / f / s ones), iNTT + coset NTT of every committed column, the quotient's gate
      evaluation over the 2^(k+3) rows (all constraint polynomials of the real circuit, y-fold), evaluations at x, and the k
      IPA rounds (two half-size MSMs + the generator collapse each).  Witness synthesis is the product's (not timed here)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import blob as Bm
    import coracle as C
    import pasta as O
    cores = _cpu_cores()
    k = runner.k
    n, ext = 1 << k, 1 << (k + 3)
    F = O.FP
    circ = Bm.decode(runner.layout.blob())
    g, gl, w, u, _ = runner.params.points()
    adv, _ = runner.layout.synthesize(runner._circuits(0, 1))          # one real witness, canonical, host
    rng = np.random.default_rng(7)

    def rnd(m):
        a = np.frombuffer(rng.bytes(m * 32), dtype=np.uint64).reshape(m, 4).copy()
        a[:, 3] &= (1 << 61) - 1
        return a

    def timed(fn, reps=1):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps
    parts = {}
    na = adv.shape[1]
    # 1. commitments: advice + instance in the Lagrange basis on the real columns; lookup A', S' hold 10-bit table values
    parts["commit_advice_lagrange_x%d" % na] = sum(timed(lambda c=c: C.msm(0, np.ascontiguousarray(adv[0, c]), gl, cores)) for c in range(na))
    small = np.zeros((n, 4), dtype=np.uint64)
    small[:, 0] = rng.integers(0, 1024, n, dtype=np.uint64)
    parts["commit_lookup_permuted_x2"] = 2 * timed(lambda: C.msm(0, small, gl, cores))
    dense = rnd(n)
    t_dense = timed(lambda: C.msm(0, dense, g, cores), 2)
    parts["commit_dense_x15 (instance, 3 z, random, 8 h, f, s)"] = 15 * t_dense
    # 2. transforms of every committed column
    col = rnd(n)
    parts["intt_n_x17"] = 17 * timed(lambda: C.ntt(0, col, F.omega(k), inverse=True, threads=cores), 2)
    e = rnd(ext)
    parts["coset_ntt_8n_x18 + ext_intt"] = 19 * timed(lambda: C.ntt(0, e, F.omega(k + 3), coset_shift=F.g, threads=cores), 2)
    # 3. quotient: all constraint polynomials of the real circuit over ALL 2^(k+3) extended rows (k <= 14; a quarter of them,
    #    scaled, above that), on random columns of the extended size -- the values do not change the work
    prog, consts, colmap = C.compile_gates(circ.gates)
    rows_timed = ext if k <= 14 else ext // 4
    cols = [rnd(ext) for _ in range(len(colmap))]
    t_q = timed(lambda: C.gate_eval(0, prog, consts, cols, 12345, 0, rows_timed, threads=cores, rot_scale=8))
    parts["quotient_gates_8n_rows"] = t_q * (ext / rows_timed)
    del cols
    # 4. evaluations at x (Horner): every queried (column, rotation) + z / lookup / sigma / h evaluations
    nq = len(circ.queries[0]) + len(circ.queries[1]) + len(circ.queries[2]) + len(circ.perm_columns) + 16 if circ.queries else 80
    parts["evaluations_x%d" % nq] = nq * timed(lambda: C.eval_poly(0, col, 987654321), 4)
    # 5. IPA: every one of the k rounds as upstream runs it -- round j works on half = n / 2^(j+1) generators: two MSMs of that
    #    size and the generator collapse g_lo + [u] g_hi
    t_ipa = 0.0
    gcur = g
    for j in range(k):
        half = gcur.shape[0] // 2
        if half < 1:
            break
        t_ipa += timed(lambda: C.msm(0, dense[:half], gcur[:half], cores)) * 2
        t0 = time.perf_counter()
        gcur = C.generator_collapse(0, gcur, 0x1234567890abcdef1234567890abcdef, cores)
        t_ipa += time.perf_counter() - t0
    parts["ipa_%d_rounds (2 MSM + generator collapse each, every round timed)" % k] = t_ipa
    per_proof = sum(parts.values())
    return {"value": 1.0 / per_proof, "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "C oracle (oracle/oracle.c: best_multiexp / best_fft / per-row gate evaluation / generator collapse restated from "
                      "halo2_proofs 0.2.0) on the real %sCircuit at k=%d: every stage of create_proof (SURVEY 3.1 steps 1-9) timed -- the quotient "
                      "over all extended rows, every IPA round, one representative of each commitment / transform class times its count; "
                      "not the Rust crate (no toolchain)" % (runner.kind.capitalize(), k),
            "seconds_per_proof": per_proof, "stages_s": {kk: round(v, 4) for kk, v in parts.items()}}


def ubench_peaks():
    """ALU yardsticks from the tracked microbench record (tools/ubench_field.hip run on this GPU model): the best rate of
    the XYZZ mixed addition and of the field multiplication in isolation, whichever limb form reaches it."""
    import glob
    import re
    best = {"xyzz_madd": None, "fe_mul": None, "source": None}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ubench_field_gfx950.txt")))
    if not files:
        return best
    best["source"] = os.path.relpath(files[-1], ROOT)
    for line in open(files[-1]):
        m = re.search(r"([\d.]+) Gop/s", line)
        if not m:
            continue
        v = float(m.group(1))
        if line.startswith(("xyzz_madd", "xyzz29_madd")):   # saturated and unsaturated-limb mixed addition: the better one is the yardstick
            best["xyzz_madd"] = max(best["xyzz_madd"] or 0, v)
        elif line.startswith(("fe_mul<Fp>", "fe29_mul<Fp>")):   # likewise the product (the builtin quotient kernels run the unsaturated one)
            best["fe_mul"] = max(best["fe_mul"] or 0, v)
    return best


def cpu_baseline_micro(name, curve):
    """msm24 / ntt22 on the host cores with the C oracle (oracle/oracle.c: best_multiexp / best_fft restated), on a BOUNDED sample:
    the NTT at its full 2^22, the MSM on 2^21 of the 2^24 points (bases by the oracle's point walk) -- GB/s of the sample
    (Pippenger's additions per point fall only slowly with n: c ~ ln n)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle as C
    import pasta as O
    import random
    cores = _cpu_cores()
    cid = {"vesta": 0, "pallas": 1, "bn254": 2}[curve]
    cv = O.CURVE_BY_ID[cid]
    rng = np.random.default_rng(5)

    def rnd(m):
        a = np.frombuffer(rng.bytes(m * 32), dtype=np.uint64).reshape(m, 4).copy()
        a[:, 3] &= (1 << 60) - 1
        return a
    if name == "ntt22":
        F, fid = {0: O.FP, 1: O.FQ, 2: O.BN_FR}[cid], cid      # the curve's scalar field; field ids follow the curve ids
        a = rnd(1 << 22)
        t0 = time.perf_counter()
        C.ntt(fid, a, F.omega(22), threads=cores)
        dt = time.perf_counter() - t0
        return {"value": (64 << 22) / dt / 1e9, "unit": "GB/s", "cores": cores, "kind": "port", "seconds": dt,
                "sample": "C oracle radix-2 FFT (best_fft restated), one full 2^22 forward NTT over the scalar field of %s" % curve}
    lg = 21
    n = 1 << lg
    g = cv.random_point(random.Random(1))
    bases = C.point_walk(cid, C.points_to_array([g])[0], n)
    sc = rnd(n)
    t0 = time.perf_counter()
    C.msm(cid, sc, bases, cores)
    dt = time.perf_counter() - t0
    return {"value": 96.0 * n / dt / 1e9, "unit": "GB/s", "cores": cores, "kind": "port", "seconds": dt,
            "sample": "C oracle chunked Pippenger (best_multiexp restated) on 2^%d of the 2^24 points of %s (bases: G, 2G, ... by the oracle's "
                      "point walk); 96 B per point / time" % (lg, curve)}


def cpu_baseline_verify(runner):
    """verify_proof for ONE proof of the runner's circuit on the host: the oracle's verifier (oracle/halo2_oracle.verify_proof,
    protocol logic in Python, its n-term MSM / commitments through the C oracle via oracle/accel.py) on a proof the product just
    made -- which must be accepted.  Key set-up (keygen_vk's fixed / permutation commitments) is not timed, as in
    benches/board.rs:80-86."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import accel as A
    import halo2_oracle as H
    import pasta as O
    from helpers import real_parity as R
    cores = _cpu_cores()
    g, _, w, u, _ = runner.params.points(want_lagrange=False)
    insts, proofs = runner.last_insts[0], runner.last_batch[0]
    with A.accelerated(cores):
        keys = R.oracle_keys(runner.layout.blob(), R.points_of(g), w, u, verifier_only=True)
        t0 = time.perf_counter()
        ok = H.verify_proof(keys, insts[0], proofs[0], O.Blake2bTranscript(O.FP))
        dt = time.perf_counter() - t0
    assert ok, "the oracle verifier rejected a proof of the bench"
    return {"value": 1.0 / dt, "unit": "verifications/s", "cores": cores, "kind": "port", "seconds": dt,
            "sample": "oracle verify_proof (Python protocol logic, MSM / commitments in the C oracle on %d threads) of one %sCircuit proof at k=%d "
                      "made by this run; accepted" % (cores, runner.kind.capitalize(), runner.k)}


def measure_region(wl, ctx, device, dist, steps, warmup):
    """W warmup steps, barrier, K timed steps between barriers (+ torch.cuda.synchronize on both sides), MAX over ranks;
    kernel-class timings summed over the workload's contexts"""
    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)
    for _ in range(warmup):
        wl.step()
    barrier()
    all_ctx = [ctx] + [w[1].ctx for w in getattr(wl, "workers", []) if w[0] is not None]
    all_ctx = list({id(c): c for c in all_ctx}.values())
    for c in all_ctx:
        c.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    barrier()
    elapsed = time.perf_counter() - t0
    timings = ctx.timings()
    for c in all_ctx[1:]:
        for kname, v in c.timings().items():
            for f in v:
                timings[kname][f] += v[f]
    for c in all_ctx:
        c.profile(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, timings, all_ctx


OTHER_WORKLOADS = [
    # (tag, workload, curve, batch, concurrency, steps, warmup)
    # (two warm-up steps for the proofs: the per-context workspace is merged into one block at the SECOND call of a shape)
    ("proof_k11", "proof_k11", "vesta", 128, 8, 4, 2),      # ShotCircuit at the reference's size (benches/shot.rs:22)
    ("proof_k12", "proof_k12", "vesta", 64, 4, 4, 2),       # BoardCircuit at the reference's size (benches/board.rs:22)
    ("proof_k17", "proof_k17", "vesta", 8, 4, 3, 2),        # the metric's third size
    ("verify_k14", "verify_k14", "vesta", 64, 1, 5, 1),     # benches/board.rs:80-86
    ("msm24_vesta", "msm24", "vesta", 1, 1, 5, 2),
    ("msm24_bn254", "msm24", "bn254", 1, 1, 5, 2),
    ("ntt22_vesta", "ntt22", "vesta", 1, 1, 20, 3),
    ("ntt22_bn254", "ntt22", "bn254", 1, 1, 20, 3),
]


def run_other_workloads(args, ctx, device, rank, world, dist):
    """The rest of BASELINE.json's metric, one short region each, sequentially in this process; every workload is freed before
    the next.  Every rank runs them (weak scaling; several ranks: proof sizes only, no CPU legs); a failure is recorded in the
    entry of that workload and does not take the line down -- the collectives of a region are reached either way."""
    out = {}
    for tag, name, curve, batch, conc, steps, warmup in OTHER_WORKLOADS:
        if world > 1 and not name.startswith("proof_k"):
            continue
        ent = {"workload": name, "steps": steps, "warmup": warmup}
        t_begin = time.perf_counter()
        wl, err = None, None
        try:
            wl = Workload(name, ctx, device, seed=4321 + rank, precompute=True, concurrency=conc, batch=batch, window_bits=0, circuit="auto",
                          rank=0, world=1, mix_divisor=1, dist=None, curve=curve)
            torch.cuda.synchronize(device)
        except Exception as e:  # noqa: BLE001
            err = "setup: %r" % (e,)
        ent["setup_s"] = round(time.perf_counter() - t_begin, 2)
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=device)
        if dist is not None:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # every rank runs the region or none does (its barriers are collectives)
        if int(ok.item()) == 0:
            ent["error"] = err or "another rank failed to set this workload up"
            out[tag] = ent
            if wl is not None:
                wl.close()
            continue
        try:
            elapsed, timings, all_ctx = measure_region(wl, ctx, device, dist, steps, warmup)
            units = wl.units_per_step * steps * world
            acc, nt = timings["msm_accumulate"], timings["ntt"]
            qt = timings.get("quotient", {"ms": 0.0, "launches": 0, "algorithmic_bytes": 0.0})
            ent["ms_per_step"] = elapsed / steps * 1e3
            solo = None
            if hasattr(wl, "solo_step") and len(all_ctx) > 1:      # the dominant kernel with ONE batch in flight, as the main line does
                wl.solo_step()
                torch.cuda.synchronize(device)
                ctx.profile(True)
                wl.solo_step()
                torch.cuda.synchronize(device)
                solo = ctx.timings()
                ctx.profile(False)
            if name.startswith("proof_k"):
                qname = "bzh_quotient" if wl.runner.quotient_codegen else "k_expr_vm2"
                src = solo if solo is not None else timings
                s_acc, s_q = src["msm_accumulate"], src.get("quotient", qt)
                dom, dom_name = (s_q, qname) if s_q["ms"] > s_acc["ms"] else (s_acc, "k_msm_accumulate")
                ent.update({"metric": "complete proofs per second (synthesis + create_proof), %sCircuit, k=%d" % (wl.runner.kind.capitalize(), wl.k),
                            "value": units / elapsed, "unit": "proofs/s", "batch": wl.batch, "concurrent_batches": conc,
                            "last_batches_verified": bool(wl.verify_last())})
                basis = "HIP events, ONE batch in flight (1 step after the region)" if solo is not None else "HIP events over the region"
            elif name.startswith("verify_k"):
                dom, dom_name, basis = acc, "k_msm_accumulate", "HIP events over the region"
                ent.update({"metric": "proof verifications per second, %sCircuit, k=%d" % (wl.runner.kind.capitalize(), wl.k),
                            "value": units / elapsed, "unit": "verifications/s", "batch": batch, "accepted_in_last_batch": wl.accepted})
            elif name == "msm24":
                dom, dom_name, basis = dict(acc, algorithmic_bytes=wl.alg_bytes_msm_launch * acc["launches"]), "k_msm_accumulate", "HIP events over the region"
                ent.update({"metric": "one 2^24-point MSM on %s: algorithmic GB/s of the whole call (96 B per point / call time)" % curve,
                            "value": wl.alg_bytes_step * steps / elapsed / 1e9, "unit": "GB/s",
                            "frac_of_hbm_peak": wl.alg_bytes_step * steps / elapsed / 1e9 / HBM_PEAK_GBS})
            else:
                dom, dom_name, basis = dict(nt, launches=steps, algorithmic_bytes=wl.alg_bytes_step * steps), "k_ntt_pass_wave (all passes of one NTT)", "HIP events over the region"
                ent.update({"metric": "one 2^22 NTT over the scalar field of %s: algorithmic GB/s (64 B per element / call time)" % curve,
                            "value": wl.alg_bytes_step * steps / elapsed / 1e9, "unit": "GB/s",
                            "frac_of_hbm_peak": wl.alg_bytes_step * steps / elapsed / 1e9 / HBM_PEAK_GBS})
            if dom["ms"] > 0 and dom["launches"]:
                a_ms = dom["ms"] / dom["launches"]
                a_bytes = dom["algorithmic_bytes"] / dom["launches"]
                ach = a_bytes / (a_ms * 1e-3) / 1e9
                ent["roofline"] = {"bound": "hbm", "kernel": dom_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                   "traffic": None, "algorithmic_bytes_per_launch": a_bytes, "avg_launch_ms": a_ms, "basis": basis}
            ent["kernel_ms_per_step"] = {kname: round(v["ms"] / steps, 3) for kname, v in timings.items() if v["launches"]}
            if world == 1 and not args.no_cpu_baseline:
                t_cpu = time.perf_counter()
                if name.startswith("proof_k"):
                    ent["cpu_baseline"] = cpu_baseline_proof(wl.runner)
                elif name.startswith("verify_k"):
                    ent["cpu_baseline"] = cpu_baseline_verify(wl.runner)
                else:
                    ent["cpu_baseline"] = cpu_baseline_micro(name, curve)
                ent["cpu_baseline_s"] = round(time.perf_counter() - t_cpu, 2)
        except Exception as e:  # noqa: BLE001
            ent["error"] = "%r" % (e,)
        try:
            wl.close()
        except Exception as e:  # noqa: BLE001
            ent.setdefault("error", "close: %r" % (e,))
        ent["wall_s"] = round(time.perf_counter() - t_begin, 2)
        out[tag] = ent
    return out



def dry_run(args, rank, world):
    """--dry-run: the launcher / rendezvous / record-gather path without a GPU (gloo)."""
    import torch.distributed as dist
    from bzh2.shard import gather_records
    grouped = world > 1 or args.force_dist
    if grouped:
        if "MASTER_PORT" not in os.environ:      # --force-dist outside a launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rec = torch.full((1, 8), rank, dtype=torch.uint8)
    t0 = time.perf_counter()
    got = gather_records(rec, [1] * world, dist) if grouped else rec
    elapsed = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "unit": "proofs/s", "n_gpus": world, "steps": 0, "warmup": 0,
                          "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                          "data": "none", "config": {"workload": "dry_run", "ranks_gathered": sorted(int(v) for v in got[:, 0]),
                                     "dist": "gloo process group, world %d" % world if grouped else "none (one rank)"}}))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    torch.set_num_threads(2)   # torch is plumbing here (device tensors, streams, the final gather): no CPU op pools per rank
    ProofRunner.EXPLICIT_RNG = args.explicit_rng
    ProofRunner.QUOTIENT_CODEGEN = not args.no_quotient_codegen
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.dist_backend == "nccl" and local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible); one rank per GPU" % (local_rank, torch.cuda.device_count()))
    if args.dist_backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # this rank's share of the node's host cores: witness synthesis threads, the lookup's sort helpers and the verifier's pool
    # (libbzh2.so reads BZH_HOST_THREADS per call) -- 8 ranks on one node must not each start threads for the whole machine
    cores_visible = len(os.sched_getaffinity(0))
    host_threads = max(2, cores_visible // max(world, 1))
    os.environ.setdefault("BZH_HOST_THREADS", str(host_threads))
    host_threads = int(os.environ["BZH_HOST_THREADS"])
    ProofRunner.SYNTH_THREADS = max(1, min(4, host_threads // max(args.concurrency, 1)))
    numa_node = pin_to_gpu_numa(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:     # --force-dist outside a launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    stream = torch.cuda.current_stream(device)
    ctx = bzh2.Context(local_rank, stream=stream.cuda_stream)
    wl = Workload(args.workload, ctx, device, seed=1234 + rank, precompute=not args.no_precompute, concurrency=args.concurrency, batch=args.batch,
                  window_bits=args.window_bits, circuit=args.circuit, rank=rank, world=world, mix_divisor=args.mix_divisor, dist=dist,
                  curve=args.curve)

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
        c.profile(not args.no_kernel_timers)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    if dist is not None:  # the one collective: gather every rank's commitments (fixed-stride records)
        from bzh2.shard import gather_records
        local = wl.records() if hasattr(wl, "records") else wl.result  # complete proofs: the proof records themselves
        counts = wl.record_counts() if hasattr(wl, "record_counts") else [local.shape[0]] * world
        gathered = gather_records(local, counts, dist)
        assert gathered.shape[0] == sum(counts)
        if hasattr(wl, "combine"):
            wl.combine(gathered)
    barrier()
    elapsed = time.perf_counter() - t0
    if rank == 0 and hasattr(wl, "records"):   # untimed: every gathered record decodes (bzh_record_decode: canonical public inputs)
        from bzh2.wire import BattleZipsRecord
        recs = (gathered if dist is not None else wl.records()).cpu().numpy()
        decoded = [BattleZipsRecord.from_fixed(recs[i].tobytes()) for i in range(recs.shape[0])]
        if getattr(wl, "fixed_total", False) or world == 1:
            assert len({(r.kind, r.index) for r in decoded}) == len(decoded), "duplicate (kind, index) among the gathered records"
        if args.records_out:
            np.save(args.records_out, recs)
    profiled_steps = args.steps
    if args.no_kernel_timers:   # the kernel classes' timings from two extra steps (untimed), scaled to the timed region's step count
        for c in all_ctx:
            c.profile(True)
        profiled_steps = 2
        for _ in range(profiled_steps):
            wl.step()
        barrier()
    timings = ctx.timings()
    per_stream = [{kname: v["ms"] for kname, v in timings.items()}]
    msm_adds = sum(c.msm_additions() for c in all_ctx)     # bucket additions actually made in the timed region (all contexts)
    for c in all_ctx[1:]:  # proofs in flight on their own ctx + stream: sum their kernel classes into the report
        tc = c.timings()
        per_stream.append({kname: v["ms"] for kname, v in tc.items()})
        for kname, v in tc.items():
            for f in v:
                timings[kname][f] += v[f]
    if profiled_steps != args.steps:   # totals as if measured over the timed region's steps (per-launch figures are unaffected)
        f = args.steps / profiled_steps
        msm_adds = int(msm_adds * f)
        for v in timings.values():
            v["ms"] *= f
            v["launches"] = int(round(v["launches"] * f))
            v["algorithmic_bytes"] *= f
        per_stream = [{kname: ms * f for kname, ms in d.items()} for d in per_stream]
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
        solo = ctx.timings()
        solo["msm_additions"] = ctx.msm_additions()
        ctx.profile(False)
    verified = None
    if hasattr(wl, "verify_last"):  # untimed: every proof of every thread's last batch through bzh_verify_batch
        verified = bool(wl.verify_last())
        assert verified, "a proof of the last batch does not verify"
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    line = None
    if rank == 0:
        units = wl.units_per_step * args.steps * (1 if getattr(wl, "fixed_total", False) else world)
        is_proof = args.workload.startswith(("board", "shot"))
        is_full = args.workload.startswith("proof_k")
        is_verify = args.workload.startswith("verify_k")
        is_mixed = args.workload == "mixed_board_shot"
        acc = timings["msm_accumulate"]
        nt = timings["ntt"]
        qt = timings.get("quotient", {"ms": 0.0, "launches": 0, "algorithmic_bytes": 0.0})
        peaks = ubench_peaks()
        # the quotient kernel's name in the profiles: the interpreter, or the key's program as compiled code
        qname = "bzh_quotient" if getattr(getattr(wl, "runner", None), "quotient_codegen", False) else "k_expr_vm2"
        if args.workload == "ntt22":
            dom_ms = nt["ms"] / max(nt["launches"], 1) * (nt["launches"] / max(args.steps, 1))  # all passes of one NTT
            alg = wl.alg_bytes_step
            dom_name = "k_ntt_pass (all passes of one 2^22 NTT)"
        elif (is_full or is_mixed) and qt["ms"] > acc["ms"]:
            # the gate evaluation over the extended coset is the largest kernel class of this run
            dom_ms = qt["ms"] / max(qt["launches"], 1)
            alg = qt["algorithmic_bytes"] / max(qt["launches"], 1)
            dom_name = "%s (quotient gate evaluation, one launch per batch)" % qname
            wl.alg_bytes_step = (acc["algorithmic_bytes"] + nt["algorithmic_bytes"] + qt["algorithmic_bytes"]) / max(args.steps, 1)
        elif is_full or is_verify or is_mixed:
            # MSM launches of a proof differ in size (28 column commits batched by phase, then the IPA rounds): average the
            # bytes the library counted per launch (bzh_ctx_work) over the same launches
            dom_ms = acc["ms"] / max(acc["launches"], 1)
            alg = acc["algorithmic_bytes"] / max(acc["launches"], 1)
            dom_name = "k_msm_accumulate (mean over the launches of a proof batch: commitments, opening rounds before and after the generator collapse)"
            wl.alg_bytes_step = (acc["algorithmic_bytes"] + nt["algorithmic_bytes"] + qt["algorithmic_bytes"]) / max(args.steps, 1)
        else:
            dom_ms = acc["ms"] / max(acc["launches"], 1)
            alg = wl.alg_bytes_msm_launch
            dom_name = "k_msm_accumulate"
        timed_region = {"avg_launch_ms": dom_ms, "algorithmic_bytes_per_launch": alg}
        roofline_basis = "HIP events on the launch stream over the timed region"
        if solo is not None and (is_full or is_mixed):
            # With several batches in flight every launch shares the GPU and its duration stretches (a per-launch average of the
            # timed region can even exceed the step it is in): that is queueing, not the kernel.  The roofline of the dominant kernel
            # is therefore taken from the SAME kernels with ONE batch in flight (two extra steps right after the timed region);
            # what the whole GPU achieved over the timed region is reported separately (roofline.whole_gpu).
            s_dom = solo["quotient"] if dom_name.startswith(qname) else solo["msm_accumulate"]
            if s_dom["ms"] > 0 and s_dom["launches"]:
                dom_ms = s_dom["ms"] / s_dom["launches"]
                alg = s_dom["algorithmic_bytes"] / s_dom["launches"]
                roofline_basis = ("HIP events on the launch stream, ONE batch in flight (2 steps after the timed region); "
                                  "the timed region's own per-launch average is in roofline.timed_region_average.  An event pair brackets "
                                  "the kernel AND the command processor's hand-over on either side (the start event completes when the previous "
                                  "kernel retires, the stop event after this one's end-of-kernel release): ~20-40 us per launch that rocprofv3's "
                                  "kernel-only durations (profiles/r*_kernel_stats.csv) do not contain.  Most of a batch's ~25 accumulate launches "
                                  "are the opening's short rounds, so the MEAN launch reads ~16 % longer here (r03: 1.88 ms against rocprof's "
                                  "1.62 ms); `frac` is therefore a slight under-statement, never an over-statement")
        achieved = alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM-side traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs of this same command), if one exists for this workload and kernel
        traffic, traffic_src = None, None
        want_kernel = qname if dom_name.startswith(qname) else "k_msm_accumulate"
        for tag in ("r04", "r03", "r02", "r01_h", "r01_e"):
            tj = os.path.join(ROOT, "profiles", "%s_%s_pmc_traffic.json" % (tag, args.workload))
            if traffic is not None or not os.path.exists(tj) or args.workload == "ntt22":
                continue
            try:
                tot, launches = 0.0, 0
                for e in json.load(open(tj))["kernels"]:
                    if want_kernel in e["kernel"]:
                        # coalesced 16-B-per-lane streams (the evaluator's column reads): FETCH_SIZE doubled as the guide prescribes;
                        # the accumulate kernel's 64-B table gathers: raw count (calibration note in the round-1 traffic file)
                        rd = e["read_bytes"] if want_kernel == qname else e["read_bytes_raw"]
                        tot += (rd + e["write_bytes"]) * e.get("launches", 1)   # the file holds one entry per grid size:
                        launches += e.get("launches", 1)                       # mean over the launches, like `achieved`
                if launches:
                    traffic = tot / launches
                    traffic_src = os.path.relpath(tj, ROOT) + (" (real circuit, one default-size batch in flight; mean over %d launches)" % launches
                                                               if tag in ("r04", "r03", "r02") else " (round-1 synthetic circuit: stale for the real one)")
            except Exception:
                traffic = None
        line = {
            "metric": ("%s proof MSM+NTT workloads per second" % args.workload) if is_proof else
                      (("complete proofs per second (synthesis + create_proof), %sCircuit, k=%d, IPA/Pasta" % (wl.runner.kind.capitalize(), wl.k)) if is_full
                       else ("%s runs per second" % args.workload)),
            "value": units / elapsed,
            "unit": "proof-workloads/s" if is_proof else ("proofs/s" if is_full else "runs/s"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if getattr(wl, "fixed_total", False) else "weak", "vs_baseline": None,
            "dtype": "u32x8 (255-bit modular integer, Montgomery)", "data": "synthetic",
            "config": dict({"workload": args.workload, "curve": wl.curve_desc,
                            "stages": "MSM commits + NTT/iNTT/coset-NTT of one proof; NOT included: synthesis, quotient, "
                                      "grand products, multiopen/IPA, transcript",
                            "form": "montgomery", "srs_window_table": not args.no_precompute, "parallelism": "independent proofs per GPU (dp%d)" % world,
                            "algorithmic_GB_per_step": wl.alg_bytes_step / 1e9,
                            "whole_step_GBps": wl.alg_bytes_step * args.steps / elapsed / 1e9}, **wl.desc),
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": dom_ms, "basis": roofline_basis,
                         "timed_region_average": timed_region},
            "kernel_ms": {kname: v for kname, v in timings.items() if v["launches"]},
            "kernel_timers": ("HIP event records around every kernel class inside the timed region" if not args.no_kernel_timers else
                              "none inside the timed region; kernel_ms and roofline from 2 extra untimed steps, scaled to the step count"),
        }
        line["config"]["host_threads_pinned_to_numa_node"] = numa_node
        line["config"]["host_threads_per_rank"] = {"cores_visible": cores_visible, "world": world, "budget (BZH_HOST_THREADS)": host_threads,
                                                   "prover_threads": args.concurrency, "synthesis_threads_per_prover": ProofRunner.SYNTH_THREADS,
                                                   "sort_helpers_max": min(8, host_threads), "verifier_pool_max": min(32, host_threads)}
        line["config"]["dist"] = (("%s process group, world %d%s" % (args.dist_backend, world, " (--force-dist)" if args.force_dist and world == 1 else ""))
                                  if dist is not None else "none (one rank)")
        # kernel_ms sums the host threads' contexts (their launches overlap on the GPU, so a class can exceed the timed region);
        # per stream: the mean over the contexts, i.e. what ONE stream spent in each kernel class over the timed region
        ns = max(len(per_stream), 1)
        line["kernel_ms_per_stream"] = {"streams": ns, "ms": {kname: sum(d.get(kname, 0.0) for d in per_stream) / ns
                                                              for kname in per_stream[0] if any(d.get(kname, 0.0) for d in per_stream)}}
        if (is_full or is_verify or is_mixed) and acc["ms"] > 0:
            # the bounds that actually bind these kernels: integer-multiplier rates against the same
            # operation in isolation (profiles/r*_ubench_field_gfx950.txt, read here); kernel time is summed over the host threads'
            # contexts, so overlapping batches understate the rates of the timed region
            wb = getattr(wl, "window_bits", 0) or 11
            nwin = (256 + wb - 1) // wb
            # hardware-derived multiplier bound beside the self-referential `peak`s: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T
            # lane-slots/s; a 256 x 256 -> 512-bit product + Montgomery reduction with a sparse modulus cannot avoid 64 + 32 = 96
            # v_mad_u64_u32 (32 x 32 + 64 -> 64); an XYZZ mixed addition is 8 M + 2 S = 10 such products
            lane_slots = 256 * 4 * 16 * 2.4e9
            hw = {"lane_slots_per_s": lane_slots, "v_mad_u64_u32_per_fe_mul": 96, "fe_mul_G_per_s": lane_slots / 96 / 1e9,
                  "xyzz_madd_G_per_s": lane_slots / 960 / 1e9,
                  "note": "multiplier instructions only (no carries, loads, address arithmetic): a bound no 32-bit-limb implementation reaches"}
            alu = {"peaks_source": peaks["source"], "hw_bound": hw}
            rate_a = msm_adds / (acc["ms"] * 1e-3) / 1e9
            alu["k_msm_accumulate"] = {"unit": "G mixed additions/s", "achieved": rate_a, "peak": peaks["xyzz_madd"],
                                       "frac": rate_a / peaks["xyzz_madd"] if peaks["xyzz_madd"] else None,
                                       "frac_of_hw_bound": rate_a / hw["xyzz_madd_G_per_s"],
                                       "additions_per_step": msm_adds / max(args.steps, 1), "table_rows": nwin,
                                       "note": "bucket additions counted by the kernel (non-zero window digits only)"}
            if is_full and qt["ms"] > 0:
                st = wl.runner.pks[0].quotient_stats()
                rows = (1 << (wl.k + 3)) * wl.units_per_step * args.steps
                rate = st["multiplications_per_row"] * rows / (qt["ms"] * 1e-3) / 1e9
                alu["quotient"] = {"unit": "G field multiplications/s", "achieved": rate, "peak": peaks["fe_mul"],
                                   "frac": rate / peaks["fe_mul"] if peaks["fe_mul"] else None,
                                   "frac_of_hw_bound": rate / hw["fe_mul_G_per_s"], "program": st}
            line["roofline"]["alu_equivalent"] = alu
            # what the whole GPU did over the timed region, against the two ALU yardsticks: independent of how launches overlap
            whole = {"G_mixed_additions_per_s": msm_adds / elapsed / 1e9,
                     "frac_of_xyzz_madd_peak": (msm_adds / elapsed / 1e9 / peaks["xyzz_madd"]) if peaks["xyzz_madd"] else None,
                     "algorithmic_GBps": wl.alg_bytes_step * args.steps / elapsed / 1e9,
                     "frac_of_hbm_peak": wl.alg_bytes_step * args.steps / elapsed / 1e9 / HBM_PEAK_GBS}
            if is_full and qt["ms"] > 0:
                stq = wl.runner.pk.quotient_stats()
                qm = stq["multiplications_per_row"] * (1 << (wl.k + 3)) * wl.units_per_step * args.steps / elapsed / 1e9
                whole["G_quotient_multiplications_per_s"] = qm
                whole["quotient_frac_of_fe_mul_peak"] = qm / peaks["fe_mul"] if peaks["fe_mul"] else None
                if peaks["xyzz_madd"] and peaks["fe_mul"]:
                    whole["accumulate_plus_quotient_frac_of_alu_time"] = whole["frac_of_xyzz_madd_peak"] + whole["quotient_frac_of_fe_mul_peak"]
            line["roofline"]["whole_gpu"] = whole
            line["roofline"]["other_kernels"] = {
                "k_msm_accumulate": {"avg_launch_ms": acc["ms"] / max(acc["launches"], 1),
                                     "achieved_GBps": acc["algorithmic_bytes"] / max(acc["ms"], 1e-9) / 1e6,
                                     "frac": acc["algorithmic_bytes"] / max(acc["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS},
                qname: {"avg_launch_ms": qt["ms"] / max(qt["launches"], 1),
                               "achieved_GBps": qt["algorithmic_bytes"] / max(qt["ms"], 1e-9) / 1e6,
                               "frac": qt["algorithmic_bytes"] / max(qt["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS} if qt["ms"] else None,
                "k_ntt_pass": {"achieved_GBps": nt["algorithmic_bytes"] / max(nt["ms"], 1e-9) / 1e6,
                               "frac": nt["algorithmic_bytes"] / max(nt["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS} if nt["ms"] else None}
            if solo is not None and solo["msm_accumulate"]["ms"] > 0:
                s_acc, s_q = solo["msm_accumulate"], solo.get("quotient", {"ms": 0, "launches": 0, "algorithmic_bytes": 0})
                one = {"note": "the same kernels with ONE batch in flight (2 untimed steps after the timed region); in the timed region "
                               "%d batches share the GPU and every launch stretches accordingly" % len(all_ctx),
                       "k_msm_accumulate": {"avg_launch_ms": s_acc["ms"] / max(s_acc["launches"], 1),
                                            "achieved_GBps": s_acc["algorithmic_bytes"] / s_acc["ms"] / 1e6,
                                            "frac": s_acc["algorithmic_bytes"] / s_acc["ms"] / 1e6 / HBM_PEAK_GBS,
                                            "G_mixed_additions_per_s": solo["msm_additions"] / (s_acc["ms"] * 1e-3) / 1e9,
                                            "frac_of_xyzz_madd_peak": (solo["msm_additions"] / (s_acc["ms"] * 1e-3) / 1e9 / peaks["xyzz_madd"])
                                            if peaks["xyzz_madd"] else None}}
                if s_q["ms"] > 0:
                    st = wl.runner.pks[0].quotient_stats()
                    rows = (1 << (wl.k + 3)) * wl.runner.batch * 2
                    rate = st["multiplications_per_row"] * rows / (s_q["ms"] * 1e-3) / 1e9
                    one[qname] = {"avg_launch_ms": s_q["ms"] / max(s_q["launches"], 1),
                                         "achieved_GBps": s_q["algorithmic_bytes"] / s_q["ms"] / 1e6,
                                         "frac": s_q["algorithmic_bytes"] / s_q["ms"] / 1e6 / HBM_PEAK_GBS,
                                         "G_field_multiplications_per_s": rate, "frac_of_fe_mul_peak": rate / peaks["fe_mul"] if peaks["fe_mul"] else None}
                line["roofline"]["single_batch_in_flight"] = one
        if is_mixed:
            line["metric"] = "complete proofs per second, fixed batch of BoardCircuit (k=14) : ShotCircuit (k=11) = 1 : 10, IPA/Pasta"
            line["unit"] = "proofs/s"
            line["config"]["stages"] = "witness synthesis + complete create_proof for every proof of the fixed batch, sharded over the ranks"
            line["config"]["last_batches_verified"] = verified
        if is_verify:
            line["metric"] = "proof verifications per second, %sCircuit, k=%d, IPA/Pasta" % (wl.runner.kind.capitalize(), wl.k)
            line["unit"] = "verifications/s"
            line["config"]["stages"] = ("complete verify_proof: instance commitments, transcript replay, expected h(x), multiopen "
                                        "recombination, IPA equation (one n-term MSM per proof)")
            line["config"]["accepted_in_last_batch"] = wl.accepted
            assert wl.accepted == wl.units_per_step, "a valid proof was rejected"
        if is_full:
            line["config"]["stages"] = ("Circuit::synthesize (host C++ witness generation, staged to HBM) + complete create_proof: commitments, "
                                        "lookup, permutation, vanishing, quotient, evaluations, multiopen, IPA, transcript -- all inside the timed region")
            line["config"]["srs"] = "Params::new(%d): hash_to_curve generators + g_lagrange (csrc/params.hip); witness / lookup / grand-product columns committed in the Lagrange basis" % wl.k
            line["config"]["proof_bytes"] = len(wl.runner.last_batch[0][0])
            line["config"]["last_batches_verified"] = verified
            line["config"]["proofs_in_flight_per_gpu"] = args.concurrency * getattr(wl, "batch", args.batch)
            line["config"]["concurrent_batches"] = args.concurrency
            line["config"]["batch"] = getattr(wl, "batch", args.batch)
            try:
                free_b, total_b = torch.cuda.mem_get_info(device)
                line["config"]["hbm_in_use_GB"] = round((total_b - free_b) / 1e9, 1)   # this rank's GPU, every tenant of it
            except Exception:
                pass
            line["config"]["srs_window_bits"] = wl.window_bits or "planner"
            line["config"]["distinct_proofs_in_last_batch"] = len(set(wl.runner.last_batch[0]))
        if world == 1 and not args.no_cpu_baseline:
            if is_full:
                line["cpu_baseline"] = cpu_baseline_proof(wl.runner)
            elif is_proof and args.workload != "shot_k11_batch":
                line["cpu_baseline"] = cpu_baseline(args.workload, wl.k)
            elif args.workload in ("msm24", "ntt22"):
                line["cpu_baseline"] = cpu_baseline_micro(args.workload, args.curve)
            elif is_verify:
                line["cpu_baseline"] = cpu_baseline_verify(wl.runner)
    # the rest of BASELINE.json's metric in the same line: short regions after the main one (every rank takes part)
    want_others = args.other_workloads == "all" or (args.other_workloads == "auto" and args.workload == "proof_k14" and args.batch == 64
                                                    and args.concurrency == 4 and not args.no_kernel_timers and not args.explicit_rng
                                                    and not args.no_quotient_codegen and not args.window_bits)
    others = None
    if want_others:
        wl.close()
        others = run_other_workloads(args, ctx, device, rank, world, dist)
    if rank == 0:
        if others is not None:
            line["config"]["other_workloads"] = others
            line["config"]["other_workloads_note"] = ("short regions run after the main timed region in the same process, each workload freed before "
                                                      "the next; `value` / `ms_per_step` of the line itself are the main workload's alone")
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
