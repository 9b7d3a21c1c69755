"""Rewrite DESIGN.md section 5 ("Measured") from the numbers under profiles/rNN_* (development tool: run after
tools/measure_round.sh has been copied into profiles/).  Usage: python tools/fill_design.py r03"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = os.path.join(ROOT, "profiles")


def J(name):
    return json.load(open(os.path.join(P, "%s_%s" % (TAG, name))))


def sp(v):
    return "{:,.0f}".format(v).replace(",", " ")


TEMPLATE = r'''## 5. Measured (MI355X, 1 GPU, `profiles/@TAG@_*`)

`bench.py` default = **proof_k14**: complete proofs of the reference's **real BoardCircuit** in a 2^14-row table on the
`Params::new(14)` SRS, 4 host threads × batches of 64 distinct witnesses on ONE shared key, **synthesis + create_proof inside
the timed region**, a 32-byte seed per proof expanded on the device (`bzh_prove_batch_seeded`), last batches verified
afterwards (untimed), every record of the last step encoded / decoded through `bzh_record_*`.

| configuration | proofs/s | note |
|---|---|---|
| BoardCircuit k = 14, 64 × 4 (default) | **@DEFAULT@** | `profiles/@TAG@_proof_k14_default_bench.json`; round 2: 508.  Boxes of the pool differ by ± 2 %: 602–617 over this round's runs of the final code |
| this round's steps on that configuration: round-2 code / + in-wave NTT, shared key, builtin quotient kernel / + generator collapse / + grand products shifted | 508 / 511 / 571–576 / 578–587 | `BZH_IPA_COLLAPSE=0`: 509; `BZH_NO_COMMIT_SHIFT=1`: 572 (same box, same run) |
| BoardCircuit k = 14, 64 × 1 (one batch in flight) | @B64C1@ | @B64C1MS@ ms per batch of 64, kernel time @KSUM@ ms of it (`@TAG@_proof_k14_b64c1_*`); round 2: 426 |
| BoardCircuit k = 14, one proof at a time (`--batch 1 --concurrency 1`) | @B1C1@ | **@B1C1MS@ ms per proof**, round 2: 16.1 (`@TAG@_proof_k14_b1c1_bench.json`) |
| BoardCircuit k = 12 (the reference's own size, `benches/board.rs:22`), 64 × 4 | **@K12@** | one at a time: @K12B1@ ms (12.7); round 2: 1 677 |
| BoardCircuit in a 2^17-row table (k = 17, the metric's third size), 8 × 4 | @K17@ | one at a time: @K17B1@ ms; round 2: 60 / 43 ms |
| ShotCircuit k = 11 (the reference's own size, `benches/shot.rs:20`), 128 × 8 | **@K11@** | one at a time: @K11B1@ ms (11.0); round 2: 3 971 |
| configs[3]: the fixed batch of 256 Board (k = 14) + 2 560 Shot (k = 11), one GPU | @MIXEDFULL@ | `@TAG@_mixed_full_bench.json`; a quarter of it per step: @MIXED@ |
| `verify_proof`, BoardCircuit k = 14, batches of 64 | @VERIFY@ verifications/s | `@TAG@_verify_k14_b64_bench.json` |
| C++ client `examples/shot_prover.cpp`, ShotCircuit, one thread, host buffers (PCIe-inclusive), `getrandom` seeds | @EXAMPLE@ (batch 64) | `@TAG@_example_cpp_client.txt` |
| CPU baseline (C oracle, 16 cores, whole proof) | @CPU@ | @CPUNOTE@ (this box; every stage timed in full — the quotient over all 2^17 rows, all 14 IPA rounds; 0.65–0.70 over the round's boxes) |
| microbenches (config 5), Vesta / Pallas / BN254: MSM 2^24 | @MSM24@ ms | `@TAG@_msm24_{vesta,pallas,bn254}_bench.json` |
| NTT 2^22 over Fp / Fq / BN254 Fr | @NTT22@ ms | `@TAG@_ntt22_*_bench.json`; round 2 (Fp): 0.61 |

(One-at-a-time figures: the product path, `--no-kernel-timers`.  With the bench's event records around every kernel class, as
round 2 measured its 16.1 / 12.7 / 11.0 / 43 ms: 11.6 / 8.4 / 7.6 / 33 ms.)

Kernel time per batch of 64, one stream (`profiles/@TAG@_proof_k14_b64c1_kernel_stats.csv`, 7 batches; round 2 in brackets):
@KTABLE@
With 4 batches in flight the VALU-bound kernels keep their rate and the latency-bound ones (reductions, collapse chains) fill
the gaps: the default sits at ≈ 1.23 × the one-stream rate, and every batch × concurrency setting from 32 × 8 to 128 × 3 lands
within 2 % of it (565–589).  HBM in use at the default: @HBM@ GB (§3).

**Where the time goes now.**  Per batch of 64 the two multiplier-bound kernels are nearly equal — `k_msm_accumulate` ≈ 39 ms and
the quotient 37 ms — and both sit at their instruction-issue bound: the quotient's 172 K issue slots per row account for its
36.4 ms to within 1 % (§4), the accumulate kernel runs the XYZZ mixed addition at 85 % of the rate the same addition reaches
in isolation at the same two waves per SIMD (13.6 G/s: its 212 VGPRs and 70 KB of LDS per workgroup set that occupancy), and the
NTT is at 90 % of its own count.  What is left of a proof's MSM work is 16 dense n-term sums (8 quotient pieces, the random
polynomial, f, S, one lookup product, 4 opening rounds) + the collapse: no further structure to exploit without changing the
protocol's messages.  The remaining levers are arithmetic-level (§7).

**Single proof** (BASELINE configs[0]/[1] as written): **@B1C1MS@ ms at k = 14, @K12B1@ ms at k = 12, @K11B1@ ms at k = 11** (round 2:
16.1 / 12.7 / 11.0).  `profiles/@TAG@_proof_k14_b1c1_kernel_stats.csv` (≈ 390 launches per proof; round 2: ≈ 555): a single proof is 22 MSM
launch chains (digits → accumulate → reduce → final sum → host), and the reductions and final sums are dependent chains of
≈ 20 XYZZ additions on ONE wave — a wave64 issues one VALU instruction per 4 cycles whatever its lanes hold, so a 3 700-slot
addition takes 6–8 µs.  Round 3's **latency mode**: when a launch has fewer reduction waves than the chip has SIMDs the
reduction and final-sum kernels put FOUR lanes on every addition (`xyzz_add_quad`, `csrc/curve.cuh`: each lane of a quad
computes one of the up-to-four independent products of each of the addition's four dependency levels and gets the others by
DPP quad broadcasts — 4 multiplications per lane instead of 14; `k_msm_reduce_quad[_wg]`, `k_msm_finalize_quad`), and the
planner cuts a vector into up to 16 × the minimal number of chunks: `msm_reduce` + `msm_finalize` 5.3 → 3.0 ms, accumulate
3.5 → 2.8 ms per proof.  (The first version chose the operand by `lane == 0 ? … :` chains, which the compiler turned into
a table in scratch memory — 64 scratch round trips per addition and no gain; lane masks fixed that.)  Also tried: the chunk
pre-sum + one fused reduction for single vectors (same chain length, no gain); the generator collapse (its per-lane chain of
≈ 900 additions is 4 ms: taken from batch 8 on only).  Later in the round, from the stage trace of one proof
(`BZH_PROVE_TRACE=1`): the three grand products share ONE batch inversion + product + scan (their three 0.25 ms Fermat chains
became one: −0.9 ms), the multiopen's 16 Kate divisions run as 4 step-wise launches over all query sets (−0.7 ms), the host's
field multiplication went to 64-bit limbs (8.7 µs instead of 28.7 per Jacobian → affine inversion, ≈ 45 per proof: −0.4 ms) and
the lookup's host sort to an integer / histogram sort (−0.15 ms): 13.4 → 11.6 ms with the bench's kernel-class timers on, and
**10.2 ms without them** (`--no-kernel-timers`: the library's HIP event records around every kernel class — what `kernel_ms` and the
roofline are made of — are stream commands too and cost a single proof ≈ 1.3 ms; the one-at-a-time lines above are measured
without them, `kernel_ms` there comes from two extra untimed steps; the batch lines keep them: +0.5 %).  What remains: the opening
is 5.1 ms of it —
14 rounds of ≈ 365 µs that do not shrink with the round (without the collapse every round is a full-width MSM over the fixed
table: accumulate ≈ 105 + reduce ≈ 100 + final sum ≈ 38 µs, five ≈ 11 µs dependent-launch gaps, ≈ 45 µs of host round trip) —
then the quotient + its commitment 1.8, the lookup 1.0, the grand products 1.3.  One proof is ≈ 120 × the 16-core CPU port, and
the batch is where this hardware is used: one proof every 1.6 ms.

Roofline of the dominant kernel `k_msm_accumulate` (HBM, as `north_star` asks), from the one-batch-in-flight durations
(`roofline.basis` in the bench line; the timed region's own per-launch average, stretched by the three other batches sharing
the GPU, is kept beside it as `timed_region_average`): @ALG@ MB algorithmic per launch / @ACCMS@ ms = @ACCGB@ GB/s = **@FRAC@ of
8 TB/s**; whole GPU over the timed region (`roofline.whole_gpu`): @WADD@ G mixed additions/s + @WMUL@ G quotient multiplications/s
= @WFRAC@ of the chip's issue slots at the two yardsticks.  PMC traffic (`profiles/@TAG@_proof_k14_pmc_traffic.json`, separate
FETCH_SIZE / WRITE_SIZE passes): `k_msm_accumulate` @TRAFFIC@ GB per launch ≈ @TRATIO@ × algorithmic (@TREAD@ GB fetched + @TWRITE@ GB
written; round 2: 1.74 GB, this round before the XCD-aware order: 0.73 + 0.59).  Every scalar becomes 24 digits and every digit
gathers its own 64-byte point (1 536 B against the 96 B the algorithmic count sees) from a 25 MB table; in plain grid order the
512 resident workgroups gather from eight 2 MB windows of it per XCD and every gather misses the 4 MB L2.  **XCD-aware order**
(§4): one run of (chunk, vector) pairs per XCD, one or two windows live per L2 — 61 % fewer bytes fetched at the same launch
time (the kernel is issue-bound).  What remains is the writes: one 1 024-bucket XYZZ set per 32 768-item chunk and vector,
re-read by the pre-sum.  The quotient fetches 14 × (§4).  Neither is the limiter: 117 GB/s of algorithmic
bytes, ≈ 3 TB/s at the fabric.

'''


def main():
    d = J("proof_k14_default_bench.json")
    b64 = J("proof_k14_b64c1_bench.json")
    b1 = J("proof_k14_b1c1_bench.json")
    rows = list(csv.DictReader(open(os.path.join(P, "%s_proof_k14_b64c1_kernel_stats.csv" % TAG))))
    nb = 7
    tot = sum(float(x["TotalDurationNs"]) for x in rows) / 1e6 / nb

    def per(name):
        return sum(float(x["TotalDurationNs"]) for x in rows if name in x["Name"]) / 1e6 / nb
    ktable = ("`k_msm_accumulate` %.1f ms (%.0f %%) [59.7], the quotient `bzh_quotient_…` %.1f (%.0f %%) [36.5], `k_ntt_pass_wave` + `k_ntt_pass` %.1f [13.3], "
              "the collapse `k_collapse_generators` %.1f + `k_expand_rows_shared_inverse` %.1f [—], chunk pre-sum + reductions + final sums %.1f [13.5], "
              "Kate division %.1f [6.1 in round 2: §4], staging copies `k_xfer16` %.1f; sum of all kernels %.0f ms [146]." % (
                  per("k_msm_accumulate"), 100 * per("k_msm_accumulate") / tot, per("bzh_quotient"), 100 * per("bzh_quotient") / tot, per("k_ntt_pass"),
                  per("k_collapse_generators"), per("k_expand_rows"), per("k_msm_chunksum") + per("k_msm_reduce") + per("k_msm_finalize"),
                  per("k_kate"), per("k_xfer16"), tot))
    r = d["roofline"]
    w = r["whole_gpu"]
    tr = J("proof_k14_pmc_traffic.json")
    acc = [e for e in tr["kernels"] if "k_msm_accumulate" in e["kernel"]]
    L = sum(e.get("launches", 1) for e in acc)
    T = sum((e["read_bytes_raw"] + e["write_bytes"]) * e.get("launches", 1) for e in acc) / L
    ex = [ln for ln in open(os.path.join(P, "%s_example_cpp_client.txt" % TAG)) if "proofs_per_s" in ln][-1]
    rep = {
        "@DEFAULT@": "%.0f" % d["value"], "@B64C1@": "%.0f" % b64["value"], "@B64C1MS@": "%.0f" % b64["ms_per_step"], "@KSUM@": "%.0f" % tot,
        "@B1C1@": "%.0f" % b1["value"], "@B1C1MS@": "%.1f" % b1["ms_per_step"], "@K12@": sp(J("proof_k12_b64c4_bench.json")["value"]),
        "@K12B1@": "%.1f" % J("proof_k12_b1c1_bench.json")["ms_per_step"], "@K17@": "%.0f" % J("proof_k17_b8c4_bench.json")["value"],
        "@K17B1@": "%.0f" % J("proof_k17_b1c1_bench.json")["ms_per_step"], "@K11@": sp(J("proof_k11_b128c8_bench.json")["value"]),
        "@K11B1@": "%.1f" % J("proof_k11_b1c1_bench.json")["ms_per_step"], "@MIXED@": sp(J("mixed_div4_bench.json")["value"]),
        "@MIXEDFULL@": sp(J("mixed_full_bench.json")["value"]), "@VERIFY@": sp(J("verify_k14_b64_bench.json")["value"]),
        "@EXAMPLE@": sp(json.loads(ex)["proofs_per_s"]), "@CPU@": "%.2f" % d["cpu_baseline"]["value"],
        "@MSM24@": " / ".join("%.1f" % J("msm24_%s_bench.json" % c)["ms_per_step"] for c in ("vesta", "pallas", "bn254")),
        "@NTT22@": " / ".join("%.2f" % J("ntt22_%s_bench.json" % c)["ms_per_step"] for c in ("vesta", "pallas", "bn254")),
        "@KTABLE@": ktable, "@ALG@": "%.1f" % (r["algorithmic_bytes_per_launch"] / 1e6), "@ACCMS@": "%.2f" % r["avg_launch_ms"],
        "@ACCGB@": "%.1f" % r["achieved"], "@FRAC@": "%.2f %%" % (100 * r["frac"]), "@WADD@": "%.2f" % w["G_mixed_additions_per_s"],
        "@WMUL@": "%.0f" % w["G_quotient_multiplications_per_s"], "@WFRAC@": "%.0f %%" % (100 * w["accumulate_plus_quotient_frac_of_alu_time"]),
        "@TREAD@": "%.2f" % (sum(e["read_bytes_raw"] * e.get("launches", 1) for e in acc) / L / 1e9),
        "@TWRITE@": "%.2f" % (sum(e["write_bytes"] * e.get("launches", 1) for e in acc) / L / 1e9),
        "@TRAFFIC@": "%.2f" % (T / 1e9), "@TRATIO@": "%.0f" % (T / r["algorithmic_bytes_per_launch"]), "@HBM@": "%s" % d["config"].get("hbm_in_use_GB"),
        "@TAG@": TAG,
    }
    st = d["cpu_baseline"]["stages_s"]
    q = sum(v for k, v in st.items() if k.startswith("quotient"))
    nt = sum(v for k, v in st.items() if "ntt" in k)
    cm = sum(v for k, v in st.items() if k.startswith("commit"))
    ip = sum(v for k, v in st.items() if k.startswith("ipa"))
    rep["@CPUNOTE@"] = "quotient %.2f s, NTT %.2f s, commits %.2f s, IPA %.2f s of %.1f s" % (q, nt, cm, ip, d["cpu_baseline"]["seconds_per_proof"])
    text = TEMPLATE
    for k, v in rep.items():
        text = text.replace(k, v)
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    i0, i1 = s.index("## 5. Measured"), s.index("## 6. Multi-GPU")
    open(path, "w").write(s[:i0] + text + s[i1:])
    print("DESIGN.md section 5 rewritten from profiles/%s_*" % TAG)


if __name__ == "__main__":
    main()
