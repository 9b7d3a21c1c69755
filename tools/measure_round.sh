#!/bin/bash
# Round measurements on the GPU box: bench lines, rocprofv3 kernel stats (same commands), PMC traffic, into gpurun_out/final/.
# Usage (from the repo root, under gpurun): bash tools/measure_round.sh
# (No compiler runs at measurement time any more: the quotient kernels are inside libbzh2.so.)
# The one-proof-at-a-time lines run with --no-kernel-timers: the library's HIP event records around every kernel class cost
# a single proof ~1.3 ms of stream time (11.7 -> 10.4 ms at k = 14); kernel_ms / roofline then come from 2 extra untimed steps.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/proof_k14_default_bench.json 2> $O/proof_k14_default_bench.err && echo default done   # the driver's command: the whole metric in one line
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none > $O/proof_k14_default_bench_under_rocprof.json 2> $O/prof_default.err && echo prof default done
python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 5 --warmup 2 > $O/proof_k14_b64c1_bench.json 2> /dev/null && echo b64c1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b64c1 -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 5 --warmup 2 > $O/proof_k14_b64c1_bench_under_rocprof.json 2> $O/prof_b64c1.err && echo prof b64c1 done
BZH_ACC_SATURATED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b64c1_sat -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 5 --warmup 2 > $O/proof_k14_b64c1_saturated_acc_bench_under_rocprof.json 2> $O/prof_b64c1_sat.err && echo prof b64c1 saturated done
python3 $R/bench.py --no-cpu-baseline --batch 1 --concurrency 1 --no-kernel-timers --steps 10 --warmup 3 > $O/proof_k14_b1c1_bench.json 2> /dev/null && echo b1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1c1 -o d -- python3 $R/bench.py --no-cpu-baseline --batch 1 --concurrency 1 --steps 10 --warmup 3 > $O/proof_k14_b1c1_bench_under_rocprof.json 2> $O/prof_b1c1.err && echo prof b1c1 done
python3 $R/bench.py --no-cpu-baseline --workload proof_k11 --batch 1 --concurrency 1 --no-kernel-timers --steps 10 --warmup 3 > $O/proof_k11_b1c1_bench.json 2> /dev/null && echo k11 b1 done
python3 $R/bench.py --no-cpu-baseline --workload proof_k11 --batch 128 --concurrency 8 --steps 8 --warmup 2 > $O/proof_k11_b128c8_bench.json 2> /dev/null && echo k11 done
python3 $R/bench.py --no-cpu-baseline --workload proof_k12 --batch 64 --concurrency 4 --steps 8 --warmup 2 > $O/proof_k12_b64c4_bench.json 2> /dev/null && echo k12 done
python3 $R/bench.py --no-cpu-baseline --workload proof_k12 --batch 1 --concurrency 1 --no-kernel-timers --steps 10 --warmup 3 > $O/proof_k12_b1c1_bench.json 2> /dev/null && echo k12 b1 done
python3 $R/bench.py --no-cpu-baseline --workload proof_k17 --batch 8 --concurrency 4 --steps 4 --warmup 2 > $O/proof_k17_b8c4_bench.json 2> /dev/null && echo k17 done
python3 $R/bench.py --no-cpu-baseline --workload verify_k14 --batch 64 --steps 5 --warmup 2 > $O/verify_k14_b64_bench.json 2> /dev/null && echo verify done
for cv in vesta pallas bn254; do
  python3 $R/bench.py --no-cpu-baseline --workload ntt22 --curve $cv --steps 10 --warmup 2 > $O/ntt22_${cv}_bench.json 2> /dev/null && echo ntt22 $cv done
  python3 $R/bench.py --no-cpu-baseline --workload msm24 --curve $cv --steps 5 --warmup 2 > $O/msm24_${cv}_bench.json 2> /dev/null && echo msm24 $cv done
done
python3 $R/bench.py --no-cpu-baseline --workload mixed_board_shot --mix-divisor 4 --steps 3 --warmup 1 > $O/mixed_div4_bench.json 2> /dev/null && echo mixed done
python3 $R/bench.py --no-cpu-baseline --workload mixed_board_shot --steps 2 --warmup 1 > $O/mixed_full_bench.json 2> /dev/null && echo mixed full done
python3 $R/bench.py --no-cpu-baseline --workload proof_k17 --batch 1 --concurrency 1 --no-kernel-timers --steps 5 --warmup 2 > $O/proof_k17_b1c1_bench.json 2> /dev/null && echo k17 b1 done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1 && echo pmc fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 3 --warmup 1 > $O/pmc_write.log 2>&1 && echo pmc write done
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/proof_k14_pmc_traffic.json && rm -rf $O/pmc_fetch $O/pmc_write && echo pmc merged
$R/battlezips-halo2_amd/tools/ubench_field > $O/ubench_field_gfx950.txt 2>&1 && echo ubench field done
$R/battlezips-halo2_amd/tools/ubench_valu > $O/ubench_valu_gfx950.txt 2>&1 && echo ubench valu done
(cd $R && bash examples/run_example.sh 64 3 > $O/example_cpp_client.txt 2>&1; tail -2 $O/example_cpp_client.txt)
ls $O
