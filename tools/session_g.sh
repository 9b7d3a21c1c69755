# quad reductions in unsaturated limbs: parity first, then same-box A/B against BZH_ACC_SATURATED=1 (single-proof latency + per-batch reduce kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_ipa.py tests/test_gpu_real_circuit_parity.py tests/test_gpu_prover.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_env_paths.py -x -q -k "ACC_SATURATED or NO_QUAD or MSM_GS" > $O/tests_env.log 2>&1 || { tail -30 $O/tests_env.log; exit 1; }
tail -2 $O/tests_env.log
cd /tmp && export TMPDIR=/tmp
for v in u29 sat; do
  unset BZH_ACC_SATURATED
  if [ $v = sat ]; then export BZH_ACC_SATURATED=1; fi
  for w in proof_k14 proof_k11 proof_k17; do
    python3 $R/bench.py --workload $w --no-cpu-baseline --other-workloads none --batch 1 --concurrency 1 --no-kernel-timers --steps 40 --warmup 5 > $O/${w}_b1c1_$v.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$O/${w}_b1c1_$v.json'));print('$v $w b1c1 ms',round(d['ms_per_step'],3))"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 5 --warmup 2 > $O/b64c1_${v}.json 2> $O/prof_$v.err
  s=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); cp $s $O/b64c1_${v}_kernel_stats.csv; rm -rf $O/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1_$v -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --batch 1 --concurrency 1 --no-kernel-timers --steps 20 --warmup 3 > $O/b1c1p_${v}.json 2> $O/prof1_$v.err
  s=$(find $O/prof1_$v -name "*kernel_stats.csv" | head -1); cp $s $O/b1c1_${v}_kernel_stats.csv; rm -rf $O/prof1_$v
  python3 - <<P
import csv
for f,div in (("$O/b64c1_${v}_kernel_stats.csv",7e6),("$O/b1c1_${v}_kernel_stats.csv",23e6)):
    rows=list(csv.DictReader(open(f)))
    tot=sum(int(r["TotalDurationNs"]) for r in rows)
    sel=[r for r in rows if any(x in r["Name"] for x in ("reduce","finalize","k_msm_accumulate","chunksum","final"))]
    print("$v", f.split('/')[-1], [(r["Name"].split('<')[0].replace('void bzh::',''), r["Name"].count('true'), round(int(r["TotalDurationNs"])/div,3)) for r in sel], "total", round(tot/div,2))
P
  python3 $R/bench.py --no-cpu-baseline --other-workloads none --steps 10 --warmup 3 > $O/default_$v.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$O/default_$v.json'));print('$v default proofs/s',d['value'])"
done
