# bucket-wise chunk sums in unsaturated limbs: parity, then per-batch kernel times next to BZH_ACC_SATURATED=1 on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_msm.py tests/test_gpu_ipa.py tests/test_gpu_real_circuit_parity.py -x -q -k "not 17" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_env_paths.py -x -q -k "ACC_SATURATED or NO_QUAD or ACC_THREADS or COLLAPSE" > $O/tests_env.log 2>&1 || { tail -30 $O/tests_env.log; exit 1; }
tail -2 $O/tests_env.log
cd /tmp && export TMPDIR=/tmp
for v in u29 sat; do
  unset BZH_ACC_SATURATED
  if [ $v = sat ]; then export BZH_ACC_SATURATED=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 5 --warmup 2 > $O/b64c1_${v}.json 2> $O/prof_$v.err
  s=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); cp $s $O/b64c1_${v}_kernel_stats.csv; rm -rf $O/prof_$v
  python3 - <<P
import csv
rows=list(csv.DictReader(open("$O/b64c1_${v}_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
sel=[r for r in rows if any(x in r["Name"] for x in ("chunksum","k_msm_accumulate","reduce_quad","expand_rows","collapse"))]
print("$v", [(r["Name"][:34], round(int(r["TotalDurationNs"])/7e6,2)) for r in sel], "total", round(tot/7e6,1))
P
  python3 $R/bench.py --no-cpu-baseline --other-workloads none --steps 10 --warmup 3 > $O/default_$v.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$O/default_$v.json'));print('$v default proofs/s',d['value'])"
done
