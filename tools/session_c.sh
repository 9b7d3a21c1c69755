# fe29 accumulate: parity tests, then A/B of one-batch-in-flight kernel stats (saturated vs unsaturated accumulator)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c
mkdir -p $O
python -m pytest tests/test_gpu_msm.py tests/test_gpu_env_paths.py tests/test_gpu_ipa.py tests/test_gpu_config_sizes.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for v in u29 sat; do
  if [ $v = sat ]; then export BZH_ACC_SATURATED=1; else unset BZH_ACC_SATURATED; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b64c1_$v -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 5 --warmup 2 > $O/b64c1_${v}_under_rocprof.json 2> $O/prof_$v.err
  s=$(find $O/prof_b64c1_$v -name "*kernel_stats.csv" | head -1)
  cp $s $O/b64c1_${v}_kernel_stats.csv
  rm -rf $O/prof_b64c1_$v
  python3 - <<P
import csv
rows=list(csv.DictReader(open("$O/b64c1_${v}_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:6]:
    print("$v", r["Name"][:70], r["Calls"], round(int(r["TotalDurationNs"])/7e6,2), "ms/batch", r["Percentage"])
print("$v total kernel ms per batch", round(tot/7e6,1))
P
  python3 $R/bench.py --no-cpu-baseline --other-workloads none --steps 10 --warmup 3 > $O/default_$v.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$O/default_$v.json'));print('$v default proofs/s',d['value'])"
done
