# one-batch-in-flight kernel stats + default throughput of the current build (quick)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04e
mkdir -p $O
python -m pytest $R/tests/test_gpu_env_paths.py $R/tests/test_gpu_ipa.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 5 --warmup 2 > $O/b64c1_under_rocprof.json 2> $O/prof.err
s=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $s $O/b64c1_kernel_stats.csv; rm -rf $O/prof
python3 - <<P
import csv
rows=list(csv.DictReader(open("$O/b64c1_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print(r["Name"][:72], r["Calls"], round(int(r["TotalDurationNs"])/7e6,2), "ms/batch")
print("total kernel ms per batch", round(tot/7e6,1))
P
python3 $R/bench.py --no-cpu-baseline --other-workloads none --steps 10 --warmup 3 > $O/default.json 2>/dev/null
python3 -c "import json;d=json.load(open('$O/default.json'));print('default proofs/s',d['value'])"
