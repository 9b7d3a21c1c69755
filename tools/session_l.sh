# single-proof latency: chunk split 16 / 32 / 64 x reduction flavour threshold
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "16 512" "32 512" "32 1024" "64 1024" "8 512"; do
  set -- $cfg
  export BZH_ACC_SPLIT=$1 BZH_RED_WG_MAX=$2
  for w in proof_k14 proof_k11 proof_k17; do
    python3 $R/bench.py --workload $w --no-cpu-baseline --other-workloads none --batch 1 --concurrency 1 --no-kernel-timers --steps 40 --warmup 5 > $O/${w}_b1c1_s$1_w$2.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$O/${w}_b1c1_s$1_w$2.json'));print('split $1 wg_max $2 $w b1c1 ms',round(d['ms_per_step'],3))"
  done
done
unset BZH_ACC_SPLIT BZH_RED_WG_MAX
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --batch 1 --concurrency 1 --no-kernel-timers --steps 20 --warmup 3 > $O/b1c1p.json 2> $O/prof1.err
s=$(find $O/prof1 -name "*kernel_stats.csv" | head -1); cp $s $O/b1c1_kernel_stats.csv; rm -rf $O/prof1
python3 - <<P
import csv
rows=list(csv.DictReader(open("$O/b1c1_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print('%-60s calls/proof %6.1f avg us %8.2f  ms/proof %6.3f'%(r['Name'][:60], int(r['Calls'])/23, float(r['AverageNs'])/1e3, int(r['TotalDurationNs'])/23e6))
print('total', tot/23e6)
P
