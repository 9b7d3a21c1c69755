# same-box A/B: fused fe29 coset output vs saturated cosets + conversion pass vs saturated quotient
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in fused unfused satq; do
  unset BZH_QUOTIENT29_UNFUSED BZH_QUOTIENT_SATURATED
  if [ $v = unfused ]; then export BZH_QUOTIENT29_UNFUSED=1; fi
  if [ $v = satq ]; then export BZH_QUOTIENT_SATURATED=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 5 --warmup 2 > $O/b64c1_${v}.json 2> $O/prof_$v.err
  s=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); cp $s $O/b64c1_${v}_kernel_stats.csv; rm -rf $O/prof_$v
  python3 - <<P
import csv
rows=list(csv.DictReader(open("$O/b64c1_${v}_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
sel=[r for r in rows if any(x in r["Name"] for x in ("quotient","k_ntt_pass_wave","k_sat_to_fe29","k_msm_accumulate"))]
print("$v", [(r["Name"][:34], round(int(r["TotalDurationNs"])/7e6,2)) for r in sel], "total", round(tot/7e6,1))
P
  python3 $R/bench.py --no-cpu-baseline --other-workloads none --steps 10 --warmup 3 > $O/default_$v.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$O/default_$v.json'));print('$v default proofs/s',d['value'], 'hbm GB', d['config'].get('hbm_in_use_GB'))"
done
