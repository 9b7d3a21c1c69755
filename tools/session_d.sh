# PMC comparison of the accumulate kernel per launch shape, saturated vs unsaturated accumulator (--pmc only passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in u29 sat; do
  if [ $v = sat ]; then export BZH_ACC_SATURATED=1; else unset BZH_ACC_SATURATED; fi
  for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR"; do
    tag=$(echo $set | cut -c1-14 | tr ' ' '_')
    rocprofv3 --pmc $set --output-format csv -d $O/pmc_${v}_$tag -o p -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 1 --warmup 1 > $O/pmc_${v}_$tag.log 2>&1
    f=$(find $O/pmc_${v}_$tag -name "*counter_collection.csv" | head -1)
    python3 - <<P
import csv, collections
rows=list(csv.DictReader(open("$f")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in rows:
    if "k_msm_accumulate" not in r["Kernel_Name"]: continue
    key=r["Grid_Size"]
    agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(key,r["Counter_Name"])]+=1
top=sorted(agg.items(), key=lambda kv:-max(kv[1].values()))[:4]
for k,d in top:
    print("$v grid",k,"launches",max(n[(k,c)] for c in d), {c:round(x/1e6,1) for c,x in d.items()})
P
    rm -rf $O/pmc_${v}_$tag
  done
done
