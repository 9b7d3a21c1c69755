# k = 17: does a larger batch than 8 pay?  (HBM: ~21 GB of cosets per worker at 8)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --no-cpu-baseline --other-workloads none --steps 10 --warmup 3 > $O/default.json 2>/dev/null && python3 -c "import json;d=json.load(open('$O/default.json'));print('default proofs/s',d['value'], d['roofline']['alu_equivalent']['quotient']['peak'])"
for cfg in "8 4" "12 4" "16 4" "16 3" "16 2"; do
  set -- $cfg
  BZH_BENCH_BIG_BATCH=$1 timeout -k 10 400 python3 $R/bench.py --no-cpu-baseline --workload proof_k17 --batch $1 --concurrency $2 --steps 4 --warmup 2 > $O/k17_b$1c$2.json 2> $O/k17_b$1c$2.err && python3 -c "import json;d=json.load(open('$O/k17_b$1c$2.json'));print('k17 batch $1 x $2: proofs/s',round(d['value'],2),'hbm GB',d['config'].get('hbm_in_use_GB'))" || { echo "k17 $1 x $2 failed"; tail -3 $O/k17_b$1c$2.err; }
done
