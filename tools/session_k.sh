# single-proof latency: the workgroup flavour of the quad reduction for up to 256 / 512 / 1024 segments
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 256 512 1024 4096; do
  export BZH_RED_WG_MAX=$v
  for w in proof_k14 proof_k11 proof_k12 proof_k17; do
    python3 $R/bench.py --workload $w --no-cpu-baseline --other-workloads none --batch 1 --concurrency 1 --no-kernel-timers --steps 40 --warmup 5 > $O/${w}_b1c1_$v.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$O/${w}_b1c1_$v.json'));print('wg_max $v $w b1c1 ms',round(d['ms_per_step'],3))"
  done
done
unset BZH_RED_WG_MAX
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_real_circuit_parity.py tests/test_gpu_msm.py -x -q -k "latency or msm" > $O/tests.log 2>&1; tail -2 $O/tests.log
