# kernel-by-kernel timeline of one batch of 64 (one batch in flight): which accumulate launches are dense, which sparse
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o d -- python3 $R/bench.py --no-cpu-baseline --other-workloads none --concurrency 1 --steps 3 --warmup 2 > $O/b64c1.json 2> $O/tr.err
python3 $R/tools/trace_batch.py $O/tr k_msm_accumulate k_msm_digits bzh_quotient > $O/batch_accumulate.txt
python3 $R/tools/trace_batch.py $O/tr > $O/batch_all.txt
rm -rf $O/tr
cat $O/batch_accumulate.txt | cut -c1-110
