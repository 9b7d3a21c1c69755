set -x
./battlezips-halo2_amd/tools/ubench_field > gpurun_out/r04_ubench_field_fe29.txt 2>&1
grep -E "29|xyzz_madd|fe_mul<Fp> dep" gpurun_out/r04_ubench_field_fe29.txt
python -m pytest tests/test_gpu_msm.py tests/test_gpu_bench_ranks.py -x -q -k "walk or rccl" > gpurun_out/r04_new_tests_b.log 2>&1; tail -5 gpurun_out/r04_new_tests_b.log
(time python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err); tail -c 800 gpurun_out/r04_bench_default.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r04_bench_default.json"))
print(d["value"], d["ms_per_step"])
for k,v in d["config"].get("other_workloads",{}).items():
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("value","unit","ms_per_step","setup_s","wall_s","cpu_baseline_s","error")}, v.get("roofline",{}).get("frac"), (v.get("cpu_baseline") or {}).get("value"))
P
