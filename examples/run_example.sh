#!/bin/bash
# Build and run the stand-alone C++ client against libbzh2.so, then check its proofs against the ctypes path.
# Usage (GPU box, repository root): bash examples/run_example.sh [k] [batch] [threads] [steps]
set -e
K=${1:-11}; B=${2:-16}; T=${3:-4}; S=${4:-5}
OUT=${GRAFT_OUT:-gpurun_out}
mkdir -p $OUT
python examples/export_case.py $OUT/case.bin $K $B
g++ -O2 -std=c++17 -I include examples/prove_batch.cpp -o $OUT/prove_batch -L battlezips-halo2_amd -lbzh2 -Wl,-rpath,$PWD/battlezips-halo2_amd -lpthread
$OUT/prove_batch $OUT/case.bin $T $S | tee $OUT/example_cpp.json
python examples/check_case.py $OUT/case.bin | tee $OUT/example_py.json
python - <<PY
import json
a = json.load(open("$OUT/example_cpp.json")); b = json.load(open("$OUT/example_py.json"))
assert a["fnv1a"] == b["fnv1a"] and a["verified"] == $B and b["verified"] == $B, (a, b)
print("C++ client and ctypes binding produced identical proofs; all verified")
PY
