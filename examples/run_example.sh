#!/bin/bash
# Build and run the stand-alone C++ client (the reference's benches/shot.rs against libbzh2.so).
# Usage (GPU box, repository root): bash examples/run_example.sh [batch] [steps]
set -e
B=${1:-16}; S=${2:-3}
OUT=${GRAFT_OUT:-gpurun_out}
mkdir -p $OUT
g++ -O2 -std=c++17 -I include examples/shot_prover.cpp -o $OUT/shot_prover -L battlezips-halo2_amd -lbzh2 -Wl,-rpath,$PWD/battlezips-halo2_amd -lpthread
$OUT/shot_prover $B $S | tee $OUT/example_cpp.json
