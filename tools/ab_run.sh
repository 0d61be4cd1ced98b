#!/bin/bash
# usage (on the GPU box): bash tools/ab_run.sh "<lib names under dhr_amd/csrc/_ab>" [reps]   -> closed / open filter GEMM time per library, interleaved
O=gpurun_out/ab; mkdir -p $O; : > $O/gemm.log
for i in $(seq ${2:-2}); do
for lib in $1; do
  export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$lib.so
  c=$(timeout 300 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 8 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  o=$(timeout 300 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 8 --open 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "$lib closed $c open $o" | tee -a $O/gemm.log
done; done
