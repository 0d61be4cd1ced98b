#!/bin/bash
# usage: build the ablation libraries first:  for a in 16 32 4; do bash tools/ab_build.sh abl$a "-DG8_ABL=$a"; done; cp dhr_amd/csrc/libdhr_hip.so dhr_amd/csrc/_ab/libdhr_hip_abl0.so
# where the in-bench time of the int8 bound GEMM goes: closed filter vs the thresholds of a real search, with ablation builds of the epilogue
O=gpurun_out/r4d; mkdir -p $O
GB="timeout 300 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 8"
export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_abl0.so
DHR_GEMM_THR_SAVE=$PWD/$O/thr.bin $GB --open 2>&1 | grep -a variant | sed 's/^/abl0 open(save) /'
unset DHR_GEMM_THR_SAVE
for i in 1 2; do
for a in 0 16 32 4; do
  export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_abl$a.so
  c=$($GB 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  o=$(DHR_GEMM_THR_LOAD=$PWD/$O/thr.bin $GB 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "abl$a closed $c open $o" | tee -a $O/abl.log
done; done
