#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -k "golden or bound or larger_random or sampled or multi_phase or overflow or zero_score or random_conf" 2>&1 | tail -3) > $O/tests.log; tail -2 $O/tests.log
GB="timeout 300 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 8"
export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_cur.so
DHR_GEMM_THR_SAVE=$PWD/$O/thr.bin $GB --open 2>&1 | grep -a variant | sed 's/^/cur open(save) /'
unset DHR_GEMM_THR_SAVE
for i in 1 2; do
for a in cur new; do
  export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$a.so
  c=$($GB 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  o=$(DHR_GEMM_THR_LOAD=$PWD/$O/thr.bin $GB 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "$a closed $c open $o" | tee -a $O/abl.log
done; done
B="timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3"
for i in 1 2; do for a in cur new; do
  DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$a.so $B > $O/$a$i.json 2> $O/$a$i.err
  python3 - <<P
import json
try:
    d=json.loads(open("$O/$a$i.json").read().strip().splitlines()[-1])
    print("$a$i", d["ms_per_step"], d["phase_ms_per_step"]["gemm_ms"], d["roofline"]["frac"], d["candidates_per_query"], d["result_checksum"]["rows"])
except Exception as e: print("$a$i", "FAILED", e)
P
done; done
