#!/bin/bash
# A/B of the bound-GEMM prologue rework (division-free tile map, 3-D grid, tile constants through LDS-DMA): base = HEAD~, pro = this tree
O=gpurun_out/r3t; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -k "golden or bound or larger_random or sampled or multi_phase or odd_shapes or dense_i8 or config1_bm25_100k or negative" 2>&1 | tail -4) > $O/tests.log; tail -2 $O/tests.log
bash tools/ab_run.sh "base pro" 2 2>&1 | tail -4
for lib in base pro base pro; do
  export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$lib.so
  d=$(timeout 300 python tools/gemm_bench.py --rows 2000000 --k 768 --iters 8 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "$lib dense-only closed $d" | tee -a $O/dense.log
done
for lib in base pro base pro; do
  export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$lib.so
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/b_$lib.json 2> $O/b_$lib.err
  python3 - <<P
import json
try:
    d=json.loads(open("$O/b_$lib.json").read().strip().splitlines()[-1])
    print("$lib hybrid", d["ms_per_step"], d["phase_ms_per_step"]["gemm_ms"], d["roofline"]["frac"], d["result_checksum"]["rows"])
except Exception as e: print("$lib", "FAILED", e)
P
done
for lib in base pro; do
  export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$lib.so
  timeout 300 python bench.py --workload dense --no-cpu-baseline --steps 5 --warmup 2 > $O/d_$lib.json 2> $O/d_$lib.err
  python3 - <<P
import json
try:
    d=json.loads(open("$O/d_$lib.json").read().strip().splitlines()[-1])
    print("$lib dense", d["ms_per_step"], d["phase_ms_per_step"]["gemm_ms"], d["roofline"]["frac"], d["result_checksum"]["rows"])
except Exception as e: print("$lib", "FAILED", e)
P
done
