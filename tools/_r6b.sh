(timeout 900 python -m pytest tests -m gpu -x -q -k "golden or bound or larger_random or dense or multi_phase or odd_shapes or bm25_100k or negative or gated_image" 2>&1 | grep -E "passed|failed")
for i in 1 2; do for a in cur new; do
  export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$a.so
  d=$(timeout 300 python tools/gemm_bench.py --rows 2000000 --k 768 --iters 8 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "$a dense-only closed $d"
done; done
for i in 1 2; do for a in cur new; do
  DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$a.so timeout 300 python bench.py --workload dense --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a dense', d['ms_per_step'], d['phase_ms_per_step']['gemm_ms'], d['result_checksum']['rows'])"
done; done
for a in cur new; do
  DHR_GATED_I8=0 DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$a.so timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a hybrid fp16-gated', d['ms_per_step'], d['phase_ms_per_step']['gemm_ms'], d['result_checksum']['rows'])"
done
for c in quora nq; do for a in cur new; do
  DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$a.so timeout 300 python bench.py --workload beir --beir-only $c --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a $c', d['ms_per_step'], d['phase_ms_per_step']['gemm_ms'])"
done; done
