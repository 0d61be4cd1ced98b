O=gpurun_out/r7c; mkdir -p $O
GB="timeout 300 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 8"
export DHR_GATED_I8=1
export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_cur.so
DHR_GEMM_THR_SAVE=$PWD/$O/thr.bin $GB --open 2>&1 | grep -a variant | sed 's/^/cur open(save) /'
for i in 1 2; do for a in cur new; do
  export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$a.so
  c=$($GB 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  o=$(DHR_GEMM_THR_LOAD=$PWD/$O/thr.bin $GB 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "$a closed $c open $o"
done; done
unset DHR_GATED_I8
for i in 1 2; do for a in cur new; do
  DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$a.so timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', d['ms_per_step'], d['phase_ms_per_step']['gemm_ms'], d['result_checksum']['rows'])"
done; done
(timeout 600 python -m pytest tests -m gpu -x -q -k "golden or bound or odd_shapes or larger_random or k_larger" 2>&1 | grep -E "passed|failed")
