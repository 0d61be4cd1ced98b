O=gpurun_out/r4_t10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "g8_bound_layout or golden or larger_random or random_configurations or bound_never or odd_shapes or query_chunking" > $O/pytest_g8.log 2>&1; tail -6 $O/pytest_g8.log
for i in 1 2; do
for pm in 0 1; do
  c=$(DHR_G8_PARTIAL=$pm timeout 300 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 8 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  o=$(DHR_G8_PARTIAL=$pm timeout 300 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 8 --open 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "partial=$pm closed $c open $o" | tee -a $O/gemm_ab.log
done; done
for pm in 0 1 0 1; do
DHR_G8_PARTIAL=$pm timeout 500 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > $O/bench_p$pm.json 2> $O/bench_p$pm.err; python - <<P
import json
d=json.loads(open("$O/bench_p$pm.json").read().strip().splitlines()[-1])
print("hybrid partial=$pm", d["ms_per_step"], d["value"], d["result_checksum"], d["phase_ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms_kernel_alone"], d["whole_job_frac_of_gemm_roofline"])
P
done
