mkdir -p gpurun_out/r3j; O=$PWD/gpurun_out/r3j
timeout 600 python -m pytest tests -m gpu -x -q -k "g8_bound_layout or gip_retrieval_golden or multi_phase_and_overflow or larger_random or sampled_threshold_fallbacks" 2>&1 | tail -3
timeout 120 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 8 2>&1 | tail -1
timeout 160 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 8 --open 2>&1 | tail -1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open("gpurun_out/r3j/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["result_checksum"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["launches"], d["roofline"]["avg_launch_ms"])
P
