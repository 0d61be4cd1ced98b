O=gpurun_out/r4_t9; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_size_properties or dense" > $O/pytest_dense.log 2>&1; tail -6 $O/pytest_dense.log
timeout 500 python bench.py --workload dense --no-cpu-baseline > $O/bench_dense.json 2> $O/bench_dense.err; python - <<P
import json
d=json.loads(open("$O/bench_dense.json").read().strip().splitlines()[-1])
print("dense", d["ms_per_step"], d["value"], d["result_checksum"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["achieved_in_timed_region"], d["whole_job_frac_of_gemm_roofline"], d["dtype"][:40])
P
