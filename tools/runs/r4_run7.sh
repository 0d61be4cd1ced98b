O=gpurun_out/r4_t7; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not full_size and not config4 and not config5 and not pq" > $O/pytest_most.log 2>&1; tail -8 $O/pytest_most.log
timeout 500 python bench.py --workload dense --no-cpu-baseline > $O/bench_dense.json 2> $O/bench_dense.err; python - <<P
import json
d=json.loads(open("$O/bench_dense.json").read().strip().splitlines()[-1])
print("dense", d["ms_per_step"], d["value"], d["result_checksum"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["achieved_in_timed_region"], d["whole_job_frac_of_gemm_roofline"])
P
timeout 500 python bench.py --workload dense --dense-i8 1 --no-cpu-baseline > $O/bench_dense_i8.json 2> $O/bench_dense_i8.err; python - <<P
import json
d=json.loads(open("$O/bench_dense_i8.json").read().strip().splitlines()[-1])
print("dense_i8", d["ms_per_step"], d["value"], d["result_checksum"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["achieved"], d["roofline"]["frac"], d["whole_job_frac_of_gemm_roofline"])
P
timeout 500 python bench.py --no-cpu-baseline > $O/bench_hybrid.json 2> $O/bench_hybrid.err; python - <<P
import json
d=json.loads(open("$O/bench_hybrid.json").read().strip().splitlines()[-1])
print("hybrid", d["ms_per_step"], d["value"], d["result_checksum"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["achieved_in_timed_region"], d["whole_job_frac_of_gemm_roofline"])
P
