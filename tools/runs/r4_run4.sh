O=gpurun_out/r4_t4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config5_pq_sharded or pq_first_stage" > $O/pytest_pq.log 2>&1; tail -15 $O/pytest_pq.log
timeout 500 python bench.py --workload dense --dense-i8 1 --no-cpu-baseline > $O/bench_dense_i8.json 2> $O/bench_dense_i8.err; python - <<P
import json
try:
    d=json.loads(open("$O/bench_dense_i8.json").read().strip().splitlines()[-1])
    print("dense_i8", d["ms_per_step"], d["value"], d["result_checksum"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["achieved"], d["roofline"]["frac"], d["whole_job_frac_of_gemm_roofline"])
except Exception as e: print("dense_i8 FAILED", e); print(open("$O/bench_dense_i8.err").read()[-1500:])
P
timeout 300 python bench.py --workload beir --beir-only scifact --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_scifact.json 2> $O/bench_scifact.err; python - <<P
import json
d=json.loads(open("$O/bench_scifact.json").read().strip().splitlines()[-1])
print("scifact", d["ms_per_step"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["launches"])
P
timeout 300 python bench.py --workload beir --beir-only nfcorpus --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_nfcorpus.json 2> $O/bench_nf.err; python - <<P
import json
d=json.loads(open("$O/bench_nfcorpus.json").read().strip().splitlines()[-1])
print("nfcorpus", d["ms_per_step"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["launches"])
P
timeout 600 python tools/shard_sim.py 2>&1 | grep -v amdgpu.ids | tee $O/shard_sim_default.log
timeout 600 python tools/shard_sim.py --sample-period 64 2>&1 | grep -v amdgpu.ids | tee $O/shard_sim_p64.log
