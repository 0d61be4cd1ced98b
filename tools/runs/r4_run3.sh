O=gpurun_out/r4_t3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sharded or bench_two_ranks or config4 or config5 or staged or enqueues" > $O/pytest_sh.log 2>&1; tail -12 $O/pytest_sh.log
for sz in "5183 300" "3633 323" "8674 1406" "25657 1000"; do timeout 120 python tools/small_probe.py $sz 5 2>&1 | grep -v amdgpu.ids | tee -a $O/small.log; done
PROFILE=0 timeout 120 python tools/small_probe.py 5183 300 5 2>&1 | grep -v amdgpu.ids | tee -a $O/small.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_small -- python $GRAFT_REPO_ROOT/tools/small_probe.py 5183 300 6 > $GRAFT_REPO_ROOT/$O/trace_small.log 2>&1
cd $GRAFT_REPO_ROOT
ls $O/trace_small/*/ | head; for f in $O/trace_small/*/*hip_api_stats.csv; do head -12 $f; done
timeout 500 python bench.py --workload dense --no-cpu-baseline > $O/bench_dense.json 2> $O/bench_dense.err; python - <<P
import json
d=json.loads(open("$O/bench_dense.json").read().strip().splitlines()[-1])
print("dense", d["ms_per_step"], d["value"], d["result_checksum"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["achieved_in_timed_region"], d["whole_job_frac_of_gemm_roofline"])
P
