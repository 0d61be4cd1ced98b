#!/bin/bash
# Round profiles: bench JSON lines (hybrid, dense, uniform idx), rocprofv3 kernel trace of the default bench,
# and a separate PMC pass (FETCH_SIZE / WRITE_SIZE) over the bound GEMM alone.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01d}
mkdir -p $R/gpurun_out
cd $R
python bench.py > gpurun_out/bench_${TAG}_hybrid.json 2> gpurun_out/bench_${TAG}_hybrid.err
python bench.py --workload dense > gpurun_out/bench_${TAG}_dense.json 2> gpurun_out/bench_${TAG}_dense.err
python bench.py --uniform-idx --no-cpu-baseline > gpurun_out/bench_${TAG}_hybrid_uniform_idx.json 2> gpurun_out/bench_${TAG}_uniform.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG} -- python $R/tools/gemm_bench.py --dlr 768 --k 1536 --idx-buckets 2 --iters 3 > $R/gpurun_out/pmc_${TAG}.log 2>&1
cd $R
DB=$(ls gpurun_out/prof_${TAG}/*/*_results.db | head -1)
python tools/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats.txt | head -12
ls gpurun_out/pmc_${TAG}/*/ | head
tail -2 gpurun_out/pmc_${TAG}.log
