#!/bin/bash
# Round profiles: bench JSON lines (hybrid, dense, uniform idx), rocprofv3 kernel trace of the default bench, and separate
# PMC passes (FETCH_SIZE; TCC hit/miss) over the bound GEMM alone through the torch-free driver (a torch process hangs under
# --pmc).  Every step has its own timeout.  Outputs under gpurun_out/; copy what should be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
mkdir -p $R/gpurun_out
cd $R
timeout 400 python bench.py > gpurun_out/bench_${TAG}_hybrid.json 2> gpurun_out/bench_${TAG}_hybrid.err
timeout 400 python bench.py --workload dense > gpurun_out/bench_${TAG}_dense.json 2> gpurun_out/bench_${TAG}_dense.err
timeout 400 python bench.py --uniform-idx --no-cpu-baseline > gpurun_out/bench_${TAG}_hybrid_uniform_idx.json 2> gpurun_out/bench_${TAG}_uniform.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_f -- $R/tools/probe/_bin/gemm_pmc 500000 6980 > $R/gpurun_out/pmc_${TAG}_f.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_t -- $R/tools/probe/_bin/gemm_pmc 500000 6980 > $R/gpurun_out/pmc_${TAG}_t.log 2>&1
cd $R
DB=$(ls gpurun_out/prof_${TAG}/*/*_results.db | head -1)
python tools/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats.txt | head -12
tail -1 gpurun_out/pmc_${TAG}_f.log
