cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
cd $R; tail -1 gpurun_out/prof_bench.log | cut -c1-400
python tools/rocpd_summary.py gpurun_out/prof_r01c 2>&1 | head -40
