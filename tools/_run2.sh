set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_abi_guard.py -m gpu -x -q -s -k "two_tier or allocation_failures or full_size_properties or clustered or real_data or golden or cli_main or k_larger or merge_sorted or sharded_local_c_abi or rccl_single or staged_sharded" 2>&1 | grep -vE "^\s*$" | tail -60 > $O/pytest_new.log
bash tools/prof_r06.sh trace > $O/prof_trace.log 2>&1
bash tools/prof_r06.sh pmc > $O/prof_pmc.log 2>&1
bash tools/prof_r06.sh shard > $O/prof_shard.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/tr_shard -- python $GRAFT_REPO_ROOT/tools/begin_trace.py > $GRAFT_REPO_ROOT/$O/tr_shard.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/tr_bm25 -- python $GRAFT_REPO_ROOT/bench.py --workload bm25 --steps 2 --warmup 1 --no-cpu-baseline --other-configs 0 > $GRAFT_REPO_ROOT/$O/tr_bm25.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(ls $O/tr_shard/*/*_results.db | head -1); python tools/timeline.py $DB 130 > $O/timeline_shard.txt
DB=$(ls $O/tr_bm25/*/*_results.db | head -1); python tools/timeline.py $DB 140 > $O/timeline_bm25.txt
rm -rf $O/tr_shard $O/tr_bm25
tail -5 $O/pytest_new.log
