R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ts_trace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/t -- python $R/tools/two_stage_time.py theta0.3 > $O/run.log 2>&1
cd $R
DB=$(ls $O/t/*/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/two_stage_theta03_kernels.txt | head -14
tail -4 $O/run.log
rm -rf $O/t
