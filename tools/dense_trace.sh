R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dense_trace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/serial -- python $R/bench.py --workload dense --steps 3 --warmup 1 --no-cpu-baseline --other-configs 0 --two-stage 0 --overlap-aux 0 > $O/serial.log 2>&1
cd $R
DB=$(ls $O/serial/*/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/dense_kernel_stats_serial.txt | head -16
tail -1 $O/serial.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['phase_ms_per_step'])"
rm -rf $O/serial
