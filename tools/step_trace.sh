#!/bin/bash
# kernel timeline of one unsharded config-3 step (and config 2 with --dense): bash tools/step_trace.sh [--dense] -> gpurun_out/step_trace/timeline*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/step_trace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T=hybrid; [ "$1" = "--dense" ] && T=dense
timeout 500 rocprofv3 --kernel-trace -d $O/tr_$T -- python $R/tools/step_trace.py $1 > $O/run_$T.log 2>&1
cd $R
DB=$(ls $O/tr_$T/*/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/timeline.py $DB 260 > $O/timeline_$T.txt
tail -n 3 $O/timeline_$T.txt; tail -n 2 $O/run_$T.log | cut -c1-600
rm -rf $O/tr_$T
