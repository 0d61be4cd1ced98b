cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
run() { name=$1; shift; echo "=== $name: $*" >> $O/shard_sweep.txt; env "$@" timeout 300 python tools/shard_sim.py --shards 8 --cand-cap 0 --mid 1 $EXTRA 2>&1 | grep -E "^shards|begin in two|mid ranks" >> $O/shard_sweep.txt; }
EXTRA="" run default X=1
EXTRA="--sample-period 16" run sample16 X=1
EXTRA="" run mid3 DHR_MID_SHARE16=3
EXTRA="" run mid1 DHR_MID_SHARE16=1
EXTRA="--main-chunks 3" run chunks3 X=1
EXTRA="" run pre4 DHR_PRE_SHARE16=4
EXTRA="" run preone DHR_PRE_ONE=1
EXTRA="" run preone_pre4 DHR_PRE_ONE=1 DHR_PRE_SHARE16=4
cat $O/shard_sweep.txt
timeout 300 python bench.py --workload bm25 --no-cpu-baseline --other-configs 0 > $O/bench_bm25.json 2> $O/bench_bm25.err
python - <<'P'
import json
j=json.loads([l for l in open('gpurun_out/r6d/bench_bm25.json') if l.startswith('{')][-1])
print('bm25', j['ms_per_step'], j['result_checksum'], j['phase_ms_per_step'], j['candidates_per_query'])
P
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "config4_full_size" 2>&1 | grep -E "config 4|passed|failed" > $O/shard_sim_test.txt; cat $O/shard_sim_test.txt
