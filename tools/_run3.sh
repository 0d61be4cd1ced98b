cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.log
timeout 600 python bench.py --quick > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest.log
python - <<'P'
import json
j=json.loads([l for l in open('gpurun_out/r6c/bench.json') if l.startswith('{')][-1])
print(j['ms_per_step'], j['result_checksum'], j['candidates_per_query'], j['phase_ms_per_step'], j['roofline']['kernel'][:40])
print(j['two_stage'])
for k,v in j['other_configs'].items(): print(k, v['ms_per_step'], v['result_checksum'], v['phase_ms_per_step'], v['candidates_per_query'], v['roofline']['kernel'][:50])
P
