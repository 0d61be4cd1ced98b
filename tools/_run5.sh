cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest.log
( while true; do echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Package Power' | tr '\n' ' ')"; sleep 0.25; done ) > $O/r06_bench.smi 2>&1 &
SMI=$!
timeout 900 python bench.py > $O/bench_r06_default_line.json 2> $O/bench_default.err
kill $SMI
bash tools/prof_r06.sh trace > $O/prof_trace.log 2>&1
bash tools/prof_r06.sh pmc > $O/prof_pmc.log 2>&1
timeout 600 python tools/two_stage_time.py > $O/r06_two_stage.txt 2>&1
timeout 400 python bench.py --workload dense --no-cpu-baseline --other-configs 0 > $O/bench_r06_dense.json 2> $O/bench_dense.err
timeout 300 python bench.py --workload bm25 --no-cpu-baseline --other-configs 0 > $O/bench_r06_bm25.json 2> $O/bench_bm25.err
timeout 400 python bench.py --overlap-aux 0 --no-cpu-baseline --other-configs 0 --two-stage 0 > $O/bench_r06_hybrid_serial.json 2> $O/bench_serial.err
timeout 400 python bench.py --data clustered --no-cpu-baseline --other-configs 0 --two-stage 0 > $O/bench_r06_hybrid_clustered.json 2> $O/bench_clustered.err
tail -3 $O/pytest.log; grep -v amdgpu.ids $O/r06_two_stage.txt | tail -8
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6e/bench_r06_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], j['ms_per_step'], j['value'], j['result_checksum'], j['phase_ms_per_step'], j['candidates_per_query'], j['roofline']['frac'])
    except Exception as e: print(f, 'FAILED', e)
P
