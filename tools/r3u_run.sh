#!/bin/bash
O=gpurun_out/r3u; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -k "pq or theta or rerank or two_stage or golden" 2>&1 | tail -4) > $O/tests.log; tail -2 $O/tests.log
DHR_DEBUG_FAIL=1 timeout 600 python tools/two_stage_time.py > $O/two_stage.txt 2> $O/two_stage.err; grep -v "will be redone" $O/two_stage.txt | tail -12
grep -c "will be redone" $O/two_stage.err; grep "will be redone" $O/two_stage.err | awk '{print $3, $10, $11}' | sort | uniq -c | sort -rn | head -8; grep "will be redone" $O/two_stage.err | head -5
timeout 600 python bench.py --workload beir --pq --beir-only hotpotqa --no-cpu-baseline > $O/beir_pq_hotpotqa.json 2> $O/beir_pq.err; cut -c1-600 $O/beir_pq_hotpotqa.json
