#!/bin/bash
O=gpurun_out/r3y; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q -k "theta or rerank or golden or random_conf or larger_random or score_rows or bm25_100k" 2>&1 | tail -4) > $O/tests.log; tail -2 $O/tests.log
timeout 600 python tools/two_stage_time.py > $O/two_stage.txt 2> $O/two_stage.err; tail -12 $O/two_stage.txt
B="timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3"
$B > $O/b_default.json 2> $O/b_default.err
$B --workload bm25 > $O/b_bm25.json 2> $O/b_bm25.err
$B --workload dense --steps 5 --warmup 2 > $O/b_dense.json 2> $O/b_dense.err
for f in default bm25 dense; do python3 - <<P
import json
try:
    d=json.loads(open("$O/b_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["device_resident"]["ms_per_step"], d["phase_ms_per_step"], d["candidates_per_query"], d["result_checksum"]["rows"])
except Exception as e: print("$f", "FAILED", e)
P
done
