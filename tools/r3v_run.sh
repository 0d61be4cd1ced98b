#!/bin/bash
O=gpurun_out/r3v; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -k "ungated_batch or theta or rerank or two_stage or golden or ip_retrieval" 2>&1 | tail -4) > $O/tests.log; tail -2 $O/tests.log
timeout 600 python tools/two_stage_time.py > $O/two_stage.txt 2> $O/two_stage.err; tail -12 $O/two_stage.txt
B="timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3"
$B > $O/b_default.json 2> $O/b_default.err
$B --main-chunks 12 > $O/b_mc12.json 2> $O/b_mc12.err
$B --main-chunks 16 > $O/b_mc16.json 2> $O/b_mc16.err
$B --sample-period 64 > $O/b_sp64.json 2> $O/b_sp64.err
for f in default mc12 mc16 sp64; do python3 - <<P
import json
try:
    d=json.loads(open("$O/b_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["device_resident"]["ms_per_step"], d["phase_ms_per_step"], d["candidates_per_query"], d["result_checksum"]["rows"])
except Exception as e: print("$f", "FAILED", e)
P
done
