#!/bin/bash
O=gpurun_out/r3w; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q -k "ungated_batch or theta or rerank or two_stage or golden or ip_retrieval or random_conf or larger_random or score_rows or fp32_queries or negative" 2>&1 | tail -4) > $O/tests.log; tail -2 $O/tests.log
timeout 600 python tools/two_stage_time.py > $O/two_stage.txt 2> $O/two_stage.err; tail -12 $O/two_stage.txt
B="timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3"
$B > $O/b_default.json 2> $O/b_default.err
for f in default; do python3 - <<P
import json
try:
    d=json.loads(open("$O/b_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["device_resident"]["ms_per_step"], d["phase_ms_per_step"], d["candidates_per_query"], d["result_checksum"]["rows"])
except Exception as e: print("$f", "FAILED", e)
P
done
./tools/probe/_bin/lds_scatter_probe > $O/lds_scatter_probe.txt 2>&1; cat $O/lds_scatter_probe.txt
for sp in 32 64; do timeout 600 python tools/shard_sim.py --sample-period $sp 2>&1 | grep "^shards" | cut -c1-300; done
