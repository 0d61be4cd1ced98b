#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
for c in quora fever fiqa nq; do
  timeout 600 python bench.py --workload beir --beir-only $c --no-cpu-baseline > $O/$c.json 2> $O/$c.err
  python3 - <<P
import json
try:
    d=json.loads(open("$O/$c.json").read().strip().splitlines()[-1])
    print("$c", d["ms_per_step"], d["phase_ms_per_step"], d["candidates_per_query"], d["sample_fallback_queries_per_step"])
except Exception as e: print("$c", "FAILED", e)
P
done
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/hyb.json 2> $O/hyb.err
python3 - <<P
import json
d=json.loads(open("$O/hyb.json").read().strip().splitlines()[-1])
print("hybrid", d["ms_per_step"], d["phase_ms_per_step"], d["candidates_per_query"], d["sample_fallback_queries_per_step"], d["result_checksum"]["rows"])
P
timeout 600 python tools/two_stage_time.py 2>/dev/null | tail -12
