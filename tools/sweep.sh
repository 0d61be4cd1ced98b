#!/bin/bash
# usage: tools/sweep.sh "<bench args 1>" "<bench args 2>" ...   (prints one summary line per run)
for cfg in "$@"; do
  timeout 700 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('CFG [$cfg]', d['value'], d['ms_per_step'], 'algTF', d['roofline']['achieved'], 'launches', d['roofline']['launches'], d['phase_ms_per_step'], d['candidates_per_query'], 'fb', d['sample_fallback_queries_per_step'])
"
done
