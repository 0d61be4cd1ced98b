#!/bin/bash
run() { timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['candidates_per_query'], d['phase_ms_per_step']['total_ms'])"; }
for i in 1 2 3; do
run --sample-period 32
run --sample-period 16
done
for i in 1 2 3; do
run --workload dense --sample-period 32
run --workload dense --sample-period 16
done
