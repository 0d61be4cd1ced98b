#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q -s -k "pq_first_stage_beir or config4 or sharded or staged or enqueues" 2>&1 | grep -E "passed|failed|Error|error|emulated|assert" | tail -12) > $O/tests.log; cat $O/tests.log
