#!/bin/bash
O=gpurun_out/r3z; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q -k "ungated or theta or rerank or golden or random_conf or ip_retrieval or two_stage or sharded_equals" 2>&1 | tail -4) > $O/tests.log; tail -2 $O/tests.log
timeout 600 python tools/two_stage_time.py > $O/two_stage.txt 2> $O/two_stage.err; tail -12 $O/two_stage.txt
