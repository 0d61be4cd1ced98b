#!/bin/bash
O=gpurun_out/ab2; mkdir -p $O
timeout 900 python tools/stress.py 300 77 2>&1 | tail -2 > $O/stress.txt
timeout 900 python tools/stress_sampled.py 2>&1 | tail -3 >> $O/stress.txt
cat $O/stress.txt
bash tools/prof_r04.sh bench 2>&1 | tail -9
bash tools/prof_r04.sh trace 2>&1 | tail -32
