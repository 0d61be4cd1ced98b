#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
timeout 900 python tools/stress.py 500 31 > $O/stress.txt 2>&1; tail -2 $O/stress.txt
DHR_DENSE_I8=1 timeout 600 python tools/stress.py 250 32 > $O/stress_i8.txt 2>&1; tail -2 $O/stress_i8.txt
timeout 1200 python tools/stress_sampled.py 80 33 > $O/stress_sampled.txt 2>&1; tail -3 $O/stress_sampled.txt
