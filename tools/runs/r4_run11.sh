O=gpurun_out/r4_t11; mkdir -p $O
timeout 1500 python tools/stress.py 1000 404 > $O/stress_small.log 2>&1; tail -3 $O/stress_small.log
timeout 2400 python tools/stress_sampled.py 160 505 > $O/stress_sampled.log 2>&1; tail -3 $O/stress_sampled.log
