cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest.log
timeout 1200 python tools/stress.py 400 61 > $O/stress.log 2>&1; tail -3 $O/stress.log
timeout 1500 python tools/stress_sampled.py 100 62 > $O/stress_sampled.log 2>&1; tail -3 $O/stress_sampled.log
head -5 $O/pytest.log
