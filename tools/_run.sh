mkdir -p gpurun_out/beir
timeout 1500 python bench.py --workload beir --no-cpu-baseline > gpurun_out/beir/bench_r02_beir_exact.jsonl 2> gpurun_out/beir/exact.err
timeout 1500 python bench.py --workload beir --pq --no-cpu-baseline > gpurun_out/beir/bench_r02_beir_pq.jsonl 2> gpurun_out/beir/pq.err
tail -n 3 gpurun_out/beir/*.err
wc -l gpurun_out/beir/*.jsonl
