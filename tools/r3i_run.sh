mkdir -p gpurun_out/r3i; O=$PWD/gpurun_out/r3i
bash tools/ab_run.sh "base abl16 abl32" 2 > $O/ab.log 2>&1; cat gpurun_out/ab/gemm.log
run() { tag=$1; shift; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err
python - <<P
import json
d=json.loads(open("gpurun_out/r3i/bench_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"], d["result_checksum"]["rows"] % 100000, d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["launches"], d["roofline"]["avg_launch_ms"])
P
}
run s16 --sample-period 16
run s64 --sample-period 64
run m16 --main-chunks 16
run s16m16 --sample-period 16 --main-chunks 16
