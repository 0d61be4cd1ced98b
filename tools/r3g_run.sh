mkdir -p gpurun_out/r3g; O=$PWD/gpurun_out/r3g
run() { tag=$1; shift; timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err
python - <<P
import json
d=json.loads(open("gpurun_out/r3g/bench_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"], d["result_checksum"]["rows"] % 100000, d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["launches"], d["roofline"]["avg_launch_ms"])
P
}
run base
run ov1 --overlap-aux 1
run ov1c64 --overlap-aux 1 --aux-cus 64
run ov1c192 --overlap-aux 1 --aux-cus 192
run ov1m4 --overlap-aux 1 --main-chunks 12
run base2
