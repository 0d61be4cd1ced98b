#!/bin/bash
# round 3, session 3: overlap of the auxiliary kernels with the int8 GEMM (no longer at the power cap), two-stage modes with stats, 8-shard emulation
O=gpurun_out/r3s; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3"
$B > $O/b_default.json 2> $O/b_default.err
$B --overlap-aux 1 > $O/b_ov128.json 2> $O/b_ov128.err
$B --overlap-aux 1 --aux-cus 64 > $O/b_ov64.json 2> $O/b_ov64.err
$B --overlap-aux 1 --aux-cus 0 > $O/b_ov0.json 2> $O/b_ov0.err
$B --overlap-aux 1 --aux-cus 192 > $O/b_ov192.json 2> $O/b_ov192.err
$B --sample-period 16 > $O/b_sp16.json 2> $O/b_sp16.err
for f in default ov128 ov64 ov0 ov192 sp16; do python3 - <<P
import json
try:
    d=json.loads(open("$O/b_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["device_resident"]["ms_per_step"], d["phase_ms_per_step"], d["candidates_per_query"], d["result_checksum"]["rows"])
except Exception as e: print("$f", "FAILED", e)
P
done
timeout 600 python tools/two_stage_time.py > $O/two_stage.txt 2>&1; cat $O/two_stage.txt | tail -14
timeout 600 python tools/shard_sim.py > $O/shard_sim.txt 2>&1; tail -4 $O/shard_sim.txt | cut -c1-600
