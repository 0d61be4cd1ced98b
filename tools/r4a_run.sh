#!/bin/bash
O=gpurun_out/r4a; mkdir -p $O
B="timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3"
pr() { python3 - <<P
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    print("$1", d["ms_per_step"], d["device_resident"]["ms_per_step"], d["phase_ms_per_step"], d["candidates_per_query"], d["result_checksum"]["rows"])
except Exception as e: print("$1", "FAILED", e)
P
}
for i in 1 2; do
  DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_cur.so $B > $O/cur$i.json 2> $O/cur$i.err; pr cur$i
  $B > $O/new$i.json 2> $O/new$i.err; pr new$i
done
for g in 64 128 256; do $B --max-growth $g > $O/grow$g.json 2> $O/grow$g.err; pr grow$g; done
./tools/probe/_bin/lds_scatter_probe > $O/lds_scatter_probe.txt 2>&1; cat $O/lds_scatter_probe.txt
for g in 32 128 512; do timeout 600 python tools/shard_sim.py --max-growth $g 2>&1 | grep "^shards" | cut -c1-300; done
