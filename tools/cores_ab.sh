#!/bin/bash
# usage (GPU box): bash tools/cores_ab.sh "<lib:prio> ..."  -> config-3 bench step per A/B library (dhr_amd/csrc/_ab) and DHR_AUX_PRIO setting
O=gpurun_out/cores; mkdir -p $O
for cfg in $1; do
  lib=${cfg%%:*}; prio=${cfg##*:}
  export DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_$lib.so DHR_AUX_PRIO=$prio
  timeout 400 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --other-configs 0 $EXTRA > $O/bench_${lib}_p$prio.json 2> $O/bench_${lib}_p$prio.err
  python3 - <<P
import json
try:
    d=json.loads(open("$O/bench_${lib}_p$prio.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("$lib prio $prio: %.2f ms/step  gemm in-region %.1f alone %.1f  phases %s  cand %s  chk %s" % (d["ms_per_step"], d["phase_ms_per_step"]["gemm_ms"], r["avg_launch_ms_kernel_alone"]*r["launches"]/d["steps"], {k:v for k,v in d["phase_ms_per_step"].items() if k!="gemm_ms"}, d["candidates_per_query"], d["result_checksum"]["rows"]))
except Exception as e: print("$lib prio $prio FAILED", e); print(open("$O/bench_${lib}_p$prio.err").read()[-800:])
P
done
