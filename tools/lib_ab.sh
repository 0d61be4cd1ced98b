#!/bin/bash
# usage (GPU box): bash tools/lib_ab.sh "name1 name2 ..." [extra bench flags] -> the A/B libraries dhr_amd/csrc/_ab/libdhr_hip_<name>.so
# (tools/ab_build.sh) on ONE box, alternating, two rounds: config-3 step in the default (overlapped) mode and with --overlap-aux 0 (every kernel alone)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do for v in $1; do
  for ov in -1 0; do
    DHR_HIP_LIB=$R/dhr_amd/csrc/_ab/libdhr_hip_$v.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --other-configs 0 --two-stage 0 --overlap-aux $ov $2 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v overlap $ov: %.2f ms/step  refine %.2f rescore %.2f select %.2f  checksum %s' % (j['ms_per_step'], j['phase_ms_per_step']['refine_ms'], j['phase_ms_per_step']['rescore_ms'], j['phase_ms_per_step']['select_ms'], j['result_checksum']['rows']))"
  done
done; done
