#!/bin/bash
# usage (GPU box): bash tools/lib_ab4.sh "name1 name2" [extra bench flags] -> like tools/lib_ab.sh, default mode only, FOUR alternating rounds + the mean per library
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2 3 4; do for v in $1; do
  DHR_HIP_LIB=$R/dhr_amd/csrc/_ab/libdhr_hip_$v.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --other-configs 0 --two-stage 0 $2 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v %.2f ms/step  refine %.2f rescore %.2f  checksum %s' % (j['ms_per_step'], j['phase_ms_per_step']['refine_ms'], j['phase_ms_per_step']['rescore_ms'], j['result_checksum']['rows']))"
done; done | tee /tmp/_ab4.txt
python3 - <<'P'
import collections
d = collections.defaultdict(list)
for l in open('/tmp/_ab4.txt'):
    p = l.split()
    d[p[0]].append(float(p[1]))
for k, v in d.items():
    print('%s: mean %.2f ms over %d runs (min %.2f max %.2f)' % (k, sum(v) / len(v), len(v), min(v), max(v)))
P
