#!/bin/bash
# usage (GPU box): bash tools/dense_clk.sh -> package power / shader clock while the bound GEMM of a DENSE-ONLY int8 index (config 2's) runs alone
rocm-smi --showmaxpower --showpowercap 2>/dev/null | grep -i -E "power|cap" | head -4
for g in 0 1; do
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Package Power' | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g'; echo; sleep 0.2; done ) > /tmp/smi_d$g.log 2>&1 &
  SMI=$!
  r=$(DHR_DENSE_G8=$g timeout 300 python tools/gemm_bench.py --k 768 --dlr 0 --rows 4000000 --iters 200 2>&1 | grep -a variant | tail -1)
  kill $SMI
  python3 - <<P
import re
P=[];C=[]
for l in open("/tmp/smi_d$g.log"):
    m=re.search(r'Power \(W\): ([\d.]+)',l); c=re.search(r'\((\d+)Mhz\)',l)
    if m and c: P.append(float(m.group(1))); C.append(int(c.group(1)))
b=[(p,c) for p,c in zip(P,C) if p>800]
print("DHR_DENSE_G8=$g: $r")
if b: print("   busy samples %d: power mean %.0f W max %.0f, sclk mean %.0f MHz min %d max %d" % (len(b), sum(x[0] for x in b)/len(b), max(x[0] for x in b), sum(x[1] for x in b)/len(b), min(x[1] for x in b), max(x[1] for x in b)))
P
done
