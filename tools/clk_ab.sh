#!/bin/bash
# usage (GPU box): bash tools/clk_ab.sh  -> package power / sclk while the bound GEMM runs alone (closed and open filter); the power cap of the board first.
# PERSIST="0 1" (with DHR_HIP_LIB pointing at a DHR_AB_VARIANTS=1 build, tools/ab): one workgroup per tile against persistent workgroups, as in rounds 4-5
export DHR_GATED_I8=1
rocm-smi --showmaxpower --showpowercap 2>/dev/null | grep -i -E "power|cap" | head -6
for pz in ${PERSIST:-0}; do for o in "" "--open"; do
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Package Power' | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g'; echo; sleep 0.2; done ) > /tmp/smi_$pz$o.log 2>&1 &
  SMI=$!
  r=$(DHR_G8_PERSIST=$pz timeout 300 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 200 $o 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  kill $SMI
  python3 - <<P
import re
P=[];C=[]
for l in open("/tmp/smi_$pz$o.log"):
    m=re.search(r'Power \(W\): ([\d.]+)',l); c=re.search(r'\((\d+)Mhz\)',l)
    if m and c: P.append(float(m.group(1))); C.append(int(c.group(1)))
b=[(p,c) for p,c in zip(P,C) if p>800]
if b:
    print("persist $pz $o: $r ms/launch; busy samples %d: power mean %.0f W max %.0f, sclk mean %.0f MHz min %d max %d" % (len(b), sum(x[0] for x in b)/len(b), max(x[0] for x in b), sum(x[1] for x in b)/len(b), min(x[1] for x in b), max(x[1] for x in b)))
else: print("persist $pz $o: $r ms/launch; no busy samples", len(P))
P
done; done
