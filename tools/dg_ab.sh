#!/bin/bash
# usage (GPU box): bash tools/dg_ab.sh "2 4 8"   -> for A/B libraries built with -DDHR_DOC_GROUP=n (tools/ab_build.sh dg<n> "-DDHR_DOC_GROUP=<n>"):
# bound GEMM alone over 2 M rows x 6 980 queries (closed filter, 200 launches): ms per launch, package power and shader clock (rocm-smi, 0.2 s),
# then one rocprofv3 --pmc FETCH_SIZE pass (torch-free driver LINKED against that library: tools/probe/_bin/gemm_pmc_dg<n>, built like gemm_pmc with
# -L dhr_amd/csrc/_ab -l:libdhr_hip_dg<n>.so; never LD_PRELOAD an A/B library onto the default driver -- two copies in one process) -> fabric-side bytes.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/dg_ab; mkdir -p $O; cd $R
for dg in $1; do
  lib=$R/dhr_amd/csrc/_ab/libdhr_hip_dg$dg.so
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Package Power' | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g'; echo; sleep 0.2; done ) > /tmp/smi_dg$dg.log 2>&1 &
  SMI=$!
  r=$(DHR_HIP_LIB=$lib timeout 300 python tools/gemm_bench.py --synth --dlr 768 --rows 2000000 --iters 200 2>&1 | grep -a variant | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  kill $SMI
  ( cd /tmp && export TMPDIR=/tmp && drv=$R/tools/probe/_bin/gemm_pmc_dg$dg; [ $dg = 4 ] && drv=$R/tools/probe/_bin/gemm_pmc; [ -x $drv ] && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_dg$dg -- $drv 500000 6980 768 > $O/pmc_dg$dg.log 2>&1 )
  python3 - <<P
import re, csv, glob
P=[];C=[]
for l in open("/tmp/smi_dg$dg.log"):
    m=re.search(r'Power \(W\): ([\d.]+)',l); c=re.search(r'\((\d+)Mhz\)',l)
    if m and c: P.append(float(m.group(1))); C.append(int(c.group(1)))
b=[(p,c) for p,c in zip(P,C) if p>800]
v=[];
for f in glob.glob("$O/pmc_dg$dg/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm_filter" in row.get("Kernel_Name","") and row["Counter_Name"]=="FETCH_SIZE": v.append(float(row["Counter_Value"]))
fetch = (sum(v)/len(v)*1000*2/1e9) if v else float("nan")      # FETCH_SIZE in KB as profiles/r05_gemm_pmc.txt reads it; x 2 = the gfx950 correction for 16-byte-per-lane streaming reads
alg = 500000*3840/1e9
pw = sum(x[0] for x in b)/max(len(b),1); ck = sum(x[1] for x in b)/max(len(b),1)
print("DOC_GROUP $dg: $r ms/launch (2 M rows)  power mean %.0f W (%d samples)  sclk mean %.0f MHz  | fabric-side read %.2f GB per 500 k-row launch = %.1f x the %.2f GB algorithmic" % (pw, len(b), ck, fetch, fetch/alg, alg))
P
  rm -rf $O/pmc_dg$dg
done
