R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dg_ab2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for b in gemm_pmc gemm_pmc_dg2 gemm_pmc_dg8; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$b -- $R/tools/probe/_bin/$b 500000 6980 768 > $O/$b.log 2>&1
  tail -1 $O/$b.log | cut -c1-120
  python3 - <<P
import csv, glob
v=[]
for f in glob.glob("$O/$b/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm_filter" in row.get("Kernel_Name","") and row["Counter_Name"]=="FETCH_SIZE": v.append(float(row["Counter_Value"]))
print("$b FETCH_SIZE per launch (KB):", sum(v)/max(len(v),1), len(v), "launches ->", sum(v)/max(len(v),1)*1000*2/1e9, "GB fabric-side (x2 gfx950 correction)")
P
  rm -rf $O/$b
done
