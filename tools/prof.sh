#!/bin/bash
# Round profiles: bench JSON lines (hybrid, dense, uniform idx, fp16-gated A/B, bm25), rocprofv3 kernel trace of the default bench, and
# separate PMC passes over the bound GEMM alone through the torch-free driver (a torch process hangs under --pmc): FETCH_SIZE; TCC hit /
# miss; SQ busy / wait counters; LDS counters -- for the gated_i8 kernel (hybrid default), the fp16-gated kernel (DHR_GATED_I8=0) and the
# dense-only kernel.  Every step has its own timeout.  Outputs under gpurun_out/; copy what should be judged into profiles/.
# usage: bash tools/prof.sh r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
( while true; do echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Package Power' | tr '\n' ' ')"; sleep 0.25; done ) > $O/${TAG}_bench.smi 2>&1 &
SMI=$!
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_hybrid.json 2> $O/bench_hybrid.err
kill $SMI
timeout 400 python bench.py --workload dense --no-cpu-baseline > $O/bench_${TAG}_dense.json 2> $O/bench_dense.err
timeout 400 python bench.py --uniform-idx --no-cpu-baseline > $O/bench_${TAG}_hybrid_uniform_idx.json 2> $O/bench_uniform.err
DHR_GATED_I8=0 timeout 400 python bench.py --no-cpu-baseline > $O/bench_${TAG}_hybrid_fp16_gated.json 2> $O/bench_hybrid_fp16.err
timeout 300 python bench.py --workload bm25 --no-cpu-baseline > $O/bench_${TAG}_bm25.json 2> $O/bench_bm25.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/trace_bench.log 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA"
# g8: the default hybrid index (gated_i8); hybf16g: the same with DHR_GATED_I8=0 (fp16 gated half, int8 ungated); dense: dense-only (fp16)
for cfg in "g8 768 1" "hybf16g 768 0" "dense 0 1"; do
  set -- $cfg
  export DHR_GATED_I8=$3
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$1_f -- $R/tools/probe/_bin/gemm_pmc 500000 6980 $2 > $O/pmc_$1_f.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc_$1_t -- $R/tools/probe/_bin/gemm_pmc 500000 6980 $2 > $O/pmc_$1_t.log 2>&1
  timeout 200 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/pmc_$1_s -- $R/tools/probe/_bin/gemm_pmc 500000 6980 $2 > $O/pmc_$1_s.log 2>&1
  timeout 200 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/pmc_$1_l -- $R/tools/probe/_bin/gemm_pmc 500000 6980 $2 > $O/pmc_$1_l.log 2>&1
done
unset DHR_GATED_I8
cd $R
DB=$(ls $O/trace/*/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/${TAG}_kernel_stats.txt | head -16
python3 tools/pmc_summary.py $O > $O/${TAG}_gemm_pmc_raw.txt
cat $O/${TAG}_gemm_pmc_raw.txt
tail -1 $O/pmc_g8_f.log; tail -1 $O/pmc_hybf16g_f.log; tail -1 $O/pmc_dense_f.log
for f in hybrid dense hybrid_uniform_idx hybrid_fp16_gated bm25; do python3 - <<P
import json
try:
    d=json.loads(open("$O/bench_${TAG}_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["value"], d["result_checksum"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["achieved"], d["roofline"]["frac"])
except Exception as e: print("$f", "FAILED", e)
P
done
