#!/bin/bash
# Round-6 profiles (same recipe as tools/prof.sh, plus what round 4 added): bench JSON lines of every configuration, rocprofv3 kernel traces of
# the default bench (refine / rescoring beside the next chunk's GEMM) AND of the serial mode (--overlap-aux 0: the kernel alone), separate PMC
# passes over the bound GEMM through the torch-free driver.  Every step has its own timeout.  Outputs under gpurun_out/prof_r04; copy what
# should be judged into profiles/.   usage: bash tools/prof_r06.sh [part]   part = bench | trace | pmc | beir | cpu32 | all
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=r06
PART=${1:-all}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
summ() { python3 - <<P
import json
for f in "$@".split():
    try:
        d=json.loads(open("$O/bench_${TAG}_%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["result_checksum"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("frac_kernel_alone"), d["whole_job_frac_of_gemm_roofline"], d.get("index_device_gb"))
    except Exception as e: print(f, "FAILED", e)
P
}
if [ $PART = bench ] || [ $PART = all ]; then
  ( while true; do echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Package Power' | tr '\n' ' ')"; sleep 0.25; done ) > $O/${TAG}_bench.smi 2>&1 &
  SMI=$!
  timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_hybrid.json 2> $O/bench_hybrid.err
  kill $SMI
  timeout 400 python bench.py --workload dense --no-cpu-baseline --other-configs 0 > $O/bench_${TAG}_dense.json 2> $O/bench_dense.err
  timeout 400 python bench.py --workload dense --dense-i8 0 --no-cpu-baseline --other-configs 0 > $O/bench_${TAG}_dense_fp16.json 2> $O/bench_dense_fp16.err
  timeout 400 python bench.py --uniform-idx --no-cpu-baseline --other-configs 0 > $O/bench_${TAG}_hybrid_uniform_idx.json 2> $O/bench_uniform.err
  DHR_GATED_I8=0 timeout 400 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_${TAG}_hybrid_fp16_gated.json 2> $O/bench_hybrid_fp16.err
  timeout 400 python bench.py --overlap-aux 0 --no-cpu-baseline --other-configs 0 > $O/bench_${TAG}_hybrid_serial.json 2> $O/bench_hybrid_serial.err
  timeout 300 python bench.py --workload bm25 --no-cpu-baseline --other-configs 0 > $O/bench_${TAG}_bm25.json 2> $O/bench_bm25.err
  timeout 400 python bench.py --data clustered --no-cpu-baseline --other-configs 0 > $O/bench_${TAG}_hybrid_clustered.json 2> $O/bench_clustered.err
  summ "hybrid dense dense_fp16 hybrid_uniform_idx hybrid_fp16_gated hybrid_serial bm25 hybrid_clustered"
fi
if [ $PART = trace ] || [ $PART = all ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-configs 0 --two-stage 0 > $O/trace_bench.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_serial -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-configs 0 --two-stage 0 --overlap-aux 0 > $O/trace_serial_bench.log 2>&1
  cd $R
  DB=$(ls $O/trace/*/*_results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/${TAG}_kernel_stats.txt | head -14
  DB=$(ls $O/trace_serial/*/*_results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/${TAG}_kernel_stats_serial.txt | head -14
  tail -1 $O/trace_bench.log | cut -c1-400; tail -1 $O/trace_serial_bench.log | cut -c1-400
  rm -rf $O/trace $O/trace_serial          # the databases are large: gpurun copies back at most 64 MiB
fi
if [ $PART = pmc ] || [ $PART = all ]; then
  cd /tmp && export TMPDIR=/tmp
  SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
  SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA"
  for cfg in "g8 768 1 -1 0" "dense 0 1 0 0" "densei8 0 1 1 0"; do
    set -- $cfg
    export DHR_GATED_I8=$3 DHR_DENSE_I8=$4
    timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$1_f -- $R/tools/probe/_bin/gemm_pmc 500000 6980 $2 > $O/pmc_$1_f.log 2>&1
    timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc_$1_t -- $R/tools/probe/_bin/gemm_pmc 500000 6980 $2 > $O/pmc_$1_t.log 2>&1
    timeout 200 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/pmc_$1_s -- $R/tools/probe/_bin/gemm_pmc 500000 6980 $2 > $O/pmc_$1_s.log 2>&1
    timeout 200 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/pmc_$1_l -- $R/tools/probe/_bin/gemm_pmc 500000 6980 $2 > $O/pmc_$1_l.log 2>&1
  done
  unset DHR_GATED_I8 DHR_DENSE_I8
  cd $R
  python3 tools/pmc_summary.py $O > $O/${TAG}_gemm_pmc_raw.txt
  cat $O/${TAG}_gemm_pmc_raw.txt
  for d in $O/pmc_*_?; do [ -d $d ] && rm -rf $d; done
  tail -1 $O/pmc_g8_f.log; tail -1 $O/pmc_dense_f.log; tail -1 $O/pmc_densei8_f.log
fi
if [ $PART = beir ] || [ $PART = all ]; then
  timeout 1500 python bench.py --workload beir --no-cpu-baseline --steps 10 --warmup 3 --per-step > $O/bench_${TAG}_beir_exact.jsonl 2> $O/bench_beir_exact.err
  [ "${BEIR_PQ:-0}" = 1 ] && timeout 2400 python bench.py --workload beir --pq --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_${TAG}_beir_pq.jsonl 2> $O/bench_beir_pq.err
  python3 - <<P
import json
for f in ("exact", "pq"):
    try:
        for l in open("$O/bench_${TAG}_beir_%s.jsonl" % f):
            d = json.loads(l)
            print(f, d["config"]["workload"][:34], d["ms_per_step"], d["phase_ms_per_step"].get("gemm_ms"), d["candidates_per_query"], d["roofline"]["achieved"], d["roofline"]["frac"])
    except Exception as e: print(f, "FAILED", e)
P
fi
if [ $PART = cpu32 ] || [ $PART = all ]; then
  timeout 1200 python bench.py --steps 5 --warmup 2 --other-configs 0 --cpu-queries 32 --cpu-queries-1t 32 > $O/bench_${TAG}_hybrid_cpu32.json 2> $O/bench_cpu32.err
  python3 -c "
import json
d=json.loads(open('$O/bench_${TAG}_hybrid_cpu32.json').read().strip().splitlines()[-1]); print(json.dumps(d['cpu_baseline'])[:1500])"
  timeout 600 python tools/two_stage_time.py > $O/${TAG}_two_stage.txt 2>&1; grep -v amdgpu.ids $O/${TAG}_two_stage.txt | tail -12
fi

if [ $PART = shard ] || [ $PART = all ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "config4_full_size" 2>&1 | grep -E "config 4|passed|failed" > $O/${TAG}_shard_sim.txt
  for ns in 2 4 8; do echo "--- tools/shard_sim.py --shards $ns --cand-cap 0 --mid 1 (two-round begin, second agreement)" >> $O/${TAG}_shard_sim.txt; timeout 900 python tools/shard_sim.py --shards $ns --cand-cap 0 --mid 1 2>&1 | grep -E "begin in two|^shards|mid ==" >> $O/${TAG}_shard_sim.txt; done
  cat $O/${TAG}_shard_sim.txt
fi
if [ $PART = power ] || [ $PART = all ]; then
  bash tools/clk_ab.sh > $O/${TAG}_power.txt 2>&1; grep -E "persist|cap|Power" $O/${TAG}_power.txt | head -12
fi
