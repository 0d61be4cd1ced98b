mkdir -p gpurun_out/r3f; O=$PWD/gpurun_out/r3f
bash tools/ab_run.sh "base abl4 abl8" 2 > $O/ab.log 2>&1; cat gpurun_out/ab/gemm.log
DHR_DEBUG_PLAN=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open("gpurun_out/r3f/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["result_checksum"], d["phase_ms_per_step"], d["candidates_per_query"], d["roofline"]["launches"], d["roofline"]["avg_launch_ms"], d["index_device_gb"])
P
grep "main pass" $O/bench.err | tail -2
R=$PWD
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_g8_f -- $R/tools/probe/_bin/gemm_pmc 500000 6980 768 > $O/pmc_g8_f.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc_g8_t -- $R/tools/probe/_bin/gemm_pmc 500000 6980 768 > $O/pmc_g8_t.log 2>&1
timeout 200 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/pmc_g8_s -- $R/tools/probe/_bin/gemm_pmc 500000 6980 768 > $O/pmc_g8_s.log 2>&1
timeout 200 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/pmc_g8_l -- $R/tools/probe/_bin/gemm_pmc 500000 6980 768 > $O/pmc_g8_l.log 2>&1
cd $R
python3 tools/pmc_summary.py $O | tee $O/pmc_summary.txt
tail -1 $O/pmc_g8_f.log
