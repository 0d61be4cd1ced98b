# per-ablation PMC passes over the bound GEMM (torch-free driver); usage: bash tools/clk.sh "0 1 14"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL"
P3="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM"
for a in ${1:-0 1 14}; do
  n=1
  for P in "$P1" "$P2" "$P3"; do
    DHR_GEMM_ABLATE=$a timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/pm_${a}_$n -- $R/tools/probe/_bin/gemm_pmc 500000 6980 > $R/gpurun_out/pm_${a}_$n.log 2>&1
    n=$((n+1))
  done
  echo "ABL $a done"
done
