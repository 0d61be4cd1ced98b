#!/bin/bash
# SQ / GRBM counter passes over the bound GEMM alone (torch-free driver tools/probe/_bin/gemm_pmc), per kernel variant.
# usage: bash tools/pmc_variants.sh "3 4" [rows]   -> gpurun_out/pmcv/v<variant>_<pass>/..., summary on stdout
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ROWS=${2:-500000}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA"
mkdir -p $R/gpurun_out/pmcv
for v in ${1:-3 4}; do
  n=1
  for P in "$P1" "$P2"; do
    DHR_GEMM_VARIANT=$v timeout 150 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/pmcv/v${v}_$n -- $R/tools/probe/_bin/gemm_pmc $ROWS 6980 > $R/gpurun_out/pmcv/v${v}_$n.log 2>&1
    n=$((n+1))
  done
  tail -1 $R/gpurun_out/pmcv/v${v}_1.log
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcv
