#!/bin/bash
# SQ counter passes over tools/sq_probe.py (run on the GPU box): gpurun -- 'bash tools/sq_pmc.sh r03'
set -e -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/sq1 $out/sq2 $out/sq3
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/sq1 -- python3 tools/sq_probe.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/sq2 -- python3 tools/sq_probe.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $out/sq3 -- python3 tools/sq_probe.py > /dev/null 2>&1
{ python3 tools/pmc_sum.py $out/sq1 igemm; python3 tools/pmc_sum.py $out/sq2 igemm; python3 tools/pmc_sum.py $out/sq3 igemm; } > $out/${tag}_sq_pmc.txt
rm -rf $out/sq1 $out/sq2 $out/sq3
cat $out/${tag}_sq_pmc.txt
