#!/bin/bash
# SQ counter passes over tools/scatter_probe.py: LDS instruction counts, bank conflicts and busy cycles of the col2im scatter
# and of the fused DCN data gradient (per-kernel means per launch)   gpurun -- 'bash tools/scatter_pmc.sh > gpurun_out/r04_scatter_pmc.txt'
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/spmc$i
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/spmc$i -- python3 tools/scatter_probe.py > /dev/null 2>&1
  python3 tools/pmc_sum.py gpurun_out/spmc$i dcn_
  rm -rf gpurun_out/spmc$i
done
