#!/bin/bash
# PMC passes over the RoIAlign probe (one counter group per run; summaries printed)
export TMPDIR=/tmp RP_NOGRAPH=1
cd "$GRAFT_REPO_ROOT"
i=0
for grp in "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc$i -- python tools/roi_probe.py > /dev/null 2>&1
  python tools/pmc_sum.py gpurun_out/pmc$i roi_align
done
