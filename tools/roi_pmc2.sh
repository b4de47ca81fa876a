#!/bin/bash
# More PMC passes over the RoIAlign probe: texture-address / L1 / L2 busy and stall counters
# (each pass under its own timeout; results appended to gpurun_out/roi_pmc2.txt)
export TMPDIR=/tmp RP_NOGRAPH=1
cd "$GRAFT_REPO_ROOT"
i=10
# (the TA_* / TD_* counter groups hang rocprofv3 on this pool -- each burnt its 100 s timeout and ended in a SIGKILL of a
#  profiler mid-collection: they are not run.  DM_ROI_PMC_TA=1 adds them back for a pool where they work.)
groups=("TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum" \
        "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
        "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_READ_sum" \
        "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES")
if [ "${DM_ROI_PMC_TA:-0}" = "1" ]; then
  groups+=("TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum")
fi
for grp in "${groups[@]}"; do
  i=$((i+1))
  echo "== $grp" >> gpurun_out/roi_pmc2.txt
  timeout -k 5 100 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc$i -- python3 tools/roi_probe.py > /dev/null 2>&1 || echo "   (pass failed or timed out: rc $?)" >> gpurun_out/roi_pmc2.txt
  python3 tools/pmc_sum.py gpurun_out/pmc$i roi_align >> gpurun_out/roi_pmc2.txt 2>&1
  echo "pass $i done"
done
