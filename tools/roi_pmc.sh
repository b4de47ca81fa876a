#!/bin/bash
# Counter passes over tools/sq_probe_roi.py (the two RoIAlign forward kernels alone; run on the GPU box):
#   gpurun -- 'bash tools/roi_pmc.sh r03'
set -e -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/rq1 $out/rq2 $out/rq3 $out/rq4 $out/rq5
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/rq1 -- python3 tools/sq_probe_roi.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM --output-format csv -d $out/rq2 -- python3 tools/sq_probe_roi.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/rq3 -- python3 tools/sq_probe_roi.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/rq4 -- python3 tools/sq_probe_roi.py > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $out/rq5 -- python3 tools/sq_probe_roi.py > /dev/null 2>&1 || echo "(TCC pass failed)"
{ for d in rq1 rq2 rq3 rq4 rq5; do python3 tools/pmc_sum.py $out/$d roi_align || true; done; } > $out/${tag}_sq_pmc_roi.txt
rm -rf $out/rq1 $out/rq2 $out/rq3 $out/rq4 $out/rq5
cat $out/${tag}_sq_pmc_roi.txt
