#!/bin/bash
# RoIAlign 14x14 ceiling evidence in one GPU call (writes gpurun_out/<tag>_roialign_ceiling.txt; copy to profiles/):
#   gpurun --timeout 900 -- 'bash tools/roi_ceiling.sh r03'
# 1. tools/micro/roi_tile_ablate: the product kernel and its seven phase ablations, event-timed;
# 2. rocprofv3 --pmc passes (counters only, no tracing) over the full kernel and the three single ablations:
#    wave-state cycles, instruction mix, FETCH_SIZE, WRITE_SIZE; the XCD-aware workgroup order beside the plain one.
set -e -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out
bin=$out/roi_tile_ablate
rois=tools/micro/rois_512_1333x800.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idynamask_amd/csrc tools/micro/roi_tile_ablate.hip -o $bin
res=$out/${tag}_roialign_ceiling.txt
{
  echo "# RoIAlign 14x14 (roi_align_tile_kernel) ceiling measurement, $(date -u +%F) -- tools/roi_ceiling.sh"
  echo "## 1. event-timed variants (tools/micro/roi_tile_ablate.hip = the library's roi_align.hip + DM_ROI_ABLATE)"
  $bin $rois
  echo
  echo "## 1b. round 2's configuration (launch order, 32 channels per workgroup), and launch order with 16 channels"
  DM_ROI_ORDER=0 DM_ROI_CT=32 ROI_ABL_ONLY=0 $bin $rois | tail -1
  DM_ROI_ORDER=0 DM_ROI_CT=16 ROI_ABL_ONLY=0 $bin $rois | tail -1
  echo
  echo "## 1c. 7x7 (bbox extraction) full kernel"
  ROI_P=7 ROI_ABL_ONLY=0 $bin $rois | tail -1
} > $res 2>&1
pmc() {   # pmc <label> <env...> -- counters...
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  local d=$out/ceil_pmc
  rm -rf $d
  env "${envs[@]}" ROI_REPS=4 rocprofv3 --pmc "$@" --output-format csv -d $d -- $bin $rois > /dev/null 2>&1 || { echo "$label: rocprofv3 failed"; return 0; }
  python3 tools/pmc_sum.py $d 2>/dev/null | sed "s/^/$label: /"
}
{
  echo
  echo "## 2. counters (sums over the launches of one run: 5 warm-up + 5 x 4 timed = 25 launches of the kernel; separate passes)"
  for v in 0 1 2 4; do
    pmc "abl=$v wave-state" ROI_ABL_ONLY=$v -- SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
    pmc "abl=$v instructions" ROI_ABL_ONLY=$v -- SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
  done
  pmc "abl=0 FETCH_SIZE (default: XCD-aware order, 16 channels)" ROI_ABL_ONLY=0 -- FETCH_SIZE
  pmc "abl=0 WRITE_SIZE (default)" ROI_ABL_ONLY=0 -- WRITE_SIZE
  pmc "abl=0 FETCH_SIZE (round 2: launch order, 32 channels)" ROI_ABL_ONLY=0 DM_ROI_ORDER=0 DM_ROI_CT=32 -- FETCH_SIZE
  pmc "abl=0 FETCH_SIZE (launch order, 16 channels)" ROI_ABL_ONLY=0 DM_ROI_ORDER=0 DM_ROI_CT=16 -- FETCH_SIZE
  pmc "abl=0 L2 hit and miss (default)" ROI_ABL_ONLY=0 -- TCC_HIT_sum TCC_MISS_sum
  pmc "abl=0 L2 hit and miss (round 2 order)" ROI_ABL_ONLY=0 DM_ROI_ORDER=0 DM_ROI_CT=32 -- TCC_HIT_sum TCC_MISS_sum
} >> $res 2>&1
rm -rf $out/ceil_pmc
cat $res
