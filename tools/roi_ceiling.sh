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
for v in 0 1 2 3 4 5 6 7; do      # one binary per variant: the ablation bits are compile-time constants
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idynamask_amd/csrc -DDM_ROI_ABLATE=$v tools/micro/roi_tile_ablate.hip -o ${bin}_$v &
  if [ $((v % 4)) = 3 ]; then wait; fi
done
wait
res=$out/${tag}_roialign_ceiling.txt
{
  echo "# RoIAlign 14x14 (roi_align_tile_kernel) ceiling measurement, $(date -u +%F) -- tools/roi_ceiling.sh"
  echo "## 1. event-timed variants (tools/micro/roi_tile_ablate.hip = the library's roi_align.hip compiled with -DDM_ROI_ABLATE=<variant>; variant 0 = the product kernel)"
  ${bin}_0 $rois
  for v in 1 2 3 4 5 6 7; do ${bin}_$v $rois | tail -1; done
  echo
  echo "## 1b. round 2's configuration (launch order, 32 channels per workgroup), and launch order with 16 channels"
  ${bin}_0 $rois | tail -1
  DM_ROI_ORDER=0 DM_ROI_CT=16 ${bin}_0 $rois | tail -1
  echo
  echo "## 1c. 7x7 (bbox extraction) full kernel"
  ROI_P=7 ${bin}_0 $rois | tail -1
} > $res 2>&1
pmc() {   # pmc <variant> <label> <env...> -- counters...
  local var=$1; shift
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  local d=$out/ceil_pmc
  rm -rf $d
  env "${envs[@]}" ROI_REPS=4 ROI_NO_RAMP=1 rocprofv3 --pmc "$@" --output-format csv -d $d -- ${bin}_$var $rois > /dev/null 2>&1 || { echo "$label: rocprofv3 failed"; return 0; }
  python3 tools/pmc_sum.py $d 2>/dev/null | sed "s/^/$label: /"
}
{
  echo
  echo "## 2. counters (sums over the launches of one run: 5 warm-up + 5 x 4 timed = 25 launches of the kernel; separate passes)"
  for v in 0 1 2 4; do
    pmc $v "abl=$v wave-state" X=0 -- SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
    pmc $v "abl=$v instructions" X=0 -- SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
  done
  pmc 0 "abl=0 FETCH_SIZE (default: XCD-aware order, 16 channels)" X=0 -- FETCH_SIZE
  pmc 0 "abl=0 WRITE_SIZE (default)" X=0 -- WRITE_SIZE
  pmc 0 "abl=0 FETCH_SIZE (round 2: launch order, 32 channels)" DM_ROI_ORDER=0 DM_ROI_CT=32 -- FETCH_SIZE
  pmc 0 "abl=0 FETCH_SIZE (launch order, 16 channels)" DM_ROI_ORDER=0 DM_ROI_CT=16 -- FETCH_SIZE
  pmc 0 "abl=0 L2 hit and miss (default)" X=0 -- TCC_HIT_sum TCC_MISS_sum
  pmc 0 "abl=0 L2 hit and miss (round 2 order)" DM_ROI_ORDER=0 DM_ROI_CT=32 -- TCC_HIT_sum TCC_MISS_sum
} >> $res 2>&1
rm -rf $out/ceil_pmc
cat $res
