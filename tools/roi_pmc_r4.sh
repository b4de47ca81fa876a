#!/bin/bash
# Round 4: fabric traffic of the RoIAlign 14x14 variants (FETCH_SIZE x 2 per the gfx950 correction, WRITE_SIZE) against
# their time: is the launch bound by what crosses the L2 boundary?   gpurun -- 'bash tools/roi_pmc_r4.sh r04'
set -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out
res=$out/${tag}_roi_traffic.txt
: > $res
run() {   # label, env...
  label=$1; shift
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf $out/rqx

    ( export "$@" PROBE_ITERS=10; timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d $out/rqx -- python3 tools/roi_pmc_probe.py > /dev/null 2>&1 )
    echo "## $label [$c]" >> $res
    python3 tools/pmc_sum.py $out/rqx roi_ >> $res 2>&1 || true
  done
  rm -rf $out/rqx
}
run "tile kernel (round 3 + waitcnt fix), 16 ch, XCD order, RoIs as given" DM_ROI_PERSIST=0
run "tile kernel, 16 ch, XCD order, RoIs sorted (level, y, x)" ROI_SORT=1
run "tile kernel, 32 ch, XCD order, as given" DM_ROI_CT=32
run "tile kernel, 32 ch, XCD order, sorted" DM_ROI_CT=32 ROI_SORT=1
run "tile kernel, 32 ch, launch order, sorted" DM_ROI_CT=32 DM_ROI_ORDER=0 ROI_SORT=1
run "plan + persistent kernel, 32 ch, as given" DM_ROI_PERSIST=1
cat $res
