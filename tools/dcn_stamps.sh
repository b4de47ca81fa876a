#!/bin/bash
# builds libdynamask_hip_stamps.so (the product's sources, dcn_bwd_fused.hip with -DDM_DCN_STAMPS) under gpurun_out/ and runs tools/dcn_stamps.py
set -e
mkdir -p gpurun_out/stamps
objs=""
for f in dynamask_amd/csrc/*.hip; do
  o=gpurun_out/stamps/$(basename $f).o
  if [ "$(basename $f)" = dcn_bwd_fused.hip ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -Idynamask_amd/csrc -DDM_DCN_STAMPS -c $f -o $o
  else
    o=dynamask_amd/build/$(basename $f).o
  fi
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/stamps/libdynamask_hip_stamps.so $objs
for args in "256 64 56 2" "256 128 28 2"; do
  DYNAMASK_HIP_LIB=gpurun_out/stamps/libdynamask_hip_stamps.so python tools/dcn_stamps.py $args
done
