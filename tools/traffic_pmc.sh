#!/bin/bash
# HBM traffic of the two roofline kernels: FETCH_SIZE and WRITE_SIZE in separate passes (gfx950)
export TMPDIR=/tmp PROBE_ITERS=6
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/traffic_fetch gpurun_out/traffic_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic_fetch -- python tools/pmc_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic_write -- python tools/pmc_probe.py > /dev/null 2>&1
python tools/pmc_sum.py gpurun_out/traffic_fetch
python tools/pmc_sum.py gpurun_out/traffic_write
