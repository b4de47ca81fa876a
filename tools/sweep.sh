#!/bin/bash
# A/B sweeps of environment knobs over tools/infer_bench.py (graph figures only): bash tools/sweep.sh OUT "ENV1" "ENV2" ...
out=$1; shift
: > $out
for e in "$@"; do
  echo "# $e" >> $out
  env $e python3 tools/infer_bench.py 3 2>&1 | grep -v amdgpu.ids | sed -E 's/\(eager all.*//' >> $out
done
