#!/bin/bash
# Run GPU steps one after the other on a gpurun box; ordinary failures do not stop the sequence, a step that had to be
# KILLED (its timeout) does: after a hung GPU step nothing else is started in the same call.
# usage: tools/run_steps.sh "<seconds> <outfile> <command...>" ...
for spec in "$@"; do
  secs=${spec%% *}; rest=${spec#* }; out=${rest%% *}; cmd=${rest#* }
  mkdir -p "$(dirname "$out")"
  echo "== [$secs s] $cmd > $out"
  timeout -k 10 "$secs" bash -c "$cmd" > "$out" 2>&1
  rc=$?
  tail -n 12 "$out"
  echo "== rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== step killed at its limit: stopping"; exit $rc; fi
done
