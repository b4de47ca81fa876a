#!/bin/bash
# One pass over everything profiles/ holds for a round; run on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r06'
# Writes gpurun_out/<tag>_*; copy what is to be kept into profiles/.  Every rocprofv3 line puts python3 itself behind `--`.
set -e -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out
rm -rf $out/traffic_fetch $out/traffic_write $out/traffic_mfma $out/prof_*
stats() {  # stats <dir> <dest>: copy the kernel_stats.csv of a rocprofv3 --stats run
  cp "$(ls -t $1/*/*kernel_stats.csv | head -1)" "$2"
}
# ---- the driver's line, the two-rank launch rehearsal (both ranks on cuda:0, gloo), the training step as the headline
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.stderr
tail -1 $out/${tag}_bench.json | cut -c1-200
DM_BENCH_REHEARSAL=1 DM_BENCH_NO_ENTRY=1 python3 bench.py --gpus 2 --steps 10 --warmup 2 --cpu-sample 0 --no-end-to-end > $out/${tag}_bench_rehearsal_gpus2.json 2>> $out/${tag}_bench.stderr
DM_BENCH_NO_ENTRY=1 python3 bench.py --leg train --steps 8 --warmup 4 --cpu-sample 0 --no-end-to-end > $out/${tag}_bench_leg_train.json 2>> $out/${tag}_bench.stderr
echo "bench done"
# ---- per-kernel times: the whole bench command (the file the line's numbers can be checked against), the three roofline
# kernels alone (bench.py reads its *_rocprof_committed figures from this one), the headline step alone, the training step
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --cpu-sample 0 > /dev/null 2>&1
stats $out/prof_bench $out/${tag}_bench_kernel_stats.csv
PROBE_ITERS=40 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_roof -- python3 tools/pmc_probe.py > /dev/null 2>&1
stats $out/prof_roof $out/${tag}_roofline_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_head -- python3 tools/headline_trace.py 20 > /dev/null 2>&1
stats $out/prof_head $out/${tag}_headline_kernel_stats.csv
TP_STEPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -- python3 tools/train_probe.py > /dev/null 2>&1
stats $out/prof_train $out/${tag}_train_step_kernel_stats.csv
DM_TRAIN_SIDE_STREAM=0 TP_STEPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_serial -- python3 tools/train_probe.py > /dev/null 2>&1
stats $out/prof_serial $out/${tag}_train_step_serial_kernel_stats.csv
# ---- the entry points as a whole (forward_train + backward, get_targets, paste + RLE): per-kernel times of tools/sync_trace.py's call
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_entry -- python3 tools/sync_trace.py > $out/${tag}_forward_train_syncs.txt 2>&1
stats $out/prof_entry $out/${tag}_forward_train_kernel_stats.csv
# ---- timelines: the training step (busy time per queue, gaps of the chain), the 100- and 16-detection inference calls
TP_STEPS=6 rocprofv3 --kernel-trace --output-format csv -d $out/prof_tl -- python3 tools/train_probe.py > /dev/null 2>&1
python3 tools/timeline.py "$(ls $out/prof_tl/*/*kernel_trace.csv | head -1)" chain > $out/${tag}_train_timeline.txt 2>&1
for nd in 100 16; do
  rm -rf $out/prof_inf
  ND=$nd rocprofv3 --kernel-trace --output-format csv -d $out/prof_inf -- python3 tools/infer100_probe.py > /dev/null 2>&1
  python3 tools/infer_timeline.py "$(ls $out/prof_inf/*/*kernel_trace.csv | head -1)" 6 order > $out/${tag}_infer${nd}_timeline.txt 2>&1
done
rm -rf $out/prof_tl $out/prof_inf
echo "traces done"
# ---- event-timed tables
python3 tools/step_shapes.py > $out/${tag}_step_shapes_solo.txt 2>&1
python3 tools/kbench.py > $out/${tag}_kbench.txt 2>&1
python3 tools/small_n.py > $out/${tag}_small_n.txt 2>&1
ROI_REPS=3 python3 tools/roi_cold.py > $out/${tag}_roi_cold.txt 2>&1
{ echo "# default (every fused launch of round 6 on)"; python3 tools/infer_bench.py;
  echo "# DM_FUSED_STAGE_HEAD=0 DM_FUSED_MERGE_TAIL=0 DM_GROUPED_SEM=0 DM_FUSED_DCN_TOUT=0: the launch sequence of round 5";
  DM_FUSED_STAGE_HEAD=0 DM_FUSED_MERGE_TAIL=0 DM_GROUPED_SEM=0 DM_FUSED_DCN_TOUT=0 python3 tools/infer_bench.py; } > $out/${tag}_infer_experiments.txt 2>&1
echo "tables done"
# ---- HBM traffic and MFMA busy of the roofline kernels: separate --pmc passes (gfx950), then the file bench.py reads
PROBE_ITERS=6 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/traffic_fetch -- python3 tools/pmc_probe.py > /dev/null 2>&1
PROBE_ITERS=6 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/traffic_write -- python3 tools/pmc_probe.py > /dev/null 2>&1
PROBE_ITERS=6 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/traffic_mfma -- python3 tools/pmc_probe.py > /dev/null 2>&1
{ python3 tools/pmc_sum.py $out/traffic_fetch; python3 tools/pmc_sum.py $out/traffic_write; python3 tools/pmc_sum.py $out/traffic_mfma; } > $out/${tag}_traffic_pmc.txt
python3 tools/make_pmc_json.py $out/traffic_fetch $out/traffic_write $out/traffic_mfma $out/${tag}_roofline_kernel_stats.csv 6 $out/${tag}_pmc_traffic.json > /dev/null
echo "pmc json done"
python3 tools/step_sensitivity.py > $out/${tag}_step_sensitivity.txt 2>&1 || echo "step_sensitivity failed (see its file)"
echo "all done"
