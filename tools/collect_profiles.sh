#!/bin/bash
# One pass over everything profiles/ holds for a round; run on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r01l'
# Writes gpurun_out/<tag>_*; copy what is to be kept into profiles/.
set -e -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/traffic_fetch $out/traffic_write $out/traffic_mfma $out/prof_bench $out/prof_roof $out/prof_head $out/prof_train
stats() {  # stats <dir> <dest>: copy the kernel_stats.csv of a rocprofv3 --stats run
  cp "$(ls -t $1/*/*kernel_stats.csv | head -1)" "$2"
}
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.stderr
tail -1 $out/${tag}_bench.json | cut -c1-200
# the N > 1 launch path rehearsed on the one GPU (both ranks on cuda:0, gloo), and the training step as the headline
DM_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 10 --warmup 2 --cpu-sample 0 > $out/${tag}_bench_rehearsal_gpus2.json 2>> $out/${tag}_bench.stderr
python3 bench.py --leg train --steps 8 --warmup 4 --cpu-sample 0 > $out/${tag}_bench_leg_train.json 2>> $out/${tag}_bench.stderr
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --cpu-sample 0 > /dev/null 2>&1
stats $out/prof_bench $out/${tag}_bench_kernel_stats.csv
PROBE_ITERS=40 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_roof -- python3 tools/pmc_probe.py > /dev/null 2>&1
stats $out/prof_roof $out/${tag}_roofline_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_head -- python3 tools/headline_trace.py 20 > /dev/null 2>&1
stats $out/prof_head $out/${tag}_headline_kernel_stats.csv
TP_STEPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -- python3 tools/train_probe.py > /dev/null 2>&1
stats $out/prof_train $out/${tag}_train_step_kernel_stats.csv
# the same step on ONE stream (no side streams): per-kernel times without co-running kernels
rm -rf $out/prof_serial
DM_TRAIN_SIDE_STREAM=0 TP_STEPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_serial -- python3 tools/train_probe.py > /dev/null 2>&1
stats $out/prof_serial $out/${tag}_train_step_serial_kernel_stats.csv
# training-step timeline: busy time per queue, what runs alone, the gaps of the chain's queue
rm -rf $out/prof_tl
TP_STEPS=6 rocprofv3 --kernel-trace --output-format csv -d $out/prof_tl -- python3 tools/train_probe.py > /dev/null 2>&1
python3 tools/timeline.py "$(ls $out/prof_tl/*/*kernel_trace.csv | head -1)" chain > $out/${tag}_train_timeline.txt 2>&1
rm -rf $out/prof_tl
python3 tools/step_shapes.py > $out/${tag}_step_shapes_solo.txt 2>&1
# the 100-detection inference call (bucketed HIP-graph replay): launches, union busy time, kernel families per replay
rm -rf $out/prof_inf
rocprofv3 --kernel-trace --output-format csv -d $out/prof_inf -- python3 tools/infer100_probe.py > /dev/null 2>&1
python3 tools/infer_timeline.py "$(ls $out/prof_inf/*/*kernel_trace.csv | head -1)" 6 > $out/${tag}_infer100_timeline.txt 2>&1
rm -rf $out/prof_inf
echo "traces done"
python3 tools/kbench.py > $out/${tag}_kbench.txt 2>&1
python3 tools/tail_probe.py > $out/${tag}_tail_probe.txt 2>&1
PROBE_ITERS=6 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/traffic_fetch -- python3 tools/pmc_probe.py > /dev/null 2>&1
PROBE_ITERS=6 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/traffic_write -- python3 tools/pmc_probe.py > /dev/null 2>&1
PROBE_ITERS=6 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/traffic_mfma -- python3 tools/pmc_probe.py > /dev/null 2>&1
{ python3 tools/pmc_sum.py $out/traffic_fetch; python3 tools/pmc_sum.py $out/traffic_write; python3 tools/pmc_sum.py $out/traffic_mfma; } > $out/${tag}_traffic_pmc.txt
cat $out/${tag}_traffic_pmc.txt
# the file bench.py reads roofline.traffic from: regenerated with every collection (copy to profiles/pmc_traffic.json)
python3 tools/make_pmc_json.py $out/traffic_fetch $out/traffic_write $out/traffic_mfma $out/${tag}_roofline_kernel_stats.csv 6 $out/${tag}_pmc_traffic.json > /dev/null
echo "pmc json done"
hipcc --offload-arch=gfx950 -O3 -w tools/micro/mfma_mix.hip -o /tmp/mfma_mix 2> /dev/null && /tmp/mfma_mix > $out/${tag}_mfma_mix.txt
python3 tools/dcn_offsets_exp.py > $out/${tag}_dcn_offsets_exp.txt 2>&1
python3 tools/small_n.py > $out/${tag}_small_n.txt 2>&1
python3 tools/col2im_exp.py > $out/${tag}_col2im_exp.txt 2>&1
python3 tools/op_probe.py wgradcat wgrad > $out/${tag}_wgrad_probe.txt 2>&1
ROI_REPS=3 python3 tools/roi_cold.py > $out/${tag}_roi_cold.txt 2>&1
python3 tools/infer_streams_exp.py > $out/${tag}_infer_streams_exp.txt 2>&1
echo "all done"
