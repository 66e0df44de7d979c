#!/bin/bash
# Copy what tools/prof_round.sh left under gpurun_out/<tag>/ into profiles/<prefix>_* (the files the docs cite) and rebuild
# the PMC records keyed to the current kernel sources.  usage: bash tools/collect_round_profiles.sh <tag> <prefix, e.g. r03>
tag=$1; pre=$2
g=gpurun_out/$tag
set -e
tail -1 $g/bench.json > profiles/${pre}_bench_default.json
tail -1 $g/bench_inflight1.json > profiles/${pre}_bench_inflight1_nostress.json
cp $g/stats/bench_kernel_stats.csv profiles/${pre}_kernel_stats_bench_default.csv
cp $g/stats1/bench_kernel_stats.csv profiles/${pre}_kernel_stats_bench_inflight1_nostress.csv
cp $g/step_timeline.txt profiles/${pre}_step_timeline_profiled.txt
cp $g/step_timeline_device.txt profiles/${pre}_step_timeline_device.txt
cp $g/pmc_sliced/pmc_summary.txt profiles/${pre}_pmc_cross_attn_sliced.txt
cp $g/pmc_agg/pmc_summary.txt profiles/${pre}_pmc_cross_attn_agg.txt
cp $g/pmc_fwd/pmc_summary.txt profiles/${pre}_pmc_cross_attn.txt
cp $g/pmc_step/pmc_summary.txt profiles/${pre}_pmc_step_inflight1.txt
cp $g/pmc_rawbwd/pmc_summary.txt profiles/${pre}_pmc_train_raw_backward.txt
python3 tools/make_pmc_record.py $g/pmc_sliced/pmc_summary.txt profiles/${pre}_pmc_cross_attn_sliced.json \
  "tools/bench_sliced.py --coarse: 900 x 24 x 4 levels, fp32 slice-planar pyramid, levels 2-3 from projected rows; one launch = the gather of one decoder layer (items form of the plan)" cross_attn_agg_items_coarse_kernel gd4d_cross_attn_sliced.hip
python3 tools/make_pmc_record.py $g/pmc_agg/pmc_summary.txt profiles/${pre}_pmc_cross_attn_agg.json \
  "tools/bench_late.py: 900 x 24 x 4 levels, fp32 channels-last pyramid; one launch = the row-form aggregate of one decoder layer" cross_attn_agg_kernel gd4d_cross_attn_late.hip
python3 tools/make_pmc_record.py $g/pmc_fwd/pmc_summary.txt profiles/${pre}_pmc_cross_attn.json \
  "tools/bench_kernel.py --order: 900 x 24 x 4 levels, fp32 projected values; one launch = the gather of one decoder layer" cross_attn_fwd_block gd4d_cross_attn.hip
python3 tools/make_pmc_record.py $g/pmc_rawbwd/pmc_summary.txt profiles/${pre}_pmc_cross_attn_dot_sliced.json \
  "tools/bench_raw_bwd.py: 900 x 24 x 4 levels, fp32 slice-planar pyramid; one launch = the backward gather of one decoder layer" cross_attn_dot_sliced_kernel gd4d_cross_attn_sliced_bwd.hip
python3 tools/make_pmc_record.py $g/pmc_rawbwd/pmc_summary.txt profiles/${pre}_pmc_pyramid_grad_reduce.json \
  "tools/bench_raw_bwd.py: 11.6 M records of 6 decoder layers, 900 x 24 x 4 levels; one launch = the pyramid's gradient of one training step" pyramid_grad_reduce_kernel gd4d_cross_attn_sliced_bwd.hip
for p in "train:bench_train" "train_criterion:bench_train_criterion" "train_vov:bench_train_vov" "distill:bench_distill" "train_projected_values:bench_train_projected_values"; do
  tail -1 $g/train/${p%%:*}.json > profiles/${pre}_${p##*:}.json
done
cp $g/train/stats/train_kernel_stats.csv profiles/${pre}_kernel_stats_train_step.csv
echo collected
cp $(find $g/hpe -name '*kernel_stats.csv' | head -1) profiles/${pre}_kernel_stats_head_pe.csv
cp $g/head_pe_breakdown.txt profiles/${pre}_head_pe_breakdown.txt
cp $g/pmc_mlp2/pmc_summary.txt profiles/${pre}_pmc_mlp2.txt
cp gpurun_out/f2b/f2b_kernel_stats.csv profiles/${pre}_kernel_stats_features_to_boxes.csv 2>/dev/null
echo collected head pe
