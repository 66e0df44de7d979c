#!/bin/bash
# Round measurement bundle (run on the GPU box through gpurun):
#   1. default bench line            -> gpurun_out/<tag>/bench.json
#   2. rocprofv3 kernel stats of the SAME command -> gpurun_out/<tag>/stats/
#   3. PMC passes (HBM traffic FETCH_SIZE / WRITE_SIZE in separate passes, L2 hits, wait buckets) of the step's gather
#      (gd4d_cross_attn_agg_sliced_fwd, tools/bench_sliced.py), of its one-workgroup-per-query form (tools/bench_late.py) and of
#      the projected-value gather (tools/bench_kernel.py); issue / MFMA / LDS counters of the row chains and the attention core
tag=${1:-round}
stages=${2:-1234}      # 1: bench lines + kernel stats + timelines, 2: PMC passes of the inference kernels, 3: training, 4: head PE + features -> boxes
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
if [[ $stages == *1* ]]; then
python3 bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -c 400 gpurun_out/$tag/bench.json
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/$tag/stats -o bench -- python3 bench.py > gpurun_out/$tag/bench_profiled.json 2> gpurun_out/$tag/bench_profiled.err
find gpurun_out/$tag/stats -name '*kernel_trace.csv' -delete
# one sample in flight, no all-visible stress launches: the run whose average duration of the aggregate kernel is the
# figure `roofline` is priced on (in the default run two samples' launches overlap and the stress launches share the name)
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/$tag/stats1 -o bench -- python3 bench.py --inflight 1 --no-stress > gpurun_out/$tag/bench_inflight1.json 2> gpurun_out/$tag/bench_inflight1.err
find gpurun_out/$tag/stats1 -name '*kernel_trace.csv' -delete
# timeline of one step with ONE sample in flight (the default bench interleaves two), profiled and from the device's own clock
rocprofv3 --kernel-trace -f csv -d gpurun_out/$tag/tl -o bench -- python3 bench.py --inflight 1 --no-roofline --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/$tag/bench_tl.json 2> gpurun_out/$tag/bench_tl.err
t=$(find gpurun_out/$tag/tl -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $t > gpurun_out/$tag/step_timeline.txt 2>&1
find gpurun_out/$tag/tl -name '*kernel_trace.csv' -delete
python3 tools/trace_step.py > gpurun_out/$tag/step_timeline_device.txt 2>&1
python3 tools/trace_step.py nhwc > gpurun_out/$tag/step_timeline_device_nhwc.txt 2>&1
fi
if [[ $stages == *2* ]]; then
for which in sliced:tools/bench_sliced.py:--iters:3:--coarse agg:tools/bench_late.py:--iters:2 fwd:tools/bench_kernel.py:--iters:5:--order; do
  name=${which%%:*}; cmd=$(echo ${which#*:} | tr ':' ' ')
  mkdir -p gpurun_out/$tag/pmc_$name
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp -f csv -d gpurun_out/$tag/pmc_$name/p$i -o pmc -- python3 $cmd > gpurun_out/$tag/pmc_$name/p$i.log 2>&1 || echo "pass $name $i failed"
    find gpurun_out/$tag/pmc_$name/p$i -name '*kernel_trace.csv' -delete
  done
  python3 tools/pmc_summary.py gpurun_out/$tag/pmc_$name > /dev/null
  find gpurun_out/$tag/pmc_$name -name '*counter_collection.csv' -delete
done
grep -A 14 "cross_attn_agg_items_coarse" gpurun_out/$tag/pmc_sliced/pmc_summary.txt | head -16
# the query side of the step (row chains, attention core): issue / MFMA / LDS / wait counters, one sample in flight
bash tools/prof_pmc.sh $tag/pmc_step bench.py --inflight 1 --no-roofline --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
grep -A 40 "row_chain_kernel\|mha_core_kernel" gpurun_out/$tag/pmc_step/pmc_summary.txt | head -100
find gpurun_out/$tag/pmc_step -name '*counter_collection.csv' -delete
fi
if [[ $stages == *3* ]]; then
# training: bench lines + kernel stats of the step, PMC passes of the raw-pyramid backward kernels (tools/bench_raw_bwd.py)
mkdir -p gpurun_out/$tag/train
python3 bench.py --mode train --steps 20 --warmup 3 > gpurun_out/$tag/train/train.json 2> gpurun_out/$tag/train/train.err
python3 bench.py --mode train --steps 10 --warmup 3 --criterion > gpurun_out/$tag/train/train_criterion.json 2> gpurun_out/$tag/train/train_criterion.err
python3 bench.py --mode train --steps 5 --warmup 2 --levels vov > gpurun_out/$tag/train/train_vov.json 2> gpurun_out/$tag/train/train_vov.err
python3 bench.py --mode distill --steps 5 --warmup 2 > gpurun_out/$tag/train/distill.json 2> gpurun_out/$tag/train/distill.err
GD4D_TRAIN_VALUES=projected python3 bench.py --mode train --steps 10 --warmup 3 --no-fuse-wgrad > gpurun_out/$tag/train/train_projected_values.json 2> gpurun_out/$tag/train/train_projected_values.err
for f in train train_criterion train_vov distill train_projected_values; do tail -1 gpurun_out/$tag/train/$f.json | cut -c1-200; done
bash tools/prof_train_stats.sh $tag/train --no-roofline | head -12
bash tools/prof_pmc.sh $tag/pmc_rawbwd tools/bench_raw_bwd.py --iters 2 > /dev/null 2>&1
grep -A 26 "dot_sliced_kernel\|grad_reduce_kernel\|grad_count_kernel" gpurun_out/$tag/pmc_rawbwd/pmc_summary.txt | head -90
find gpurun_out/$tag/pmc_rawbwd -name '*counter_collection.csv' -delete
fi
if [[ $stages == *4* ]]; then
# the head's feature position embedding: kernel stats of the library's own route, PMC of the fused position MLP
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/$tag/hpe -o hpe -- python3 tools/bench_head_pe.py --hip-only > gpurun_out/$tag/hpe.log 2>&1
find gpurun_out/$tag/hpe -name '*kernel_trace.csv' -delete
python3 tools/bench_head_pe.py 2>/dev/null | grep -v amdgpu > gpurun_out/$tag/head_pe_breakdown.txt
bash tools/prof_pmc.sh $tag/pmc_mlp2 tools/bench_mlp2.py > /dev/null 2>&1
grep -A 30 "mlp2_kernel" gpurun_out/$tag/pmc_mlp2/pmc_summary.txt | head -40
# features -> boxes as one request (head position embedding -> decoder -> box epilogue -> decode): kernel statistics
bash tools/prof_f2b.sh | tail -3
find gpurun_out/$tag/pmc_mlp2 -name '*counter_collection.csv' -delete
fi
find gpurun_out -name '*agent_info.csv' -delete; find gpurun_out -name '*domain_stats.csv' -delete
du -sh gpurun_out | tail -1
