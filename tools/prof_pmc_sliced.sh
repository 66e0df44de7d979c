#!/bin/bash
# PMC passes of the step's gather alone (tools/bench_sliced.py) -> gpurun_out/<tag>/pmc_sliced/pmc_summary.txt
tag=${1:-pmc_sliced}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag/pmc_sliced
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -f csv -d gpurun_out/$tag/pmc_sliced/p$i -o pmc -- python3 tools/bench_sliced.py --iters 3 --coarse > gpurun_out/$tag/pmc_sliced/p$i.log 2>&1 || echo "pass $i failed"
  find gpurun_out/$tag/pmc_sliced/p$i -name '*kernel_trace.csv' -delete
done
python3 tools/pmc_summary.py gpurun_out/$tag/pmc_sliced | grep -A 14 "cross_attn_agg_items_coarse" | head -16
