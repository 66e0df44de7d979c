#!/bin/bash
# memory-side PMC passes only (FETCH_SIZE, WRITE_SIZE, TCC hit/miss, SQ wait buckets) for a dev-tool command
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -f csv -d gpurun_out/$tag/p$i -o pmc -- python3 "$@" > gpurun_out/$tag/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 gpurun_out/$tag/p$i.log)"
  find gpurun_out/$tag/p$i -name '*kernel_trace.csv' -delete
done
python3 tools/pmc_summary.py gpurun_out/$tag
