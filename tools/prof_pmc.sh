#!/bin/bash
# PMC passes (one group per pass; no tracing domains besides kernel-trace) for a dev-tool command.
# usage: bash tools/prof_pmc.sh <tag> <python tool + args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -f csv -d gpurun_out/$tag/p$i -o pmc -- python3 "$@" > gpurun_out/$tag/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 gpurun_out/$tag/p$i.log)"
  find gpurun_out/$tag/p$i -name '*kernel_trace.csv' -delete
done
python3 tools/pmc_summary.py gpurun_out/$tag
