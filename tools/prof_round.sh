#!/bin/bash
# Round measurement bundle (run on the GPU box through gpurun):
#   1. default bench line            -> gpurun_out/<tag>/bench.json
#   2. rocprofv3 kernel stats of the SAME command -> gpurun_out/<tag>/stats/
#   3. PMC HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the fused kernel alone
tag=${1:-round}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
python3 bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -c 600 gpurun_out/$tag/bench.json
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/$tag/stats -o bench -- python3 bench.py > gpurun_out/$tag/bench_profiled.json 2> gpurun_out/$tag/bench_profiled.err
find gpurun_out/$tag/stats -name '*kernel_trace.csv' -delete
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -f csv -d gpurun_out/$tag/pmc_$n -o pmc -- python3 tools/bench_kernel.py --iters 5 --order > gpurun_out/$tag/pmc_$n.log 2>&1
  find gpurun_out/$tag/pmc_$n -name '*kernel_trace.csv' -delete
done
python3 tools/pmc_summary.py gpurun_out/$tag
grep -h "alg_bytes" gpurun_out/$tag/pmc_FETCH_SIZE.log | tail -1
