#!/bin/bash
# round 4, run 2: new timed-size parity tests; row-chain K-permutation A/B (rebuilds gd4d_rowchain.o on the box)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run2; mkdir -p $o
python3 -m pytest tests/test_timed_size_parity_gpu.py -x -q -m gpu --durations=8 > $o/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -25 $o/pytest_parity.log
python3 -m pytest tests/test_rowchain_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $o/pytest_chain.log 2>&1; echo "pytest chain rc=$?"; tail -5 $o/pytest_chain.log
b1() { python3 bench.py --inflight 1 --no-stress --no-cpu-baseline --no-roofline --no-nhwc-figure --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_sample_batch1'],4), 'ms per sample')"; }
b1 kperm; b1 kperm
python3 tools/bench_chain.py 2>&1 | tail -12 | tee $o/chain_kperm.txt
touch graph-detr4d_amd/csrc/gd4d_rowchain.hip
make -s -C graph-detr4d_amd/csrc EXTRA="-DRC_KPERM=0 -DRC_LD_N=516" 2>&1 | grep -i error
b1 r3layout; b1 r3layout
python3 tools/bench_chain.py 2>&1 | tail -12 | tee $o/chain_r3.txt
touch graph-detr4d_amd/csrc/gd4d_rowchain.hip
make -s -C graph-detr4d_amd/csrc 2>&1 | grep -i error
b1 kperm
