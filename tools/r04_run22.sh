#!/bin/bash
# (a) TN-GEMM reduce re-check, (b) timeline of one replayed TRAINING step (graph) with and without dropout
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run22; mkdir -p $o
timeout 900 python3 -m pytest tests/test_head_pe_gpu.py tests/test_timed_size_parity_gpu.py -x -q -m gpu -p no:cacheprovider -k "head or gemm or se_fuse or pe" > $o/pytest_pe.log 2>&1; echo "pe rc=$? $(tail -1 $o/pytest_pe.log)"
CAMS=24 timeout 300 python3 tools/time_head_pe_train.py 2>&1 | head -4
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -f csv -d $o/tl -o train -- python3 bench.py --mode train --steps 4 --warmup 2 --no-roofline > $o/train.json 2> $o/train.err
tail -1 $o/train.json | cut -c1-300
t=$(find $o/tl -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $t pyramid_slice > $o/timeline_train.txt
find $o/tl -name '*kernel_trace.csv' -delete
head -5 $o/timeline_train.txt
