#!/bin/bash
# PMC passes of the raw-pyramid backward kernels again (their source file changed: the records are keyed to its hash)
cd "$GRAFT_REPO_ROOT"
true
grep -A 3 "dot_sliced_kernel" gpurun_out/r04b/pmc_rawbwd/pmc_summary.txt | head -6
mkdir -p gpurun_out/r04b/train
python3 bench.py --mode train --steps 20 --warmup 3 > gpurun_out/r04b/train/train.json 2> gpurun_out/r04b/train/train.err
python3 bench.py --mode train --steps 20 --warmup 3 --dropout > gpurun_out/r04b/train/train_dropout.json 2> gpurun_out/r04b/train/train_dropout.err
for f in train train_dropout; do tail -1 gpurun_out/r04b/train/$f.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["traffic"])'; done
