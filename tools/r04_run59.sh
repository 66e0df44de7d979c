#!/bin/bash
# what-if: the attention core without its K / V operand splits (wrong results; timing only)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run59; mkdir -p $o
ms() { tail -1 $1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("value_batch1"))'; }
for rep in 1 2 3; do
python3 bench.py --inflight 1 --steps 200 --warmup 10 --no-roofline --no-cpu-baseline > $o/b_base_$rep.json 2> $o/b_base_$rep.err; echo "base $(ms $o/b_base_$rep.json)"
GD4D_LIB_PATH=$PWD/build_ab/libgd4d_whatif.so python3 bench.py --inflight 1 --steps 200 --warmup 10 --no-roofline --no-cpu-baseline > $o/b_whatif_$rep.json 2> $o/b_whatif_$rep.err; echo "whatif $(ms $o/b_whatif_$rep.json)"
done
GD4D_LIB_PATH=$PWD/build_ab/libgd4d_whatif.so python3 tools/trace_step.py 2>&1 | grep mha_core | head -6
