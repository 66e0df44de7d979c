#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run29; mkdir -p $o
ulimit -c 0
for i in 1 2; do timeout 1500 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > $o/pytest_$i.log 2>&1; echo "all $i rc=$? $(tail -1 $o/pytest_$i.log)"; done
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for rep in 1 2; do
for v in "chains GD4D_TRAIN_CHAINS=1" "generic GD4D_TRAIN_CHAINS=0"; do set -- $v; name=$1; shift
env "$@" python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline > $o/${name}_$rep.json 2> $o/${name}_$rep.err
echo "$name $(tail -1 $o/${name}_$rep.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
env "$@" python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline --dropout > $o/${name}_drop_$rep.json 2> $o/${name}_drop_$rep.err
echo "$name dropout $(tail -1 $o/${name}_drop_$rep.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done; done
python3 bench.py > $o/bench_default.json 2> $o/bench_default.err; tail -1 $o/bench_default.json | cut -c1-400
