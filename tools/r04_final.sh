#!/bin/bash
# Round 4, final tree: the whole measurement bundle (tools/prof_round.sh), the training lines beside it, the other configurations,
# the training timeline and the full GPU test run.  Collected into profiles/ with tools/collect_round_profiles.sh r04c r04; the
# dropout / per-module training lines, the training timeline and the configs/ lines are copied by hand (profiles/r04_bench_*.json,
# r04_step_timeline_train_dropout_profiled.txt).
cd "$GRAFT_REPO_ROOT"
bash tools/prof_round.sh r04c > /dev/null 2>&1
o=gpurun_out/r04c/train; mkdir -p $o
python3 bench.py --mode train --steps 20 --warmup 3 --dropout > $o/train_dropout.json 2> $o/train_dropout.err
GD4D_TRAIN_CHAINS=0 python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline > $o/train_generic.json 2> $o/train_generic.err
GD4D_TRAIN_CHAINS=0 python3 bench.py --mode train --steps 20 --warmup 3 --no-roofline --dropout > $o/train_generic_dropout.json 2> $o/train_generic_dropout.err
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -f csv -d $o/tl -o train -- python3 bench.py --mode train --steps 6 --warmup 2 --no-roofline --dropout > $o/train_tl.json 2> $o/train_tl.err
t=$(find $o/tl -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $t pyramid_slice > $o/timeline_train_dropout.txt
find $o/tl -name '*kernel_trace.csv' -delete
for f in train train_dropout train_generic train_generic_dropout train_criterion train_vov distill train_projected_values; do echo "$f $(tail -1 $o/$f.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3))')"; done
c=gpurun_out/r04c/configs; mkdir -p $c
run() { name=$1; shift; python3 bench.py --no-stress --no-cpu-baseline --no-roofline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 > $c/$name.json; python3 -c "
import json
d=json.loads(open('$c/$name.json').read()); c=d.get('channels_last_input') or {}
print('$name:', round(d['value'],1), 'samples/s;', round(d['ms_per_sample_batch1'],4), 'ms one at a time;', 'channels-last', round(c.get('value',0),1), round(c.get('ms_per_sample_batch1',0),4))"; }
run cfg1_bf16_6cams --frames 1 --value-dtype bf16
run cfg1_fp32_6cams --frames 1
run bf16_24cams --value-dtype bf16
run vov_24cams --levels vov
run hdetr_2700q --queries 2700
tail -1 gpurun_out/r04c/bench.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('default:', round(d['value'],1), d['ms_per_step'], 'batch1', d.get('value_batch1'), 'frac', r['frac'], 'traffic', r['traffic'], 'nhwc', (d.get('channels_last_input') or {}).get('value'))"
tail -1 gpurun_out/r04c/bench_inflight1.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('inflight1:', round(d['value'],1), d['ms_per_step'], 'frac', r['frac'], 'us', r.get('us_per_launch'))"
ulimit -c 0
timeout 600 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r04c/pytest_final.log 2>&1; echo "all rc=$? $(tail -1 gpurun_out/r04c/pytest_final.log)"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
