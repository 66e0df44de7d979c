#!/bin/bash
# the other configurations, for orientation (docs/measurements_r04.md section 5)
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run17; mkdir -p $o
run() { name=$1; shift; python3 bench.py --no-stress --no-cpu-baseline --no-roofline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 > $o/$name.json; python3 -c "
import json
d=json.loads(open('$o/$name.json').read()); c=d.get('channels_last_input') or {}
print('$name:', round(d['value'],1), 'samples/s;', round(d['ms_per_sample_batch1'],4), 'ms one at a time;', 'channels-last', round(c.get('value',0),1), round(c.get('ms_per_sample_batch1',0),4))"; }
run cfg1_bf16_6cams --frames 1 --value-dtype bf16
run cfg1_fp32_6cams --frames 1
run bf16_24cams --value-dtype bf16
run vov_24cams --levels vov
run hdetr_2700q --queries 2700
python3 bench.py --mode train --dropout --steps 20 --warmup 3 --no-roofline 2>/dev/null | tail -1 > $o/train_dropout.json; python3 -c "
import json; d=json.loads(open('$o/train_dropout.json').read()); print('train --dropout:', round(d['ms_per_step'],3), 'ms')"
