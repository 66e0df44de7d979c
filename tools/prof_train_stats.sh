#!/bin/bash
# rocprofv3 kernel stats of the eager training step -> gpurun_out/<tag>/stats/ (top of the table printed)
tag=${1:-train}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/$tag/stats -o train -- python3 bench.py --mode train --steps 5 --warmup 2 --no-graph "$@" > gpurun_out/$tag/profiled.json 2> gpurun_out/$tag/profiled.err
find gpurun_out/$tag/stats -name '*kernel_trace.csv' -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/$tag/stats/train_kernel_stats.csv')))
import json
# timed + warm-up steps as the bench line reports them (the script's own --steps 5 --warmup 2 unless "$@" overrode them) + the two
# untimed passes bench.py runs before them (the eager check pass and the first pass that binds the gradient buffer)
line=json.loads([l for l in open('gpurun_out/$tag/profiled.json') if l.startswith('{')][-1])
steps=int(line['steps'])+int(line['warmup'])+2
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step (ms):', round(tot/steps/1e6,3))
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/steps/1e3:9.1f} us/step {int(r['Calls'])/steps:7.1f} calls {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
