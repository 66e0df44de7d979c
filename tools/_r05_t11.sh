cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_t11; mkdir -p $o
timeout 300 python3 tools/bench_hungarian.py 40 100 2>&1 | grep problems
timeout 900 python3 -m pytest tests/test_head_loss_gpu.py -q -m gpu > $o/pytest2.log 2>&1; echo "pytest2 rc=$?"; grep -E "^E  |^FAILED|^ERROR|passed|failed" $o/pytest2.log | tail -12
for i in 1 2; do timeout 600 python3 bench.py --mode train --criterion --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train criterion', round(d['ms_per_step'],3), d['config']['launch'])"; done
