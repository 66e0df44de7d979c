cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_t5; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_head_loss_gpu.py tests/test_modules_gpu.py tests/test_rowchain_gpu.py tests/test_train_chains_gpu.py tests/test_configs_gpu.py -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |^FAILED|^ERROR|passed|failed" $o/pytest.log | tail -30
for m in "--criterion" "--criterion --no-graph"; do timeout 600 python3 bench.py --mode train $m --no-roofline 2>$o/err.txt | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train $m', round(d['ms_per_step'],3), d['config']['launch'])"; tail -2 $o/err.txt; done
timeout 600 python3 bench.py --mode train --criterion --optimizer sgd --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train criterion sgd', round(d['ms_per_step'],3), d['config']['launch'])"
