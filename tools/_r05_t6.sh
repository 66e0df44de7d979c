cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_t6; mkdir -p $o
timeout 900 python3 -m pytest tests/test_head_pe_gpu.py tests/test_head_loss_gpu.py -q -m gpu -k "mlp or device_assign" > $o/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |^FAILED|^ERROR|passed|failed" $o/pytest.log | tail -20
timeout 300 python3 tools/bench_mlp2.py 2>&1 | tail -3
timeout 600 python3 bench.py --mode train --criterion --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train criterion', round(d['ms_per_step'],3), d['config']['launch'])"
