cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_t7; mkdir -p $o
timeout 900 python3 -m pytest tests/test_head_pe_gpu.py -q -m gpu -k "mlp" > $o/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |^FAILED|^ERROR|passed|failed" $o/pytest.log | tail -20
timeout 300 python3 tools/bench_mlp2.py 2>&1 | tail -2
