cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_t9; mkdir -p $o
timeout 900 python3 -m pytest tests/test_modules_gpu.py -q -m gpu -k "any_num_points" > $o/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |^FAILED|^ERROR|passed|failed" $o/pytest.log | tail -30
timeout 2400 python3 -m pytest tests -q -m gpu -x > $o/pytest_all.log 2>&1; echo "pytest all rc=$?"; grep -E "^E  |^FAILED|^ERROR|passed|failed" $o/pytest_all.log | tail -12
timeout 300 python3 tools/bench_head_pe.py 2>&1 | grep -v "^E2026\|^W2026\|amdgpu" | tail -5
