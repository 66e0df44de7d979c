cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_t3; mkdir -p $o
timeout 2700 python3 -m pytest tests -q -m gpu -s > $o/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR|passed|failed|flipped mask bit:" $o/pytest.log | tail -40
