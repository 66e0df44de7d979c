cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05_t1; mkdir -p $o
timeout 2400 python3 -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $o/pytest.log
