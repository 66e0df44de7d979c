#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_training_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
