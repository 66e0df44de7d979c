#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 tools/dev_image_group_time.py 2>&1 | tail -1
timeout 600 python3 -m pytest tests/test_train_chains_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1
