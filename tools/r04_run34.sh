#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python3 tools/dev_count_scaling.py 2>&1 | tail -5
