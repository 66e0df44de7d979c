#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python3 tools/dev_coarse_proj.py 2>&1 | tail -6
