#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run24; mkdir -p $o
timeout 600 python3 tools/dev_chain_debug.py > $o/debug.txt 2>&1; tail -80 $o/debug.txt
