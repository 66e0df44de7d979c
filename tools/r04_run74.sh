#!/bin/bash
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04_run74; mkdir -p $o
timeout 150 python3 -u -m pytest tests/test_cross_attn_sliced_bwd_gpu.py -x -v -m gpu -p no:cacheprovider -k "ride or count" 2>&1 | tail -15
