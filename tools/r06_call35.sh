#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c35; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_train_step.py -q -p no:cacheprovider 2>&1 | grep -E "^FAILED|AssertionError|passed|failed" > $O/t_full.txt
MFX_TARGET_ARENA=0 timeout 2400 python -m pytest tests/test_gpu_train_step.py -q -p no:cacheprovider 2>&1 | grep -E "^FAILED|AssertionError|passed|failed" > $O/t_full_noarena.txt
