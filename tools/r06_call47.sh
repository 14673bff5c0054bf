#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c47; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_step.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" > $O/t.txt
