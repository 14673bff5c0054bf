#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_train_fullsize.py -x -q > gpurun_out/train_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/train_tests.log | tail -5
