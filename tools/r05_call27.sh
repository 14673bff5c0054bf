#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "gram_regression_heads_vs_torch" 2>&1 | grep -E "^E  |passed|failed|assert " | head -20
