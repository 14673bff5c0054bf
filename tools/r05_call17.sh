#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "dcn" 2>&1 | tail -5
