#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
( timeout 600 python tools/pointwise_bench.py 8 2>&1 | grep -v amdgpu
  timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv2d or root_cat or split_k or conv_cases or rowmap or heads" 2>&1 | tail -2 ) > gpurun_out/igemm_pf2.md 2>&1
cat gpurun_out/igemm_pf2.md
