#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "pointwise or root_cat or conv2d" 2>&1 | tail -4
  timeout 300 python tools/pw_layers_bench.py 8 2>&1 | grep -v amdgpu.ids ) > gpurun_out/pw_layers.md 2>&1
cat gpurun_out/pw_layers.md
