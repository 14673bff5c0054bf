#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_cw" 2>&1 | tail -3
  for o in "cw_rows6=0" "cw_rows6=1"; do echo "== $o"; timeout 300 python tools/conv_cw_bench.py 8 $o 2>&1 | grep -v amdgpu.ids | grep "512\|12x40\|layer"; done
  timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | tail -3 ) > gpurun_out/cw_rows6.md 2>&1
cat gpurun_out/cw_rows6.md
